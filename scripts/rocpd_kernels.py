"""Per-kernel totals and the time line of one kernel from a rocprofv3 run that wrote a rocpd database (ROCm 7 default output).
    python scripts/rocpd_kernels.py <results.db> [substring of the kernel to list launch by launch]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select k.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d "
                  "join rocpd_info_kernel_symbol k on d.kernel_id = k.id order by d.start").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, s, e, g, w in rows:
    agg[name][0] += 1
    agg[name][1] += (e - s) / 1e6
print("%-90s %6s %10s %10s" % ("kernel", "calls", "total ms", "avg ms"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-90s %6d %10.3f %10.4f" % (k[:90], v[0], v[1], v[1] / v[0]))
if len(sys.argv) > 2:
    t0 = rows[0][1]
    for name, s, e, g, w in rows:
        if sys.argv[2] in name:
            print("t = %9.3f ms  %9.4f ms  grid %d  workgroup %d" % ((s - t0) / 1e6, (e - s) / 1e6, g, w))
