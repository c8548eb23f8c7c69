#!/bin/bash
# The kernels of one sweep in launch order with their durations (rocprofv3 --kernel-trace; the timed call only: everything after the last
# launch over fewer than MIN problems' worth of threads is the warm-up's):  bash scripts/probes/sweep_timeline.sh <tag> <sweep arguments ...>
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
TAG=$1; shift
OUT=gpurun_out; rm -rf $OUT/tl_$TAG
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$TAG -- python3 -m socp_amd.sweep "$@" > /dev/null 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, re, sys
tag = sys.argv[1]
rows = []
for f in glob.glob("gpurun_out/tl_%s/**/*kernel_trace.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
rows.sort()
t0 = rows[0][0]
out = open("gpurun_out/%s_timeline.txt" % tag, "w")
for s, e, k, g, wg in rows:
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", k)
    name = (m.group(1) + (m.group(2) or "")) if m else k[:40]
    out.write("%10.1f us  %8.1f us  grid %8d x %4d  %s\n" % ((s - t0) * 1e-3, (e - s) * 1e-3, g // max(wg, 1), wg, name))
PY
rm -rf $OUT/tl_$TAG
