#!/bin/bash
# PMC counters of every launch of the refresh's chain (separate passes; no trace domain):  bash scripts/probes/factor_chain_pmc.sh <tag> [n] [count]
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
TAG=${1:-r06}; N=${2:-253}; COUNT=${3:-2048}
OUT=gpurun_out; mkdir -p $OUT; rm -rf $OUT/pc_*
M="python3 scripts/measure_factor.py $N $COUNT 1"
export SOCP_MEASURE_ONLY=fast
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pc_fetch -- $M > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pc_write -- $M > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pc_sq -- $M > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pc_sq2 -- $M > /dev/null 2>&1
python3 - "$TAG" "$N" "$COUNT" <<'PY'
import csv, glob, json, os, sys
tag, n, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
launches = {}
for d in ("pc_fetch", "pc_write", "pc_sq", "pc_sq2"):
    for f in sorted(glob.glob("gpurun_out/%s/**/*_counter_collection.csv" % d, recursive=True), key=os.path.getmtime)[-1:]:
        seen = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("qrfac_panel", "qrfac_trail", "factor_fast_kernel")) or float(r["Grid_Size"]) / max(1.0, float(r["Workgroup_Size"])) < count:
                continue
            name = (lambda k: "panel" if "qrfac_panel" in k else "trail" if "qrfac_trail" in k else ("qform" if __import__("re").search(r"factor_fast_kernel<\d+, \d+, 2>", k) else "qrfac_single") if "factor_fast_kernel" in k else None)(k)
            did = int(r["Dispatch_Id"])
            seen.setdefault(did, name)
            launches.setdefault(d, {}).setdefault(did, {"kernel": name})[r["Counter_Name"]] = float(r["Counter_Value"])
rows = []
for d, per in launches.items():
    for i, did in enumerate(sorted(per)):
        while len(rows) <= i:
            rows.append({})
        rows[i].update(per[did])
out = {"tag": tag, "n": n, "count": count, "launches": rows}
tot = {}
for r in rows:
    for k, v in r.items():
        if k != "kernel":
            tot.setdefault(r["kernel"], {}).setdefault(k, 0.0)
            tot[r["kernel"]][k] += v
for name, c in tot.items():
    c["hbm_bytes"] = 2.0 * c.get("FETCH_SIZE", 0) * 1024 + c.get("WRITE_SIZE", 0) * 1024
out["totals"] = tot
out["hbm_bytes"] = sum(c["hbm_bytes"] for c in tot.values())
out["algorithmic_bytes"] = 8.0 * count * (2 * n * n + n * (n + 1) / 2)
out["traffic_over_algorithmic"] = out["hbm_bytes"] / out["algorithmic_bytes"]
json.dump(out, open("gpurun_out/%s_factor_chain_pmc_n%d.json" % (tag, n), "w"), indent=1)
for i, r in enumerate(rows):
    hb = 2.0 * r.get("FETCH_SIZE", 0) * 1024 + r.get("WRITE_SIZE", 0) * 1024
    print("%2d %-6s hbm %7.1f MB  matrix-core time %6.1f us (SQ_INSTS_MFMA x 64 cycles / 1024 SIMDs at 2.3 GHz)  wait_any/wave_cycles %4.2f  valu %9d mfma %8d lds %8d  waves %d" % (
        i, r["kernel"], hb / 1e6, r.get("SQ_INSTS_MFMA", 0) * 64.0 / 1024.0 / 2.3e3,
        r.get("SQ_WAIT_ANY", 0) / max(1.0, r.get("SQ_WAVE_CYCLES", 0)), r.get("SQ_INSTS_VALU", 0), r.get("SQ_INSTS_MFMA", 0), r.get("SQ_INSTS_LDS", 0), r.get("SQ_WAVES", 0)))
print("total hbm %.2f GB = %.2f x algorithmic" % (out["hbm_bytes"] / 1e9, out["traffic_over_algorithmic"]))
PY
rm -rf $OUT/pc_*
