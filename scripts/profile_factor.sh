#!/bin/bash
# Evidence for the matrix-core factorisation (on the GPU box, through gpurun):  bash scripts/profile_factor.sh <tag> [n] [count]
#   1. kernel-trace stats of both flavours (scripts/measure_factor.py)           -> gpurun_out/<tag>_factor_kernel_stats.csv
#   2. PMC, separate passes, fast flavour only: FETCH_SIZE | WRITE_SIZE | SQ ... -> gpurun_out/<tag>_factor_pmc.json
#   3. a -DSOCP_FACTOR_PROFILE build: clock ticks per phase of wave 0            -> gpurun_out/<tag>_factor_phases.txt
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
TAG=${1:-r04}; N=${2:-253}; COUNT=${3:-2048}
OUT=gpurun_out
mkdir -p $OUT
M="python3 scripts/measure_factor.py $N $COUNT 3"
$M > $OUT/${TAG}_factor_unprofiled.json 2> /dev/null; cat $OUT/${TAG}_factor_unprofiled.json
rm -rf $OUT/pf_*
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf_stats -- $M > /dev/null 2>&1
f=$(find $OUT/pf_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|factor" $f > $OUT/${TAG}_factor_kernel_stats.csv; cat $OUT/${TAG}_factor_kernel_stats.csv | cut -c1-200
# (the stats' averages mix the warm-up launch of 8 problems with the measured ones: per-dispatch durations of the MEASURED launches here)
python3 - "$TAG" "$N" "$COUNT" <<'PY'
import csv, glob, json, sys
tag, n, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out = {"tag": tag, "n": n, "count": count, "source": "rocprofv3 --kernel-trace of scripts/measure_factor.py %d %d 3 (launches over all %d problems only: the warm-up launch of 8 is left out)" % (n, count, count)}
for f in glob.glob("gpurun_out/pf_stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = "factor_fast_kernel" if "factor_fast_kernel" in k else ("factor_only_kernel" if "factor_only_kernel" in k else None)
        if key and int(r["Grid_Size_X"]) >= 64 * count:
            out.setdefault(key, {"kernel": k.split("(")[0], "dispatch_ms": []})["dispatch_ms"].append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6, 4))
# (the throughput refresh is two launches: their dispatches alternate; a refresh = one of each)
for v in out.values():
    if isinstance(v, dict) and v.get("dispatch_ms"):
        if v is out.get("factor_fast_kernel") and len(v["dispatch_ms"]) % 2 == 0 and len(v["dispatch_ms"]) >= 2:
            d = v["dispatch_ms"]
            v["dispatch_ms_qrfac_qform"] = [[d[i], d[i + 1]] for i in range(0, len(d), 2)]
            v["dispatch_ms"] = [round(d[i] + d[i + 1], 4) for i in range(0, len(d), 2)]
        v["average_ms"] = sum(v["dispatch_ms"]) / len(v["dispatch_ms"])
        v["tflops"] = 8.0 / 3.0 * n ** 3 * count / (v["average_ms"] * 1e-3) / 1e12
        v["frac_of_fp64_peak_78.6"] = v["tflops"] / 78.6
try:
    out["hip_events_unprofiled"] = json.load(open("gpurun_out/%s_factor_unprofiled.json" % tag))
except Exception:
    pass
json.dump(out, open("gpurun_out/%s_factor_timing.json" % tag, "w"), indent=1)
print(json.dumps(out)[:600])
PY
export SOCP_MEASURE_ONLY=fast
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf_fetch -- $M > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pf_write -- $M > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pf_sq -- $M > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pf_sq2 -- $M > /dev/null 2>&1
python3 - "$TAG" "$N" "$COUNT" <<'PY'
import csv, glob, json, os, sys
tag, n, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out = {"tag": tag, "n": n, "count": count, "kernel": "factor_fast_kernel"}
c = {}
for d in ("pf_fetch", "pf_write", "pf_sq", "pf_sq2"):
    for f in sorted(glob.glob("gpurun_out/%s/**/*_counter_collection.csv" % d, recursive=True), key=os.path.getmtime)[-1:]:
        # (a refresh is TWO launches since round 5 -- qrfac, qform: two instantiations of the kernel --: per counter the median over the
        # dispatches of each instantiation, summed over the instantiations)
        vals = {}
        for r in csv.DictReader(open(f)):
            if "factor_fast_kernel" in r["Kernel_Name"] and float(r["Grid_Size"]) >= 256 * count:
                vals.setdefault(r["Counter_Name"], {}).setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        for k, per_kernel in vals.items():
            c[k] = sum(sorted(v)[len(v) // 2] for v in per_kernel.values())
            out.setdefault("launches_per_refresh", len(per_kernel))
out["counters_median_per_launch"] = c
fetch_b, write_b = 2.0 * c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
alg = 8.0 * count * (2 * n * n + n * (n + 1) / 2)
out.update({"hbm_read_bytes(2x FETCH_SIZE KiB)": fetch_b, "hbm_write_bytes(WRITE_SIZE KiB)": write_b, "hbm_bytes": fetch_b + write_b,
            "algorithmic_bytes": alg, "traffic_over_algorithmic": (fetch_b + write_b) / alg if alg else None})
json.dump(out, open("gpurun_out/%s_factor_pmc.json" % tag, "w"), indent=1)
print(json.dumps(out))
PY
unset SOCP_MEASURE_ONLY
rm -rf $OUT/pf_*
# 3. the phase clocks: a profile build in its own directory, selected with SOCP_LIB_PATH (the product library is never rebuilt in place)
bash scripts/factor_phases.sh "$TAG" "$N" "$COUNT" > /dev/null
cat $OUT/${TAG}_factor_phases.txt
