#!/bin/bash
# Evidence for the matrix-core factorisation (on the GPU box, through gpurun):  bash scripts/profile_factor.sh <tag> [n] [count]
# A refresh is a CHAIN of launches since round 6 (qrfac: a panel launch and a trailing launch per pair of panels -- from 640 problems up, one
# launch below --, then qform): every figure below is per REFRESH, the launches of the chain summed.
#   1. HIP-event time, un-profiled (scripts/measure_factor.py)                                 -> gpurun_out/<tag>_factor_unprofiled.json
#   2. kernel-trace: stats of both flavours + the per-launch durations of every measured refresh -> <tag>_factor_kernel_stats.csv, <tag>_factor_timing.json
#   3. PMC, separate passes (never with a trace domain), fast flavour only                     -> <tag>_factor_pmc.json
#   4. a -DSOCP_FACTOR_PROFILE build in its own directory: clock ticks per phase                -> <tag>_factor_phases.txt
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
TAG=${1:-r06}; N=${2:-253}; COUNT=${3:-2048}
OUT=gpurun_out
mkdir -p $OUT
M="python3 scripts/measure_factor.py $N $COUNT 3"
$M > $OUT/${TAG}_factor_unprofiled.json 2> /dev/null; cat $OUT/${TAG}_factor_unprofiled.json
rm -rf $OUT/pf_*
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf_stats -- $M > /dev/null 2>&1
f=$(find $OUT/pf_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|factor|qrfac" $f > $OUT/${TAG}_factor_kernel_stats.csv; cut -c1-200 $OUT/${TAG}_factor_kernel_stats.csv
python3 - "$TAG" "$N" "$COUNT" <<'PY'
import csv, glob, json, sys
tag, n, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out = {"tag": tag, "n": n, "count": count,
       "source": "rocprofv3 --kernel-trace of scripts/measure_factor.py %d %d 3: the launches over all %d problems only (the warm-up call of 8 is left out); "
                 "a throughput refresh = the chain's launches from one panel launch of pair 0 to the qform launch, summed" % (n, count, count)}
rows = []
for f in glob.glob("gpurun_out/pf_stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        kind = ("exact" if "factor_only_kernel" in k else (lambda k: "panel" if "qrfac_panel" in k else "trail" if "qrfac_trail" in k else ("qform" if __import__("re").search(r"factor_fast_kernel<\d+, \d+, 2>", k) else "qrfac_single") if "factor_fast_kernel" in k else None)(k))
        if kind and int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) >= count:          # (workgroups = problems: the warm-up calls are smaller)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
refreshes, cur = [], []
for s, e, kind in rows:
    if kind == "exact":
        out.setdefault("factor_only_kernel", {"dispatch_ms": []})["dispatch_ms"].append(round((e - s) * 1e-6, 4))
        continue
    cur.append((kind, round((e - s) * 1e-3, 1)))
    if kind == "qform":
        refreshes.append(cur); cur = []
fast = {"refreshes": [{"launches_us": r, "sum_ms": round(sum(d for _, d in r) * 1e-3, 4),
                       "panel_ms": round(sum(d for k, d in r if k == "panel") * 1e-3, 4), "trail_ms": round(sum(d for k, d in r if k == "trail") * 1e-3, 4),
                       "qrfac_single_ms": round(sum(d for k, d in r if k == "qrfac_single") * 1e-3, 4),
                       "qform_ms": round(sum(d for k, d in r if k == "qform") * 1e-3, 4)} for r in refreshes]}
if refreshes:
    fast["average_ms"] = sum(r["sum_ms"] for r in fast["refreshes"]) / len(refreshes)
    fast["tflops"] = 8.0 / 3.0 * n ** 3 * count / (fast["average_ms"] * 1e-3) / 1e12
    fast["frac_of_fp64_peak_78.6"] = fast["tflops"] / 78.6
out["factor_fast"] = fast
v = out.get("factor_only_kernel")
if v:
    v["average_ms"] = sum(v["dispatch_ms"]) / len(v["dispatch_ms"])
try:
    out["hip_events_unprofiled"] = json.load(open("gpurun_out/%s_factor_unprofiled.json" % tag))
except Exception:
    pass
json.dump(out, open("gpurun_out/%s_factor_timing.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in fast.items() if k != "refreshes"}), [r["sum_ms"] for r in fast["refreshes"]])
PY
rm -rf $OUT/pf_*
bash scripts/probes/factor_chain_pmc.sh "$TAG" "$N" "$COUNT" > $OUT/${TAG}_factor_pmc_launches.txt 2>&1; tail -3 $OUT/${TAG}_factor_pmc_launches.txt
# (the file the bench line quotes as its RECORDED figure: n, count, hbm_bytes)
python3 - "$TAG" "$N" <<'PY'
import json, sys
tag, n = sys.argv[1], int(sys.argv[2])
d = json.load(open("gpurun_out/%s_factor_chain_pmc_n%d.json" % (tag, n)))
d["kernel"] = "qrfac_panel_kernel + qrfac_trail_kernel (the chain) + factor_fast_kernel<., ., 2> (qform): every launch of one refresh summed"
json.dump(d, open("gpurun_out/%s_factor_pmc.json" % tag, "w"), indent=1)
PY
# the phase clocks: a profile build in its own directory, selected with SOCP_LIB_PATH (the product library is never rebuilt in place)
bash scripts/factor_phases.sh "$TAG" "$N" "$COUNT" > /dev/null
cat $OUT/${TAG}_factor_phases.txt | cut -c1-600
