#!/bin/bash
# Per-launch durations of the chained qrfac + qform (rocprofv3 --kernel-trace):  bash scripts/probes/factor_chain_trace.sh <tag> [n] [count] [lib]
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
TAG=${1:-r06}; N=${2:-253}; COUNT=${3:-2048}
[ -n "$4" ] && export SOCP_LIB_PATH=$4
OUT=gpurun_out; mkdir -p $OUT; rm -rf $OUT/pf_chain
SOCP_MEASURE_ONLY=fast timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/pf_chain -- python3 scripts/measure_factor.py $N $COUNT 2 > /dev/null 2>&1
python3 - "$TAG" "$N" "$COUNT" <<'PY'
import csv, glob, sys
tag, n, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = []
for f in glob.glob("gpurun_out/pf_chain/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if ("qrfac_" in k or "factor_fast_kernel" in k) and int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) >= count:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (lambda k: "panel" if "qrfac_panel" in k else "trail" if "qrfac_trail" in k else ("qform" if __import__("re").search(r"factor_fast_kernel<\d+, \d+, 2>", k) else "qrfac_single") if "factor_fast_kernel" in k else None)(k)))
rows.sort()
out = open("gpurun_out/%s_factor_chain_trace_n%d.txt" % (tag, n), "w")
prev = None
for s, e, k in rows:
    line = "%-40s %9.1f us   gap before %7.1f us" % (k, (e - s) * 1e-3, (s - prev) * 1e-3 if prev else 0.0)
    print(line); out.write(line + "\n")
    prev = e
PY
rm -rf $OUT/pf_chain
