#!/bin/bash
# HBM traffic of the device solver's launches in the config-5 sweep (on the GPU box, through gpurun):  bash scripts/profile_trial_round.sh <tag>
# Separate PMC passes (FETCH_SIZE | WRITE_SIZE), per-kernel medians -> gpurun_out/<tag>_trial_round_pmc.json
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
TAG=${1:-r04}; OUT=gpurun_out; mkdir -p $OUT
M="python3 -m socp_amd.sweep --model interceptor --starts 2048 --solver device_fast"
rm -rf $OUT/pt_*
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pt_stats -- $M > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pt_fetch -- $M > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pt_write -- $M > /dev/null 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
n, P = 253, 2048
def kind_of(k):
    # the launches of the matrix-core refresh: the chain's panel / trailing launches, the single qrfac launch, qform
    import re
    if "qrfac_panel" in k: return "refresh: qrfac_panel (chain)"
    if "qrfac_trail" in k: return "refresh: qrfac_trail (chain)"
    return "refresh: qform" if re.search(r"factor_fast_kernel<\d+, \d+, 2>", k) else "refresh: qrfac (single launch)"
def rows(d, pat):
    f = sorted(glob.glob("gpurun_out/%s/**/*%s" % (d, pat), recursive=True))
    return list(csv.DictReader(open(f[-1]))) if f else []
dur = {}
for r in rows("pt_stats", "kernel_trace.csv"):
    k = r["Kernel_Name"]
    if "advance_kernel" in k or "factor_fast" in k or "qrfac_" in k:
        key = ("advance<%s>" % k.split("advance_kernel<")[1].split(">")[0]) if "advance_kernel" in k else kind_of(k)
        if int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) >= P:
            dur.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
pmc = {}
for d, name in (("pt_fetch", "FETCH_SIZE"), ("pt_write", "WRITE_SIZE")):
    for r in rows(d, "counter_collection.csv"):
        k = r["Kernel_Name"]
        if ("advance_kernel" in k or "factor_fast" in k or "qrfac_" in k) and float(r["Grid_Size"]) / max(1.0, float(r["Workgroup_Size"])) >= P and r["Counter_Name"] == name:
            key = ("advance<%s>" % k.split("advance_kernel<")[1].split(">")[0]) if "advance_kernel" in k else kind_of(k)
            pmc.setdefault(key, {}).setdefault(name, []).append(float(r["Counter_Value"]))
out = {"tag": tag, "command": "python3 -m socp_amd.sweep --model interceptor --starts 2048 --solver device_fast", "n": n, "problems": P,
       "note": "launches over all 2048 problems only; bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (MI355X_MICROARCH.md); per-launch lists in dispatch order",
       "matrix_bytes": {"Q": 8.0 * n * n * P, "R": 8.0 * n * (n + 1) / 2 * P}, "kernels": {}}
for key in sorted(set(dur) | set(pmc)):
    f, w = pmc.get(key, {}).get("FETCH_SIZE", []), pmc.get(key, {}).get("WRITE_SIZE", [])
    by = [2048.0 * a + 1024.0 * b for a, b in zip(f, w)]
    out["kernels"][key] = {"ms": [round(x, 4) for x in dur.get(key, [])], "hbm_bytes": by,
                           "GBps_if_same_order": [round(b / (t * 1e-3) / 1e9, 1) for b, t in zip(by, dur.get(key, []))]}
json.dump(out, open("gpurun_out/%s_trial_round_pmc.json" % tag, "w"), indent=1)
print(json.dumps(out)[:3000])
PY
rm -rf $OUT/pt_*
