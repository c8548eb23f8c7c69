#!/bin/bash
# A/B on the GPU box: the matrix-core refresh (qrfac + qform, scripts/measure_factor.py, ms, HIP events) of the product library against variant
# builds.  Build a variant here, before the call -- in its own directory, never in place:
#   bash scripts/variant_build.sh blk4 FACTOR_DEFS="-DSOCP_FACTOR_BLK=4"       (or -DSOCP_FACTOR_WPE_MID=3, -DSOCP_FACTOR_PANEL_WPE8=3, ...)
# and name the variants (default: every socp_amd/_build_* that holds a library):   bash scripts/probes/factor_variants_ab.sh [name ...]
# SIZES="n count; ..." overrides the sizes; SOCP_FACTOR_CHAIN_MIN=0 / 99999999 in the environment forces / forbids the chain of launches.
# (the one harness behind profiles/r0[4-6]_factor_*_ab.txt: round 5's per-experiment copies -- blk, wpe, split, staged -- are gone)
cd "$(dirname "$0")/../.."
[ $# -gt 0 ] && VARIANTS="$*" || VARIANTS=$(ls -d socp_amd/_build_* 2>/dev/null | sed 's|socp_amd/_build_||')
IFS=';' read -ra CFGS <<< "${SIZES:-253 2048;200 2048;127 4096;85 4096;48 4096}"
for rep in 1 2; do for T in "" $VARIANTS; do for cfg in "${CFGS[@]}"; do
  L=$PWD/socp_amd/_build${T:+_$T}/libsocp_hip.so; [ -f $L ] || continue
  echo "build_${T:-product} | $cfg | $(SOCP_MEASURE_ONLY=fast SOCP_LIB_PATH=$L python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done; done
