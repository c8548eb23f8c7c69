#!/bin/bash
# A/B on the GPU box: the matrix-core refresh of the product library against variant builds (socp_amd/_build_<name>; scripts/variant_build.sh):
#   bash scripts/probes/factor_chain_ab.sh [name ...]
cd "$(dirname "$0")/../.."
for T in "" "$@"; do for cfg in "253 2048" "200 2048" "127 4096" "85 4096" "48 4096"; do
  L=$PWD/socp_amd/_build${T:+_$T}/libsocp_hip.so; [ -f $L ] || continue
  echo "build_${T:-product} | $cfg | $(SOCP_MEASURE_ONLY=fast SOCP_LIB_PATH=$L python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done
