#!/bin/bash
# TIMING probes of the matrix-core qrfac launch (on the GPU box): what its two halves would cost as launches of their own.
#   probe1 = the trailing passes [E] alone, probe2 = the panel phases [A] / [C] alone (results of both are garbage, only the time counts);
#   variants built beforehand:  bash scripts/variant_build.sh probe1 FACTOR_DEFS=-DSOCP_FACTOR_PROBE=1   (and probe2)
cd "$(dirname "$0")/../.."
for T in "" _probe1 _probe2; do for cfg in "253 2048" "127 4096" "85 4096"; do
  L=$PWD/socp_amd/_build$T/libsocp_hip.so; [ -f $L ] || continue
  echo "build${T:-_product} | $cfg | $(SOCP_MEASURE_ONLY=fast SOCP_LIB_PATH=$L python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done
