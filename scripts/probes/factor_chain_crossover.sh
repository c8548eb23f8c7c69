#!/bin/bash
# Where the chain of launches starts to pay: the refresh (qrfac + qform) of `count` problems with the chain forced on (SOCP_FACTOR_CHAIN_MIN=0)
# and off (a huge minimum), ms, HIP events, average of 3.   bash scripts/probes/factor_chain_crossover.sh
cd "$(dirname "$0")/../.."
for n in 85 127 253; do for count in 128 256 512 768 1024 1536 2048 4096; do
  a=$(SOCP_FACTOR_CHAIN_MIN=0 SOCP_MEASURE_ONLY=fast python3 scripts/measure_factor.py $n $count 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")
  b=$(SOCP_FACTOR_CHAIN_MIN=99999999 SOCP_MEASURE_ONLY=fast python3 scripts/measure_factor.py $n $count 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")
  echo "n = $n  count = $count  chain $a  single launch $b"
done; done
