#!/bin/bash
# A/B on the GPU box: loads issued ahead of the serial recurrences (SOCP_SOLVER_BATCH / _BATCH3) now that the trial launches are built
# for 128 registers.  Every variant is built in its own directory (scripts/variant_build.sh) and selected with SOCP_LIB_PATH.
cd "$(dirname "$0")/../.."
one() { tag=$1; shift; python3 -m socp_amd.sweep "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', '|', '$*', '|', round(r['wall_s'],4), r['converged'])"; }
for B in "8 6" "4 4" "6 4" "8 4" "12 6"; do
  set -- $B
  LIB=$(bash scripts/variant_build.sh batch_$1_$2 SOLVER_DEFS="-DSOCP_SOLVER_BATCH=$1 -DSOCP_SOLVER_BATCH3=$2") || continue
  export SOCP_LIB_PATH=$LIB
  for rep in 1 2; do
    for s in device device_fast; do
      one "batch=$1/$2 $s" --model interceptor --starts 2048 --solver $s
      one "batch=$1/$2 $s" --starts 4096 --continuation kd --rk4-steps 10 --solver $s
      one "batch=$1/$2 $s" --starts 4096 --segments 9 --rk4-steps 10 --solver $s
    done
  done
done
