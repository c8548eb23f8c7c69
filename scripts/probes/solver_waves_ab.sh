#!/bin/bash
# A/B on the GPU box: the advance kernels built for W wavefronts per SIMD (-DSOCP_SOLVER_WAVES=W: registers capped, the rest spilled).
# Every variant is built in its own directory (scripts/variant_build.sh) and selected with SOCP_LIB_PATH; W = 0 is the product library.
cd "$(dirname "$0")/../.."
one() { tag=$1; shift; python3 -m socp_amd.sweep "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', '|', '$*', '|', round(r['wall_s'],4), r['converged'])"; }
for W in 0 4 5 6; do
  if [ $W = 0 ]; then unset SOCP_LIB_PATH; else LIB=$(bash scripts/variant_build.sh waves_$W SOLVER_DEFS=-DSOCP_SOLVER_WAVES=$W) || continue; export SOCP_LIB_PATH=$LIB; fi
  for rep in 1 2; do
    one "waves=$W" --model interceptor --starts 2048 --solver device_fast
    one "waves=$W" --model interceptor --starts 16384 --solver device_fast
    one "waves=$W" --starts 4096 --continuation kd --rk4-steps 10 --solver device_fast
    one "waves=$W" --starts 16384 --continuation kd --rk4-steps 10 --solver device_fast
    one "waves=$W" --starts 4096 --segments 9 --rk4-steps 10 --solver device_fast
  done
done
