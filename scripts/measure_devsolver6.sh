#!/bin/bash
# threads per problem of the device solver's launches: bash scripts/measure_devsolver6.sh <outdir>
export SOCP_MULTISTART_TRACE=1
out=gpurun_out/$1; shift; mkdir -p $out
run() { tag=$1; shift; "$@" > $out/$tag.json 2> $out/$tag.trace; echo "$tag: $(python3 -c "import json; r=json.load(open('$out/$tag.json')); print(round(r['wall_s'],4), r['converged'])") | $(grep 'set-up [0-9]' $out/$tag.trace | tail -1 | sed 's/.*set-up/set-up/' | cut -c1-140)"; }
for cfg in "0 0" "0 64" "0 128" "64 64" "128 128" "128 64"; do
  set -- $cfg
  export SOCP_SOLVER_THREADS_FACTOR=$1 SOCP_SOLVER_THREADS_TRIAL=$2
  v=f$1_t$2
  run kd_$v python3 -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver device
  run M9_$v python3 -m socp_amd.sweep --starts 4096 --segments 9 --rk4-steps 10000 --solver device
  run int2048_$v python3 -m socp_amd.sweep --model interceptor --starts 2048 --solver device
  run int256_$v python3 -m socp_amd.sweep --model interceptor --starts 256 --solver device
done
