#!/bin/bash
# LDS-resident matrix for the factor work: SOCP_SOLVER_LDS_BYTES sweep
export SOCP_MULTISTART_TRACE=1
out=gpurun_out/$1; mkdir -p $out
run() { tag=$1; shift; "$@" > $out/$tag.json 2> $out/$tag.trace; echo "$tag: $(python -c "import json; r=json.load(open('$out/$tag.json')); print(round(r['wall_s'],4), r['converged'])") | $(grep 'set-up' $out/$tag.trace | sed 's/.*set-up/set-up/' | cut -c1-110)"; }
for lds in 0 70000 150000; do
  export SOCP_SOLVER_LDS_BYTES=$lds
  run kd_lds$lds python -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver device
  run M6_lds$lds python -m socp_amd.sweep --starts 4096 --segments 6 --rk4-steps 10000 --solver device
  run M9_lds$lds python -m socp_amd.sweep --starts 4096 --segments 9 --rk4-steps 10000 --solver device
done
