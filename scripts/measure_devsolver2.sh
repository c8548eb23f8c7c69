#!/bin/bash
# device-solver runs on the sweeps VERDICT r2 #2 names; run on the GPU box from the repo root
export SOCP_MULTISTART_TRACE=1
out=gpurun_out/${1:-r03_devsolver3}
mkdir -p $out
run() { tag=$1; shift; "$@" > $out/$tag.json 2> $out/$tag.trace; echo "$tag: $(python -c "import json; r=json.load(open('$out/$tag.json')); print(r['wall_s'], r['converged'])") | $(grep 'set-up' $out/$tag.trace | sed 's/.*set-up/set-up/' | cut -c1-150)"; }
run int2048 python -m socp_amd.sweep --model interceptor --starts 2048 --solver device
run int256 python -m socp_amd.sweep --model interceptor --starts 256 --solver device
run M6 python -m socp_amd.sweep --starts 4096 --segments 6 --rk4-steps 10000 --solver device
run M9 python -m socp_amd.sweep --starts 4096 --segments 9 --rk4-steps 10000 --solver device
run kd python -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver device
run kd1e4 python -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10000 --solver device
