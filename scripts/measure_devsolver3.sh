#!/bin/bash
export SOCP_MULTISTART_TRACE=1
out=gpurun_out/$1; mkdir -p $out
run() { tag=$1; shift; "$@" > $out/$tag.json 2> $out/$tag.trace; echo "$tag: $(python -c "import json; r=json.load(open('$out/$tag.json')); print(round(r['wall_s'],4), r['converged'])") | $(grep 'set-up' $out/$tag.trace | sed 's/.*set-up/set-up/' | cut -c1-110)"; }
run int2048 python -m socp_amd.sweep --model interceptor --starts 2048 --solver device
run int256 python -m socp_amd.sweep --model interceptor --starts 256 --solver device
run M9 python -m socp_amd.sweep --starts 4096 --segments 9 --rk4-steps 10000 --solver device
run kd python -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver device
