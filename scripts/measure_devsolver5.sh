#!/bin/bash
# A/B of library builds for the device solver: bash scripts/measure_devsolver5.sh <outdir> <variant.so> [<variant.so> ...]
# (variants live in socp_amd/_build/variants/; each is copied over the library before its runs -- on the GPU box's scratch copy)
export SOCP_MULTISTART_TRACE=1
out=gpurun_out/$1; shift; mkdir -p $out
run() { tag=$1; shift; "$@" > $out/$tag.json 2> $out/$tag.trace; echo "$tag: $(python3 -c "import json; r=json.load(open('$out/$tag.json')); print(round(r['wall_s'],4), r['converged'])") | $(grep 'set-up [0-9]' $out/$tag.trace | tail -1 | sed 's/.*set-up/set-up/' | cut -c1-140)"; }
for so in "$@"; do
  v=$(basename $so .so)
  cp $so socp_amd/_build/libsocp_hip.so
  run kd_$v python3 -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver device
  run M6_$v python3 -m socp_amd.sweep --starts 4096 --segments 6 --rk4-steps 10000 --solver device
  run M9_$v python3 -m socp_amd.sweep --starts 4096 --segments 9 --rk4-steps 10000 --solver device
  run int2048_$v python3 -m socp_amd.sweep --model interceptor --starts 2048 --solver device
  run int256_$v python3 -m socp_amd.sweep --model interceptor --starts 256 --solver device
done
