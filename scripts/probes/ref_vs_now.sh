# same-box A/B of a reference build of the library (REF, default socp_amd/_build_head: `git archive <rev> socp_amd/csrc include | tar -x -C
# gpurun_out/headsrc; make -C gpurun_out/headsrc/socp_amd/csrc OUT=$PWD/socp_amd/_build_head`) against the product build: device-solver sweeps
REF=${REF:-_build_head}
python3 -m socp_amd.sweep --model interceptor --starts 2048 --solver device_fast > /dev/null 2>&1      # (the box's first large allocation)
for rep in 1 2; do
for B in $REF _build; do
  export SOCP_LIB_PATH=$PWD/socp_amd/$B/libsocp_hip.so
  for w in "--model interceptor --starts 2048 --solver device_fast" "--model interceptor --starts 16384 --solver device_fast" "--model interceptor --starts 2048 --solver device" "--starts 4096 --continuation kd --rk4-steps 10 --solver device_fast" "--starts 4096 --continuation kd --rk4-steps 10 --solver device" "--starts 4096 --segments 9 --rk4-steps 10 --solver device_fast" "--starts 4096 --segments 6 --rk4-steps 10 --solver device_fast" "--starts 4096 --segments 6 --rk4-steps 10 --solver device"; do
    echo "$B | $w | $(for r in 1 2 3; do python3 -m socp_amd.sweep $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['wall_s'],4), d['converged'], d.get('rounds_rank0'), end='  ')"; done)"
  done
done
done
