# same-box A/B of the round-4 library (socp_amd/_build_r04: `git worktree add /tmp/r04tree 3b5c044; make -C /tmp/r04tree/socp_amd/csrc OUT=$PWD/socp_amd/_build_r04`)
# against the product build: the device-solver sweeps and the Jacobian refresh alone
python3 -m socp_amd.sweep --model interceptor --starts 2048 --solver device_fast > /dev/null 2>&1      # (the box's first large allocation)
for B in _build_r04 _build; do
  export SOCP_LIB_PATH=$PWD/socp_amd/$B/libsocp_hip.so
  for w in "--model interceptor --starts 2048" "--model interceptor --starts 16384" "--starts 4096 --continuation kd --rk4-steps 10" "--starts 4096 --segments 9 --rk4-steps 10" "--starts 4096 --segments 6 --rk4-steps 10" "--starts 4096 --segments 9 --rk4-steps 10000"; do
    for S in device_fast device; do
    echo "$B $S | $w | $(for rep in 1 2 3; do python3 -m socp_amd.sweep $w --solver $S 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['wall_s'],4), d['converged'], d.get('rounds_rank0'), end='  ')"; done)"
    done
  done
  for cfg in "253 2048" "200 2048" "127 4096" "85 4096" "64 4096" "48 4096"; do
    echo "$B factor | $cfg | $(SOCP_MEASURE_ONLY=fast python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
  done
done
