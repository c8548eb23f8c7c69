# A/B: Q kept as factorised (SOCP_SOLVER_LAZY_Q=1) or updated eagerly (0) for the problem sizes below the default threshold (n >= 192)
python3 -m socp_amd.sweep --starts 4096 --segments 9 --rk4-steps 10 --solver device_fast > /dev/null 2>&1
for rep in 1 2; do for R in 0 1; do
  for w in "--starts 4096 --segments 9 --rk4-steps 10" "--starts 4096 --segments 6 --rk4-steps 10" "--starts 4096 --continuation kd --rk4-steps 10" "--starts 4096 --segments 9 --rk4-steps 10000"; do
    echo "lazy=$R | $w | $(for r in 1 2 3; do SOCP_SOLVER_LAZY_Q=$R python3 -m socp_amd.sweep $w --solver device_fast 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['wall_s'],4), d['converged'], d.get('rounds_rank0'), end='  ')"; done)"
  done
done; done
