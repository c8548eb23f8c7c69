# A/B of the throughput flavour's parallel back-substitution sums on the n = 85 sweeps (default: from n = 112 up)
python3 -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver device_fast > /dev/null 2>&1
for F in 0 1; do for w in "--starts 4096 --continuation kd --rk4-steps 10" "--starts 4096 --segments 6 --rk4-steps 10" "--starts 16384 --continuation kd --rk4-steps 10"; do
echo "fast_sums=$F | $w | $(for rep in 1 2 3; do SOCP_SOLVER_FAST_SUMS=$F python3 -m socp_amd.sweep $w --solver device_fast 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['wall_s'],4), d['converged'], d.get('rounds_rank0'), end='  ')"; done)"
done; done
