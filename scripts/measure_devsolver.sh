#!/bin/bash
# host vs device solvers on the sweeps VERDICT r2 #2 names; run on the GPU box from the repo root:  bash scripts/measure_devsolver.sh
export SOCP_MULTISTART_TRACE=1
out=gpurun_out/r03_devsolver
mkdir -p $out
for solver in host device; do
  python -m socp_amd.sweep --starts 4096 --segments 6 --rk4-steps 10000 --solver $solver > $out/sweep_4096_M6_$solver.json 2> $out/sweep_4096_M6_$solver.trace
  python -m socp_amd.sweep --starts 4096 --segments 9 --rk4-steps 10000 --solver $solver > $out/sweep_4096_M9_$solver.json 2> $out/sweep_4096_M9_$solver.trace
  python -m socp_amd.sweep --model interceptor --starts 2048 --solver $solver > $out/sweep_interceptor_2048_$solver.json 2> $out/sweep_interceptor_2048_$solver.trace
  python -m socp_amd.sweep --model interceptor --starts 256 --solver $solver > $out/sweep_interceptor_256_$solver.json 2> $out/sweep_interceptor_256_$solver.trace
  python -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver $solver > $out/chains_kd_4096_N10_$solver.json 2> $out/chains_kd_4096_N10_$solver.trace
done
for f in $out/*.json; do echo "$f: $(python -c "import json,sys; r=json.load(open('$f')); print(r['wall_s'], r['converged'], r.get('rounds_rank0'), r.get('solution_spread_rel'))")"; done
grep -h "total" $out/*.trace | grep -v round
