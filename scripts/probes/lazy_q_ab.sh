#!/bin/bash
# A/B of Config::lazy_q (Q kept as factorised in the throughput flavour of the device solver) on the solver-bound sweeps:
#   old = a build of the commit before (socp_amd/_build_old/libsocp_hip.so, when present), lazy_q=0 / 1 = this build
cd "$(dirname "$0")/../.."
one() { tag=$1; shift; python3 -m socp_amd.sweep "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', '|', '$*', '|', round(r['wall_s'],4), r['converged'], r.get('rounds_rank0'), r.get('mean_nfev'))"; }
for rep in 1 2; do
  for args in "--model interceptor --starts 2048" "--model interceptor --starts 16384" "--starts 4096 --continuation kd --rk4-steps 10" "--starts 4096 --segments 9 --rk4-steps 10" "--starts 4096 --segments 6 --rk4-steps 10"; do
    for s in device device_fast; do
      [ -f socp_amd/_build_old/libsocp_hip.so ] && SOCP_LIB_PATH=$PWD/socp_amd/_build_old/libsocp_hip.so one "old    $s" $args --solver $s
      SOCP_SOLVER_LAZY_Q=0 one "lazy_q=0 $s" $args --solver $s
      [ $s = device_fast ] && SOCP_SOLVER_LAZY_Q=1 one "lazy_q=1 $s" $args --solver $s
    done
  done
done
