#!/bin/bash
# Per-phase clock totals of the device solver (a -DSOCP_SOLVER_PROFILE build of the library must be in place):
#   bash scripts/probes/solver_phases.sh [<library.so>]
[ -n "$1" ] && cp "$1" socp_amd/_build/libsocp_hip.so
export SOCP_MULTISTART_TRACE=1
for w in "--starts 4096 --continuation kd --rk4-steps 10" "--starts 4096 --segments 9 --rk4-steps 10000" "--model interceptor --starts 2048" "--model interceptor --starts 256"; do
  echo "== $w"
  timeout -k 10 120 python3 -m socp_amd.sweep $w --solver device 2>&1 >/dev/null | grep -E "solver phases|set-up|inside the factor" | sed 's/.*(a -DSOCP_SOLVER_PROFILE build): //; s/\[socp_chains\/device\] //'
done
