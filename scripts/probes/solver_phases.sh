#!/bin/bash
# Per-phase clock totals of the device solver with the bit-equal solvers (a -DSOCP_SOLVER_PROFILE build of the library, selected with
# SOCP_LIB_PATH -- scripts/variant_build.sh builds one; never copied over the product library):
#   bash scripts/probes/solver_phases.sh [<library.so>]
cd "$(dirname "$0")/../.."
export SOCP_LIB_PATH=${1:-$(bash scripts/variant_build.sh prof_solver SOLVER_DEFS=-DSOCP_SOLVER_PROFILE)} || exit 1
export SOCP_MULTISTART_TRACE=1
for w in "--starts 4096 --continuation kd --rk4-steps 10" "--starts 4096 --segments 9 --rk4-steps 10000" "--model interceptor --starts 2048" "--model interceptor --starts 256"; do
  echo "== $w"
  timeout -k 10 120 python3 -m socp_amd.sweep $w --solver device 2>&1 >/dev/null | grep -E "solver phases|set-up|inside the factor" | sed 's/.*(a -DSOCP_SOLVER_PROFILE build): //; s/\[socp_chains\/device\] //'
done
