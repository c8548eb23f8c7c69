#!/bin/bash
# Per-phase clock totals of the device solver's kernels in the sweeps it is measured on (on the GPU box, through gpurun):
#   bash scripts/solver_phases.sh <tag> [solver = device_fast]
# A -DSOCP_SOLVER_PROFILE build (thread 0 of every workgroup adds the ticks between marks to per-phase totals) in ITS OWN DIRECTORY
# (scripts/variant_build.sh -> socp_amd/_build_prof_solver), selected with SOCP_LIB_PATH: the product library in socp_amd/_build is
# never touched, so a run killed at a time limit cannot leave a profiling build behind for later steps.
# -> gpurun_out/<tag>_solver_phases.txt
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
TAG=${1:-r06}; SOLVER=${2:-device_fast}
OUT=gpurun_out; mkdir -p $OUT
LIB=$(bash scripts/variant_build.sh prof_solver SOLVER_DEFS="-DSOCP_SOLVER_PROFILE $SOLVER_EXTRA") || { echo "solver_phases.sh: the profile build failed"; exit 1; }
export SOCP_LIB_PATH=$LIB
export SOCP_MULTISTART_TRACE=1
{
for w in "--model interceptor --starts 2048" "--starts 4096 --continuation kd --rk4-steps 10" "--starts 4096 --segments 9 --rk4-steps 10"; do
  echo "== $w --solver $SOLVER"
  timeout -k 10 120 python3 -m socp_amd.sweep $w --solver $SOLVER 2>&1 >/dev/null | grep -E "solver phases|set-up|inside the" | sed 's/.*(a -DSOCP_SOLVER_PROFILE build): //; s/\[socp_chains\/device\] //' | cut -c1-900
done
} > $OUT/${TAG}_solver_phases.txt
cat $OUT/${TAG}_solver_phases.txt
