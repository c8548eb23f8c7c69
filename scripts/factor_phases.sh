#!/bin/bash
# Clock ticks per phase of the matrix-core factorisation (on the GPU box, through gpurun):  bash scripts/factor_phases.sh <tag> [n] [count]
# A -DSOCP_FACTOR_PROFILE build (lane 0 of wave 0 adds the ticks between marks to per-phase totals) in ITS OWN directory
# (scripts/variant_build.sh -> socp_amd/_build_prof_factor), selected with SOCP_LIB_PATH: the product library is never rebuilt in
# place.
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
TAG=${1:-r06}; N=${2:-253}; COUNT=${3:-2048}
OUT=gpurun_out; mkdir -p $OUT
LIB=$(bash scripts/variant_build.sh prof_factor FACTOR_DEFS="-DSOCP_FACTOR_PROFILE $FACTOR_EXTRA") || { echo "factor_phases.sh: the profile build failed"; exit 1; }
SOCP_LIB_PATH=$LIB SOCP_MULTISTART_TRACE=1 SOCP_MEASURE_ONLY=fast python3 scripts/measure_factor.py $N $COUNT 3 2>&1 | grep -E "clock ticks|inside wave|kernel_ms" | cut -c1-700 > $OUT/${TAG}_factor_phases.txt
cat $OUT/${TAG}_factor_phases.txt
