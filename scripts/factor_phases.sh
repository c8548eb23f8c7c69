#!/bin/bash
# Clock ticks per phase of the matrix-core factorisation (on the GPU box, through gpurun):  bash scripts/factor_phases.sh <tag> [n] [count]
# A -DSOCP_FACTOR_PROFILE build (lane 0 of wave 0 adds the ticks between marks to per-phase totals), then the product build restored.
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
TAG=${1:-r05}; N=${2:-253}; COUNT=${3:-2048}
OUT=gpurun_out; mkdir -p $OUT
touch socp_amd/csrc/kernels_factor_fast.hip
make -s -C socp_amd/csrc FACTOR_DEFS="-DSOCP_FACTOR_PROFILE $FACTOR_EXTRA" > /dev/null 2>&1
SOCP_MULTISTART_TRACE=1 SOCP_MEASURE_ONLY=fast python3 scripts/measure_factor.py $N $COUNT 3 2>&1 | grep -E "clock ticks|inside wave|kernel_ms" | cut -c1-700 > $OUT/${TAG}_factor_phases.txt
cat $OUT/${TAG}_factor_phases.txt
touch socp_amd/csrc/kernels_factor_fast.hip
make -s -C socp_amd/csrc FACTOR_DEFS="$FACTOR_EXTRA" > /dev/null 2>&1
