#!/bin/bash
# The sweeps the device-solver work is measured on (interceptor n = 253 x 2048 / 256, Goddard M = 6 / M = 9 x 4096, KD chains at 10 and
# 10^4 steps), under one of the A/B axes the solver has had -- one script instead of the six near-copies of round 3.  On the GPU
# box, from the repo root (through gpurun):
#   bash scripts/measure_devsolver.sh <outdir> solvers [host device device_fast ...]   socp_chain_options.solver values side by side
#   bash scripts/measure_devsolver.sh <outdir> lds 0 70000 150000                      SOCP_SOLVER_LDS_BYTES (factor work on an LDS copy)
#   bash scripts/measure_devsolver.sh <outdir> threads "0 0" "0 64" "128 64"           SOCP_SOLVER_THREADS_FACTOR / _TRIAL per problem
#   bash scripts/measure_devsolver.sh <outdir> builds a.so b.so                        library builds (each becomes SOCP_LIB_PATH)
# Results: gpurun_out/<outdir>/<sweep>_<variant>.json (+ the engine's trace line); one summary line per run on stdout.
export SOCP_MULTISTART_TRACE=1
out=gpurun_out/$1; axis=${2:-solvers}; shift; shift
mkdir -p $out
[ $# -eq 0 ] && set -- host device
run() {
  tag=$1; shift
  "$@" > $out/$tag.json 2> $out/$tag.trace
  echo "$tag: $(python3 -c "import json; r=json.load(open('$out/$tag.json')); print(round(r['wall_s'],4), r['converged'], r.get('rounds_rank0'))") | $(grep 'set-up [0-9]' $out/$tag.trace | tail -1 | sed 's/.*set-up/set-up/' | cut -c1-160)"
  rm -f $out/$tag.trace
}
sweeps() {   # $1 = variant tag, $2 = --solver value
  run int2048_$1 python3 -m socp_amd.sweep --model interceptor --starts 2048 --solver $2
  run int256_$1  python3 -m socp_amd.sweep --model interceptor --starts 256 --solver $2
  run M6_$1      python3 -m socp_amd.sweep --starts 4096 --segments 6 --rk4-steps 10000 --solver $2
  run M9_$1      python3 -m socp_amd.sweep --starts 4096 --segments 9 --rk4-steps 10000 --solver $2
  run kd_$1      python3 -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver $2
  run kd1e4_$1   python3 -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10000 --solver $2
}
for v in "$@"; do
  case $axis in
    solvers) sweeps $v $v ;;
    lds)     SOCP_SOLVER_LDS_BYTES=$v sweeps lds$v device ;;
    threads) set -- $v; SOCP_SOLVER_THREADS_FACTOR=$1 SOCP_SOLVER_THREADS_TRIAL=$2 sweeps f$1_t$2 device ;;
    builds)  SOCP_LIB_PATH=$PWD/$v sweeps $(basename $v .so) device ;;
    *) echo "unknown axis $axis"; exit 64 ;;
  esac
done
