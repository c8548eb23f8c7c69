#!/bin/bash
# Profile the bench workload on the GPU box (run through gpurun): kernel stats and, in SEPARATE passes (never combined with a
# trace domain), the PMC counters; summaries go to profiles/<round>_<variant>_p13107_* via scripts/summarize_prof.py.
#   scripts/profile_bench.sh [round-tag, default r02]
export TMPDIR=/tmp
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r02}
for v in fast exact; do
  st=3; [ $v = exact ] && st=2
  # --lean: only the headline flavour's timed region in the profiled process (the other legs have their own kernels)
  B="python3 bench.py --variant $v --steps $st --warmup 1 --cpu-seconds 0 --lean"
  rm -rf gpurun_out/p_${v}_*
  # the same command UN-profiled first, same box, same process order: its HIP-event kernel time is what the profiled average is to be
  # read against (a profiled run holds lower clocks: MI355X_MICROARCH.md, DVFS give-back item 2)
  $B > gpurun_out/p_${v}_unprofiled.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_${v}_stats -- $B > gpurun_out/p_${v}_bench.json 2>gpurun_out/p_${v}.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_${v}_fetch -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_${v}_write -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/p_${v}_sq -- $B > /dev/null 2>&1
  python3 scripts/summarize_prof.py ${TAG}_${v}_p13107 gpurun_out/p_${v}_stats gpurun_out/p_${v}_fetch gpurun_out/p_${v}_write gpurun_out/p_${v}_sq variant=$v starts=13107 rk4_steps=10000 source=profiles/${TAG}_${v}_p13107_pmc.json unprofiled=gpurun_out/p_${v}_unprofiled.json profiled=gpurun_out/p_${v}_bench.json
  cp gpurun_out/p_${v}_bench.json gpurun_out/${TAG}_${v}_p13107_bench.json
  mkdir -p gpurun_out/profiles_${TAG}; cp profiles/${TAG}_${v}_p13107_* profiles/traffic_latest.json gpurun_out/profiles_${TAG}/ 2>/dev/null || true
  echo done $v
done
