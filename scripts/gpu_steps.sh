#!/bin/bash
# Run GPU steps one after the other on the GPU box (through gpurun):  bash scripts/gpu_steps.sh <tag> "<cmd 1>" "<cmd 2>" ...
# Each step runs under its own time limit (STEP_TIMEOUT seconds, default 500) with stdout/stderr in gpurun_out/<tag>_<k>.log.
# An ordinary failure (tests red, non-zero exit) does not stop the later steps; a step that is KILLED at its limit does --
# a hung GPU step says something is wrong with the card, and nothing more is started on it.
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TAG=$1; shift
k=0
for cmd in "$@"; do
  k=$((k+1))
  log=gpurun_out/${TAG}_${k}.log
  echo "[$(date +%T)] step $k: $cmd"
  timeout -k 10 ${STEP_TIMEOUT:-500} bash -c "$cmd" > $log 2>&1
  rc=$?
  echo "[$(date +%T)] step $k rc=$rc  $(tail -n 1 $log | cut -c1-200)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $k hit its time limit: stopping"; exit 1; fi
done
exit 0
