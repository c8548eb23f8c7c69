#!/bin/bash
# kernel-trace stats of the solver-bound sweeps (on the GPU box): TAG=<round> bash scripts/profile_sweeps.sh -> gpurun_out/<TAG>_devsolver_fast_{c5,kd,m9}_kernel_stats.csv
export TMPDIR=/tmp
cd /root/repo 2>/dev/null || cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ks_*
for W in "c5:--model interceptor --starts 2048 --solver device_fast" "kd:--starts 4096 --continuation kd --rk4-steps 10 --solver device_fast" "m9:--starts 4096 --segments 9 --rk4-steps 10 --solver device_fast"; do
  tag=${W%%:*}; args=${W#*:}
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -- python3 -m socp_amd.sweep $args > /dev/null 2>&1
  f=$(find gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/${TAG:-r06}_devsolver_fast_${tag}_kernel_stats.csv && head -12 $f | cut -c1-200
done
