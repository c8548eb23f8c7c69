#!/bin/bash
# workgroup size of the trial launches (SOCP_SOLVER_THREADS_TRIAL) on the config-5 sweeps, throughput flavour
cd "$(dirname "$0")/../.."
for rep in 1 2; do for t in 0 64 128 256; do for P in 2048 16384; do
  SOCP_SOLVER_THREADS_TRIAL=$t python3 -m socp_amd.sweep --model interceptor --starts $P --solver device_fast 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('trial_threads=$t starts=$P', round(r['wall_s'],4), r['converged'])"
done; done; done
