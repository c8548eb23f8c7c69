#!/bin/bash
# Host factor work of a Jacobian refresh on the GPU box's CPU: MINPACK's scalar column algorithm (1 / 16 threads) against the
# columns-in-SIMD-lanes form, bit-compared; then what it does to the programs that are host-bound: the doubleIntegrator M = 64
# (n = 832) program and the multiple-shooting sweeps (4096 solvers of n = 85 / 127).  Writes gpurun_out/r02_host_factor.json.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02_host_factor.json
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -O3 -std=c++17 -ffp-contract=off -Wno-psabi -Iinclude -o /tmp/qr_bench tests/tools/qr_bench.cpp -lpthread
{
echo '{"qr_bench": ['
for a in "85 1 20" "127 1 20" "832 1 3" "832 4 3" "832 16 3"; do /tmp/qr_bench $a; echo ','; done
SOCP_LINALG_TRACE=1 /tmp/qr_bench 832 1 1 2>&1 >/dev/null | tail -1 1>&2
SOCP_LINALG_TRACE=1 /tmp/qr_bench 832 16 1 2>&1 >/dev/null | tail -1 1>&2
echo 'null],'
for v in 0 1; do
  for order in 0 1; do
    echo "\"dint_M64_order${order}_vector${v}_wall_s\": $(SOCP_LINALG_VECTOR=$v python3 -c "
import subprocess, time
t = time.perf_counter(); subprocess.run(['socp_amd/_build/bin/dint_flow', 'wp', '$order', '1e-8', '64'], capture_output=True); print(time.perf_counter() - t)"),"
  done
done
for v in 0 1; do
  for M in 6 9; do
    echo "\"sweep_M${M}_vector${v}\": $(SOCP_LINALG_VECTOR=$v python -m socp_amd.sweep --starts 4096 --segments $M --variant fast),"
  done
done
echo '"cpu": "'$(grep -m1 "model name" /proc/cpuinfo | cut -d: -f2)'", "nproc": '$(nproc)'}'
} > $OUT
python3 -c "import json; d=json.load(open('$OUT')); print(json.dumps({k:(v if not isinstance(v,dict) else {kk:v[kk] for kk in ('wall_s','converged','rounds_rank0')}) for k,v in d.items()}, indent=1))"
