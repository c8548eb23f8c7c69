#!/usr/bin/env python3
"""covid19 workload of tests/testCovid19.cpp (M = 20 segments, n = 160 unknowns, 1000 RK4 steps per segment):
one FD Jacobian on the GPU (full and with the segment dedup) beside the CPU restatement on one core, and the
whole test program through the C++ mirror.  Run on the GPU box; prints one JSON object."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import Oracle, Problem, MODEL_COVID, FIXED, FREE, CONTINUOUS  # noqa: E402
from socp_amd import capi  # noqa: E402

PARAMS = [3.4, 14, 5, 1, 0.1, 1, -10, 20]
o = Oracle(MODEL_COVID, params=PARAMS)
M, d = 20, 4
mode_t = [FIXED] + [CONTINUOUS] * (M - 1) + [FIXED]
mode_x = np.full((M + 1, d), CONTINUOUS, dtype=np.int32)
mode_x[0] = FIXED
mode_x[M] = [FREE, FREE, FREE, FIXED]
Xi = np.array([0.93, 0.003, 0.01, 0.057, -0.001, 0.001, 0.0, 0.0])
tn = np.array([30.0 * i / M for i in range(M + 1)])
X = np.zeros((M + 1, 8))
X[0] = Xi
X[M, 3] = 0.6
for i in range(1, M):
    X[i] = o.traj(0.0, Xi, tn[i])
prob = Problem(d, mode_t, mode_x, tn, X)
z = X[:M].ravel().copy()
ctx = capi.Context(capi.MODEL_COVID19)
ctx.set_params(PARAMS)
n = ctx.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode)
F = ctx.residual(z)
out = {"n": n, "segments": M, "rk4_steps": 1000}
t = time.perf_counter()
Jo = o.fdjac(prob, z, o.residual(prob, z))
out["cpu_restatement_1core_fd_jacobian_ms"] = 1e3 * (time.perf_counter() - t)
for dd in (False, True):
    ctx.fd_jacobian(z, F, dedup=dd)
    c0 = ctx.counters()[0]
    t = time.perf_counter()
    for _ in range(5):
        J = ctx.fd_jacobian(z, F, dedup=dd)
    out["gpu_fd_jacobian_%s" % ("dedup" if dd else "full")] = {"ms": 1e3 * (time.perf_counter() - t) / 5,
                                                               "trajectories": int((ctx.counters()[0] - c0) // 5)}
out["bit_identical_to_cpu"] = bool(np.array_equal(J, Jo))
ctx.set_variant(capi.VARIANT_LANE_FAST)
Ff = ctx.residual(z)
ctx.fd_jacobian(z, Ff, dedup=True)
t = time.perf_counter()
for _ in range(5):
    ctx.fd_jacobian(z, Ff, dedup=True)
out["gpu_fd_jacobian_dedup_fast_flavour_ms"] = 1e3 * (time.perf_counter() - t) / 5
out["fast_flavour_residual_max_abs_diff"] = float(np.max(np.abs(Ff - F)))
ctx.set_variant(capi.VARIANT_AUTO)
exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "covid_flow")
t = time.perf_counter()
r = subprocess.run([exe, "1e-8", "3"], capture_output=True, text=True)
out["test_program_wall_s"] = time.perf_counter() - t
out["test_program_stages"] = [(s["stage"], s["info"], s["nfev"]) for s in (json.loads(l) for l in r.stdout.splitlines() if l.startswith("{"))]
print(json.dumps(out, indent=1))
