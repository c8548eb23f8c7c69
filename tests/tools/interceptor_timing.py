#!/usr/bin/env python3
"""BASELINE config 5 class: interceptor, M = 21 segments, n = 253 unknowns (final time + velocity free).
One forward-difference Jacobian = 254 residual rows x 21 segments = 5334 trajectories of 50 RK4 steps
(100 where a segment spans the burn-out time) -- GPU (fixed-step and adaptive Dormand-Prince, with and
without the segment dedup) beside the CPU restatement on one host core; then a batch of independent
problems for throughput.  Run on the GPU box; prints one JSON object (-> profiles/)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import Oracle, MODEL_INTERCEPTOR  # noqa: E402
from socp_amd import capi  # noqa: E402
from test_gpu_interceptor import multi_shooting_problem  # noqa: E402


def timeit(fn, reps=5):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps


o = Oracle(MODEL_INTERCEPTOR)
# nodes along the CONVERGED scenario-1 trajectory of the test program (tests/golden/interceptor_flow.json): the
# Jacobian a multiple-shooting solve of that scenario evaluates
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "interceptor_flow.json")))["scenario1_xtol1e-12"][-1]["z"]
RE = 6378145.0
Xf = np.zeros(12)
Xf[:6] = [12000, 1000, 0.0, np.pi / 8, 5475000 / RE, 42000 / RE]
prob, z = multi_shooting_problem(o, 21, tf=gold[12], X0=np.array(gold[:12]), Xf=Xf)
print("problem ready: n =", prob.n, file=sys.stderr, flush=True)
ctx = capi.Context(capi.MODEL_INTERCEPTOR)
n = ctx.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode)
F0 = ctx.residual(z)
out = {"n": n, "segments": 21}
t = time.perf_counter()
Jo = o.fdjac(prob, z, o.residual(prob, z))
out["cpu_restatement_1core"] = {"fd_jacobian_ms": 1e3 * (time.perf_counter() - t), "trajectories": (n + 1) * 21}
for name, kind, tol in (("rk4", capi.INT_RK4, 0.0), ("rk4_fast_flavour", capi.INT_RK4, 0.0), ("dopri5_tol1e-8", capi.INT_DOPRI5, 1e-8),
                        ("dopri5_tol1e-8_fast_flavour", capi.INT_DOPRI5, 1e-8)):
    ctx.set_variant(capi.VARIANT_LANE_FAST if name.endswith("fast_flavour") else capi.VARIANT_AUTO)
    if kind == capi.INT_DOPRI5:
        ctx.set_integrator(kind, tol)
    else:
        ctx.set_integrator(kind)
    F = ctx.residual(z)
    print(name, "residual max |F| =", float(np.max(np.abs(F))), file=sys.stderr, flush=True)
    r = {"max_abs_residual_at_converged_nodes": float(np.max(np.abs(F)))}
    for dd in (False, True):
        c0 = ctx.counters()[0]
        sec = timeit(lambda: ctx.fd_jacobian(z, F, dedup=dd), 5)
        r["dedup" if dd else "full"] = {"trajectories": int((ctx.counters()[0] - c0) // 6), "ms": 1e3 * sec}
    if kind == capi.INT_RK4:
        J = ctx.fd_jacobian(z, F, dedup=False)
        assert np.array_equal(J, ctx.fd_jacobian(z, F, dedup=True))
        r["max_abs_diff_vs_cpu_jacobian_entries_scaled"] = float(np.max(np.abs(J - Jo) / np.maximum(1.0, np.abs(Jo))))
    # throughput: P independent problems (FD rows of P perturbed starts) in one launch
    P = 64
    Z = z[None, :] * (1 + 1e-6 * np.random.default_rng(0).uniform(-1, 1, (P, n)))
    sec = timeit(lambda: ctx.fd_rows(Z), 3)
    r["batch_%d_problems" % P] = {"trajectories": P * (n + 1) * 21, "ms": 1e3 * sec, "traj_per_s": P * (n + 1) * 21 / sec}
    out[name] = r
print(json.dumps(out, indent=1))
