#!/usr/bin/env python3
"""Randomised parity stress (run on the GPU box): Goddard Model / Control / Hamiltonian at many random states over
wide ranges, every control-law branch, both arithmetic flavours, against the CPU oracle; and short trajectories.
Prints the worst relative deviations (-> DESIGN.md 5)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, MODEL_GODDARD  # noqa: E402
from socp_amd import capi  # noqa: E402

rng = np.random.default_rng(12345)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
out = {"states_per_case": B}
for mu2, label in ((1.0, "smooth_law"), (0.0, "bang_singular_off")):
    params = [3.5, 7.0, 310.0, 500.0, 1.0, 1.0, mu2, -1.0]
    o = Oracle(MODEL_GODDARD, params=params)
    o.set_switching([0.02, 0.08])
    X = np.empty((B, 14))
    dirs = rng.normal(size=(B, 3))
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    X[:, 0:3] = dirs * rng.uniform(0.98, 1.05, (B, 1))
    X[:, 3:6] = rng.normal(size=(B, 3)) * 10.0 ** rng.uniform(-10, -0.5, (B, 1))
    X[:, 6] = rng.uniform(0.2, 1.0, B)
    X[:, 7:10] = rng.normal(size=(B, 3)) * 5
    X[:, 10:13] = rng.normal(size=(B, 3)) * 10.0 ** rng.uniform(-3, 0.5, (B, 1))
    X[:, 13] = rng.uniform(-0.5, 0.5, B)
    t = rng.uniform(0.0, 0.12, B)
    ref_f = np.array([o.rhs(t[b], X[b]) for b in range(B)])
    ref_u = np.array([o.control(t[b], X[b]) for b in range(B)])
    ref_h = np.array([o.hamiltonian(t[b], X[b])[0] for b in range(B)])
    for variant, tag in ((capi.VARIANT_LANE_EXACT, "exact"), (capi.VARIANT_LANE_FAST, "fast")):
        c = capi.Context(capi.MODEL_GODDARD)
        c.set_params(params)
        c.set_switching_times([0.02, 0.08])
        c.set_variant(variant)
        f = c.eval_batch(capi.EVAL_RHS, t, X)
        u = c.eval_batch(capi.EVAL_CONTROL, t, X)
        h = c.eval_batch(capi.EVAL_HAMILTONIAN, t, X)[:, 0]
        scale = np.maximum(np.abs(ref_f).max(axis=1, keepdims=True) * 1e-3, np.abs(ref_f))
        out["%s_%s" % (label, tag)] = {
            "rhs_max_rel": float(np.nanmax(np.abs(f - ref_f) / scale)),
            "rhs_bitwise_equal_fraction": float(np.mean(np.all(f == ref_f, axis=1))),
            "control_max_abs": float(np.nanmax(np.abs(u - ref_u))),
            "hamiltonian_max_rel": float(np.nanmax(np.abs(h - ref_h) / np.maximum(1.0, np.abs(ref_h)))),
            "non_finite_rows_gpu_vs_cpu": [int(np.sum(~np.isfinite(f).all(axis=1))), int(np.sum(~np.isfinite(ref_f).all(axis=1)))]}
        c.close()
print(json.dumps(out, indent=1))
