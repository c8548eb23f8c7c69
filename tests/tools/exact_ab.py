#!/usr/bin/env python3
"""A/B harness for changes that must not move a single bit of the reference-order (exact) flavour:
    exact_ab.py save <file.npz>      run a fixed set of evaluations / trajectories / residuals, store the outputs
    exact_ab.py check <file.npz>     run the same set and compare bit for bit with the stored outputs
Run on the GPU box (the stored file must travel inside the repository, e.g. tests/tools/_ab/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from socp_amd import capi, sweep  # noqa: E402


def collect():
    out = {}
    rng = np.random.default_rng(2024)
    B = 40000
    X = np.empty((B, 14))
    dirs = rng.normal(size=(B, 3))
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    X[:, 0:3] = dirs * rng.uniform(0.98, 1.05, (B, 1))
    X[:, 3:6] = rng.normal(size=(B, 3)) * 10.0 ** rng.uniform(-10, -0.5, (B, 1))
    X[:, 6] = rng.uniform(0.2, 1.0, B)
    X[:, 7:10] = rng.normal(size=(B, 3)) * 5
    X[:, 10:13] = rng.normal(size=(B, 3)) * 10.0 ** rng.uniform(-3, 0.5, (B, 1))
    X[:, 13] = rng.uniform(-0.5, 0.5, B)
    X[:50, 3:6] = 0.0                      # degenerate rows: zero speed, zero p_v, huge / tiny magnitudes
    X[50:100, 10:13] = 0.0
    X[100:150] *= 1e150
    X[150:200] *= 1e-150
    t = rng.uniform(0.0, 0.12, B)
    for mu2 in (1.0, 0.0):
        c = capi.Context(capi.MODEL_GODDARD)
        c.set_params([3.5, 7.0, 310.0, 500.0, 1.0, 1.0, mu2, -1.0])
        c.set_switching_times([0.02, 0.08])
        c.set_variant(capi.VARIANT_LANE_EXACT)
        out["g_rhs_%g" % mu2] = c.eval_batch(capi.EVAL_RHS, t, X)
        out["g_ctl_%g" % mu2] = c.eval_batch(capi.EVAL_CONTROL, t, X)
        out["g_ham_%g" % mu2] = c.eval_batch(capi.EVAL_HAMILTONIAN, t, X)
        c.set_step_number(200)
        Z = sweep.goddard_starts(4096, 1e-2)
        out["g_traj_%g" % mu2] = c.integrate_batch(0.0, 0.26, Z)
        c.set_integrator(capi.INT_DOPRI5, 1e-9)
        out["g_dopri_%g" % mu2] = c.integrate_batch(0.0, 0.26, Z[:512])
        c.set_integrator(capi.INT_RK4)
        if mu2 > 0:
            c.set_step_number(10)
            sweep.goddard_multiple_shooting_problem(c, 6)
            Zm = sweep.goddard_multiple_shooting_starts(c, sweep.goddard_starts(64, 0.05), 6)
            out["g_res"] = c.residual_batch(Zm)
            out["g_rows"] = c.fd_rows(Zm[:4])
        c.close()
    # covid19 and the double integrator
    c = capi.Context(capi.MODEL_COVID19)
    c.set_params([3.4, 14, 5, 1, 0.1, 1, -10, 20])
    Xc = np.abs(rng.normal(size=(5000, 8))) * [1, 0.01, 0.05, 0.1, 1, 1, 1, 1]
    Xc[:, 4:] = rng.normal(size=(5000, 4))
    out["c_rhs"] = c.eval_batch(capi.EVAL_RHS, 0.0, Xc)
    out["c_traj"] = c.integrate_batch(0.0, 3.0, Xc[:512])
    c.close()
    c = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    Xd = rng.normal(size=(5000, 12))
    out["d_rhs"] = c.eval_batch(capi.EVAL_RHS, 0.0, Xd)
    out["d_traj"] = c.integrate_batch(0.0, 7.0, Xd[:512])
    c.close()
    return out


mode, path = sys.argv[1], sys.argv[2]
got = collect()
if mode == "save":
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **got)
    print("saved", {k: v.shape for k, v in got.items()})
else:
    want = np.load(path)
    bad = 0
    for k, v in got.items():
        same = np.array_equal(v, want[k], equal_nan=True)
        nd = int(np.sum(~((v == want[k]) | (np.isnan(v) & np.isnan(want[k])))))
        print("%-12s %s  differing entries: %d of %d" % (k, "IDENTICAL" if same else "DIFFERENT", nd, v.size))
        bad += not same
    sys.exit(1 if bad else 0)
