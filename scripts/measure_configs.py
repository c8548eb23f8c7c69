#!/usr/bin/env python3
"""Measure the BASELINE configurations that are not the bench line (run on the GPU box):
  C2   Goddard single shooting n = 14: one problem = 15 trajectories per Jacobian (latency-bound)
  C128 Goddard M = 9, FREE tf + one FREE interior time -> n = 128: Jacobian batch at fixed z,
       129 residual rows x 9 segments, with and without the segment dedup
  C3   doubleIntegrator M = 64 way-points (n = 832): FD Jacobian batch and variational Jacobian
Prints one JSON object; numbers go into DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socp_amd import capi, sweep  # noqa: E402


def timeit(fn, reps=5):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps


def goddard_multi(ctx, M, free_interior):
    d, s = 7, 14
    mode_t = [capi.FIXED] + [capi.CONTINUOUS] * (M - 1) + [capi.FREE]
    if free_interior:
        mode_t[M // 2] = capi.FREE
    mode_x = np.full((M + 1, d), capi.CONTINUOUS, dtype=np.int32)
    mode_x[0] = capi.FIXED
    mode_x[M] = capi.FIXED
    mode_x[M, 3:7] = capi.FREE
    time_nodes = np.linspace(0.0, sweep.TF, M + 1)
    X = np.zeros((M + 1, s))
    X[0] = np.concatenate([sweep.X0_STATE, sweep.PSTAR])
    X[M, 0] = 1.01
    # node states along the converged single-shooting trajectory (integrated on the device)
    for i in range(1, M):
        X[i] = ctx.integrate_batch(0.0, time_nodes[i], X[0][None, :])[0]
    n = ctx.problem_set(mode_t, mode_x, time_nodes, X)
    z = np.concatenate([X[:M].ravel(), [time_nodes[j] for j in range(M + 1) if mode_t[j] == capi.FREE]])
    assert len(z) == n
    return n, z


def main():
    out = {}
    for variant, tag in ((capi.VARIANT_LANE_EXACT, "exact"), (capi.VARIANT_LANE_FAST, "fast")):
        ctx = capi.Context(capi.MODEL_GODDARD)
        ctx.set_params(sweep.GODDARD_PARAMS)
        ctx.set_step_number(10000)
        ctx.set_variant(variant)
        # C2
        sweep.goddard_single_shooting_problem(ctx)
        z = np.concatenate([sweep.X0_STATE, sweep.PSTAR])
        F0 = ctx.residual(z)
        sec = timeit(lambda: ctx.fd_rows(z[None, :]), 3)
        out["C2_%s" % tag] = {"trajectories": 15, "ms": 1e3 * sec, "traj_per_s": 15 / sec}
        # C128
        n, z = goddard_multi(ctx, 9, True)
        F0 = ctx.residual(z)
        c0 = ctx.counters()[0]
        sec_full = timeit(lambda: ctx.fd_jacobian(z, F0, dedup=False), 3)
        per_full = (ctx.counters()[0] - c0) // 4
        c0 = ctx.counters()[0]
        sec_ded = timeit(lambda: ctx.fd_jacobian(z, F0, dedup=True), 3)
        per_ded = (ctx.counters()[0] - c0) // 4
        out["C128_%s" % tag] = {"n": n, "segments": 9, "trajectories_full": int(per_full), "ms_full": 1e3 * sec_full,
                                "traj_per_s_full": per_full / sec_full, "trajectories_dedup": int(per_ded),
                                "ms_dedup": 1e3 * sec_ded, "jacobians_per_s_dedup": 1 / sec_ded}
        ctx.close()
    # C3
    ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    M = 64
    mode_t = [capi.FIXED] + [capi.FREE] * M
    mode_x = np.zeros((M + 1, 6), dtype=np.int32)
    mode_x[1:M, 3:6] = capi.CONTINUOUS
    X = np.zeros((M + 1, 12))
    X[:, 0] = 20.0 * np.arange(M + 1) / M
    X[:M, 6:] = 0.001
    tn = 60.0 * np.arange(M + 1) / M
    n = ctx.problem_set(mode_t, mode_x, tn, X)
    z = np.concatenate([X[:M].ravel(), tn[1:]])
    F0 = ctx.residual(z)
    c0 = ctx.counters()[0]
    sec_fd = timeit(lambda: ctx.fd_jacobian(z, F0, dedup=False), 3)
    per_fd = (ctx.counters()[0] - c0) // 4
    c0 = ctx.counters()[0]
    sec_dd = timeit(lambda: ctx.fd_jacobian(z, F0, dedup=True), 3)
    per_dd = (ctx.counters()[0] - c0) // 4
    sec_var = timeit(lambda: ctx.var_jacobian(z), 3)
    out["C3_dint_M64"] = {"n": n, "fd_trajectories_full": int(per_fd), "fd_ms_full": 1e3 * sec_fd,
                          "fd_trajectories_dedup": int(per_dd), "fd_ms_dedup": 1e3 * sec_dd,
                          "variational_trajectories": M, "variational_ms": 1e3 * sec_var}
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
