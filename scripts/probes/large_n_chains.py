"""Probe: host vs device solvers for FEW LARGE problems (doubleIntegrator way-points, M = 64: n = 832, hybrj with the batched
variational Jacobian): where does the AUTO rule's P n^2 >= 1.6e6 (P >= 3 at n = 832) stand?"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from socp_amd import capi  # noqa: E402


def main():
    M = 64
    ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    mode_t = [capi.FIXED] + [capi.FREE] * M
    mode_x = np.zeros((M + 1, 6), dtype=np.int32)
    mode_x[1:M, 3:6] = capi.CONTINUOUS
    X = np.zeros((M + 1, 12))
    X[:, 0] = 20.0 * np.arange(M + 1) / M
    X[:M, 6:] = 0.001
    tn = 60.0 * np.arange(M + 1) / M
    n = ctx.problem_set(mode_t, mode_x, tn, X)
    z = np.concatenate([X[:M].ravel(), tn[1:]])
    rng = np.random.default_rng(1)
    out = {"n": n}
    ctx.aux_stream()
    for P in [int(p) for p in os.environ.get("PROBE_P", "1,2,4,16,64").split(",")]:
        Z0 = np.tile(z, (P, 1))
        Z0[:, 6:12] *= 1 + 0.1 * rng.uniform(-1, 1, (P, 6))
        rec = {}
        for name, solver in (("host", capi.SOLVER_HOST), ("device", capi.SOLVER_DEVICE)):
            best = None
            for _rep in range(2):
                t0 = time.perf_counter()
                r = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-8, analytic_jac=True, max_rounds=40, solver=solver)
                w = time.perf_counter() - t0
                best = w if best is None else min(best, w)
            rec[name] = dict(wall_s=round(best, 4), rounds=int(r["stats"]["rounds"]), converged=int(np.sum(r["info"] == 1)), z=r["z"])
        same = bool(np.array_equal(rec["host"].pop("z"), rec["device"].pop("z")))
        out[P] = dict(rec, same_iterates=same)
        print(P, out[P], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
