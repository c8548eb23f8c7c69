"""Probe: what do the interceptor's per-problem-block adaptive kernels (<InterceptorT, 1, 1, true>: 512 VGPRs, 100-164 B of scratch
per lane) cost against the shared-parameter ones (scratch-free)?  The SAME 256-start config-5 sweep (n = 253, Dormand-Prince) run
once as a plain multi-start (shared parameters) and once as parameter chains whose goal is the parameter's current value (one
solve each, identical arithmetic, but every launch through the per-problem instantiations)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from socp_amd import capi, sweep  # noqa: E402


def main():
    P = int(os.environ.get("PROBE_STARTS", "256"))
    ctx = capi.Context(capi.MODEL_INTERCEPTOR)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    n, z = sweep.interceptor_config5_problem(ctx)
    ctx.set_integrator(capi.INT_DOPRI5, 1e-8)
    rng = np.random.default_rng(5)
    Z0 = np.tile(z, (P, 1))
    Z0[:, 6:12] *= 1 + 1e-3 * rng.uniform(-1, 1, (P, 6))
    ctx.warm_up()
    MU = capi.INTERCEPTOR_PARAM_NAMES.index("mu_gft")
    p0 = np.array(ctx.get_params())
    for name, kw in (("shared", dict(kind=capi.CHAIN_PLAIN)),
                     ("per-problem", dict(kind=capi.CHAIN_PARAM, param_index=MU, step=1.0, goal=np.full(P, p0[MU]), params=np.tile(p0, (P, 1))))):
        ctx.chains_solve(Z0[:8], xtol=1e-8, solver=capi.SOLVER_DEVICE, **{k: (v[:8] if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
        best = None
        for _rep in range(3):
            t0 = time.perf_counter()
            r = ctx.chains_solve(Z0, xtol=1e-8, solver=capi.SOLVER_DEVICE, **kw)
            w = time.perf_counter() - t0
            best = w if best is None else min(best, w)
        print(name, "wall %.4f s" % best, "converged", int(np.sum(r["info"] == 1)), "rounds", r["stats"]["rounds"], flush=True)


if __name__ == "__main__":
    main()
