import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from socp_amd import capi
M = 64
ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
mode_t = [capi.FIXED] + [capi.FREE] * M
mode_x = np.zeros((M + 1, 6), dtype=np.int32)
mode_x[1:M, 3:6] = capi.CONTINUOUS
X = np.zeros((M + 1, 12)); X[:, 0] = 20.0 * np.arange(M + 1) / M; X[:M, 6:] = 0.001
tn = 60.0 * np.arange(M + 1) / M
n = ctx.problem_set(mode_t, mode_x, tn, X)
z = np.concatenate([X[:M].ravel(), tn[1:]])
rng = np.random.default_rng(1)
P = int(sys.argv[1])
Z0 = np.tile(z, (P, 1)); Z0[:, 6:12] *= 1 + 0.1 * rng.uniform(-1, 1, (P, 6))
ctx.aux_stream()
for solver in (capi.SOLVER_DEVICE, capi.SOLVER_DEVICE, capi.SOLVER_HOST):
    r = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-8, analytic_jac=True, max_rounds=40, solver=solver)
print(r["info"], r["nfev"], r["njev"])
