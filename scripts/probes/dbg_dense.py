import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from socp_amd import capi, sweep
X0 = np.concatenate([sweep.X0_STATE, sweep.PSTAR * (1 + 1e-3 * np.array([0.3, -0.2, 0.5, 0.1, -0.7, 0.9, -0.4]))])
for var in (capi.VARIANT_LANE_EXACT, capi.VARIANT_LANE_FAST):
    ctx = capi.Context(capi.MODEL_GODDARD); ctx.set_params(sweep.GODDARD_PARAMS); ctx.set_step_number(10); ctx.set_variant(var)
    t, d = ctx.integrate_dense(0.0, sweep.TF, X0); p = ctx.integrate_batch(0.0, sweep.TF, X0[None])[0]
    print("fixed", var, len(t), np.max(np.abs(d[-1] - p)))
    for tol in (1e-6, 1e-9):
        ctx.set_integrator(capi.INT_DOPRI5, tol)
        t, d = ctx.integrate_dense(0.0, sweep.TF, X0); p = ctx.integrate_batch(0.0, sweep.TF, X0[None])[0]
        p64 = ctx.integrate_batch(0.0, sweep.TF, np.repeat(X0[None], 70, axis=0))
        print("adaptive", var, tol, len(t), np.max(np.abs(d[-1] - p)), np.max(np.abs(p64 - p[None])), t[:4], t[-3:])
    ctx.close()
