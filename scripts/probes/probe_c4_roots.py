import json, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from socp_amd import capi, sweep
gold = json.load(open("tests/golden/c2_root.json")); zg = np.array(gold["z"])
ctx = capi.Context(capi.MODEL_GODDARD); ctx.set_params(sweep.GODDARD_PARAMS); ctx.set_step_number(10000)
ctx.set_variant(capi.VARIANT_LANE_FAST); sweep.goddard_single_shooting_problem(ctx)
Z0 = sweep.goddard_starts(4096, 1e-3)
for xtol in (1e-8, 1e-10, 1e-12):
    out = ctx.chains_solve(Z0, kind=0, xtol=xtol)
    ok = out["info"] == 1
    err = np.max(np.abs(out["z"] - zg[None, :]), axis=1) / np.max(np.abs(zg))
    print("xtol", xtol, "conv", ok.sum(), "rounds", out["stats"]["rounds"], "wall", out["stats"]["wall_ms"], np.unique(out["info"], return_counts=True))
    bad = np.where(ok & (err > 1e-8))[0]
    print(" outliers", len(bad), [(int(p), float(err[p]), float(out["fnorm"][p]), int(out["nfev"][p])) for p in bad[:20]])
    print(" max fnorm of converged", out["fnorm"][ok].max(), "err quantiles", np.quantile(err[ok], [0.5, 0.99, 1.0]))
