"""Which starts of the 4096-start config-4 sweep end with info = 1 although |F| did not vanish (MINPACK's delta <= xtol |x| exit)?
Both flavours; for the reference-order flavour the same starts solved on the CPU (oracle residual + this library's hybrd) beside.
Writes gpurun_out/c4_outliers.json."""
import json, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from socp_amd import capi, sweep
from oracle import oracle as orc
res = {}
Z0 = sweep.goddard_starts(4096, 1e-3)
for name, var in (("fast", capi.VARIANT_LANE_FAST), ("exact", capi.VARIANT_LANE_EXACT)):
    ctx = capi.Context(capi.MODEL_GODDARD); ctx.set_params(sweep.GODDARD_PARAMS); ctx.set_step_number(10000)
    ctx.set_variant(var); sweep.goddard_single_shooting_problem(ctx)
    out = ctx.chains_solve(Z0, kind=0, xtol=1e-12)
    ok = out["info"] == 1
    bad = np.where(ok & (out["fnorm"] > 1e-9))[0]
    res[name] = {"converged": int(ok.sum()), "outliers": [{"start": int(p), "fnorm": float(out["fnorm"][p]), "nfev": int(out["nfev"][p]),
                                                              "z": out["z"][p].tolist()} for p in bad]}
    print(name, ok.sum(), [(int(p), float(out["fnorm"][p]), int(out["nfev"][p])) for p in bad], flush=True)
    ctx.close()
o = orc.Oracle(orc.MODEL_GODDARD, step_nbr=10000, params=sweep.GODDARD_PARAMS)
mode_x = np.zeros((2, 7), dtype=np.int32); mode_x[1, 3:7] = orc.FREE
X = np.zeros((2, 14)); X[0, :7] = sweep.X0_STATE; X[1, 0] = 1.01
prob = orc.Problem(7, [orc.FIXED, orc.FIXED], mode_x, np.array([0.0, sweep.TF]), X)
cpu = []
for rec in (res["exact"]["outliers"] + res["fast"]["outliers"])[:4]:
    p = rec["start"]
    r = capi.hybrd(lambda v: o.residual(prob, v), Z0[p], xtol=1e-12, epsfcn=1e-15)
    f = float(np.linalg.norm(o.residual(prob, r["x"])))
    cpu.append({"start": p, "info": int(r["info"]), "nfev": int(r["nfev"]), "fnorm": f, "z": np.asarray(r["x"]).tolist()})
    print("cpu", p, r["info"], r["nfev"], f, flush=True)
res["cpu"] = cpu
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/c4_outliers.json", "w"), indent=1)
