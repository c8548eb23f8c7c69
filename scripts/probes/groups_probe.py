"""Probe: does splitting a lock-step sweep into G groups ON ONE GPU (socp_sweep_solve with G contexts on device 0, one host
thread each) hide one group's solver rounds under the other's trajectory launches?  Prints wall per G for three workloads."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from socp_amd import capi, sweep  # noqa: E402


class SweepStats(C.Structure):
    _fields_ = [("ndev", C.c_int), ("wall_ms", C.c_double), ("device_wall_ms", C.c_double * 16), ("device_rounds", C.c_longlong * 16),
                ("trajectories", C.c_longlong)]


def run(ctx, G, Z0, opt, params=None, goal=None):
    L = capi.lib()
    P, n = Z0.shape
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int)
    L.socp_sweep_solve.restype = C.c_int
    L.socp_sweep_solve.argtypes = [C.c_void_p, ip, C.c_int, C.c_int, C.c_void_p] + [dp] * 7 + [dp] + [ip] * 4 + [dp] * 3 + [C.c_void_p]
    devs = (C.c_int * G)(*([0] * G))
    Z = np.empty_like(Z0)
    info = np.zeros(P, np.int32); nl = np.zeros(P, np.int32); nt = np.zeros(P, np.int32); sv = np.zeros(P, np.int32)
    b = np.zeros(P); pf = np.zeros(P); fn = np.zeros(P)
    st = SweepStats()
    d = lambda a: None if a is None else a.ctypes.data_as(dp)  # noqa: E731
    i = lambda a: a.ctypes.data_as(ip)  # noqa: E731
    t0 = time.perf_counter()
    rc = L.socp_sweep_solve(ctx.h, devs, G, P, C.byref(opt), d(Z0), d(params), d(goal), None, None, None, None, d(Z), i(info), i(nl), i(nt),
                            i(sv), d(b), d(pf), d(fn), C.byref(st))
    wall = time.perf_counter() - t0
    assert rc == 0, rc
    return wall, dict(z=Z, info=info, nfev=nl)


def main():
    out = {}
    starts = int(os.environ.get("PROBE_STARTS", "4096"))
    steps = int(os.environ.get("PROBE_STEPS", "10000"))
    groups = [int(g) for g in os.environ.get("PROBE_GROUPS", "1,2,3,4").split(",")]
    for name in os.environ.get("PROBE_WORK", "m9,m6,kd").split(","):
        ctx = capi.Context(capi.MODEL_GODDARD)
        ctx.set_params(sweep.GODDARD_PARAMS)
        ctx.set_step_number(steps)
        ctx.set_variant(capi.VARIANT_LANE_FAST)
        params = goal = None
        opt = capi.ChainOptions(capi.CHAIN_PLAIN, 0, 1.0, 1e-12, 1e-8, 10000, 1e-15, 1.0, 1, -1, 0, 0, capi.SOLVER_AUTO)
        if name in ("m9", "m6"):
            M = 9 if name == "m9" else 6
            sweep.goddard_multiple_shooting_problem(ctx, M)
            Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(starts, 1e-3), M)
        else:
            gold = json.load(open(os.path.join(os.path.dirname(capi.__file__), "..", "tests", "golden", "goddard_flow.json")))
            z_nd = np.array([g for g in gold["goddard_single_stage"] if g["stage"] == 2 and g["xtol"] == 1e-6][0]["init_z"])
            sweep.goddard_multiple_shooting_problem(ctx, 6, tf=z_nd[-1])
            S = sweep.goddard_starts(starts, 1e-3)
            xi = (S[:, 7:] / sweep.PSTAR - 1.0) / 1e-3
            Z0 = np.tile(z_nd, (starts, 1))
            Z0[:, 7:14] *= 1.0 + 1e-3 * xi
            params = np.tile(np.array(sweep.GODDARD_PARAMS), (starts, 1))
            params[:, 2] = 0.0
            goal = np.full(starts, 310.0)
            opt.kind = capi.CHAIN_PARAM
            opt.param_index = 2
        Z0 = np.ascontiguousarray(Z0)
        ref = None
        for G in groups:
            best = None
            for _rep in range(3):
                wall, r = run(ctx, G, Z0, opt, params, goal)
                best = wall if best is None else min(best, wall)
            if ref is None:
                ref = r
            same = bool(np.array_equal(ref["z"], r["z"]) and np.array_equal(ref["nfev"], r["nfev"]))
            out.setdefault(name, {})[G] = dict(wall_s=round(best, 4), same_as_one_group=same, converged=int(np.sum(r["info"] == 1)))
            print(name, G, out[name][G], flush=True)
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
