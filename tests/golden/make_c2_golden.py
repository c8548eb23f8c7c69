"""Golden root of BASELINE config 2 / 4's problem (Goddard single shooting, n = 14, fixed tf, mu2 = 1, KD = 310, 1e4 RK4 steps),
computed on the CPU: the oracle's residual (oracle/socp_oracle.c, pinned against the reference's objects) driven by the
library's host hybrd (bit-equal to SciPy's MINPACK, tests/test_minpack.py) with the reference's knobs (shooting.cpp:95-105) at
xtol = 1e-12, from the benchmark's centre p* and from the first six synthetic starts.  Run in the authoring
container:  python tests/golden/make_c2_golden.py  ->  tests/golden/c2_root.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.oracle import Oracle, Problem, MODEL_GODDARD, FIXED, FREE   # noqa: E402
from socp_amd import capi, sweep                                         # noqa: E402


def main():
    o = Oracle(MODEL_GODDARD, step_nbr=10000, params=sweep.GODDARD_PARAMS)
    mode_x = np.zeros((2, 7), dtype=np.int32)
    mode_x[1, 3:7] = FREE
    X = np.zeros((2, 14))
    X[0, :7] = sweep.X0_STATE
    X[1, 0] = 1.01
    prob = Problem(7, [FIXED, FIXED], mode_x, np.array([0.0, sweep.TF]), X)
    z0 = np.concatenate([sweep.X0_STATE, sweep.PSTAR])
    centre = capi.hybrd(lambda v: o.residual(prob, v), z0, xtol=1e-12, epsfcn=1e-15)
    assert centre["info"] == 1
    starts = []
    Z = sweep.goddard_starts(6, 1e-3)
    for p in range(6):
        r = capi.hybrd(lambda v: o.residual(prob, v), Z[p], xtol=1e-12, epsfcn=1e-15)
        starts.append({"start": p, "info": int(r["info"]), "nfev": int(r["nfev"]), "z": [float(v) for v in r["x"]],
                       "fnorm": float(np.linalg.norm(r["fvec"]))})
    spread = max(np.max(np.abs(np.array(s["z"]) - centre["x"])) for s in starts) / np.max(np.abs(centre["x"]))
    out = {"problem": "goddard single shooting n=14, tf=%r fixed, params %r, 10000 RK4 steps" % (sweep.TF, sweep.GODDARD_PARAMS),
           "xtol": 1e-12, "epsfcn": 1e-15, "z": [float(v) for v in centre["x"]], "info": int(centre["info"]), "nfev": int(centre["nfev"]),
           "fnorm": float(np.linalg.norm(centre["fvec"])), "starts": starts, "spread_rel": float(spread)}
    with open(os.path.join(ROOT, "tests", "golden", "c2_root.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("root", centre["x"][7:], "spread over starts", spread)


if __name__ == "__main__":
    main()
