#!/usr/bin/env python3
"""Observed parity errors of the GPU flavours against the CPU oracle (run on the GPU box) -> DESIGN.md."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle, MODEL_GODDARD  # noqa: E402
from socp_amd import capi, sweep  # noqa: E402

out = {}
ctx = capi.Context(capi.MODEL_GODDARD)
ctx.set_params(sweep.GODDARD_PARAMS)
o = Oracle(MODEL_GODDARD, params=sweep.GODDARD_PARAMS)
X0 = sweep.goddard_starts(64, 1e-3)
for N in (10, 1000, 10000):
    ctx.set_step_number(N)
    o.m.step_nbr = N
    Xc = o.integrate_batch(0.0, sweep.TF, X0)
    for tag, v in (("exact", capi.VARIANT_LANE_EXACT), ("fast", capi.VARIANT_LANE_FAST)):
        ctx.set_variant(v)
        Xg = ctx.integrate_batch(0.0, sweep.TF, X0)
        out["%s_N%d" % (tag, N)] = float(np.max(np.abs(Xg - Xc)) / max(1.0, np.max(np.abs(Xc))))
ctx.close()
print(json.dumps(out, indent=1))
