#!/usr/bin/env python3
"""Golden converged solutions of the reference's test programs: the reference's flow restated over
the CPU oracle (tests/flow_oracle.py), Newton solve by SciPy's MINPACK (the independent solver the
survey used against the real reference residual).  The per-stage nfev equal the
survey's record of the reference run (1186/190/640/103 through fsolve = 1184/188/638/101 raw).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from flow_oracle import goddard_test_flow  # noqa: E402

out = {"goddard_N10_M6": [dict(stage=s["stage"], info=int(s["info"]), nfev=int(s["nfev"]), z=[float(v) for v in s["z"]])
                           for s in goddard_test_flow("scipy", 10, 6)]}
json.dump(out, open(os.path.join(HERE, "goddard_flow.json"), "w"), indent=0)
print({k: [(s["stage"], s["info"], s["nfev"]) for s in v] for k, v in out.items()})
