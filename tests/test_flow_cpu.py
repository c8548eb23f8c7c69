"""CPU: the reference's testGoddard flow on the oracle.  Pins the oracle's shooting layer and the
library's hybrd against the record of the real reference run (residual evaluations per
stage 1186 / 190 / 640 / 103 through scipy.fsolve, i.e. 1184 / 188 / 638 / 101 raw MINPACK nfev;
stage-3 costates of the user manual / survey)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
from flow_oracle import goddard_test_flow  # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "goddard_flow.json")))["goddard_N10_M6"]


def test_goddard_flow_with_library_hybrd(built):
    stages = goddard_test_flow("socp")
    assert [s["info"] for s in stages] == [1, 1, 1, 1]
    assert [s["nfev"] for s in stages] == [1184, 188, 638, 101]
    for s, g in zip(stages, GOLD):
        assert np.max(np.abs(s["z"] - np.array(g["z"]))) <= 1e-12 * np.max(np.abs(g["z"]))


def test_golden_matches_survey_record():
    # converged stage-3 costates and tf of the reference run (7 significant digits)
    z = np.array(GOLD[2]["z"])
    ref = np.array([-7.101936, 0.006771117, 0.6791204, -0.3064235, 0.0004630762, 0.04640836, 0.06019189])
    assert np.allclose(z[7:14], ref, rtol=2e-6, atol=0)
    assert abs(z[-1] - 0.2310855) <= 1e-7
    # stage 2 (KD = 310, mu2 = 1): the costates the benchmark batch is centred on
    z2 = np.array(GOLD[1]["z"])
    p = np.array([-8.121947733, 7.775439382e-3, 0.7775438809, -0.4779369965, 5.715013318e-4, 5.715009222e-2, 9.958404873e-2])
    assert np.allclose(z2[7:14], p, rtol=1e-9, atol=0)


def test_double_integrator_flows_match_survey_record(built):
    """testDoubleIntegrator (nfev, njev) = (32,4), (14,1), (127,2); testDoubleIntegrator_WP first
    solve ier = 4, then (60,5), (115,4) -- through scipy.fsolve, whose wrapper adds two calls."""
    from flow_oracle import dint_basic_flow, dint_wp_flow
    for solver in ("scipy", "socp"):
        b = dint_basic_flow(solver, 1)
        assert [(s["info"], s["nfev"] + 2, s["njev"]) for s in b] == [(1, 32, 4), (1, 14, 1), (1, 127, 2)]
        w = dint_wp_flow(solver, 1)
        assert w[0]["info"] == 4
        assert [(s["info"], s["nfev"] + 2, s["njev"]) for s in w[1:]] == [(1, 60, 5), (1, 115, 4)]
