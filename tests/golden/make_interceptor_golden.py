#!/usr/bin/env python3
"""Frozen outputs of the interceptor restatement (oracle/interceptor_oracle.c).

PARITY UNPINNED: the reference's interceptor.cpp cannot be compiled in this image (it includes Eigen/Dense) and
the reference ships no output of it, so these vectors come from the CPU restatement, not from the reference.
They freeze the restatement (a later edit that changes its results fails tests/test_oracle_interceptor.py) and
give the GPU flow test its expected solutions.  What ties them to the reference: the restated test program
(tests/flow_oracle.py: interceptor_flow = tests/testInterceptor.cpp) converges with info = 1 in all three
scenarios from the reference's own analytical guess, as the reference's test expects ("OK = 1").

  interceptor_vectors.npz : Model/Control/Hamiltonian at 12 states x (chart 1|2) x (stage 1|0), chart changes,
                            five ComputeTraj end states (incl. chart switches and both stages), residuals
  interceptor_flow.json   : per scenario and xtol, the stages of the test program: info, nfev, z
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import Oracle, MODEL_INTERCEPTOR  # noqa: E402
from flow_oracle import interceptor_flow  # noqa: E402
from test_gpu_interceptor import scenario_state, states_both_charts, single_shooting_problem, multi_shooting_problem  # noqa: E402

o = Oracle(MODEL_INTERCEPTOR)
out = {}
X1, X2 = states_both_charts(o, 12)
out["X1"], out["X2"] = X1, X2
for chart, X in ((1, X1), (2, X2)):
    for stage, t in ((1, 3.0), (0, 27.0)):
        o.set_flags(chart, stage)
        key = "c%d_s%d" % (chart, stage)
        out["rhs_" + key] = np.array([o.rhs(t, x) for x in X])
        out["ctl_" + key] = np.array([o.control(t, x) for x in X])
        out["ham_" + key] = np.array([o.hamiltonian(t, x)[0] for x in X])
out["chart21_of_X2"] = np.array([o.chart21(x) for x in X2])
X0, _ = scenario_state()
Xs, _ = scenario_state(gamma=1.49)
cases = [(0.0, 10.0, X0), (0.0, 30.0, X0), (22.0, 31.0, X0), (0.0, 6.0, Xs), (0.0, 25.0, Xs)]
out["traj_t0"] = np.array([c[0] for c in cases])
out["traj_tf"] = np.array([c[1] for c in cases])
out["traj_X0"] = np.array([c[2] for c in cases])
res = [(o.traj(a, x, e), o.flags()) for a, e, x in cases]
out["traj_Xf"] = np.array([r[0] for r in res])
out["traj_flags"] = np.array([r[1] for r in res])          # (chart, stage) left behind
o.set_param("mu_gft", 0.6)
for M in (1, 4):
    prob, z = single_shooting_problem(o) if M == 1 else multi_shooting_problem(o, M)
    out["res_z_M%d" % M] = z
    out["res_F_M%d" % M] = o.residual(prob, z)
np.savez_compressed(os.path.join(HERE, "interceptor_vectors.npz"), **out)

flows = {}
for sc in (1, 2, 3):
    for xtol in (1e-8, 1e-12):
        st = interceptor_flow("scipy", xtol, sc)
        flows["scenario%d_xtol%g" % (sc, xtol)] = [dict(stage=s["stage"], info=int(s["info"]), nfev=int(s["nfev"]),
                                                        z=[float(v) for v in s["z"]]) for s in st]
        print(sc, xtol, [(s["stage"], s["info"], s["nfev"]) for s in st])
json.dump(flows, open(os.path.join(HERE, "interceptor_flow.json"), "w"), indent=0)
