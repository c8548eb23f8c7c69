"""GPU: covid19 (SEIR) device model -- IEEE +,-,*,/ only, so everything is bit-identical to the CPU path:
model evaluations and 1000-step segments against vectors produced by the reference's own object, the
n = 160 residual / FD Jacobian against the oracle, and the whole testCovid19 Newton history."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import Oracle, Problem, MODEL_COVID, FIXED, FREE, CONTINUOUS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFV = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
PARAMS = [3.4, 14, 5, 1, 0.1, 1, -10, 20]


@pytest.fixture(scope="module")
def cctx():
    from socp_amd import capi
    c = capi.Context(capi.MODEL_COVID19)
    c.set_params(PARAMS)
    yield c
    c.close()


def test_model_against_reference_vectors(cctx):
    from socp_amd import capi
    X = REFV["c_X"]
    assert np.array_equal(cctx.eval_batch(capi.EVAL_RHS, 0.0, X), REFV["c_rhs"])
    assert np.array_equal(cctx.eval_batch(capi.EVAL_CONTROL, 0.0, X), REFV["c_ctl"])
    assert np.array_equal(cctx.eval_batch(capi.EVAL_HAMILTONIAN, 0.0, X)[:, 0], REFV["c_ham"])
    assert np.array_equal(cctx.integrate_batch(0.0, 1.5, REFV["c_traj_X0"]), REFV["c_traj"])
    from socp_amd import capi as _c
    cctx.set_variant(_c.VARIANT_LANE_FAST)          # same IEEE operations under contraction: rounding-level agreement
    Xf = cctx.integrate_batch(0.0, 1.5, REFV["c_traj_X0"])
    cctx.set_variant(_c.VARIANT_AUTO)
    assert np.max(np.abs(Xf - REFV["c_traj"])) <= 1e-12


def test_residual_and_fd_jacobian_n160(cctx, built):
    o = Oracle(MODEL_COVID, params=PARAMS)
    M, d = 20, 4
    mode_t = [FIXED] + [CONTINUOUS] * (M - 1) + [FIXED]
    mode_x = np.full((M + 1, d), CONTINUOUS, dtype=np.int32)
    mode_x[0] = FIXED
    mode_x[M] = [FREE, FREE, FREE, FIXED]
    Xi = np.array([0.93, 0.003, 0.01, 0.057, -0.001, 0.001, 0.0, 0.0])
    time = np.array([30.0 * i / M for i in range(M + 1)])
    X = np.zeros((M + 1, 8))
    X[0] = Xi
    X[M, 3] = 0.6
    for i in range(1, M):
        X[i] = o.traj(0.0, Xi, time[i])
    prob = Problem(d, mode_t, mode_x, time, X)
    z = X[:M].ravel().copy()
    assert cctx.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode) == prob.n == 160
    F = cctx.residual(z)
    assert np.array_equal(F, o.residual(prob, z))
    J = cctx.fd_jacobian(z, F, dedup=True)
    assert np.array_equal(J, cctx.fd_jacobian(z, F, dedup=False))
    assert np.array_equal(J, o.fdjac(prob, z, F))


def test_covid_program_matches_cpu_history():
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "covid_flow.json")))
    exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "covid_flow")
    out = subprocess.run([exe, "1e-8", "3"], capture_output=True, text=True, timeout=900)
    stages = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(stages) == len(gold), out.stderr
    for s, g in zip(stages, gold):
        assert (s["stage"], s["info"], s["nfev"]) == (g["stage"], g["info"], g["nfev"])
        assert np.max(np.abs(np.array(s["z"]) - np.array(g["z"]))) <= 1e-13 * np.max(np.abs(g["z"]))
