"""GPU: variational (hybrj) path of the double integrator -- one wavefront per augmented trajectory,
analytic shooting Jacobian assembled on the device.  The path contains only IEEE +,-,*,/ and sqrt,
so the comparison with the CPU oracle is bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import Oracle, Problem, MODEL_DINT, FIXED, FREE, CONTINUOUS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "dint_flow.json")))
REFV = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


@pytest.fixture(scope="module")
def dctx():
    from socp_amd import capi
    c = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    yield c
    c.close()


@pytest.fixture(scope="module")
def doracle(built):
    return Oracle(MODEL_DINT)


def test_state_and_augmented_rhs_against_reference_vectors(dctx):
    from socp_amd import capi
    X = REFV["d_X"]
    assert np.array_equal(dctx.eval_batch(capi.EVAL_RHS, 0.0, X), REFV["d_rhs"])
    assert np.array_equal(dctx.eval_batch(capi.EVAL_CONTROL, 0.0, X), REFV["d_ctl"])
    assert np.array_equal(dctx.eval_batch(capi.EVAL_HAMILTONIAN, 0.0, X)[:, 0], REFV["d_ham"])
    assert np.array_equal(dctx.eval_batch(capi.EVAL_HAMILTONIAN, 0.0, X, is_jac=1), REFV["d_dham"])
    got = dctx.eval_batch(capi.EVAL_RHS, 0.0, REFV["d_Xaug"], is_jac=1)
    assert np.array_equal(got, REFV["d_rhs_aug"])          # -0.0 == 0.0: the reference sums explicit zeros


def test_segments_against_reference_vectors(dctx):
    Xg = dctx.integrate_batch(0.0, 7.5, REFV["d_traj_X0"])
    assert np.array_equal(Xg, REFV["d_traj"])
    Xa = dctx.integrate_batch(0.0, 7.5, REFV["d_traj_aug_X0"], is_jac=1)
    assert np.array_equal(Xa, REFV["d_traj_aug"])


def _wp_problem(M):
    mode_t = [FIXED] + [FREE] * M
    mode_x = np.zeros((M + 1, 6), dtype=np.int32)
    mode_x[1:M, 3:6] = CONTINUOUS
    X = np.zeros((M + 1, 12))
    X[:, 0] = 20.0 * np.arange(M + 1) / M
    X[:M, 6:] = 0.001
    time = 60.0 * np.arange(M + 1) / M
    prob = Problem(6, mode_t, mode_x, time, X)
    z = np.concatenate([X[:M].ravel(), time[1:]])
    return prob, z


@pytest.mark.parametrize("M", [1, 2, 5, 64])
def test_variational_jacobian_bitwise(dctx, doracle, M):
    """M = 2 is testDoubleIntegrator_WP; M = 5 exercises the reference's column spill at FREE interior
    times (shooting.cpp:1070); M = 64 is BASELINE config 3 (n = 832)."""
    rng = np.random.default_rng(M)
    if M == 1:
        X = np.zeros((2, 12))
        X[0, 6:] = 0.01
        X[1, :3] = [10.0, 15.0, 0.0]
        prob = Problem(6, [FIXED, FREE], np.zeros((2, 6), dtype=np.int32), np.array([0.0, 10.0]), X)
        z = np.concatenate([X[0], [10.0]])
    else:
        prob, z = _wp_problem(M)
    z = z + 1e-3 * rng.uniform(-1, 1, prob.n)
    assert dctx.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode) == prob.n
    assert np.array_equal(dctx.residual(z), doracle.residual(prob, z))
    Jg = dctx.var_jacobian(z)
    Jc = doracle.jacobian(prob, z)
    assert np.array_equal(Jg, Jc)
    # the FD path of the same problem (north_star path, modelOrder = 0)
    F0 = dctx.residual(z)
    Jfd = dctx.fd_jacobian(z, F0, dedup=False)
    assert np.array_equal(Jfd, dctx.fd_jacobian(z, F0, dedup=True))
    assert np.array_equal(Jfd, doracle.fdjac(prob, z, F0))


def _run(args):
    exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "dint_flow")
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    return out.returncode, [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")], out.stderr


@pytest.mark.parametrize("flow", ["basic", "wp"])
@pytest.mark.parametrize("order", [1, 0])
def test_double_integrator_programs(flow, order):
    """The reference's two doubleIntegrator programs through the C++ mirror: no exp anywhere, so the
    whole Newton history (info, nfev, njev, solution) equals the CPU path's exactly."""
    rc, stages, err = _run([flow, order, 1e-8])
    gold = GOLD["%s_order%d_xtol1e-08" % (flow, order)]
    assert len(stages) == len(gold), err
    for s, g in zip(stages, gold):
        assert (s["info"], s["nfev"]) == (g["info"], g["nfev"]), s["stage"]
        if order == 1:
            assert s["njev"] == g["njev"]
        if g["info"] == 1:
            assert np.max(np.abs(np.array(s["z"]) - np.array(g["z"]))) <= 1e-13 * np.max(np.abs(g["z"]))


@pytest.mark.parametrize("order", [1, 0])
def test_config3_64_segments_history(order):
    """BASELINE config 3 (doubleIntegrator, 64 segments, n = 832), hybrj and FD/hybrd: from the WP-style
    guess the CPU path stalls with info = 5; the device path must report the same info, counts and iterate."""
    rc, stages, err = _run(["wp", order, 1e-8, 64])
    gold = GOLD["wp_M64_order%d_xtol1e-08" % order]
    assert len(stages) == 1 and stages[0]["n"] == 832, err
    s, g = stages[0], gold[0]
    assert (s["info"], s["nfev"]) == (g["info"], g["nfev"]) and (order == 0 or s["njev"] == g["njev"])
    assert np.max(np.abs(np.array(s["z"]) - np.array(g["z"]))) <= 1e-13 * np.max(np.abs(g["z"]))


@pytest.mark.parametrize("M", [7, 100])
def test_row_tiles_and_direct_stores_agree_with_cpu(doracle, M, monkeypatch):
    """The residual kernels write whole rows through an LDS tile when M <= 64 (several rows per workgroup, the
    last workgroup partly filled) and store directly when M > 64; `SOCP_ROW_TILES=0` forces the direct form.
    All of them must give the CPU path's bits, for batches that do not divide evenly."""
    from socp_amd import capi
    prob, z = _wp_problem(M)
    rng = np.random.default_rng(M)
    B = 11
    Z = z[None, :] + 1e-3 * rng.uniform(-1, 1, (B, prob.n))
    want = doracle.residual_batch(prob, Z)
    c = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    assert c.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode) == prob.n
    assert np.array_equal(c.residual_batch(Z), want)
    rows = c.fd_rows(Z[:2])
    assert np.array_equal(rows[:, 0, :], want[:2])
    eps = np.sqrt(1e-15)
    for j in (0, 5, prob.n - 1):
        zp = Z[1].copy()
        zp[j] += (eps * abs(zp[j])) or eps
        assert np.array_equal(rows[1, j + 1], doracle.residual(prob, zp)), j
    c.close()
    # the same through the direct-store form, in a fresh process (the switch is read once)
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from socp_amd import capi; "
            "d = np.load(sys.argv[1]); c = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR); "
            "c.problem_set(d['mt'], d['mx'], d['t'], d['x']); np.save(sys.argv[2], c.residual_batch(d['Z']))" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), mt=prob.mode_t, mx=prob.mode_x, t=prob.time, x=prob.xnode, Z=Z)
        subprocess.run(["python", "-c", code, os.path.join(td, "in.npz"), os.path.join(td, "out.npy")], check=True,
                       env=dict(os.environ, SOCP_ROW_TILES="0"), timeout=300)
        assert np.array_equal(np.load(os.path.join(td, "out.npy")), want)
