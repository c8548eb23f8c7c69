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


# ---- the variational path under the ADAPTIVE integrator.  With -D_USE_BOOST the reference sends EVERY integrate() call
# through Dormand-Prince, the isJac = 1 trajectories of the hybrj path included (odeTools.cpp:129-134, model.hpp:395-414,
# shooting.cpp:996-1130).  Boost.Odeint is absent and the reference holds no vectors: PARITY UNPINNED, as for the state-only
# adaptive path -- the comparisons below are with the restatement (oracle/socp_oracle.c: orc_integrate_dopri5_jac) and with
# independent routes to the same numbers.

def test_adaptive_augmented_trajectories_against_the_restatement(doracle):
    """(i) is_jac = 1 segments under SOCP_INT_DOPRI5: one wavefront per augmented trajectory, per-wave step control, against the
    restatement's adaptive integration of the same 156-element state: <= 100 tol (observed: rounding level -- the per-element
    arithmetic is the same, only pow() in the step-size controller differs between device and host libm)."""
    from socp_amd import capi
    rng = np.random.default_rng(5)
    B = 9
    X0 = np.zeros((B, 156))
    X0[:, :12] = rng.uniform(-1, 1, (B, 12)) * np.array([5, 5, 5, 1, 1, 1, 0.5, 0.5, 0.5, 2, 2, 2])
    X0[:, 12:] = np.eye(12).ravel()[None, :]
    t0 = np.zeros(B)
    tf = np.array([7.5, 0.3, 20.0, 1.0, 3.0, 0.0, -1.0, 12.0, 60.0])     # incl. a zero-length and a backward segment: no step
    for tol in (1e-6, 1e-10):
        ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
        ctx.set_integrator(capi.INT_DOPRI5, tol)
        got = ctx.integrate_batch(t0, tf, X0, is_jac=1)
        ctx.close()
        doracle.set_integrator(1, tol)
        want = doracle.integrate_batch(t0, tf, X0, is_jac=1)
        doracle.set_integrator(0)
        scale = np.maximum(1.0, np.max(np.abs(want), axis=1, keepdims=True))
        assert np.max(np.abs(got - want) / scale) <= 100 * tol
        assert np.array_equal(got[5], X0[5]) and np.array_equal(got[6], X0[6])
    # ... and the adaptive result is the fine fixed-step one to its tolerance (so the adaptive kernel is what ran, and it integrates)
    ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    ctx.set_step_number(20000)
    fine = ctx.integrate_batch(t0[:5], tf[:5], X0[:5], is_jac=1)
    ctx.close()
    assert np.max(np.abs(got[:5] - fine) / np.maximum(1.0, np.max(np.abs(fine), axis=1, keepdims=True))) <= 1e-7


@pytest.mark.parametrize("M", [1, 2, 5])
def test_adaptive_variational_jacobian(dctx, doracle, M):
    """(ii) socp_var_jacobian under the adaptive integrator: against the restatement's hybrj Jacobian with the same integrator, and
    against a forward-difference Jacobian of adaptive STATE-ONLY trajectories at tol 1e-11 (an independent route: other kernels,
    no variational equations) to forward-difference accuracy."""
    from socp_amd import capi
    rng = np.random.default_rng(40 + M)
    if M == 1:
        X = np.zeros((2, 12))
        X[0, 6:] = 0.01
        X[1, :3] = [10.0, 15.0, 0.0]
        prob = Problem(6, [FIXED, FREE], np.zeros((2, 6), dtype=np.int32), np.array([0.0, 10.0]), X)
        z = np.concatenate([X[0], [10.0]])
    else:
        prob, z = _wp_problem(M)
    z = z + 1e-3 * rng.uniform(-1, 1, prob.n)
    tol = 1e-11
    ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    ctx.set_integrator(capi.INT_DOPRI5, tol)
    assert ctx.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode) == prob.n
    Jv = ctx.var_jacobian(z)
    F0 = ctx.residual(z)
    Jfd = ctx.fd_jacobian(z, F0, epsfcn=1e-12, dedup=True)
    ctx.close()
    doracle.set_integrator(1, tol)
    Jc = doracle.jacobian(prob, z)
    doracle.set_integrator(0)
    scale = np.max(np.abs(Jc))
    assert np.isfinite(Jv).all() and np.max(np.abs(Jv - Jc)) <= 100 * tol * scale
    # the reference's variational Jacobian leaves out a segment's dependence on its START time (shooting.cpp:996-1130 has d/dt_end
    # terms only): compare the columns of the state unknowns, where both routes compute the same thing
    # -- minus the columns into which the reference's copy loop drops a FREE interior node's time term one block too far
    # (shooting.cpp:1070, kept on purpose: column 12 (k + 1) for the node after next; present from M = 3 on)
    cols = np.array([c for c in range(12 * M) if not (c % 12 == 0 and c >= 24)])
    assert np.max(np.abs(Jv[:, cols] - Jfd[:, cols])) <= 2e-5 * scale


def test_double_integrator_program_with_adaptive_hybrj():
    """(iii) testDoubleIntegrator (modelOrder = 1: hybrj, variational Jacobians) through the C++ mirror with the adaptive integrator
    -- which the round-3 library refused (SOCP_ERR_UNSUPPORTED): all three solves converge, to the fixed-step roots within the
    discretisation difference (1e-6)."""
    exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "dint_flow")
    out = subprocess.run([exe, "basic", "1", "1e-8"], capture_output=True, text=True, timeout=600, env=dict(os.environ, SOCP_FLOW_ADAPTIVE="1"))
    stages = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    gold = GOLD["basic_order1_xtol1e-08"]
    assert out.returncode == 0 and len(stages) == len(gold) == 3, out.stderr[-2000:]
    for s, g in zip(stages, gold):
        assert s["info"] == 1 and g["info"] == 1 and s["njev"] >= 1
        assert np.max(np.abs(np.array(s["z"]) - np.array(g["z"]))) <= 1e-6 * np.max(np.abs(g["z"])), s["stage"]


def test_adaptive_hybrj_chains_device_and_host_solvers_agree():
    """analytic_jac chains (batched variational Jacobians, one wavefront per (chain, segment)) under the adaptive integrator: the
    lock-step engine with the solvers on the host and on the device, bit for bit, and every chain on the fixed-step root."""
    from socp_amd import capi
    prob, z = _wp_problem(2)
    ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    ctx.set_integrator(capi.INT_DOPRI5, 1e-10)
    assert ctx.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode) == prob.n
    rng = np.random.default_rng(3)
    Z0 = np.tile(z, (24, 1)) * (1.0 + 1e-3 * rng.uniform(-1, 1, (24, prob.n)))
    host = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10, analytic_jac=True, solver=capi.SOLVER_HOST)
    dev = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10, analytic_jac=True, solver=capi.SOLVER_DEVICE)
    for k in ("z", "info", "nfev", "njev"):
        assert np.array_equal(host[k], dev[k]), k
    assert np.all(host["njev"] >= 1)
    ctx.close()
