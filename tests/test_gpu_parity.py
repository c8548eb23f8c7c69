"""GPU parity: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (SURVEY 8d "Parity tolerance"):
  * reference-order variant, N = 10 / 30 steps: |Xf_gpu - Xf_cpu|_inf <= 1e-10 * max(1, |Xf|_inf)
    (observed: a few ulp -- the only rounding that differs from x86 is inside exp);
  * N = 1e4 steps: <= 1e-8 on the benchmark distribution;
  * residual rows: same bound scaled by the row magnitude.
"""
import numpy as np
import pytest

from conftest import (GODDARD_TF, goddard_c1_problem, goddard_costate_batch, goddard_single_problem)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gctx():
    from socp_amd import capi
    c = capi.Context(capi.MODEL_GODDARD)
    yield c
    c.close()


@pytest.fixture(scope="module")
def goracle(built):
    from oracle.oracle import Oracle, MODEL_GODDARD
    return Oracle(MODEL_GODDARD)


def relerr(a, b):
    return np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))


@pytest.mark.parametrize("mu2", [1.0, 0.2, 0.0])
def test_eval_rhs_control_hamiltonian(gctx, goracle, mu2):
    from socp_amd import capi
    rng = np.random.default_rng(3)
    B = 200
    X = goddard_costate_batch(B, 0.3) * (1 + 0.05 * rng.uniform(-1, 1, (B, 14)))
    X[:, 3:6] = rng.uniform(-0.1, 0.1, (B, 3))
    t = rng.uniform(0, 0.12, B)        # straddles both default switching times (mu2 = 0 branches)
    gctx.set_param("mu2", mu2)
    goracle.set_param("mu2", mu2)
    rhs = gctx.eval_batch(capi.EVAL_RHS, t, X)
    ctl = gctx.eval_batch(capi.EVAL_CONTROL, t, X)
    ham = gctx.eval_batch(capi.EVAL_HAMILTONIAN, t, X)
    for b in range(B):
        ref = goracle.rhs(t[b], X[b])
        assert np.allclose(rhs[b], ref, rtol=1e-13, atol=1e-13 * np.max(np.abs(ref))), (b, rhs[b] - ref)
        assert np.allclose(ctl[b], goracle.control(t[b], X[b]), rtol=1e-13, atol=1e-15)
        h = goracle.hamiltonian(t[b], X[b])[0]
        assert abs(ham[b, 0] - h) <= 1e-12 * max(1.0, abs(h))


@pytest.mark.parametrize("N,tol", [(10, 1e-10), (1000, 1e-10)])
def test_integrate_batch_short(gctx, goracle, N, tol):
    gctx.set_param("mu2", 1.0)
    goracle.set_param("mu2", 1.0)
    gctx.set_step_number(N)
    goracle.m.step_nbr = N
    B = 130                               # ragged: 2 full waves + 2 lanes
    X0 = goddard_costate_batch(B, 1e-3)
    tf = np.full(B, GODDARD_TF)
    Xg = gctx.integrate_batch(0.0, tf, X0)
    Xc = goracle.integrate_batch(0.0, tf, X0)
    assert relerr(Xg, Xc) <= tol


def test_integrate_batch_full_size(gctx, goracle):
    """BASELINE metric unit: 1e4 RK4 steps per trajectory."""
    gctx.set_param("mu2", 1.0)
    goracle.set_param("mu2", 1.0)
    gctx.set_step_number(10000)
    goracle.m.step_nbr = 10000
    B = 64
    X0 = goddard_costate_batch(B, 1e-3)
    Xg = gctx.integrate_batch(0.0, GODDARD_TF, X0)
    Xc = goracle.integrate_batch(0.0, GODDARD_TF, X0)
    assert np.all(np.isfinite(Xg))
    assert relerr(Xg, Xc) <= 1e-8


def test_integrate_edge_cases(gctx, goracle):
    """Empty batch, zero-length and backward segments (odeTools.cpp:136: zero steps, input returned)."""
    gctx.set_step_number(10)
    goracle.m.step_nbr = 10
    assert gctx.integrate_batch(0.0, 0.1, np.empty((0, 14))).shape == (0, 14)
    X0 = goddard_costate_batch(5, 1e-3)
    t0 = np.array([0.0, 0.1, 0.2, 0.0, 0.05])
    tf = np.array([0.0, 0.1, 0.1, 0.1, 0.05])      # zero, zero, backward, normal, zero
    Xg = gctx.integrate_batch(t0, tf, X0)
    Xc = goracle.integrate_batch(t0, tf, X0)
    for b in (0, 1, 2, 4):
        assert np.array_equal(Xg[b], X0[b])
    assert relerr(Xg, Xc) <= 1e-12


def test_switching_times_per_row(gctx, goracle):
    """mu2 = 0: bang / singular / off selected by t vs per-row switching times (goddard.cpp:146-162)."""
    gctx.set_param("mu2", 0.0)
    goracle.set_param("mu2", 0.0)
    gctx.set_step_number(10)
    goracle.m.step_nbr = 10
    B = 70
    rng = np.random.default_rng(5)
    X0 = goddard_costate_batch(B, 1e-2)
    sw = np.stack([rng.uniform(0.005, 0.04, B), rng.uniform(0.05, 0.1, B)], axis=1)
    tf = rng.uniform(0.02, 0.12, B)
    Xg = gctx.integrate_batch(0.0, tf, X0, sw=sw)
    Xc = goracle.integrate_batch(0.0, tf, X0, aux_sw=sw)
    assert relerr(Xg, Xc) <= 1e-10
    gctx.set_param("mu2", 1.0)
    goracle.set_param("mu2", 1.0)


def _setup_problem(gctx, prob):
    n = gctx.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode)
    assert n == prob.n


def test_residual_c1_stage1(gctx, goracle):
    """testGoddard.cpp as shipped: M = 6, free tf, n = 85; KD = 0 first solve (testGoddard.cpp:94-99)."""
    goracle.set_param("mu2", 1.0)
    goracle.set_param("KD", 310.0)
    goracle.m.step_nbr = 10
    prob, z = goddard_c1_problem(goracle)
    for KD in (0.0, 310.0):
        goracle.set_param("KD", KD)
        gctx.set_param("KD", KD)
        gctx.set_param("mu2", 1.0)
        gctx.set_step_number(10)
        _setup_problem(gctx, prob)
        Fc = goracle.residual(prob, z)
        Fg = gctx.residual(z)
        assert np.allclose(Fg, Fc, rtol=1e-12, atol=1e-12 * np.max(np.abs(Fc)))
        if KD == 0.0:
            # the reference's own value of this residual (probe of the reference build)
            assert abs(Fg[84] - (-1240.5248135826923)) <= 1e-9
    gctx.set_param("KD", 310.0)
    goracle.set_param("KD", 310.0)


def test_residual_batch_and_fd_jacobian_c2(gctx, goracle):
    """BASELINE config 2: n = 14 single shooting; the FD batch is base + 14 perturbed rows."""
    for c in (gctx, goracle):
        c.set_param("mu2", 1.0)
        c.set_param("KD", 310.0)
    gctx.set_step_number(100)
    goracle.m.step_nbr = 100
    prob, z = goddard_single_problem()
    _setup_problem(gctx, prob)
    eps = np.sqrt(1e-15)
    Z = np.tile(z, (15, 1))
    for j in range(14):
        h = eps * abs(z[j]) or eps
        Z[j + 1, j] += h
    Fg = gctx.residual_batch(Z)
    Fc = goracle.residual_batch(prob, Z)
    assert np.allclose(Fg, Fc, rtol=1e-11, atol=1e-11 * np.max(np.abs(Fc)))
    # the fused FD kernel must reproduce (F_j - F_0)/h from the SAME implementation bit for bit
    Jg = gctx.fd_jacobian(z, Fg[0], epsfcn=1e-15, dedup=False)
    Jman = np.stack([(Fg[j + 1] - Fg[0]) / (eps * abs(z[j]) or eps) for j in range(14)], axis=1)
    assert np.array_equal(Jg, Jman)
    Jd = gctx.fd_jacobian(z, Fg[0], epsfcn=1e-15, dedup=True)
    assert np.array_equal(Jg, Jd)


def test_fd_jacobian_dedup_c1(gctx, goracle):
    """n = 85: integrating only the segments a column can change gives the identical Jacobian."""
    for c in (gctx, goracle):
        c.set_param("mu2", 1.0)
        c.set_param("KD", 310.0)
    gctx.set_step_number(10)
    goracle.m.step_nbr = 10
    prob, z = goddard_c1_problem(goracle)
    _setup_problem(gctx, prob)
    F0 = gctx.residual(z)
    t0, _ = gctx.counters()
    Jfull = gctx.fd_jacobian(z, F0, dedup=False)
    t1, _ = gctx.counters()
    Jded = gctx.fd_jacobian(z, F0, dedup=True)
    t2, _ = gctx.counters()
    assert np.array_equal(Jfull, Jded)
    assert (t1 - t0) == 85 * 6 and (t2 - t1) < (t1 - t0) / 2
    # against MINPACK fdjac1 on the oracle: FD noise is amplified by 1/h, compare loosely
    Jc = goracle.fdjac(prob, z, goracle.residual(prob, z))
    scale = np.max(np.abs(Jc))
    assert np.max(np.abs(Jfull - Jc)) <= 1e-5 * scale


def test_fd_rows_many_problems(gctx, goracle):
    """One launch for the (n+1) residual rows of many starts; rows must equal the residual of the
    explicitly perturbed vectors, and the difference kernel the fused-FD Jacobian, bit for bit."""
    import torch
    for c in (gctx, goracle):
        c.set_param("mu2", 1.0)
        c.set_param("KD", 310.0)
    gctx.set_step_number(50)
    goracle.m.step_nbr = 50
    prob, z = goddard_single_problem()
    _setup_problem(gctx, prob)
    P = 9
    Z = np.tile(z, (P, 1))
    Z[:, 7:] = goddard_costate_batch(P, 1e-3)[:, 7:]
    rows = gctx.fd_rows(Z, epsfcn=1e-15)
    eps = np.sqrt(1e-15)
    for p in range(P):
        Zp = np.tile(Z[p], (15, 1))
        for j in range(14):
            Zp[j + 1, j] += eps * abs(Z[p, j]) or eps
        assert np.array_equal(rows[p], gctx.residual_batch(Zp))
        J = gctx.fd_jacobian(Z[p], rows[p, 0], epsfcn=1e-15)
        dZ = torch.from_numpy(Z[p:p + 1].copy()).cuda()
        dR = torch.from_numpy(rows[p:p + 1].copy()).cuda()
        dJ = torch.empty((1, 14, 14), dtype=torch.float64, device="cuda")
        gctx.fd_diff_dev(1, dZ.data_ptr(), 1e-15, dR.data_ptr(), dJ.data_ptr())
        gctx.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(dJ.cpu().numpy()[0].T, J)      # device J is column-major
    Fc = goracle.residual(prob, Z[3])
    assert np.allclose(rows[3, 0], Fc, rtol=1e-11, atol=1e-11 * np.max(np.abs(Fc)))


# ---- throughput flavour (restructured arithmetic, FMA contraction): tolerance, not bit equality ----

@pytest.fixture()
def gfast(gctx):
    from socp_amd import capi
    gctx.set_variant(capi.VARIANT_LANE_FAST)
    yield gctx
    gctx.set_variant(capi.VARIANT_AUTO)


@pytest.mark.parametrize("mu2", [1.0, 0.2, 0.0])
def test_fast_rhs_matches_oracle(gfast, goracle, mu2):
    from socp_amd import capi
    rng = np.random.default_rng(11)
    B = 256
    X = goddard_costate_batch(B, 0.3) * (1 + 0.05 * rng.uniform(-1, 1, (B, 14)))
    X[:, 3:6] = rng.uniform(-0.1, 0.1, (B, 3))
    t = rng.uniform(0, 0.12, B)
    gfast.set_param("mu2", mu2)
    goracle.set_param("mu2", mu2)
    rhs = gfast.eval_batch(capi.EVAL_RHS, t, X)
    ref = np.stack([goracle.rhs(t[b], X[b]) for b in range(B)])
    # relative to the size of each derivative vector: the factored gravity-gradient terms cancel
    # differently from the reference's term-by-term sum
    scale = np.max(np.abs(ref), axis=1, keepdims=True)
    assert np.max(np.abs(rhs - ref) / scale) <= 5e-13
    gfast.set_param("mu2", 1.0)
    goracle.set_param("mu2", 1.0)


@pytest.mark.parametrize("N,tol", [(10, 1e-10), (1000, 1e-10), (10000, 1e-8)])
def test_fast_trajectory_tolerance(gfast, goracle, N, tol):
    """SURVEY 8d parity tolerance (2): <= 1e-10 for short segments, <= 1e-8 at 1e4 RK4 steps."""
    for c in (gfast, goracle):
        c.set_param("mu2", 1.0)
        c.set_param("KD", 310.0)
    gfast.set_step_number(N)
    goracle.m.step_nbr = N
    B = 96
    X0 = goddard_costate_batch(B, 1e-3)
    Xg = gfast.integrate_batch(0.0, GODDARD_TF, X0)
    Xc = goracle.integrate_batch(0.0, GODDARD_TF, X0)
    assert np.all(np.isfinite(Xg))
    assert relerr(Xg, Xc) <= tol


def test_fast_bang_singular_off(gfast, goracle):
    for c in (gfast, goracle):
        c.set_param("mu2", 0.0)
    gfast.set_step_number(20)
    goracle.m.step_nbr = 20
    B = 70
    rng = np.random.default_rng(5)
    X0 = goddard_costate_batch(B, 1e-2)
    sw = np.stack([rng.uniform(0.005, 0.04, B), rng.uniform(0.05, 0.1, B)], axis=1)
    tf = rng.uniform(0.02, 0.12, B)
    Xg = gfast.integrate_batch(0.0, tf, X0, sw=sw)
    Xc = goracle.integrate_batch(0.0, tf, X0, aux_sw=sw)
    assert relerr(Xg, Xc) <= 1e-10
    for c in (gfast, goracle):
        c.set_param("mu2", 1.0)


def test_fast_residual_c1(gfast, goracle):
    for c in (gfast, goracle):
        c.set_param("mu2", 1.0)
        c.set_param("KD", 310.0)
    gfast.set_step_number(10)
    goracle.m.step_nbr = 10
    prob, z = goddard_c1_problem(goracle)
    _setup_problem(gfast, prob)
    Fc = goracle.residual(prob, z)
    Fg = gfast.residual(z)
    assert np.max(np.abs(Fg - Fc)) <= 1e-11 * np.max(np.abs(Fc))


# ---- adaptive Dormand-Prince integrator (what the reference runs with -D_USE_BOOST) ---------------------------

def test_dopri5_against_oracle_and_tolerance(gctx, goracle):
    """No Boost offline => the specification is the oracle's restatement of odeint's controlled stepper;
    it is itself checked against a 1e5-step RK4 solution and SciPy's RK45 in tests/test_oracle.py.  The GPU
    follows the same algorithm with per-lane step control; `pow` in the step controller is the only
    non-IEEE operation besides exp, so agreement is to ~100 x rounding of a tol-accurate solution."""
    from socp_amd import capi
    for c in (gctx, goracle):
        c.set_param("mu2", 1.0)
        c.set_param("KD", 310.0)
    gctx.set_step_number(10)
    goracle.m.step_nbr = 10
    B = 70
    X0 = goddard_costate_batch(B, 1e-3)
    fine = Oracle_fine_solution(X0[:4])
    for tol in (1e-6, 1e-9):
        gctx.set_integrator(capi.INT_DOPRI5, tol)
        Xg = gctx.integrate_batch(0.0, GODDARD_TF, X0)
        Xc = np.stack([goracle.traj_dopri5(0.0, X0[b], GODDARD_TF, tol)[0] for b in range(B)])
        assert np.all(np.isfinite(Xg))
        # same algorithm, but `pow` and exp round differently, so a step-size or an accept/reject decision
        # can differ: two tol-accurate solutions agree at the tolerance level, not at rounding level
        # (global error of this pair on this problem is ~30 x tol: tests/test_oracle.py)
        assert relerr(Xg, Xc) <= 100 * tol
        assert relerr(Xg[:4], fine) <= 200 * tol           # and a tol-accurate solution
    # zero-length and backward segments take no step
    t0 = np.array([0.1, 0.2]); tf = np.array([0.1, 0.1])
    assert np.array_equal(gctx.integrate_batch(t0, tf, X0[:2]), X0[:2])
    gctx.set_integrator(capi.INT_RK4)


def Oracle_fine_solution(X0):
    from oracle.oracle import Oracle, MODEL_GODDARD
    o = Oracle(MODEL_GODDARD, step_nbr=100000)
    o.set_param("mu2", 1.0)
    return np.stack([o.traj(0.0, x, GODDARD_TF) for x in X0])


def test_dopri5_residual_and_newton(gctx, goracle):
    """The shooting residual and a full Newton solve with the adaptive integrator (n = 14)."""
    from socp_amd import capi
    for c in (gctx, goracle):
        c.set_param("mu2", 1.0)
        c.set_param("KD", 310.0)
    gctx.set_step_number(10)
    prob, z = goddard_single_problem()
    _setup_problem(gctx, prob)
    gctx.set_integrator(capi.INT_DOPRI5, 1e-10)
    F = gctx.residual(z)
    J = gctx.fd_jacobian(z, F, dedup=True)
    assert np.array_equal(J, gctx.fd_jacobian(z, F, dedup=False))
    out = capi.hybrd(lambda v: gctx.residual(v), z, xtol=1e-10, epsfcn=1e-15,
                     fdjac=lambda x, f, e: gctx.fd_jacobian(x, f, epsfcn=e))
    assert out["info"] == 1
    # the root of the adaptive problem (local error 1e-10) agrees with the 1e4-step RK4 root it approximates
    gctx.set_integrator(capi.INT_RK4)
    gctx.set_step_number(10000)
    ref = capi.hybrd(lambda v: gctx.residual(v), z, xtol=1e-10, epsfcn=1e-15,
                     fdjac=lambda x, f, e: gctx.fd_jacobian(x, f, epsfcn=e))
    assert ref["info"] == 1
    assert np.max(np.abs(out["x"] - ref["x"])) <= 1e-5 * np.max(np.abs(ref["x"]))
    gctx.set_step_number(10)


def test_dopri5_fast_flavour_agrees_with_reference_order(gctx, gfast):
    """The adaptive integrator on the restructured right-hand side: same trajectories to the tolerance level."""
    from socp_amd import capi
    X0 = goddard_costate_batch(64, 1e-3)
    out = []
    for c in (gctx, gfast):
        c.set_param("mu2", 1.0)
        c.set_step_number(10)
        c.set_integrator(capi.INT_DOPRI5, 1e-10)
        out.append(c.integrate_batch(0.0, GODDARD_TF, X0))
        c.set_integrator(capi.INT_RK4)
    assert np.max(np.abs(out[0] - out[1]) / np.maximum(1.0, np.abs(out[0]))) < 1e-8


@pytest.mark.parametrize("flavour", ["exact", "fast"])
def test_dopri5_dense_output_is_the_integrators_own_steps(goracle, flavour):
    """The observer form of the reference's adaptive integrate() (odeTools.cpp:103-123, Boost branch at :108): the trace of an
    adaptive segment is the state at t0 and at the end of every ACCEPTED step.  Parity UNPINNED (no Boost
    offline): checked by construction -- in the reference-order flavour the last row is the plain adaptive result bit for bit
    (same arithmetic, same step-size decisions; the throughput flavour is contracted per kernel instantiation: rounding level),
    the rows sit on the trajectory (a fine fixed-step solution integrated to each row's time) within 100 x tol, and the number of
    steps is the restatement's (+-1: `pow` in the controller rounds differently on the device)."""
    from socp_amd import capi
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_variant(capi.VARIANT_LANE_EXACT if flavour == "exact" else capi.VARIANT_LANE_FAST)
    for c in (ctx, goracle):
        c.set_param("mu2", 1.0)
        c.set_param("KD", 310.0)
    ctx.set_step_number(10)
    goracle.m.step_nbr = 10
    X0 = goddard_costate_batch(3, 1e-3)[2]
    for tol in (1e-6, 1e-9):
        ctx.set_integrator(capi.INT_DOPRI5, tol)
        times, dense = ctx.integrate_dense(0.0, GODDARD_TF, X0)
        plain = ctx.integrate_batch(0.0, GODDARD_TF, X0[None, :])[0]
        assert times[0] == 0.0 and np.array_equal(dense[0], X0)
        assert times[-1] == GODDARD_TF
        if flavour == "exact":
            assert np.array_equal(dense[-1], plain)
        else:
            assert relerr(dense[-1][None, :], plain[None, :]) <= 1e-9
        assert np.all(np.diff(times) > 0) and len(times) >= 4
        _, accepted, _rej = goracle.traj_dopri5(0.0, X0, GODDARD_TF, tol)
        assert abs((len(times) - 1) - accepted) <= 1
        # a cap smaller than the trajectory: rows are counted, not stored, and the stored ones are the first ones
        t_short, d_short = ctx.integrate_dense(0.0, GODDARD_TF, X0, cap=3)
        assert len(t_short) == 3 and np.array_equal(d_short, dense[:3])
        ctx.set_integrator(capi.INT_RK4)
        ctx.set_step_number(20000)
        fine = ctx.integrate_batch(np.zeros(len(times) - 1), times[1:], np.repeat(X0[None, :], len(times) - 1, axis=0))
        ctx.set_step_number(10)
        assert relerr(dense[1:], fine) <= 100 * tol
    ctx.close()
