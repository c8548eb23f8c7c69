"""CPU: the interceptor restatement (oracle/interceptor_oracle.c).  PARITY UNPINNED -- the reference's
interceptor.cpp needs Eigen, which the image lacks, and the reference ships no output of it.  These tests are
what stands in for a pin: internal consistency that a transcription slip would break (the costate equations
are -dH/dx of the separately restated Hamiltonian in BOTH charts; a chart change is invertible and leaves H
unchanged), the stage/chart logic of ComputeTraj, the overridden final rows, the restated test program
converging from the reference's own analytical guess, and the frozen vectors under tests/golden/."""
import json
import os

import numpy as np
import pytest

from oracle.oracle import Oracle, MODEL_INTERCEPTOR
from test_gpu_interceptor import scenario_state, states_both_charts, single_shooting_problem, multi_shooting_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "interceptor_vectors.npz"))
FLOW = json.load(open(os.path.join(ROOT, "tests", "golden", "interceptor_flow.json")))


def hamiltonian_gradient(o, t, X):
    """Partial derivatives of H at fixed control: the thrust terms use sin/cos(alpha) while the control law is
    their small-angle optimum, so dH/du != 0 in the powered stage and the total derivative would not do."""
    ub = o.control(t, X)
    g = np.zeros(12)
    for i in range(12):
        h = 1e-6 * max(1.0, abs(X[i]))
        Xp, Xm = X.copy(), X.copy()
        Xp[i] += h
        Xm[i] -= h
        g[i] = (o.hamiltonian_at(t, Xp, ub) - o.hamiltonian_at(t, Xm, ub)) / (2 * h)
    return g


@pytest.mark.parametrize("chart", [1, 2])
@pytest.mark.parametrize("stage,t", [(1, 3.0), (0, 27.0)])
def test_dynamics_are_the_hamiltonian_system(built, chart, stage, t):
    """x' = dH/dp, p' = -dH/dx with H and the right-hand side restated from different parts of the source
    (interceptor.cpp:275-335/:440-503 vs :388-437/:555-604)."""
    o = Oracle(MODEL_INTERCEPTOR)
    X1, X2 = states_both_charts(o, 8)
    o.set_flags(chart, stage)
    worst = 0.0
    for X in (X1 if chart == 1 else X2):
        f = o.rhs(t, X)
        g = hamiltonian_gradient(o, t, X)
        ref = np.concatenate([g[6:], -g[:6]])
        worst = max(worst, np.max(np.abs(f - ref) / np.maximum(1e-4 * np.abs(ref).max(), np.abs(ref))))
    assert worst < 1e-6


def test_chart_change_round_trip_and_invariance(built):
    o = Oracle(MODEL_INTERCEPTOR)
    X1, X2 = states_both_charts(o, 12)
    for a, b in zip(X1, X2):
        back = o.chart21(b)
        back[3] = a[3] + (back[3] - a[3] + np.pi) % (2 * np.pi) - np.pi      # heading is an angle
        assert np.max(np.abs(back - a) / np.maximum(1.0, np.abs(a))) < 1e-9
        for stage, t in ((1, 3.0), (0, 27.0)):
            o.set_flags(1, stage)
            h1 = o.hamiltonian(t, a)[0]
            o.set_flags(2, stage)
            h2 = o.hamiltonian(t, b)[0]
            assert abs(h1 - h2) <= 1e-9 * max(1.0, abs(h1))


def test_lu6_against_numpy(built):
    o = Oracle(MODEL_INTERCEPTOR)
    rng = np.random.default_rng(0)
    for _ in range(20):
        A = rng.normal(size=(6, 6))
        A[3:, :3] = 0.0                                  # the block shape the chart Jacobians have
        b = rng.normal(size=6)
        x = o.lu6_solve(A, b)
        assert np.max(np.abs(x - np.linalg.solve(A, b))) <= 1e-12 * np.linalg.cond(A)


def test_compute_traj_stages_and_flags(built):
    o = Oracle(MODEL_INTERCEPTOR)
    X0, _ = scenario_state()
    o.traj(0.0, X0, 10.0)
    assert o.flags()[1] == 1                             # ends inside the powered stage (t1 = 200/10 = 20 s)
    Xf, rows = o.traj_trace(0.0, X0, 30.0)
    assert o.flags()[1] == 0 and len(rows) == 2 * 51
    assert [r[3] for r in rows[:51]] == [1] * 51 and [r[3] for r in rows[51:]] == [0] * 51
    assert rows[50][0] == pytest.approx(20.0, abs=1e-12) and rows[51][0] == 20.0
    o.traj(22.0, X0, 31.0)
    assert o.flags()[1] == 0
    # a start beyond the chart limit changes chart before the first step and the result comes back in chart 1
    Xs, _ = scenario_state(gamma=1.49)
    Xf, rows = o.traj_trace(0.0, Xs, 6.0)
    assert rows[0][2] == 1 and rows[1][2] == 2
    o.set_flags(1, 1)
    assert np.isfinite(Xf).all()


def test_final_rows_override(built):
    o = Oracle(MODEL_INTERCEPTOR)
    prob, z = single_shooting_problem(o)
    F = o.residual(prob, z)
    Xtf = o.traj(0.0, z[:12], z[12])
    assert F[6] == (Xtf[0] - prob.xnode[1, 0]) / 7500.0          # altitude row scaled by hr
    assert F[7] == Xtf[7] + 1.0                                  # free final velocity: p_v + muV
    assert F[8] == Xtf[2] - prob.xnode[1, 2]
    assert F[12] == o.hamiltonian(z[12], Xtf)[0] + 0.0           # free final time: H + muT, flags as ComputeTraj left them


def test_frozen_vectors(built):
    o = Oracle(MODEL_INTERCEPTOR)
    for chart, X in ((1, GOLD["X1"]), (2, GOLD["X2"])):
        for stage, t in ((1, 3.0), (0, 27.0)):
            o.set_flags(chart, stage)
            key = "c%d_s%d" % (chart, stage)
            assert np.array_equal(np.array([o.rhs(t, x) for x in X]), GOLD["rhs_" + key])
            assert np.array_equal(np.array([o.control(t, x) for x in X]), GOLD["ctl_" + key])
            assert np.array_equal(np.array([o.hamiltonian(t, x)[0] for x in X]), GOLD["ham_" + key])
    assert np.array_equal(np.array([o.chart21(x) for x in GOLD["X2"]]), GOLD["chart21_of_X2"])
    for a, e, x, want, fl in zip(GOLD["traj_t0"], GOLD["traj_tf"], GOLD["traj_X0"], GOLD["traj_Xf"], GOLD["traj_flags"]):
        assert np.array_equal(o.traj(a, x, e), want)
        assert tuple(o.flags()) == tuple(fl)
    o.set_param("mu_gft", 0.6)
    for M in (1, 4):
        prob, _ = single_shooting_problem(o) if M == 1 else multi_shooting_problem(o, M)
        assert np.array_equal(o.residual(prob, GOLD["res_z_M%d" % M]), GOLD["res_F_M%d" % M])


@pytest.mark.parametrize("solver", ["scipy", "socp"])
def test_restated_test_program_converges(built, solver):
    """tests/testInterceptor.cpp restated over the oracle: every SolveOCP returns 1 ("OK = 1") from the
    reference's analytical guess; SciPy's MINPACK and the library's own hybrd walk the same path."""
    from flow_oracle import interceptor_flow
    for sc in (1, 2, 3):
        got = interceptor_flow(solver, 1e-8, sc)
        want = FLOW["scenario%d_xtol1e-08" % sc]
        assert [(s["stage"], s["info"], s["nfev"]) for s in got] == [(s["stage"], s["info"], s["nfev"]) for s in want]
        for s, w in zip(got, want):
            assert np.array_equal(s["z"], np.array(w["z"]))


def test_adaptive_residual_of_the_config5_problem_converges_to_the_fixed_step_limit(built):
    """BASELINE config 5 (interceptor + adaptive Dormand-Prince + M = 21, n = 253) on the CPU restatement: with the
    adaptive integrator selected, the residual approaches the fine fixed-step residual as the tolerance shrinks (5(4) pair:
    error ~ tol), the chart hook included.  This is the checker the GPU test of the same configuration compares with."""
    from test_gpu_interceptor import config5_problem
    o = Oracle(MODEL_INTERCEPTOR)
    prob, z = config5_problem(o)
    assert prob.n == 253
    fine = Oracle(MODEL_INTERCEPTOR, step_nbr=2000)
    F_fine = fine.residual(prob, z)
    scale = np.maximum(1.0, np.abs(F_fine))
    errs = []
    for tol in (1e-6, 1e-8, 1e-10):
        o.set_integrator(1, tol)
        F = o.residual(prob, z)
        assert np.all(np.isfinite(F))
        errs.append(float(np.max(np.abs(F - F_fine) / scale)))
    assert errs[0] < 1e-3 and errs[1] < 1e-5 and errs[2] < 1e-7, errs
    assert errs[2] < errs[0]
    o.set_integrator(0)
    assert np.max(np.abs(o.residual(prob, z) - F_fine) / scale) < 1e-6      # the 50-step RK4 default, for scale
