"""GPU: interceptor device model (models_interceptor.hpp) against the CPU restatement
(oracle/interceptor_oracle.c -- PARITY UNPINNED, see its header: the reference TU needs Eigen).

The model calls sin/cos/tan/atan2/acos/exp, which differ between the device library and libm in the last
place, so nothing here is bit-exact.  Tolerances: 1e-12 relative on single evaluations, 1e-10 on
50/100-step trajectories (SURVEY 8d, N = 10/30 class), 1e-9 through a chart change (6x6 solve with
condition ~1e7 from mixing metres and radians)."""
import numpy as np
import pytest

from oracle.oracle import Oracle, Problem, MODEL_INTERCEPTOR, FIXED, FREE, CONTINUOUS

pytestmark = pytest.mark.gpu

R_E = 6378145.0


def scenario_state(gamma=np.pi / 4, chi=0.0):
    """testInterceptor.cpp:170-176 initial state + the analytical costate guess (InitAnalytical)."""
    o = Oracle(MODEL_INTERCEPTOR)
    Xi = np.zeros(12)
    Xi[:6] = [1000, 1000, gamma, chi, 5454661 / R_E, 46086 / R_E]
    Xf = np.zeros(12)
    Xf[:6] = [6000, 1000, 0.01 * np.pi, 0.01 * np.pi, (5454661 + 27829.0) / R_E, 46086 / R_E]
    return o.init_analytical(0.0, Xi, 10.0, Xf), Xf


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


@pytest.fixture(scope="module")
def ictx():
    from socp_amd import capi
    c = capi.Context(capi.MODEL_INTERCEPTOR)
    yield c
    c.close()


def states_both_charts(o, B, seed=3):
    rng = np.random.default_rng(seed)
    X1, _ = scenario_state()
    rows1, rows2 = [], []
    for _ in range(B):
        x = X1 * (1 + 0.2 * rng.uniform(-1, 1, 12))
        x[2] = rng.uniform(-1.3, 1.3)
        x[3] = rng.uniform(-3, 3)
        rows1.append(x)
        rows2.append(o.chart12(x))
    return np.array(rows1), np.array(rows2)


def test_model_control_hamiltonian_both_charts_both_stages(ictx, built):
    from socp_amd import capi
    o = Oracle(MODEL_INTERCEPTOR)
    X1, X2 = states_both_charts(o, 16)
    for chart, X in ((1, X1), (2, X2)):
        for stage, t in ((1, 3.0), (0, 27.0)):
            o.set_flags(chart, stage)
            sw = np.tile([float(stage), float(chart)], (len(X), 1))
            f = ictx.eval_batch(capi.EVAL_RHS, t, X, sw=sw)
            u = ictx.eval_batch(capi.EVAL_CONTROL, t, X, sw=sw)
            H = ictx.eval_batch(capi.EVAL_HAMILTONIAN, t, X, sw=sw)[:, 0]
            for b in range(len(X)):
                fo = o.rhs(t, X[b])
                assert np.max(np.abs(f[b] - fo) / np.maximum(1e-6 * np.abs(fo).max(), np.abs(fo))) < 1e-9, (chart, stage, b)
                assert rel(u[b], o.control(t, X[b])) < 1e-12
                assert abs(H[b] - o.hamiltonian(t, X[b])[0]) <= 1e-11 * max(1.0, np.abs(fo).max())


def test_trajectories_two_stages_and_chart_switch(ictx, built):
    o = Oracle(MODEL_INTERCEPTOR)
    X0, _ = scenario_state()
    cases = [(0.0, 10.0, X0),                      # powered stage only
             (0.0, 30.0, X0),                      # powered then coasting (t1 = 20 s)
             (22.0, 31.0, X0)]                     # coasting only
    Xs, _ = scenario_state(gamma=1.49)             # |cos(gamma)| < chartLimit: starts with a chart change
    cases.append((0.0, 6.0, Xs))
    cases.append((0.0, 25.0, Xs))
    t0 = np.array([c[0] for c in cases])
    tf = np.array([c[1] for c in cases])
    X = np.array([c[2] for c in cases])
    Xf = ictx.integrate_batch(t0, tf, X)
    switched = 0
    for b, (a, e, x) in enumerate(cases):
        ref, rows = o.traj_trace(a, x, e)
        switched += any(r[2] == 2 for r in rows)
        assert rel(Xf[b], ref) < (1e-9 if any(r[2] == 2 for r in rows) else 1e-10), (b, Xf[b], ref)
    assert switched >= 2                           # the chart-change path really ran


def test_dense_rows_match_trace_rows(ictx, built):
    o = Oracle(MODEL_INTERCEPTOR)
    Xs, _ = scenario_state(gamma=1.49)
    ref, rows = o.traj_trace(0.0, Xs, 25.0)
    t, X, aux = ictx.integrate_dense_aux(0.0, 25.0, Xs, cap=256)
    assert len(t) == len(rows) + 1 == 2 * 50 + 2 + 1
    for k, (tr, Xr, chart, stage) in enumerate(rows):
        assert abs(t[k] - tr) <= 1e-12 * max(1, abs(tr))
        assert (aux[k, 0], aux[k, 1]) == (stage, chart), k
        assert rel(X[k], Xr) < 1e-9, k
    assert rel(X[-1], ref) < 1e-9 and tuple(aux[-1]) == (float(o.flags()[1]), float(o.flags()[0]))


def single_shooting_problem(o):
    """testInterceptor.cpp initState(): M = 1, tf free, final velocity free -> n = 13."""
    X0, Xf = scenario_state()
    mode_t = [FIXED, FREE]
    mode_x = np.zeros((2, 6), dtype=np.int32)
    mode_x[1, 1] = FREE
    X = np.array([X0, Xf])
    prob = Problem(6, mode_t, mode_x, np.array([0.0, 10.0]), X)
    return prob, np.concatenate([X0, [10.0]])


def multi_shooting_problem(o, M, tf=24.0, X0=None, Xf=None):
    """M segments over [0, tf]; node states along the trajectory from X0 (default: the analytical guess -- not a
    solution, the later nodes are far from anything physical, which is what a first Newton iterate looks like)."""
    if X0 is None:
        X0, Xf = scenario_state()
    mode_t = [FIXED] + [CONTINUOUS] * (M - 1) + [FREE]
    mode_x = np.zeros((M + 1, 6), dtype=np.int32)
    mode_x[1:M] = CONTINUOUS
    mode_x[M, 1] = FREE
    time = np.array([tf * i / M for i in range(M + 1)])
    X = np.zeros((M + 1, 12))
    X[0], X[M] = X0, Xf
    for i in range(1, M):
        X[i] = o.traj(0.0, X0, time[i])
    return Problem(6, mode_t, mode_x, time, X), np.concatenate([X[:M].ravel(), [tf]])


@pytest.mark.parametrize("M", [1, 4, 21])
def test_residual_and_fd_rows(ictx, built, M):
    o = Oracle(MODEL_INTERCEPTOR)
    o.set_param("mu_gft", 0.6)
    ictx.set_param("mu_gft", 0.6)
    prob, z = single_shooting_problem(o) if M == 1 else multi_shooting_problem(o, M)
    assert ictx.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode) == prob.n
    if M == 21:
        assert prob.n == 253                      # BASELINE config 5 class: ~256 unknowns
    Fo = o.residual(prob, z)
    F = ictx.residual(z)
    scale = np.maximum(1.0, np.abs(Fo))
    assert np.max(np.abs(F - Fo) / scale) < 1e-9
    # custom final rows really in play: altitude row is scaled by hr, velocity row is p_v + muV
    Xtf = o.traj(prob.time[M - 1] if M > 1 else 0.0, z[12 * (M - 1):12 * M], z[-1])
    assert abs(Fo[6] - (Xtf[0] - prob.xnode[M, 0]) / 7500.0) < 1e-12
    assert abs(Fo[7] - (Xtf[7] + 1.0)) < 1e-12
    # FD rows = residuals of the perturbed unknown vectors
    rows = ictx.fd_rows(z[None, :], epsfcn=1e-15)[0]
    eps = np.sqrt(1e-15)
    for j in (0, 7, 8, prob.n - 1):
        zp = z.copy()
        h = eps * abs(z[j]) or eps
        zp[j] += h
        Fj = o.residual(prob, zp)
        assert np.max(np.abs(rows[j + 1] - Fj) / np.maximum(1.0, np.abs(Fj))) < 1e-9, j
    assert np.max(np.abs(rows[0] - Fo) / scale) < 1e-9
    ictx.set_param("mu_gft", 1.0)


def config5_problem(o):
    """BASELINE config 5: M = 21 segments -> n = 253, nodes along the CONVERGED scenario-1 trajectory of the test program
    (tests/golden/interceptor_flow.json): the Jacobian a multiple-shooting solve of that scenario evaluates."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = json.load(open(os.path.join(root, "tests", "golden", "interceptor_flow.json")))["scenario1_xtol1e-12"][-1]["z"]
    Xf = np.zeros(12)
    Xf[:6] = [12000, 1000, 0.0, np.pi / 8, 5475000 / R_E, 42000 / R_E]
    return multi_shooting_problem(o, 21, tf=gold[12], X0=np.array(gold[:12]), Xf=Xf)


@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_config5_interceptor_dopri5_253_unknowns(built, variant):
    """BASELINE config 5 AS STATED: interceptor model, adaptive Dormand-Prince (per-lane step control), multiple shooting
    with M = 21 segments -> n = 253 unknowns.  Residual and all 254 FD rows against the CPU restatement running the same
    adaptive integrator (oracle/interceptor_oracle.c: orc_interceptor_compute_traj_adaptive -- PARITY UNPINNED: no Boost,
    no Eigen, no reference-held vectors, and the reference's interceptor is RK4-only, so this combination is an
    extrapolation of the reference; the tolerance is the one the single-trajectory adaptive tests use, 100 x tol),
    the FD Jacobian identical with and without the segment dedup, and every entry finite."""
    from socp_amd import capi
    tol = 1e-8
    o = Oracle(MODEL_INTERCEPTOR)
    prob, z = config5_problem(o)                     # nodes integrated with the fixed-step path
    assert prob.n == 253
    o.set_integrator(1, tol)
    c = capi.Context(capi.MODEL_INTERCEPTOR)
    c.set_variant(capi.VARIANT_LANE_EXACT if variant == "exact" else capi.VARIANT_LANE_FAST)
    c.set_integrator(capi.INT_DOPRI5, tol)
    assert c.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode) == 253
    Fo = o.residual(prob, z)
    F = c.residual(z)
    scale = np.maximum(1.0, np.abs(Fo))
    assert np.all(np.isfinite(F)) and np.max(np.abs(F - Fo) / scale) <= 100 * tol
    # the adaptive integrator is really the one that ran: the fixed-step residual differs by far more than rounding
    c.set_integrator(capi.INT_RK4)
    F_rk4 = c.residual(z)
    c.set_integrator(capi.INT_DOPRI5, tol)
    assert np.max(np.abs(F - F_rk4) / scale) > 1e-12
    # all n + 1 rows of the FD batch in one launch vs residuals of the perturbed vectors
    rows = c.fd_rows(z[None, :], epsfcn=1e-15)[0]
    assert rows.shape == (254, 253) and np.all(np.isfinite(rows))
    eps = np.sqrt(1e-15)
    Zp = np.repeat(z[None, :], 254, axis=0)
    for j in range(253):
        h = eps * abs(z[j]) or eps
        Zp[j + 1, j] += h
    want = o.residual_batch(prob, Zp)
    assert np.max(np.abs(rows - want) / np.maximum(1.0, np.abs(want))) <= 100 * tol
    assert np.array_equal(rows[0], F)                       # same kernel arithmetic for the base row
    # FD Jacobian: dedup on / off identical, and equal to the differences of the rows
    J_full = c.fd_jacobian(z, F, epsfcn=1e-15, dedup=False)
    J_ded = c.fd_jacobian(z, F, epsfcn=1e-15, dedup=True)
    assert np.array_equal(J_full, J_ded) and np.all(np.isfinite(J_full))
    hs = np.array([eps * abs(v) or eps for v in z])
    assert np.array_equal(J_full, ((rows[1:] - rows[0]) / hs[:, None]).T)
    c.close()


def test_adaptive_integrator_matches_fine_fixed_step(ictx):
    """Dormand-Prince with the per-step chart choice (extension; the reference's interceptor is RK4-only):
    converges to what a very fine fixed-step run gives."""
    from socp_amd import capi
    X0, _ = scenario_state()
    ictx.set_step_number(4000)
    fine = ictx.integrate_batch(0.0, 30.0, X0[None, :])[0]
    ictx.set_step_number(50)
    ictx.set_integrator(capi.INT_DOPRI5, 1e-10)
    ada = ictx.integrate_batch(0.0, 30.0, X0[None, :])[0]
    ictx.set_integrator(capi.INT_RK4)
    assert rel(ada, fine) < 1e-7


# ---- the reference's test program through the C++ host mirror ---------------------------------------------
import json  # noqa: E402
import os  # noqa: E402
import subprocess  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "socp_amd", "_build", "bin", "interceptor_flow")


def run_flow(xtol, scenario, trace=None):
    cmd = [EXE, repr(xtol), str(scenario)] + ([str(trace)] if trace else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    return out.returncode, [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")], out.stderr


@pytest.mark.parametrize("scenario", [1, 2, 3])
def test_interceptor_program_converged_solution(scenario):
    """tests/testInterceptor.cpp through interceptor + shooting of the host mirror, every residual and FD Jacobian
    on the GPU.  north_star tolerance: converged solution within 1e-8 relative of the CPU path, asserted at
    xtol = 1e-12 where the root is defined that sharply; at the test's own 1e-8 every stage must report 1."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "interceptor_flow.json")))
    rc, stages, err = run_flow(1e-12, scenario)
    want = gold["scenario%d_xtol1e-12" % scenario]
    assert rc == 0 and len(stages) == len(want), err
    for s, g in zip(stages, want):
        assert (s["stage"], s["info"]) == (g["stage"], 1)
        z, zg = np.array(s["z"]), np.array(g["z"])
        assert np.max(np.abs(z - zg)) <= 1e-8 * np.max(np.abs(zg)), s["stage"]
        # per component, on the components that are not (numerically) zero
        big = np.abs(zg) > 1e-6 * np.max(np.abs(zg))
        assert np.max(np.abs(z[big] - zg[big]) / np.abs(zg[big])) <= 1e-6, s["stage"]
    rc, stages, err = run_flow(1e-8, scenario)
    assert rc == 0 and [s["info"] for s in stages] == [1, 1, 1], err


def test_interceptor_trace_file(tmp_path, built):
    """shooting::Trace through interceptor::ComputeTraj(isTrace = 1): one row per stage start and step,
    t, X[12], u, beta, H, chart (interceptor.cpp:131-151); the rows replay the converged trajectory."""
    trace = tmp_path / "trace_S1.dat"
    rc, stages, err = run_flow(1e-8, 1, trace)
    assert rc == 0, err
    rows = np.loadtxt(trace)
    z = np.array(stages[-1]["z"])
    tf = z[12]
    n_expected = 2 * 51 if tf > 20.0 else 51
    assert rows.shape == (n_expected, 1 + 12 + 2 + 1 + 1)
    assert rows[0, 0] == 0.0 and abs(rows[-1, 0] - tf) < 1e-5 * tf
    assert np.max(np.abs(rows[0, 1:13] - z[:12]) / np.maximum(1e-3, np.abs(z[:12]))) < 1e-5     # 6 printed digits
    o = Oracle(MODEL_INTERCEPTOR)
    Xf, ref = o.traj_trace(0.0, z[:12], tf)
    for k in (1, 25, 50, n_expected - 1):
        t, X, chart, stage = ref[k]
        assert rows[k, -1] == chart
        o.set_flags(chart, stage)
        want = np.concatenate([[t], X, o.control(t, X), o.hamiltonian(t, X)])
        assert np.all(np.abs(rows[k, :-1] - want) <= 2e-5 * np.abs(want) + 1e-9), k


def test_adaptive_budget_bounds_a_singular_trajectory(ictx, built):
    """Nodes taken along the raw analytical guess run into v -> 0 (a singularity of the dynamics) on the late
    segments; adaptive stepping there would take millions of steps.  The trial-step budget ends such a lane with
    NaN (odeint's step_adjustment_error) instead of holding the wave; healthy segments are unaffected."""
    import time
    from socp_amd import capi
    o = Oracle(MODEL_INTERCEPTOR)
    prob, z = multi_shooting_problem(o, 21)
    ictx.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode)
    F_rk4 = ictx.residual(z)
    ictx.set_integrator(capi.INT_DOPRI5, 1e-8)
    t = time.perf_counter()
    F = ictx.residual(z)
    dt = time.perf_counter() - t
    ictx.set_integrator(capi.INT_RK4)
    assert dt < 60.0
    assert np.isnan(F).any() or np.max(np.abs(F - F_rk4) / np.maximum(1.0, np.abs(F_rk4))) < 1e-2
    early = slice(12, 12 * 6)                              # continuity rows of the first segments: well-behaved
    assert np.all(np.isfinite(F[early]))
    assert np.max(np.abs(F[early] - F_rk4[early]) / np.maximum(1.0, np.abs(F_rk4[early]))) < 1e-4


# ---- throughput flavour (InterceptorT<true>: reciprocals, short sincos, no atan2; contraction on) --------------
def test_fast_flavour_matches_cpu_path(built):
    from socp_amd import capi
    o = Oracle(MODEL_INTERCEPTOR)
    c = capi.Context(capi.MODEL_INTERCEPTOR)
    c.set_variant(capi.VARIANT_LANE_FAST)
    X1, X2 = states_both_charts(o, 16)
    for chart, X in ((1, X1), (2, X2)):
        for stage, t in ((1, 3.0), (0, 27.0)):
            o.set_flags(chart, stage)
            sw = np.tile([float(stage), float(chart)], (len(X), 1))
            f = c.eval_batch(capi.EVAL_RHS, t, X, sw=sw)
            for b in range(len(X)):
                fo = o.rhs(t, X[b])
                assert np.max(np.abs(f[b] - fo) / np.maximum(1e-6 * np.abs(fo).max(), np.abs(fo))) < 1e-9, (chart, stage, b)
    X0, _ = scenario_state()
    Xs, _ = scenario_state(gamma=1.49)
    cases = [(0.0, 10.0, X0), (0.0, 30.0, X0), (22.0, 31.0, X0), (0.0, 6.0, Xs), (0.0, 25.0, Xs)]
    Xf = c.integrate_batch(np.array([k[0] for k in cases]), np.array([k[1] for k in cases]), np.array([k[2] for k in cases]))
    for b, (a, e, x) in enumerate(cases):
        assert rel(Xf[b], o.traj(a, x, e)) < 1e-9, b
    o.set_param("mu_gft", 0.6)
    c.set_param("mu_gft", 0.6)
    prob, z = multi_shooting_problem(o, 21)
    c.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode)
    Fo = o.residual(prob, z)
    assert np.max(np.abs(c.residual(z) - Fo) / np.maximum(1.0, np.abs(Fo))) < 1e-9
    F = c.residual(z)
    assert np.array_equal(c.fd_jacobian(z, F, dedup=True), c.fd_jacobian(z, F, dedup=False))
    # adaptive integrator in the throughput flavour agrees with the adaptive integrator in reference order
    c.set_integrator(capi.INT_DOPRI5, 1e-10)
    Xa = c.integrate_batch(0.0, 30.0, X0[None, :])[0]
    c.set_variant(capi.VARIANT_AUTO)
    Xb = c.integrate_batch(0.0, 30.0, X0[None, :])[0]
    assert rel(Xa, Xb) < 1e-8
    c.close()


@pytest.mark.parametrize("scenario", [1, 3])
def test_fast_flavour_converged_solution(scenario):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "interceptor_flow.json")))
    out = subprocess.run([EXE, "1e-12", str(scenario)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, SOCP_VARIANT="fast"))
    stages = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    want = gold["scenario%d_xtol1e-12" % scenario]
    assert out.returncode == 0 and len(stages) == len(want), out.stderr
    for s, g in zip(stages, want):
        z, zg = np.array(s["z"]), np.array(g["z"])
        assert s["info"] == 1 and np.max(np.abs(z - zg)) <= 1e-8 * np.max(np.abs(zg)), s["stage"]
