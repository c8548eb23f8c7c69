"""GPU: batched continuation chains (socp_chains_solve) and the per-problem blocks under them.

The reference's continuation loops are sequential (shooting.cpp:598-692 on the boundary data, :695-778 on one model parameter);
the engine runs P chains of one problem structure in lock-step.  The bar: every chain's solves are bit-identical to the
sequential loop run for that chain alone -- checked against (a) the C++ host mirror itself (`goddard_flow stage 2` =
shooting::SolveOCP(step, "KD", goal), tests/cpp/goddard_flow.cpp) and (b) a sequential restatement of the same loop over the
one-problem C-ABI entry points the mirror calls (socp_problem_set / socp_hybrd_batched / socp_residual_batch / socp_fd_jacobian),
for every chain; bisection after a failed solve included."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "goddard_flow.json")))
STAGE2_INIT = np.array([g for g in GOLD["goddard_single_stage"] if g["stage"] == 2 and g["xtol"] == 1e-6][0]["init_z"])
PARAMS0 = [3.5, 7.0, 0.0, 500.0, 1.0, 1.0, 1.0, -1.0]            # testGoddard before its KD continuation: KD = 0, mu2 = 1
KD = 2


def goddard_m6(ctx, x_final0=1.01):
    """The testGoddard layout (testGoddard.cpp:24-82): M = 6, tf FREE, final velocity and mass free; node times from the
    stage-2 start (uniform grid up to its tf)."""
    from socp_amd import capi
    M, d = 6, 7
    mode_t = [capi.FIXED] + [capi.CONTINUOUS] * (M - 1) + [capi.FREE]
    mode_x = np.zeros((M + 1, d), dtype=np.int32)
    mode_x[1:M] = capi.CONTINUOUS
    mode_x[M, 3:7] = capi.FREE
    tf = STAGE2_INIT[-1]
    time = np.array([0.0 + i * (tf - 0.0) / M for i in range(M + 1)])
    X = np.zeros((M + 1, 14))
    X[:M] = STAGE2_INIT[:84].reshape(M, 14)
    X[M, 0] = x_final0
    assert ctx.problem_set(mode_t, mode_x, time, X) == 85
    return mode_t, mode_x, time, X


def make_ctx(variant="exact", steps=10):
    from socp_amd import capi
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(PARAMS0)
    ctx.set_step_number(steps)
    ctx.set_variant(capi.VARIANT_LANE_EXACT if variant == "exact" else capi.VARIANT_LANE_FAST)
    return ctx


def sequential_solve(ctx, z, xtol):
    """One Newton solve exactly as shooting::SolveShootingFunction of the mirror performs it (shooting.cpp:433-452 there)."""
    from socp_amd import capi
    r = capi.hybrd(lambda v: ctx.residual(v), z, xtol=xtol, epsfcn=1e-15, factor=1.0,
                   fdjac=lambda v, f, e: ctx.fd_jacobian(v, f, epsfcn=e, dedup=True))
    return r["info"], r["x"], r["nfev"]


def sequential_chain(ctx, z0, step, step_min, xtol, set_b):
    """shooting::SolveShootingContinuation (both forms share this skeleton): set_b(b) installs the blended data."""
    b, b_prec = min(step, 1.0), 0.0
    set_b(b)
    committed, temp = z0.copy(), z0.copy()
    solves, nfev_total = 0, 0
    while True:
        info, x, nfev = sequential_solve(ctx, temp, xtol)
        solves += 1
        nfev_total += nfev
        if info != 1:
            stop = abs(b - b_prec) < step_min
            b = b_prec + (b - b_prec) / 2
            temp = committed.copy()
            set_b(b)
            if stop:
                break
        elif b == 1:
            committed = x.copy()
            break
        else:
            b_prec = b
            b = min(b + step, 1.0)
            committed = x.copy()
            temp = x.copy()
            set_b(b)
    return dict(z=committed, info=info, nfev=nfev, nfev_total=nfev_total, solves=solves, b=b if info == 1 else b_prec)


@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_per_problem_blocks_equal_one_problem_at_a_time(variant):
    """Row q of a batch with its own (parameters, boundary data) == the residual of that problem evaluated alone with the
    shared-parameter kernels: bit for bit in the reference-order flavour (the PERPROB instantiation changes where the
    parameters live, not the arithmetic), to rounding in the throughput flavour; including a row whose control law differs
    from the others' (mu2 = 0: bang / singular / off)."""
    ctx = make_ctx(variant)
    mode_t, mode_x, time, X = goddard_m6(ctx)
    rng = np.random.default_rng(5)
    B = 9
    Z = STAGE2_INIT[None, :] * (1 + 1e-3 * rng.uniform(-1, 1, (B, 85)))
    P = np.tile(np.array(PARAMS0 + [0.0227, 0.08]), (B, 1))
    P[:, KD] = np.linspace(0.0, 400.0, B)
    P[3, 6] = 0.0                                     # one row on the imposed bang / singular / off law
    P[4, 6] = 0.2
    T = np.tile(time, (B, 1)) * (1 + 1e-2 * rng.uniform(-1, 1, (B, 1)))
    T[:, 0] = 0.0
    XN = np.tile(X.ravel(), (B, 1))
    XN[:, 6 * 14] = 1.01 + 1e-3 * np.arange(B)        # final altitude target per row
    F = ctx.residual_batch_blocks(Z, params=P, time=T, xnode=XN)
    F_p = ctx.residual_batch_blocks(Z, params=P)      # parameters only
    F_b = ctx.residual_batch_blocks(Z, time=T, xnode=XN)

    def same(a, b):
        # exact flavour: no contraction, IEEE operations in one order => the same bits from any instantiation.  The
        # throughput flavour is compiled with FMA contraction, which the compiler applies per instantiation (the per-problem
        # kernel is the general-law one, the shared-parameter launch may pick the smooth-law specialisation): rounding level.
        if variant == "exact":
            return np.array_equal(a, b)
        return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= 1e-12
    for q in range(B):
        ctx.set_params(P[q, :8])
        ctx.set_switching_times(P[q, 8:])
        ctx.problem_set(mode_t, mode_x, T[q], XN[q].reshape(7, 14))
        assert same(F[q], ctx.residual(Z[q])), q
        ctx.problem_set(mode_t, mode_x, time, X)
        assert same(F_p[q], ctx.residual(Z[q])), q
        ctx.set_params(PARAMS0)
        ctx.set_switching_times([0.0227, 0.08])
        ctx.problem_set(mode_t, mode_x, T[q], XN[q].reshape(7, 14))
        assert same(F_b[q], ctx.residual(Z[q])), q
    ctx.problem_set(mode_t, mode_x, time, X)
    assert not np.array_equal(F[1], F[2])
    ctx.close()


def test_kd_continuation_chains_equal_the_sequential_loop_and_the_cpp_mirror(tmp_path):
    """P chains of testGoddard's KD continuation (SolveOCP(step, "KD", goal)), every chain its own goal; some with a step
    < 1, one with a goal far enough that a solve fails and the step is bisected."""
    ctx = make_ctx("exact")
    goddard_m6(ctx)
    goals = np.array([310.0, 250.0, 400.0, 310.0, 120.0, 5000.0, 310.0, 600.0])
    P = len(goals)
    Z0 = np.tile(STAGE2_INIT, (P, 1))
    Z0[3, 7:14] *= 1 + 1e-6                          # one chain starts elsewhere
    for step in (1.0, 0.4):
        res = ctx.chains_solve(Z0, kind=1, param_index=KD, step=step, goal=goals, params=np.tile(PARAMS0, (P, 1)), xtol=1e-6)
        assert res["stats"]["rounds"] > 0
        seq = []
        for p in range(P):
            def set_b(b, goal=goals[p]):
                ctx.set_param("KD", (1 - b) * 0.0 + b * goal)
            seq.append(sequential_chain(ctx, Z0[p], step, 1e-12, 1e-6, set_b))
            ctx.set_params(PARAMS0)
        for p in range(P):
            assert res["info"][p] == seq[p]["info"], (step, p)
            assert np.array_equal(res["z"][p], seq[p]["z"]), (step, p)
            assert res["solves"][p] == seq[p]["solves"] and res["nfev_total"][p] == seq[p]["nfev_total"], (step, p)
            assert res["nfev"][p] == seq[p]["nfev"]
            assert res["b_reached"][p] == seq[p]["b"]
        assert np.all(res["info"][[0, 1, 2, 3, 4]] == 1) and np.all(res["param_final"][[0, 1, 2]] == goals[[0, 1, 2]])
        if step == 1.0:
            assert np.max(res["solves"]) > 1 or np.any(res["info"] != 1)            # the far goal needed bisection (or gave up)
            gold2 = [g for g in GOLD["goddard_single_stage"] if g["stage"] == 2 and g["xtol"] == 1e-6][0]
            assert np.array_equal(res["z"][0], np.array(gold2["z"])) and res["nfev"][0] == gold2["nfev"]     # testGoddard's own stage 2 (CPU golden)
    # the C++ host mirror itself, for three of the chains (goal 310 is testGoddard's own stage 2)
    exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "goddard_flow")
    res = ctx.chains_solve(Z0, kind=1, param_index=KD, step=1.0, goal=goals, params=np.tile(PARAMS0, (P, 1)), xtol=1e-6)
    for p in (0, 2, 5):
        zf = tmp_path / ("z%d.txt" % p)
        zf.write_text(" ".join(repr(float(v)) for v in Z0[p]))
        out = subprocess.run([exe, "stage", "2", "10", "1", "1e-6", str(zf)], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, SOCP_VARIANT="exact", SOCP_FLOW_KD_GOAL=repr(float(goals[p]))))
        recs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        assert recs and recs[0]["info"] == res["info"][p], (p, out.returncode, out.stdout[-500:], out.stderr[-500:])
        assert np.array_equal(np.array(recs[0]["z"]), res["z"][p]), p
        assert recs[0]["nfev"] == res["nfev"][p]
        assert recs[1]["KD_final"] == res["param_final"][p]
    ctx.close()


def test_boundary_data_chains_equal_the_sequential_loop():
    """SolveOCP(step): homotopy on the boundary data, (1 - b) previous + b desired (shooting.cpp:598-692) -- here the final
    altitude target of every chain moves from 1.01 to its own goal, KD = 310 (testGoddard's state after its stage 2)."""
    ctx = make_ctx("exact")
    ctx.set_param("KD", 310.0)
    mode_t, mode_x, time, X = goddard_m6(ctx)
    z_conv = np.array(GOLD["goddard_N10_M6"][1]["z"])              # converged with drag
    goals = np.array([1.0102, 1.0105, 1.011, 1.03])
    P = len(goals)
    Xp = np.tile(X.ravel(), (P, 1))
    Xg = Xp.copy()
    Xg[:, 6 * 14] = goals
    Tp = np.tile(time, (P, 1))
    Z0 = np.tile(z_conv, (P, 1))
    for step in (1.0, 0.5):
        res = ctx.chains_solve(Z0, kind=2, step=step, time_prev=Tp, x_prev=Xp, time_goal=Tp, x_goal=Xg, xtol=1e-6)
        for p in range(P):
            def set_b(b, p=p):
                Xb = X.copy()
                Xb[6, 0] = (1 - b) * 1.01 + b * goals[p]
                ctx.problem_set(mode_t, mode_x, (1 - b) * time + b * time, Xb)
            s = sequential_chain(ctx, Z0[p], step, 1e-12, 1e-6, set_b)
            assert res["info"][p] == s["info"] and np.array_equal(res["z"][p], s["z"]), (step, p)
            assert res["solves"][p] == s["solves"] and res["nfev_total"][p] == s["nfev_total"], (step, p)
        assert np.all(res["info"][:3] == 1)
        ctx.problem_set(mode_t, mode_x, time, X)
    ctx.close()


@pytest.mark.parametrize("variant", ["exact", "fast"])
@pytest.mark.parametrize("solver", ["host", "device"])
def test_speculative_jacobians_change_no_iterate(variant, solver):
    """Residual requests evaluated as whole FD batches (speculate = 1) / never (0) / when the chip has idle SIMDs (-1): same
    solutions, same evaluation counts, fewer launch rounds and no launched Jacobian when every request is speculated -- with the
    solvers on the host and (round 4: the device engine has the cache too) on the device, and both engines agree bit for bit."""
    from socp_amd import capi, sweep
    ctx = make_ctx(variant, steps=200)
    ctx.set_params(sweep.GODDARD_PARAMS)
    sweep.goddard_single_shooting_problem(ctx)
    Z0 = sweep.goddard_starts(96, 1e-3)
    which = capi.SOLVER_HOST if solver == "host" else capi.SOLVER_DEVICE
    runs = {s: ctx.chains_solve(Z0, kind=0, xtol=1e-8, speculate=s, solver=which) for s in (0, 1, -1)}
    other = ctx.chains_solve(Z0, kind=0, xtol=1e-8, speculate=1, solver=capi.SOLVER_DEVICE if solver == "host" else capi.SOLVER_HOST)
    for key in ("z", "info", "nfev", "fnorm"):
        assert np.array_equal(other[key], runs[1][key]), key
    assert other["stats"]["rounds"] == runs[1]["stats"]["rounds"] and other["stats"]["jacobians_from_cache"] == runs[1]["stats"]["jacobians_from_cache"]
    for s in (1, -1):
        for key in ("z", "info", "nfev", "fnorm"):
            assert np.array_equal(runs[0][key], runs[s][key]), (s, key)
    assert runs[0]["stats"]["jacobians_from_cache"] == 0 and runs[0]["stats"]["jacobians_launched"] > 0
    assert runs[1]["stats"]["jacobians_launched"] == 0 and runs[1]["stats"]["jacobians_from_cache"] == runs[0]["stats"]["jacobians_launched"]
    assert runs[1]["stats"]["rounds"] < runs[0]["stats"]["rounds"]
    assert runs[-1]["stats"]["rounds"] == runs[1]["stats"]["rounds"]              # 96 starts: everything fits the idle SIMDs
    # multi-start through the old entry point = the same engine
    ms = ctx.multistart_solve(Z0, xtol=1e-8)
    assert np.array_equal(ms["z"], runs[0]["z"]) and np.array_equal(ms["info"], runs[0]["info"])
    ctx.close()


def test_round_limit_stops_stragglers_only():
    """max_rounds: chains still solving after the budget get info = -3 (the negative-callback abort of shooting.cpp:873);
    every chain that finished within the budget has exactly the result it has without a limit."""
    from socp_amd import sweep
    ctx = make_ctx("exact", steps=200)
    ctx.set_params(sweep.GODDARD_PARAMS)
    sweep.goddard_single_shooting_problem(ctx)
    Z0 = sweep.goddard_starts(64, 3e-3)               # wide enough that the starts need different numbers of rounds
    full = ctx.chains_solve(Z0, kind=0, xtol=1e-8)
    limit = int(np.sort(full["nfev"])[len(Z0) // 2] // 14 + 6)          # about the median start's number of rounds
    cut = ctx.chains_solve(Z0, kind=0, xtol=1e-8, max_rounds=limit)
    assert cut["stats"]["rounds"] == limit < full["stats"]["rounds"]
    stopped = cut["info"] == -3
    assert 0 < stopped.sum() < len(Z0)
    assert np.array_equal(cut["z"][~stopped], full["z"][~stopped]) and np.array_equal(cut["info"][~stopped], full["info"][~stopped])
    assert np.array_equal(cut["nfev"][~stopped], full["nfev"][~stopped])
    ctx.close()


def test_covid_program_continuations_as_chains():
    """The two continuations of the reference's tests/testCovid19.cpp -- SolveOCP(0.1) on the final target (R: 0.6 -> 0.7) and
    SolveOCP(0.01) on the horizon (tf: 30 -> 365 days, 100 homotopy steps) -- M = 20, n = 160, as boundary-data chains
    (shooting.cpp:598-692).  The chain that reproduces the program's own data must end on the program's unknowns bit for bit
    (covid19 has + - * / only: the C++ mirror's solves are the CPU path's); the chains beside it (other targets / horizons)
    must equal the sequential loop run for each of them alone."""
    from socp_amd import capi
    exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "covid_flow")
    out = subprocess.run([exe, "1e-8", "3"], capture_output=True, text=True, timeout=900, env=dict(os.environ, SOCP_VARIANT="exact"))
    prog = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [s["info"] for s in prog] == [1, 1, 1], out.stderr

    M, d, s = 20, 4, 8
    ctx = capi.Context(capi.MODEL_COVID19)
    ctx.set_params([3.4, 14, 5, 1, 0.1, 1, -10, 20])
    ctx.set_variant(capi.VARIANT_LANE_EXACT)
    mode_t = [capi.FIXED] + [capi.CONTINUOUS] * (M - 1) + [capi.FIXED]
    mode_x = np.zeros((M + 1, d), dtype=np.int32)
    mode_x[1:M] = capi.CONTINUOUS
    mode_x[M, :3] = capi.FREE
    Xi = np.array([0.93, 0.003, 0.01, 0.057, -0.001, 0.001, 0.0, 0.0])
    time30 = np.array([0.0 + i * (30.0 - 0.0) / M for i in range(M + 1)])
    X = np.zeros((M + 1, s))
    X[0] = Xi
    X[1:M] = ctx.integrate_batch(np.zeros(M - 1), time30[1:M], np.repeat(Xi[None, :], M - 1, axis=0))     # InitShooting's Move
    X[M, 3] = 0.6
    assert ctx.problem_set(mode_t, mode_x, time30, X) == 160

    def data(tf, target):
        T = time30.copy()
        T[M] = tf
        Xd = X.copy()
        Xd[M, 3] = target
        return T, Xd.ravel()

    # ---- stage 2: target continuation, step 0.1; chain 0 is the program's, the others aim elsewhere
    z1 = np.array(prog[0]["z"])
    targets = [0.7, 0.65, 0.72]
    Tp, Xp = data(30.0, 0.6)
    goal = [data(30.0, t) for t in targets]
    P = len(targets)
    res = ctx.chains_solve(np.tile(z1, (P, 1)), kind=2, step=0.1, time_prev=np.tile(Tp, (P, 1)), x_prev=np.tile(Xp, (P, 1)),
                           time_goal=np.array([g[0] for g in goal]), x_goal=np.array([g[1] for g in goal]), xtol=1e-8)
    # 11 solves, not 10: b = 0.1 + 0.1 + ... reaches 0.9999999999999999 after ten steps and the loop adds min(b + step, 1)
    assert np.all(res["info"] == 1) and np.all(res["solves"] == 11)
    assert np.array_equal(res["z"][0], np.array(prog[1]["z"])) and res["nfev"][0] == prog[1]["nfev"]
    for p in (1, 2):
        def set_b(b, p=p):
            ctx.problem_set(mode_t, mode_x, (1 - b) * Tp + b * goal[p][0], ((1 - b) * Xp + b * goal[p][1]).reshape(M + 1, s))
        q = sequential_chain(ctx, z1, 0.1, 1e-12, 1e-8, set_b)
        assert q["info"] == 1 and np.array_equal(res["z"][p], q["z"]) and res["nfev_total"][p] == q["nfev_total"], p
    # ---- stage 3: horizon continuation, step 0.01 (100 solves per chain); chain 0 is the program's
    z2 = np.array(prog[1]["z"])
    horizons = [365.0, 200.0]
    Tp, Xp = data(30.0, 0.7)
    goal = [data(h, 0.7) for h in horizons]
    P = len(horizons)
    ctx.problem_set(mode_t, mode_x, Tp, Xp.reshape(M + 1, s))
    res = ctx.chains_solve(np.tile(z2, (P, 1)), kind=2, step=0.01, time_prev=np.tile(Tp, (P, 1)), x_prev=np.tile(Xp, (P, 1)),
                           time_goal=np.array([g[0] for g in goal]), x_goal=np.array([g[1] for g in goal]), xtol=1e-8)
    assert np.all(res["info"] == 1) and np.all(res["solves"] >= 100) and np.all(res["b_reached"] == 1.0)
    assert np.array_equal(res["z"][0], np.array(prog[2]["z"])) and res["nfev"][0] == prog[2]["nfev"]
    ctx.close()


def test_interceptor_program_continuations_as_chains():
    """The reference's tests/testInterceptor.cpp (scenario 1): continuation on the model parameter mu_gft 0 -> 1, step 0.1
    (shooting.cpp:695-778), then on the boundary data towards the scenario, step 0.1, with a FREE final time
    (shooting.cpp:598-692) -- as chains.  The interceptor is a table-driven model with its own ComputeTraj and final rows: this
    drives the per-problem instantiations of ITS kernels.  The C++ mirror's program runs the same device arithmetic through the
    shared-parameter kernels, so the chain must end on its unknowns bit for bit."""
    from socp_amd import capi
    exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "interceptor_flow")
    out = subprocess.run([exe, "1e-8", "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ, SOCP_VARIANT="exact"))
    prog = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [s["stage"] for s in prog] == ["analytical_guess", "mu_gft_continuation", "scenario_continuation"], out.stderr
    assert [s["info"] for s in prog] == [1, 1, 1]

    RE = 6378145.0
    ctx = capi.Context(capi.MODEL_INTERCEPTOR)
    ctx.set_variant(capi.VARIANT_LANE_EXACT)
    names = capi.INTERCEPTOR_PARAM_NAMES
    MU = names.index("mu_gft")
    mode_t = [capi.FIXED, capi.FREE]
    mode_x = np.zeros((2, 6), dtype=np.int32)
    mode_x[1, 1] = capi.FREE
    X = np.zeros((2, 12))
    X[0, :6] = [1000, 1000, np.pi / 4, 0.0, 5454661 / RE, 46086 / RE]
    X[1, :6] = [6000, 1000, 0.01 * np.pi, 0.01 * np.pi, (5454661 + 27829.0) / RE, 46086 / RE]
    assert ctx.problem_set(mode_t, mode_x, [0.0, 10.0], X) == 13
    # ---- stage 2: mu_gft 0 -> 1, step 0.1, from the converged mu_gft = 0 solution; a second chain aims at 0.6
    z1 = np.array(prog[0]["z"])
    p0 = ctx.get_params()
    p0[MU] = 0.0
    goals = np.array([1.0, 0.6])
    res = ctx.chains_solve(np.tile(z1, (2, 1)), kind=1, param_index=MU, step=0.1, goal=goals, params=np.tile(p0, (2, 1)), xtol=1e-8)
    assert np.all(res["info"] == 1) and np.all(res["param_final"] == goals)
    assert np.array_equal(res["z"][0], np.array(prog[1]["z"])) and res["nfev"][0] == prog[1]["nfev"]
    # ---- stage 3: boundary data towards scenario 1, FREE tf; previous data = the stage-2 solution (GetSolution: final node =
    #      the final state of the stored trajectory)
    z2 = res["z"][0]
    ctx.set_param("mu_gft", 1.0)
    t_end = z2[12]
    X_end = ctx.integrate_batch(0.0, t_end, z2[None, :12])[0]
    Xp = np.zeros((2, 12))
    Xp[0], Xp[1] = z2[:12], X_end
    Tp = np.array([0.0, t_end])
    Xg = np.zeros((2, 12))
    Xg[0, :6] = [3000, 1000, -np.pi / 6, 0.0, 5454661 / RE, 46086 / RE]
    Xg[1, :6] = [12000, 1000, 0.0, np.pi / 8, 5475000 / RE, 42000 / RE]
    Tg = np.array([0.0, 20.0])
    ctx.problem_set(mode_t, mode_x, Tp, Xp)
    res3 = ctx.chains_solve(z2[None, :], kind=2, step=0.1, time_prev=Tp[None, :], x_prev=Xp.ravel()[None, :], time_goal=Tg[None, :],
                            x_goal=Xg.ravel()[None, :], xtol=1e-8)
    assert res3["info"][0] == 1 and res3["b_reached"][0] == 1.0
    assert np.array_equal(res3["z"][0], np.array(prog[2]["z"])) and res3["nfev"][0] == prog[2]["nfev"]
    ctx.close()


@pytest.mark.parametrize("order", [1, 0])
def test_double_integrator_program_continuations_as_chains(order):
    """tests/testDoubleIntegrator.cpp: SolveOCP(1.0) on the boundary data (target y: 15 -> 20), then SolveOCP(1.0, muT, 0.02), with
    the variational Jacobian (modelOrder 1, hybrj -- what the reference's test runs; `analytic_jac`) and with forward differences
    (modelOrder 0, hybrd).  No transcendental function anywhere: the C++ mirror's program IS the CPU path, and the chain must end on
    its unknowns, nfev and njev exactly.  A second chain with another goal runs beside it."""
    from socp_amd import capi
    exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "dint_flow")
    out = subprocess.run([exe, "basic", str(order), "1e-8"], capture_output=True, text=True, timeout=900)
    prog = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [s["stage"] for s in prog] == ["solve", "data_continuation", "muT_continuation"] and all(s["info"] == 1 for s in prog), out.stderr

    ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    mode_t = [capi.FIXED, capi.FREE]
    mode_x = np.zeros((2, 6), dtype=np.int32)
    X = np.zeros((2, 12))
    X[0, 6:] = 0.01
    X[1, 0], X[1, 1] = 10.0, 15.0
    T = np.array([0.0, 10.0])
    assert ctx.problem_set(mode_t, mode_x, T, X) == 13
    kw = dict(xtol=1e-8, analytic_jac=bool(order))
    # the program's first solve, as a one-solve chain
    z0 = np.concatenate([X[0], [10.0]])
    r1 = ctx.chains_solve(z0[None, :], kind=0, **kw)
    assert r1["info"][0] == 1 and np.array_equal(r1["z"][0], np.array(prog[0]["z"])) and r1["nfev"][0] == prog[0]["nfev"]
    if order:
        assert r1["njev"][0] == prog[0]["njev"]
    # data continuation: y target 15 -> 20 (chain 0, the program's) and 15 -> 12 (chain 1)
    Xg = np.tile(X.ravel(), (2, 1))
    Xg[0, 12 + 1] = 20.0
    Xg[1, 12 + 1] = 12.0
    z1 = np.array(prog[0]["z"])
    r2 = ctx.chains_solve(np.tile(z1, (2, 1)), kind=2, step=1.0, time_prev=np.tile(T, (2, 1)), x_prev=np.tile(X.ravel(), (2, 1)),
                          time_goal=np.tile(T, (2, 1)), x_goal=Xg, **kw)
    assert np.all(r2["info"] == 1)
    assert np.array_equal(r2["z"][0], np.array(prog[1]["z"])) and r2["nfev"][0] == prog[1]["nfev"]
    if order:
        assert r2["njev"][0] == prog[1]["njev"]
    assert not np.array_equal(r2["z"][1], r2["z"][0])
    # parameter continuation muT 0.01 -> 0.02 on the new boundary data
    ctx.problem_set(mode_t, mode_x, T, Xg[0].reshape(2, 12))
    z2 = np.array(prog[1]["z"])
    r3 = ctx.chains_solve(np.tile(z2, (2, 1)), kind=1, param_index=2, step=1.0, goal=np.array([0.02, 0.05]),
                          params=np.tile(ctx.get_params(), (2, 1)), **kw)
    assert r3["info"][0] == 1 and r3["param_final"][0] == 0.02
    assert np.array_equal(r3["z"][0], np.array(prog[2]["z"])) and r3["nfev"][0] == prog[2]["nfev"]
    if order:
        assert r3["njev"][0] == prog[2]["njev"]
    ctx.close()


def test_batched_variational_jacobian_equals_one_problem_at_a_time():
    """socp_var_jacobian_multi_dev through the chain engine's per-problem blocks: np problems with their own parameters and
    boundary data in one launch == socp_var_jacobian of each problem alone, bit for bit (WP layout: FREE interior time, FIXED /
    CONTINUOUS interior modes, M = 5)."""
    import torch
    from socp_amd import capi
    ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    M = 5
    mode_t = [capi.FIXED] + [capi.FREE] * M
    mode_x = np.zeros((M + 1, 6), dtype=np.int32)
    mode_x[1:M, 3:6] = capi.CONTINUOUS
    X = np.zeros((M + 1, 12))
    X[:, 0] = 20.0 * np.arange(M + 1) / M
    X[:M, 6:] = 0.001
    tn = 60.0 * np.arange(M + 1) / M
    n = ctx.problem_set(mode_t, mode_x, tn, X)
    rng = np.random.default_rng(2)
    P = 7
    Z = np.tile(np.concatenate([X[:M].ravel(), tn[1:]]), (P, 1)) * (1 + 1e-2 * rng.uniform(-1, 1, (P, n)))
    params = np.tile(np.concatenate([ctx.get_params(), [0.0, 0.0]]), (P, 1))
    params[:, 1] = 1.0 + 0.1 * np.arange(P)            # a_max per problem
    params[:, 2] = 0.01 * (1 + np.arange(P))            # muT per problem
    dev = torch.device("cuda", 0)
    dZ = torch.from_numpy(Z).to(dev)
    dP = torch.from_numpy(params).to(dev)
    dJ = torch.empty((P, n * n), dtype=torch.float64, device=dev)
    L = capi.lib()
    assert L.socp_problem_set_blocks_dev(ctx.h, dP.data_ptr(), params.shape[1], None, None) == 0
    assert L.socp_var_jacobian_multi_dev(ctx.h, P, dZ.data_ptr(), dJ.data_ptr()) == 0
    ctx.synchronize()
    L.socp_problem_set_blocks_dev(ctx.h, None, 0, None, None)
    got = dJ.cpu().numpy().reshape(P, n, n).transpose(0, 2, 1)              # column-major -> J[row, col]
    for q in range(P):
        ctx.set_params(params[q, :3])
        assert np.array_equal(got[q], ctx.var_jacobian(Z[q])), q
    ctx.close()


@pytest.mark.parametrize("order", [1, 0])
def test_double_integrator_waypoint_program_continuations_as_chains(order):
    """tests/testDoubleIntegrator_WP.cpp (M = 2, FREE interior and final times, FIXED positions / CONTINUOUS velocities at the
    way-point): its first solve does not converge and the reference ignores that (the unknowns stay at the initial guess,
    shooting.cpp:588-592), then SolveOCP(1.0) moves the way-point data and SolveOCP(1.0, muT, 0.02) the parameter."""
    from socp_amd import capi
    exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "dint_flow")
    out = subprocess.run([exe, "wp", str(order), "1e-8"], capture_output=True, text=True, timeout=900)
    prog = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [s["stage"] for s in prog] == ["solve", "data_continuation", "muT_continuation"], out.stderr
    assert prog[0]["info"] != 1 and prog[1]["info"] == 1 and prog[2]["info"] == 1

    M = 2
    ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    mode_t = [capi.FIXED, capi.FREE, capi.FREE]
    mode_x = np.zeros((M + 1, 6), dtype=np.int32)
    mode_x[1, 3:6] = capi.CONTINUOUS
    vt = np.array([60.0 * i / M for i in range(M + 1)])
    vX = np.zeros((M + 1, 12))
    vX[:, 0] = [20.0 * i / M for i in range(M + 1)]
    vX[:M, 6:] = 0.001
    assert ctx.problem_set(mode_t, mode_x, vt, vX) == 26
    z0 = np.concatenate([vX[:M].ravel(), vt[1:]])
    kw = dict(xtol=1e-8, analytic_jac=bool(order))
    r1 = ctx.chains_solve(z0[None, :], kind=0, **kw)
    assert r1["info"][0] == prog[0]["info"] and r1["nfev"][0] == prog[0]["nfev"]          # the same failure, after the same work
    # the program's unknowns are still the initial guess (GetParameters after a failed SolveOCP)
    assert np.array_equal(np.array(prog[0]["z"]), z0)
    Xg = vX.copy()
    Xg[1, 1], Xg[2, 1], Xg[2, 2] = 15.0, 5.0, 10.0
    r2 = ctx.chains_solve(z0[None, :], kind=2, step=1.0, time_prev=vt[None, :], x_prev=vX.ravel()[None, :], time_goal=vt[None, :],
                          x_goal=Xg.ravel()[None, :], **kw)
    assert r2["info"][0] == 1 and np.array_equal(r2["z"][0], np.array(prog[1]["z"])) and r2["nfev"][0] == prog[1]["nfev"]
    ctx.problem_set(mode_t, mode_x, vt, Xg)
    r3 = ctx.chains_solve(r2["z"], kind=1, param_index=2, step=1.0, goal=np.array([0.02]), params=ctx.get_params()[None, :], **kw)
    assert r3["info"][0] == 1 and np.array_equal(r3["z"][0], np.array(prog[2]["z"])) and r3["nfev"][0] == prog[2]["nfev"]
    if order:
        assert r2["njev"][0] == prog[1]["njev"] and r3["njev"][0] == prog[2]["njev"]
    ctx.close()


def test_groups_take_turns_without_changing_an_iterate(monkeypatch):
    """The engine splits the chains into groups whose launches and host work overlap (SOCP_CHAINS_GROUPS; 2 from 2048 chains
    up): a chain only ever meets its own group, so solutions, counts and the continuation bookkeeping do not depend on G."""
    from socp_amd import sweep
    ctx = make_ctx("exact", steps=100)
    goddard_m6(ctx)
    P = 11
    rng = np.random.default_rng(9)
    Z0 = np.tile(STAGE2_INIT, (P, 1))
    Z0[:, 7:14] *= 1 + 1e-3 * rng.uniform(-1, 1, (P, 7))
    goals = np.linspace(150.0, 900.0, P)
    runs = {}
    for G in ("1", "2", "3"):
        monkeypatch.setenv("SOCP_CHAINS_GROUPS", G)
        runs[G] = ctx.chains_solve(Z0, kind=1, param_index=KD, step=0.5, goal=goals, params=np.tile(PARAMS0, (P, 1)), xtol=1e-6)
    for G in ("2", "3"):
        for key in ("z", "info", "nfev", "nfev_total", "solves", "b_reached", "param_final", "fnorm"):
            assert np.array_equal(runs["1"][key], runs[G][key]), (G, key)
    assert np.all(runs["1"]["info"] == 1) and np.all(runs["1"]["solves"] >= 2)          # step 0.5: two solves, more where a step was bisected
    # single shooting with speculative Jacobians, groups on: same again
    ctx.set_params(sweep.GODDARD_PARAMS)
    sweep.goddard_single_shooting_problem(ctx)
    Zs = sweep.goddard_starts(50, 1e-3)
    ref = None
    for G in ("1", "2"):
        monkeypatch.setenv("SOCP_CHAINS_GROUPS", G)
        r = ctx.chains_solve(Zs, kind=0, xtol=1e-8, speculate=1)
        assert r["stats"]["jacobians_launched"] == 0
        if ref is None:
            ref = r
        else:
            assert np.array_equal(ref["z"], r["z"]) and np.array_equal(ref["nfev"], r["nfev"]) and np.array_equal(ref["info"], r["info"])
    ctx.close()


def test_throughput_flavour_chains_within_north_star_tolerance():
    """north_star: converged results within 1e-8 relative of the reference path.  The KD continuation chains in the throughput
    flavour (restructured arithmetic, per-problem general-law kernels) against the reference-order flavour, at a solver tolerance
    where the root is defined that sharply (xtol = 1e-12, as tests/test_host_flow.py does for single solves)."""
    goals = np.array([310.0, 200.0, 450.0, 800.0])
    P = len(goals)
    Z0 = np.tile(STAGE2_INIT, (P, 1))
    out = {}
    for variant in ("exact", "fast"):
        ctx = make_ctx(variant)
        goddard_m6(ctx)
        out[variant] = ctx.chains_solve(Z0, kind=1, param_index=KD, step=0.5, goal=goals, params=np.tile(PARAMS0, (P, 1)), xtol=1e-12)
        ctx.close()
    assert np.all(out["exact"]["info"] == 1) and np.all(out["fast"]["info"] == 1)
    rel = np.max(np.abs(out["fast"]["z"] - out["exact"]["z"]), axis=1) / np.max(np.abs(out["exact"]["z"]), axis=1)
    assert np.all(rel <= 1e-8), rel
    assert np.array_equal(out["fast"]["solves"], out["exact"]["solves"])


def test_chain_engine_edge_cases():
    """Empty batch, one chain, bad options, and a chain whose goal equals its start (b = 1 at once: a single solve)."""
    import ctypes as C
    from socp_amd import capi
    ctx = make_ctx("exact")
    goddard_m6(ctx)
    r = ctx.chains_solve(np.zeros((0, 85)), kind=1, param_index=KD, step=1.0, goal=np.zeros(0), params=np.zeros((0, 8)), xtol=1e-6)
    assert r["z"].shape == (0, 85) and r["stats"]["rounds"] == 0
    one = ctx.chains_solve(STAGE2_INIT[None, :], kind=1, param_index=KD, step=1.0, goal=[0.0], params=np.array(PARAMS0)[None, :], xtol=1e-6)
    assert one["info"][0] == 1 and one["solves"][0] == 1 and one["param_final"][0] == 0.0 and one["b_reached"][0] == 1.0
    with pytest.raises(capi.SocpError):
        ctx.chains_solve(STAGE2_INIT[None, :], kind=1, param_index=99, step=1.0, goal=[1.0], xtol=1e-6)          # no such parameter
    with pytest.raises(capi.SocpError):
        ctx.chains_solve(STAGE2_INIT[None, :], kind=1, param_index=KD, step=0.0, goal=[1.0], xtol=1e-6)          # step must be > 0
    with pytest.raises(capi.SocpError):
        ctx.chains_solve(STAGE2_INIT[None, :], kind=2, step=1.0, xtol=1e-6)                                      # data homotopy without data
    with pytest.raises(capi.SocpError):
        ctx.chains_solve(STAGE2_INIT[None, :], kind=0, xtol=1e-6, analytic_jac=True)                             # goddard has no variational equations
    # the context is still usable after the refusals
    assert np.all(np.isfinite(ctx.residual(STAGE2_INIT)))
    ctx.close()


def test_a_nan_start_ends_its_own_chain_and_nothing_else():
    """NaNs propagate silently in the reference: a chain that starts on NaN must come to an end by MINPACK's own
    rules (info 4 / 5 after the slow-progress counters run out) without touching its neighbours' iterates."""
    ctx = make_ctx("exact")
    goddard_m6(ctx)
    Z0 = np.tile(STAGE2_INIT, (3, 1))
    Z0[1, 20] = np.nan
    goals = np.array([310.0, 310.0, 310.0])
    r = ctx.chains_solve(Z0, kind=1, param_index=KD, step=1.0, step_min=1e-3, goal=goals, params=np.tile(PARAMS0, (3, 1)), xtol=1e-6, max_rounds=400)
    assert r["info"][0] == 1 and r["info"][2] == 1 and r["info"][1] != 1
    assert np.array_equal(r["z"][0], r["z"][2])
    alone = ctx.chains_solve(Z0[:1], kind=1, param_index=KD, step=1.0, goal=goals[:1], params=np.array(PARAMS0)[None, :], xtol=1e-6)
    assert np.array_equal(alone["z"][0], r["z"][0]) and alone["nfev_total"][0] == r["nfev_total"][0]
    ctx.close()


def test_config5_interceptor_solve_sweep():
    """BASELINE config 5 as a SOLVE sweep (python -m socp_amd.sweep --model interceptor): interceptor, adaptive Dormand-Prince,
    M = 21, n = 253 -- a handful of starts around the converged scenario-1 trajectory, full Newton solves in lock-step, all on
    the same root; the throughput flavour agrees with the reference-order flavour within north_star's 1e-8."""
    from socp_amd import capi, sweep
    roots = {}
    for variant in ("exact", "fast"):
        ctx = capi.Context(capi.MODEL_INTERCEPTOR)
        ctx.set_variant(capi.VARIANT_LANE_EXACT if variant == "exact" else capi.VARIANT_LANE_FAST)
        n, z = sweep.interceptor_config5_problem(ctx)
        assert n == 253
        # the adaptive integrator's step-size decisions put noise of the order of its tolerance into the FD Jacobian: the solver
        # tolerance has to stay above it (xtol 1e-9 over an ODE tolerance of 1e-11)
        ctx.set_integrator(capi.INT_DOPRI5, 1e-11)
        rng = np.random.default_rng(4)
        Z0 = np.tile(z, (6, 1))
        Z0[:, 6:12] *= 1 + 1e-3 * rng.uniform(-1, 1, (6, 6))
        r = ctx.chains_solve(Z0, kind=0, xtol=1e-9)
        assert np.all(r["info"] == 1), r["info"]
        spread = np.max(np.abs(r["z"] - r["z"][0])) / np.max(np.abs(r["z"][0]))
        assert spread < 1e-8, spread
        roots[variant] = r["z"][0]
        ctx.close()
    assert np.max(np.abs(roots["fast"] - roots["exact"])) / np.max(np.abs(roots["exact"])) <= 1e-8


@pytest.mark.parametrize("solver", ["host", "device"])
def test_speculative_rows_with_continuation_chains_that_restart_at_different_times(solver):
    """Speculative FD rows and chains of SEVERAL solves each (round 4: with the solvers on the device a restarted chain's start list
    shared a pinned buffer with the advance list, which differs from it exactly when cached Jacobians are waiting -- chains were
    restarted from the wrong list and their iterates changed; one chain alone, or no speculation, hid it): nine KD chains with their
    own goals, one of them bisecting 50 times, every output equal with the rows never / always / automatically speculated."""
    from socp_amd import capi
    ctx = make_ctx("exact")
    goddard_m6(ctx)
    goals = np.array([310.0, 250.0, 400.0, 310.0, 120.0, 5000.0, 310.0, 600.0, 280.0])
    P = len(goals)
    Z0 = np.tile(STAGE2_INIT, (P, 1))
    Z0[3, 7:14] *= 1 + 1e-6
    which = capi.SOLVER_HOST if solver == "host" else capi.SOLVER_DEVICE
    kw = dict(kind=capi.CHAIN_PARAM, param_index=KD, step=0.4, goal=goals, params=np.tile(PARAMS0, (P, 1)), xtol=1e-6, solver=which)
    runs = {s: ctx.chains_solve(Z0, speculate=s, **kw) for s in (0, 1, -1)}
    for s in (1, -1):
        for key in ("z", "info", "nfev", "nfev_total", "solves", "b_reached", "param_final", "fnorm"):
            assert np.array_equal(runs[0][key], runs[s][key], equal_nan=True), (s, key)
    assert runs[1]["stats"]["jacobians_from_cache"] > 0 and np.max(runs[0]["solves"]) > 20
    ctx.close()


def test_engines_agree_on_seeded_random_mixtures_of_chains(monkeypatch):
    """Differential test of the lock-step engines' bookkeeping (request lists, restarts, cached rows, groups): seeded random
    mixtures of KD continuation chains -- numbers of chains, goals (some unreachable: dozens of bisections), continuation steps --
    solved by the host engine without speculative rows (the baseline: bit-equal to the sequential loop, tests above) and by the
    device engine with the rows never / always / automatically speculated and the chains in one, two or three groups.  Every
    output of every chain must be the baseline's, bit for bit."""
    from socp_amd import capi
    ctx = make_ctx("exact")
    goddard_m6(ctx)
    rng = np.random.default_rng(20250905)
    keys = ("z", "info", "nfev", "nfev_total", "solves", "b_reached", "param_final", "fnorm")
    for trial in range(6):
        P = int(rng.integers(2, 24))
        goals = rng.choice([120.0, 250.0, 280.0, 310.0, 400.0, 600.0, 2500.0], size=P)
        Z0 = np.tile(STAGE2_INIT, (P, 1))
        Z0[:, 7:14] *= 1 + 1e-6 * rng.uniform(-1, 1, (P, 7))
        step = float(rng.choice([1.0, 0.4, 0.25]))
        kw = dict(kind=capi.CHAIN_PARAM, param_index=KD, step=step, step_min=1e-2, goal=goals, params=np.tile(PARAMS0, (P, 1)), xtol=1e-6)
        monkeypatch.setenv("SOCP_CHAINS_DEVICE_GROUPS", "1")
        base = ctx.chains_solve(Z0, solver=capi.SOLVER_HOST, speculate=0, **kw)
        for speculate, groups in ((0, "1"), (1, "1"), (-1, "2"), (1, "3"), (0, "2")):
            monkeypatch.setenv("SOCP_CHAINS_DEVICE_GROUPS", groups)
            r = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE, speculate=speculate, **kw)
            for k in keys:
                assert np.array_equal(base[k], r[k], equal_nan=True), (trial, P, step, speculate, groups, k)
        r = ctx.chains_solve(Z0, solver=capi.SOLVER_HOST, speculate=1, **kw)
        for k in keys:
            assert np.array_equal(base[k], r[k], equal_nan=True), (trial, "host speculate=1", k)
    ctx.close()
