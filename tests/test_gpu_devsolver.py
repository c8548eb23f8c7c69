"""GPU: the lock-step engine with the Newton solvers ON THE DEVICE (socp_chain_options.solver = SOCP_SOLVER_DEVICE:
kernels_solver.hip, one workgroup per chain) against the same engine with the solvers on the host.  The bar is the one every
engine feature has: no iterate changes -- unknowns, info, evaluation counts, homotopy state and |F| of every chain equal TO THE
LAST BIT, for plain multi-start solves, both continuation kinds (bisection after failed solves included), forward-difference
and variational Jacobians, fixed-step and adaptive integration, n from 14 to 253.  (The arithmetic of the device solver is
pinned on the CPU by tests/test_devsolver_sim.py; this file checks the many-thread kernels and the engine around them.)"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = ("z", "info", "nfev", "nfev_total", "njev", "solves", "b_reached", "param_final", "fnorm")


def both(ctx, Z0, spec=True, **kw):
    from socp_amd import capi
    host = ctx.chains_solve(Z0, solver=capi.SOLVER_HOST, speculate=0, **kw)
    dev = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE, speculate=0, **kw)      # (speculative rows, both engines: tests/test_gpu_chains.py)
    for k in KEYS:
        assert np.array_equal(host[k], dev[k], equal_nan=True), k
    assert host["stats"]["rounds"] == dev["stats"]["rounds"]
    assert host["stats"]["jacobians_launched"] == dev["stats"]["jacobians_launched"]
    # ... and with every residual request evaluated as a forward-difference batch (the cached-Jacobian path, restarts included:
    # the round-4 bug of the restart list showed only with several chains of several solves each, which some callers of this are)
    # (not under a tight round budget: a cached Jacobian saves a round, so the budget ends elsewhere)
    if spec:
        rows = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE, speculate=1, **kw)
        for k in KEYS:
            assert np.array_equal(host[k], rows[k], equal_nan=True), ("speculate=1", k)
    return host, dev


@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_single_shooting_sweep_n14(variant):
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(200)
    ctx.set_variant(capi.VARIANT_LANE_EXACT if variant == "exact" else capi.VARIANT_LANE_FAST)
    sweep.goddard_single_shooting_problem(ctx)
    Z0 = sweep.goddard_starts(96, 3e-3)               # wide enough for failures (info 4 / 5) beside converged starts
    host, dev = both(ctx, Z0, kind=capi.CHAIN_PLAIN, xtol=1e-8)
    assert (dev["info"] == 1).sum() >= 60 and (dev["info"] != 1).sum() >= 1
    ctx.close()


@pytest.mark.parametrize("M,P", [(6, 70), (9, 40)])
def test_multiple_shooting_sweeps(M, P):
    """The testGoddard layout with M segments (n = 85, 127): Jacobian refreshes of thousands of columns, Broyden updates,
    every start on the CPU path's root."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    n = sweep.goddard_multiple_shooting_problem(ctx, M)
    assert n == 14 * M + 1
    Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(P, 0.05), M)
    host, dev = both(ctx, Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10)
    assert np.all(dev["info"] == 1)
    ctx.close()


def test_kd_continuation_chains_with_bisection():
    """testGoddard's drag continuation as chains with their own goals (one far enough that solves fail and the step is halved):
    restarts, per-chain parameter blocks and the bisection logic around the device solvers."""
    from socp_amd import capi
    from test_gpu_chains import STAGE2_INIT, PARAMS0, KD, goddard_m6, make_ctx
    ctx = make_ctx("exact")
    goddard_m6(ctx)
    goals = np.array([310.0, 250.0, 400.0, 310.0, 120.0, 5000.0, 310.0, 600.0])
    P = len(goals)
    Z0 = np.tile(STAGE2_INIT, (P, 1))
    Z0[3, 7:14] *= 1 + 1e-6
    for step in (1.0, 0.4):
        host, dev = both(ctx, Z0, kind=capi.CHAIN_PARAM, param_index=KD, step=step, goal=goals, params=np.tile(PARAMS0, (P, 1)), xtol=1e-6)
        assert np.all(dev["info"][[0, 1, 2, 3, 4]] == 1)
    assert np.max(dev["solves"]) > 1
    gold2 = [g for g in json.load(open(os.path.join(ROOT, "tests", "golden", "goddard_flow.json")))["goddard_single_stage"]
             if g["stage"] == 2 and g["xtol"] == 1e-6][0]
    r = ctx.chains_solve(Z0[:1], kind=capi.CHAIN_PARAM, param_index=KD, step=1.0, goal=goals[:1], params=np.array(PARAMS0)[None, :], xtol=1e-6,
                         solver=capi.SOLVER_DEVICE)
    assert np.array_equal(r["z"][0], np.array(gold2["z"])) and r["nfev"][0] == gold2["nfev"]      # testGoddard's own stage 2, CPU golden
    ctx.close()


def test_boundary_data_chains():
    from socp_amd import capi
    from test_gpu_chains import GOLD, goddard_m6, make_ctx
    ctx = make_ctx("exact")
    ctx.set_param("KD", 310.0)
    mode_t, mode_x, time, X = goddard_m6(ctx)
    z_conv = np.array(GOLD["goddard_N10_M6"][1]["z"])
    goals = np.array([1.0102, 1.0105, 1.011, 1.03])
    P = len(goals)
    Xp = np.tile(X.ravel(), (P, 1))
    Xg = Xp.copy()
    Xg[:, 6 * 14] = goals
    Tp = np.tile(time, (P, 1))
    for step in (1.0, 0.5):
        both(ctx, np.tile(z_conv, (P, 1)), kind=capi.CHAIN_DATA, step=step, time_prev=Tp, x_prev=Xp, time_goal=Tp, x_goal=Xg, xtol=1e-6)
    ctx.close()


@pytest.mark.parametrize("order", [1, 0])
def test_double_integrator_chains_hybrj_and_hybrd(order):
    """The way-point layout with M = 5 segments (FREE interior and final times, n = 65): hybrj chains with the batched variational
    Jacobian, and the same with forward differences, per-chain a_max / muT."""
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
    z = np.concatenate([X[:M].ravel(), tn[1:]])
    P = 6
    rng = np.random.default_rng(order)
    Z0 = np.tile(z, (P, 1))
    Z0[:, 6:12] *= 1 + 0.2 * rng.uniform(-1, 1, (P, 6))
    params = np.tile(ctx.get_params(), (P, 1))
    params[:, 2] = 0.01 * (1 + 0.3 * np.arange(P))
    host, dev = both(ctx, Z0, kind=capi.CHAIN_PLAIN, params=params, xtol=1e-8, analytic_jac=bool(order))
    if order:
        assert np.all(dev["njev"] >= 1)
    assert n == 65
    ctx.close()


def test_interceptor_adaptive_n253():
    """BASELINE config 5's layout: 253 unknowns, adaptive integrator, table-driven model -- 4 threads' worth of columns per
    workgroup (256), Jacobians of 64 009 entries scattered into the solver matrices on the device."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_INTERCEPTOR)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    n, z = sweep.interceptor_config5_problem(ctx)
    ctx.set_integrator(capi.INT_DOPRI5, 1e-11)
    rng = np.random.default_rng(4)
    Z0 = np.tile(z, (8, 1))
    Z0[:, 6:12] *= 1 + 1e-3 * rng.uniform(-1, 1, (8, 6))
    host, dev = both(ctx, Z0, kind=capi.CHAIN_PLAIN, xtol=1e-9)
    assert n == 253 and np.all(dev["info"] == 1)
    ctx.close()


def test_nan_start_round_limit_and_refusals():
    from socp_amd import capi
    from test_gpu_chains import STAGE2_INIT, PARAMS0, KD, goddard_m6, make_ctx
    ctx = make_ctx("exact")
    goddard_m6(ctx)
    Z0 = np.tile(STAGE2_INIT, (3, 1))
    Z0[1, 20] = np.nan
    goals = np.array([310.0, 310.0, 310.0])
    kw = dict(kind=capi.CHAIN_PARAM, param_index=KD, step=1.0, step_min=1e-3, goal=goals, params=np.tile(PARAMS0, (3, 1)), xtol=1e-6)
    host, dev = both(ctx, Z0, max_rounds=400, **kw)
    assert dev["info"][0] == 1 and dev["info"][2] == 1 and dev["info"][1] != 1
    # a round budget: the stragglers stop with info = -3, everyone else is untouched
    full = ctx.chains_solve(Z0[[0, 2]], solver=capi.SOLVER_DEVICE, kind=capi.CHAIN_PARAM, param_index=KD, step=1.0, goal=goals[:2],
                            params=np.tile(PARAMS0, (2, 1)), xtol=1e-6)
    cut_h, cut_d = both(ctx, Z0[[0, 2]], spec=False, kind=capi.CHAIN_PARAM, param_index=KD, step=1.0, goal=goals[:2], params=np.tile(PARAMS0, (2, 1)),
                        xtol=1e-6, max_rounds=3)
    assert np.all(cut_d["info"] == -3) and full["stats"]["rounds"] > 3
    with pytest.raises(capi.SocpError):
        ctx.chains_solve(Z0, solver=7, **kw)
    # an empty batch, and the context is still good for a residual afterwards
    r = ctx.chains_solve(np.zeros((0, 85)), solver=capi.SOLVER_DEVICE, kind=capi.CHAIN_PLAIN, xtol=1e-6)
    assert r["z"].shape == (0, 85)
    assert np.all(np.isfinite(ctx.residual(STAGE2_INIT)))
    ctx.close()


def test_auto_picks_the_device_for_large_sweeps_only():
    """socp_chain_options.solver = AUTO: the host engine for small sweeps, the device engine where the host side is the bottleneck
    (P n^2 >= 1.6e6; from 4e5 for n <= 32) -- and, on a reference-order context, the same iterates either way."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    sweep.goddard_single_shooting_problem(ctx)
    small = ctx.chains_solve(sweep.goddard_starts(64, 1e-3), kind=capi.CHAIN_PLAIN, xtol=1e-8)
    assert small["stats"]["jacobians_from_cache"] > 0 and small["stats"]["wall_ms"] > 0
    # (reference-order context: AUTO's device choice is the bit-equal solver; on a throughput-flavour context it is the matrix-core
    # factorisation, whose iterates differ at rounding level -- tests/test_gpu_factor_fast.py)
    ctx.set_variant(capi.VARIANT_LANE_EXACT)
    sweep.goddard_multiple_shooting_problem(ctx, 6)
    Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(300, 0.05), 6)
    big = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-8)
    assert np.all(big["info"] == 1)
    forced = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-8, solver=capi.SOLVER_HOST)
    assert np.array_equal(forced["z"], big["z"]) and np.array_equal(forced["nfev"], big["nfev"])
    ctx.close()


@pytest.mark.parametrize("M,order", [(64, 1), (64, 0), (90, 1)])
def test_large_problems_on_the_device_solver(M, order):
    """BASELINE config 3's layout (doubleIntegrator way-points, n = 13 M: 832 unknowns = 13 wavefronts per workgroup) and one
    beyond the workgroup size (M = 90: n = 1170 > 1024 threads, so a thread owns more than one column and the LDS vectors
    exceed 64 KB): a few chains, host solvers vs device solvers, bit for bit -- hybrj with the batched variational Jacobian and
    hybrd with forward differences."""
    from socp_amd import capi
    ctx = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    mode_t = [capi.FIXED] + [capi.FREE] * M
    mode_x = np.zeros((M + 1, 6), dtype=np.int32)
    mode_x[1:M, 3:6] = capi.CONTINUOUS
    X = np.zeros((M + 1, 12))
    X[:, 0] = 20.0 * np.arange(M + 1) / M
    X[:M, 6:] = 0.001
    tn = 60.0 * np.arange(M + 1) / M
    n = ctx.problem_set(mode_t, mode_x, tn, X)
    assert n == 13 * M
    z = np.concatenate([X[:M].ravel(), tn[1:]])
    rng = np.random.default_rng(M + order)
    Z0 = np.tile(z, (3, 1))
    Z0[:, 6:12] *= 1 + 0.1 * rng.uniform(-1, 1, (3, 6))
    host, dev = both(ctx, Z0, spec=False, kind=capi.CHAIN_PLAIN, xtol=1e-8, analytic_jac=bool(order), max_rounds=12)
    assert np.all(np.isfinite(dev["z"]))
    ctx.close()


@pytest.mark.parametrize("case", ["n14", "n85", "n127", "n253"])
def test_blocked_factor_kernel_changes_no_bit(case):
    """The LDS-blocked factor kernel (one wavefront per problem; panels of 8 reflectors, 64-column blocks: solver_dev.hpp
    factor_blocked) against the in-place factorisation inside the advance kernel and against the host solvers: identical chains.
    The switch is by size (n >= 128); here it is forced on (SOCP_SOLVER_BLOCKED_MIN_N=1) and off (=0) for sizes on both sides."""
    from socp_amd import capi, sweep
    if case == "n253":
        ctx = capi.Context(capi.MODEL_INTERCEPTOR)
        ctx.set_variant(capi.VARIANT_LANE_FAST)
        n, z = sweep.interceptor_config5_problem(ctx)
        ctx.set_integrator(capi.INT_DOPRI5, 1e-11)
        rng = np.random.default_rng(9)
        Z0 = np.tile(z, (5, 1))
        Z0[:, 6:12] *= 1 + 1e-3 * rng.uniform(-1, 1, (5, 6))
        kw = dict(kind=capi.CHAIN_PLAIN, xtol=1e-9)
    else:
        ctx = capi.Context(capi.MODEL_GODDARD)
        ctx.set_params(sweep.GODDARD_PARAMS)
        ctx.set_step_number(10)
        ctx.set_variant(capi.VARIANT_LANE_FAST)
        if case == "n14":
            sweep.goddard_single_shooting_problem(ctx)
            Z0 = sweep.goddard_starts(40, 3e-3)
        else:
            M = 6 if case == "n85" else 9
            sweep.goddard_multiple_shooting_problem(ctx, M)
            Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(24, 0.05), M)
        kw = dict(kind=capi.CHAIN_PLAIN, xtol=1e-10)
    old = os.environ.get("SOCP_SOLVER_BLOCKED_MIN_N")
    try:
        os.environ["SOCP_SOLVER_BLOCKED_MIN_N"] = "1"
        host, blocked = both(ctx, Z0, **kw)
        os.environ["SOCP_SOLVER_BLOCKED_MIN_N"] = "0"
        inplace = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE, **kw)
    finally:
        if old is None:
            os.environ.pop("SOCP_SOLVER_BLOCKED_MIN_N", None)
        else:
            os.environ["SOCP_SOLVER_BLOCKED_MIN_N"] = old
    for k in KEYS:
        assert np.array_equal(blocked[k], inplace[k], equal_nan=True), k
    ctx.close()


@pytest.mark.parametrize("case", ["n85", "n127", "n253"])
def test_workgroup_size_changes_no_bit(case, monkeypatch):
    """The solver launches may use fewer threads than columns (a thread then owns several columns; the trial-step launches do so
    by themselves when a launch does not fit the chip at full size, kernels_solver.hip: launch_advance): one wavefront per problem
    forced for both kinds of launch, and two, against the default size and against the host solvers -- identical chains."""
    from socp_amd import capi, sweep
    if case == "n253":
        ctx = capi.Context(capi.MODEL_INTERCEPTOR)
        ctx.set_variant(capi.VARIANT_LANE_FAST)
        n, z = sweep.interceptor_config5_problem(ctx)
        ctx.set_integrator(capi.INT_DOPRI5, 1e-11)
        rng = np.random.default_rng(11)
        Z0 = np.tile(z, (5, 1))
        Z0[:, 6:12] *= 1 + 1e-3 * rng.uniform(-1, 1, (5, 6))
        kw = dict(kind=capi.CHAIN_PLAIN, xtol=1e-9)
    else:
        ctx = capi.Context(capi.MODEL_GODDARD)
        ctx.set_params(sweep.GODDARD_PARAMS)
        ctx.set_step_number(10)
        ctx.set_variant(capi.VARIANT_LANE_FAST)
        M = 6 if case == "n85" else 9
        sweep.goddard_multiple_shooting_problem(ctx, M)
        Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(24, 0.05), M)
        kw = dict(kind=capi.CHAIN_PLAIN, xtol=1e-10)
    host, dev = both(ctx, Z0, **kw)
    for threads in ("64", "128"):
        monkeypatch.setenv("SOCP_SOLVER_THREADS_FACTOR", threads)
        monkeypatch.setenv("SOCP_SOLVER_THREADS_TRIAL", threads)
        small = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE, **kw)
        for k in KEYS:
            assert np.array_equal(small[k], dev[k], equal_nan=True), (k, threads)
    ctx.close()


def test_trial_launches_that_exceed_the_chip_shrink_their_workgroups():
    """4096 problems of n = 85 at two wavefronts each do not fit the chip's 3072 resident wavefronts: the trial-step launches then
    run one wavefront per problem (launch_advance's rule, no switch set).  Same chains as the host solvers, bit for bit."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    sweep.goddard_multiple_shooting_problem(ctx, 6)
    Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(4096, 0.05), 6)
    host, dev = both(ctx, Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10)
    assert np.sum(dev["info"] == 1) >= 4000
    ctx.close()


def test_kept_workspace_changes_no_result():
    """The device engine keeps its arenas between calls (include/socp_solver.h, socp_workspace_*): a call that runs in a block an
    earlier, larger, different sweep left behind gives the bits of a call on fresh memory; release returns what was kept."""
    from socp_amd import capi, sweep
    capi.workspace_release()
    assert capi.workspace_cached_bytes() == 0
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    n = sweep.goddard_multiple_shooting_problem(ctx, 6)
    Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(48, 0.05), 6)
    kw = dict(kind=capi.CHAIN_PLAIN, xtol=1e-10)
    fresh = {s: ctx.chains_solve(Z0[:16], solver=s, **kw) for s in (capi.SOLVER_DEVICE, capi.SOLVER_DEVICE_FAST)}
    small = capi.workspace_cached_bytes()
    assert small > 16 * n * n * 8
    big = ctx.chains_solve(Z0 * (1 + 1e-3), solver=capi.SOLVER_DEVICE, **kw)             # a larger sweep of other numbers: the block grows
    grown = capi.workspace_cached_bytes()
    assert grown > small and np.all(big["info"] == 1)
    for s, ref in fresh.items():
        again = ctx.chains_solve(Z0[:16], solver=s, **kw)                                 # now in the big block's leftovers
        for k in KEYS:
            assert np.array_equal(ref[k], again[k], equal_nan=True), (s, k)
        assert capi.workspace_cached_bytes() == grown                                     # grown, never shrunk
    assert capi.workspace_release(0) == grown and capi.workspace_cached_bytes() == 0
    after = ctx.chains_solve(Z0[:16], solver=capi.SOLVER_DEVICE, **kw)
    assert np.array_equal(after["z"], fresh[capi.SOLVER_DEVICE]["z"])
    ctx.close()


def test_two_groups_of_chains_side_by_side_change_no_result(monkeypatch):
    """Large sweeps on the device solvers run as two groups of chains on two host threads, the second on a clone of the context
    (batchsolve.cpp; by itself from 262 144 chains up, forced here): a chain only meets its own group, so every output is the
    one-group output bit for bit -- plain sweeps and parameter chains with their per-chain arrays --, the statistics add up and the
    clone's trajectories are counted on the caller's context."""
    from socp_amd import capi, sweep
    from test_gpu_chains import STAGE2_INIT, PARAMS0, KD, goddard_m6, make_ctx
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    sweep.goddard_multiple_shooting_problem(ctx, 6)
    Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(301, 0.05), 6)
    for solver in (capi.SOLVER_DEVICE, capi.SOLVER_DEVICE_FAST):
        monkeypatch.setenv("SOCP_CHAINS_DEVICE_GROUPS", "1")
        c0 = ctx.counters()[0]
        one = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10, solver=solver)
        c1 = ctx.counters()[0]
        for G in ("2", "3"):
            monkeypatch.setenv("SOCP_CHAINS_DEVICE_GROUPS", G)
            c2 = ctx.counters()[0]
            two = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10, solver=solver)
            for k in KEYS:
                assert np.array_equal(one[k], two[k], equal_nan=True), (solver, G, k)
            assert two["stats"]["jacobians_launched"] + two["stats"]["jacobians_from_cache"] == \
                one["stats"]["jacobians_launched"] + one["stats"]["jacobians_from_cache"]
            assert two["stats"]["rounds"] <= one["stats"]["rounds"]
            assert ctx.counters()[0] - c2 >= 0.9 * (c1 - c0)             # (speculative FD batches depend on the group size: not equal)
    ctx.close()
    # continuation chains with per-chain parameters and goals: the per-chain arrays are offset per group
    ctx = make_ctx("exact")
    goddard_m6(ctx)
    goals = np.array([310.0, 250.0, 400.0, 310.0, 120.0, 5000.0, 310.0, 600.0, 280.0])
    P = len(goals)
    Z0 = np.tile(STAGE2_INIT, (P, 1))
    Z0[3, 7:14] *= 1 + 1e-6
    kw = dict(kind=capi.CHAIN_PARAM, param_index=KD, step=0.4, goal=goals, params=np.tile(PARAMS0, (P, 1)), xtol=1e-6, solver=capi.SOLVER_DEVICE)
    monkeypatch.setenv("SOCP_CHAINS_DEVICE_GROUPS", "1")
    one = ctx.chains_solve(Z0, **kw)
    monkeypatch.setenv("SOCP_CHAINS_DEVICE_GROUPS", "2")
    two = ctx.chains_solve(Z0, **kw)
    for k in KEYS:
        assert np.array_equal(one[k], two[k], equal_nan=True), k
    ctx.close()
