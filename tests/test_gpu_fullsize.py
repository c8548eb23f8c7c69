"""GPU: the named workloads AT THEIR STATED SIZE: north_star's 128-unknown Goddard layout against the oracle,
BASELINE config 4 at 4096 starts and config 5's solve sweep at 256 starts -- so that the code paths that only switch on at
size (host threads of the lock-step engine, chain groups, Jacobian read-back in passes, the speculation budget) run under the
driver's test run and not only in builder-run scripts."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fd_points(z, eps=np.sqrt(1e-15)):
    """Base point + the n perturbed points of MINPACK's fdjac1: h_j = eps |z_j|, or eps where z_j = 0."""
    n = len(z)
    Z = np.tile(z, (n + 1, 1))
    h = eps * np.abs(z)
    h[h == 0] = eps
    for j in range(n):
        Z[j + 1, j] = z[j] + h[j]
    return Z, h


@pytest.mark.parametrize("rk4_steps", [100, 10000])
@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_north_star_128_unknown_layout_against_the_oracle(built, variant, rk4_steps):
    """Goddard, M = 9, FREE tf + one FREE interior time, n = 128 (the size north_star's >= 10x target is quoted on): the residual
    and all 129 forward-difference rows against the CPU oracle's residual_batch -- reference-order flavour bit for bit,
    throughput flavour within north_star's 1e-8; the fused FD Jacobian equals the differences of those rows and is identical
    with and without the segment dedup.  Reference rows exercised together only here: a free interior time under the smooth
    law (goddard.cpp:343-370 -> H(X-) row; shooting.cpp:961-973) with CONTINUOUS nodes spaced between two FREE junctions
    (shooting.cpp:1592-1609)."""
    from socp_amd import capi, sweep
    from oracle import oracle as orc
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(rk4_steps)
    ctx.set_variant(capi.VARIANT_LANE_EXACT if variant == "exact" else capi.VARIANT_LANE_FAST)
    n, z, mode_t, mode_x, tn, X = sweep.goddard_north_star_128_problem(ctx)
    assert n == 128 and len(z) == 128
    # move the point off the p* trajectory so that every continuity row is non-trivial, and the free times off the grid
    rng = np.random.default_rng(128)
    z = z * (1 + 1e-3 * rng.uniform(-1, 1, n))
    Zp, h = _fd_points(z)

    o = orc.Oracle(orc.MODEL_GODDARD, step_nbr=rk4_steps, params=sweep.GODDARD_PARAMS)
    prob = orc.Problem(7, mode_t, mode_x, tn, X)
    assert prob.n == 128
    want = o.residual_batch(prob, Zp)                    # 129 rows x 9 segments on the CPU
    assert np.all(np.isfinite(want))

    rows = ctx.fd_rows(z[None, :])[0]                    # ONE launch: 129 x 9 trajectories
    F = ctx.residual(z)
    batch = ctx.residual_batch(Zp)
    assert np.array_equal(F, rows[0]) and np.array_equal(batch, rows)      # three kernels, one arithmetic
    scale = np.maximum(1.0, np.max(np.abs(want), axis=1, keepdims=True))
    err = np.max(np.abs(rows - want) / scale)
    if variant == "exact":
        assert np.array_equal(rows, want), err                             # bit-identical to the CPU path
    else:
        assert err <= 1e-8, err
    # timeline: the interior free time and the uniformly spaced nodes on either side of it
    assert np.array_equal(ctx.timeline(z), o.timeline(prob, z))

    J_full = ctx.fd_jacobian(z, F, dedup=False)
    J_dedup = ctx.fd_jacobian(z, F, dedup=True)
    assert np.array_equal(J_full, J_dedup)
    J_rows = ((rows[1:] - rows[0][None, :]) / h[:, None]).T                 # J[row, col]
    assert np.array_equal(J_full, J_rows)
    if variant == "exact":
        assert np.array_equal(J_full, o.fdjac(prob, z, want[0]))
    ctx.close()


def test_config4_at_4096_starts(built):
    """BASELINE config 4 as stated: 4096 independent initial-costate starts of the n = 14 single-shooting problem, 1e4 RK4 steps,
    full Newton solves in lock-step, throughput flavour.  Converged count; sampled chains bit-equal to the same start solved
    alone (through the engine with one chain and speculation off -- other launches, same arithmetic -- and through the blocking
    hybrd driven from the host); every converged start on the CPU path's root (golden: oracle residual + host hybrd) within
    north_star's 1e-8; the GPU residual at that root against the oracle's."""
    from socp_amd import capi, sweep
    from oracle import oracle as orc
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "c2_root.json")))
    zg = np.array(gold["z"])
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10000)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    assert sweep.goddard_single_shooting_problem(ctx) == 14
    P = 4096
    Z0 = sweep.goddard_starts(P, 1e-3)
    out = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-12)
    ok = out["info"] == 1
    assert ok.sum() >= 4080, (ok.sum(), np.unique(out["info"], return_counts=True))
    assert out["stats"]["rounds"] > 0 and out["stats"]["jacobians_from_cache"] + out["stats"]["jacobians_launched"] >= P
    # MINPACK reports info = 1 when the trust region has collapsed (delta <= xtol |x|), whatever |F| is: of the 4087 starts that
    # end that way here, one (start 4093) does so with |F| = 4e-8 after 56 evaluations.  The root comparison is made on the starts
    # whose residual actually vanished; test_config4_outlier_start_is_minpacks_rule (below) shows WHY that start ends there.
    well = ok & (out["fnorm"] <= 1e-9)
    assert well.sum() >= 4080, well.sum()
    assert (ok & ~well).sum() <= 2, np.where(ok & ~well)[0]
    err = np.max(np.abs(out["z"][well] - zg[None, :]), axis=1) / np.max(np.abs(zg))
    assert np.max(err) <= 1e-8, np.max(err)
    # 32 sampled chains: alone == in the batch of 4096
    sample = np.linspace(0, P - 1, 32).astype(int)
    for p in sample:
        alone = ctx.chains_solve(Z0[p:p + 1], kind=capi.CHAIN_PLAIN, xtol=1e-12, speculate=0)
        assert alone["info"][0] == out["info"][p] and alone["nfev"][0] == out["nfev"][p]
        assert np.array_equal(alone["z"][0], out["z"][p])
    for p in sample[:2]:
        def fd(x, fvec, eps):
            return ctx.fd_jacobian(x, fvec, epsfcn=eps, dedup=True)
        alone = capi.hybrd(lambda v: ctx.residual(v), Z0[p], xtol=1e-12, epsfcn=1e-15, fdjac=fd)
        assert alone["info"] == out["info"][p] and alone["nfev"] == out["nfev"][p] and np.array_equal(alone["x"], out["z"][p])
    # the residual the device reports at the golden root against the oracle's (1e4 steps): north_star's 1e-8 on the residual
    o = orc.Oracle(orc.MODEL_GODDARD, step_nbr=10000, params=sweep.GODDARD_PARAMS)
    mode_x = np.zeros((2, 7), dtype=np.int32)
    mode_x[1, 3:7] = orc.FREE
    X = np.zeros((2, 14))
    X[0, :7] = sweep.X0_STATE
    X[1, 0] = 1.01
    prob = orc.Problem(7, [orc.FIXED, orc.FIXED], mode_x, np.array([0.0, sweep.TF]), X)
    zs = out["z"][well][:8]
    Fc = o.residual_batch(prob, zs)
    Fg = ctx.residual_batch(zs)
    assert np.max(np.abs(Fg - Fc)) <= 1e-8
    ctx.close()


def test_config4_outlier_start_is_minpacks_rule(built):
    """test_config4_at_4096_starts compares with the golden root only the starts whose residual vanished,
    because ONE start of the throughput flavour (4093; scripts/probes/probe_c4_outliers.py -> profiles/r04_c4_outliers.json) ends
    with info = 1 and |F| = 4e-8.  What that is, under test:
      * it is MINPACK's own exit: the same start driven through the HOST solver (socp_hybr_*, the algorithm pinned to SciPy's
        MINPACK bit for bit) with the throughput flavour's residuals reproduces the engine's result bit for bit, and at its exit
        the trust region has collapsed -- delta <= xtol |diag x| -- while |F| is not zero: info = 1 by the rule;
      * it is a matter of the rounding path, not of the device: the reference-order flavour on the GPU and the CPU path (oracle
        residual + the same host solver) agree on this start BIT FOR BIT -- info 1, 38 evaluations, on the golden root;
      * the throughput flavour itself reaches the root from where it stopped (restart: |F| < 1e-9, within 1e-8 of the golden)."""
    from socp_amd import capi, sweep
    from oracle import oracle as orc
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "c2_root.json")))
    zg = np.array(gold["z"])
    p, xtol = 4093, 1e-12
    Z0 = sweep.goddard_starts(4096, 1e-3)

    def ctx_of(variant):
        c = capi.Context(capi.MODEL_GODDARD)
        c.set_params(sweep.GODDARD_PARAMS)
        c.set_step_number(10000)
        c.set_variant(variant)
        assert sweep.goddard_single_shooting_problem(c) == 14
        return c

    # -- throughput flavour: engine result, then the same start through the host state machine with the trust region read out
    fast = ctx_of(capi.VARIANT_LANE_FAST)
    eng = fast.chains_solve(Z0[p:p + 1], kind=capi.CHAIN_PLAIN, xtol=xtol, speculate=0)
    assert eng["info"][0] == 1 and eng["fnorm"][0] > 1e-9            # the outlier (if this ever stops being one, the filter can go)
    h = capi.HybrSolver(14, xtol=xtol, maxfev=10000, epsfcn=1e-15)
    h.start(Z0[p])
    req, xe, out_buf = h.advance(0)
    while req != capi.REQ_DONE:
        if req == capi.REQ_FVEC:
            out_buf[:] = fast.residual(xe.copy())
        else:
            x = xe.copy()
            out_buf[:] = fast.fd_jacobian(x, h.fvec, epsfcn=1e-15, dedup=True).T.ravel()        # column-major
        req, xe, out_buf = h.advance(0)
    delta, xnorm, fnorm = h.trust_region
    assert h.info == 1 and h.nfev == eng["nfev"][0] and np.array_equal(h.x, eng["z"][0])
    assert delta <= xtol * xnorm and fnorm > 1e-9 and abs(fnorm - eng["fnorm"][0]) <= 1e-12 * fnorm
    # ... and from there the flavour does reach the root
    again = fast.chains_solve(eng["z"], kind=capi.CHAIN_PLAIN, xtol=xtol, speculate=0)
    assert again["info"][0] == 1 and again["fnorm"][0] <= 1e-9
    assert np.max(np.abs(again["z"][0] - zg)) / np.max(np.abs(zg)) <= 1e-8
    fast.close()

    # -- reference order on the GPU == the CPU path, on this start
    exact = ctx_of(capi.VARIANT_LANE_EXACT)
    ge = exact.chains_solve(Z0[p:p + 1], kind=capi.CHAIN_PLAIN, xtol=xtol, speculate=0)
    exact.close()
    o = orc.Oracle(orc.MODEL_GODDARD, step_nbr=10000, params=sweep.GODDARD_PARAMS)
    mode_x = np.zeros((2, 7), dtype=np.int32)
    mode_x[1, 3:7] = orc.FREE
    X = np.zeros((2, 14))
    X[0, :7] = sweep.X0_STATE
    X[1, 0] = 1.01
    prob = orc.Problem(7, [orc.FIXED, orc.FIXED], mode_x, np.array([0.0, sweep.TF]), X)
    cpu = capi.hybrd(lambda v: o.residual(prob, v), Z0[p], xtol=xtol, epsfcn=1e-15)
    assert cpu["info"] == 1 and ge["info"][0] == 1 and cpu["nfev"] == ge["nfev"][0]
    assert np.array_equal(np.asarray(cpu["x"]), ge["z"][0])
    assert ge["fnorm"][0] <= 1e-9 and np.max(np.abs(ge["z"][0] - zg)) / np.max(np.abs(zg)) <= 1e-8


@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_config5_sweep_at_256_starts(variant):
    """BASELINE config 5 as a solve sweep at the size DESIGN quotes: interceptor, adaptive Dormand-Prince, M = 21, n = 253,
    256 starts around the converged scenario-1 trajectory: all converge, to one root (spread < 1e-8).  256 x 253^2 doubles per
    Jacobian refresh go through the engine's threaded host side and its chunked read-back.  Parity of this configuration is
    unpinned (no Boost, no Eigen; tests/test_gpu_interceptor.py has the comparison with the restatement)."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_INTERCEPTOR)
    ctx.set_variant(capi.VARIANT_LANE_FAST if variant == "fast" else capi.VARIANT_LANE_EXACT)
    n, z = sweep.interceptor_config5_problem(ctx)
    assert n == 253
    ctx.set_integrator(capi.INT_DOPRI5, 1e-11)           # the FD Jacobian's noise must stay below the solver tolerance
    P = 256
    raw = sweep.mt19937_64(20250905, 6 * P)
    xi = ((raw >> np.uint64(11)).astype(np.float64) * 2.0 ** -53 * 2.0 - 1.0).reshape(P, 6)
    Z0 = np.tile(z, (P, 1))
    Z0[:, 6:12] *= 1.0 + 1e-3 * xi
    r = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-9)
    assert np.all(r["info"] == 1), np.unique(r["info"], return_counts=True)
    spread = np.max(np.abs(r["z"] - np.median(r["z"], axis=0))) / np.max(np.abs(r["z"]))
    assert spread < 1e-8, spread
    alone = ctx.chains_solve(Z0[100:101], kind=capi.CHAIN_PLAIN, xtol=1e-9, speculate=0)
    if variant == "exact":
        assert alone["nfev"][0] == r["nfev"][100] and np.array_equal(alone["z"][0], r["z"][100])
    else:
        # throughput flavour: the sweep's Jacobian refreshes go through the matrix-core factorisation (AUTO, round 4), the lone
        # chain through the host solver -- the same iteration to rounding, the same root to north_star's 1e-8
        assert np.max(np.abs(alone["z"][0] - r["z"][100])) <= 1e-8 * np.max(np.abs(r["z"][100]))
        same = ctx.chains_solve(Z0[:256], kind=capi.CHAIN_PLAIN, xtol=1e-9, solver=capi.SOLVER_DEVICE)
        assert np.array_equal(same["info"], r["info"]) and np.max(np.abs(same["z"] - r["z"])) <= 1e-8 * np.max(np.abs(r["z"]))
    ctx.close()


@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_full_batch_properties_that_need_no_oracle(variant):
    """BASELINE's full size (13 107 starts x 15 rows = 196 605 trajectories of 10^4 RK4 steps: one bench step) is beyond what the CPU
    oracle finishes in a test, so the batch is checked through properties that do not depend on its size:
      * lanes are independent -- the rows of a shuffled batch are the shuffled rows, bit for bit (no lane reads a neighbour's data,
        nothing depends on where in the grid a trajectory runs);
      * composition -- 0 -> tf in 10^4 steps equals 0 -> tf/2 -> tf in 5000 + 5000 steps of the same dt (tf/2 and dt are exact
        binary halves) to rounding (1e-10): the step loop carries no hidden state besides (t, X).  Not bit for bit, and rightly so: the
        reference's loop accumulates t += dt (odeTools.cpp:136-145), 5000 additions of dt stop a rounding error short of tf/2 and
        the loop then takes one more step of that size -- which the single call does not take in the middle;
      * an interval that is empty or runs backwards leaves every state as it is, bit for bit -- the reference's `while (t < tf)`
        never enters (odeTools.cpp:136);
      * the rows sampled against the oracle in bench.py's parity leg are rows of this same launch."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_variant(capi.VARIANT_LANE_FAST if variant == "fast" else capi.VARIANT_LANE_EXACT)
    P = 13107 if variant == "fast" else 4096                 # the reference-order flavour runs at 0.28 of the rate: a third of the batch
    Z0 = sweep.goddard_starts(P, 1e-3)
    z_rows = np.repeat(Z0, 15, axis=0)                       # 15 rows per start, as an FD batch has (here: identical copies + jitter)
    rng = np.random.default_rng(3)
    z_rows[:, 7:] *= 1.0 + 1e-6 * rng.uniform(-1, 1, (len(z_rows), 7))
    B = len(z_rows)
    tf = 0.2
    t0 = np.zeros(B)
    ctx.set_step_number(10000)
    X1 = ctx.integrate_batch(t0, np.full(B, tf), z_rows)
    assert np.all(np.isfinite(X1))
    perm = rng.permutation(B)
    X1p = ctx.integrate_batch(t0, np.full(B, tf), z_rows[perm])
    assert np.array_equal(X1p, X1[perm])
    ctx.set_step_number(5000)
    Xh = ctx.integrate_batch(t0, np.full(B, tf / 2), z_rows)
    X2 = ctx.integrate_batch(np.full(B, tf / 2), np.full(B, tf), Xh)
    scale = np.max(np.abs(X1), axis=0)
    assert np.max(np.abs(X2 - X1) / scale) <= 1e-10             # (observed 1e-12: one rounding-size step, carried through 5000 more)
    ctx.set_step_number(10000)
    Xb = ctx.integrate_batch(np.full(B, tf), t0, X1)
    assert np.array_equal(Xb, X1)
    assert np.array_equal(ctx.integrate_batch(np.full(B, tf), np.full(B, tf), X1), X1)
    ctx.close()
