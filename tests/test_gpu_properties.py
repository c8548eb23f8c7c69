"""GPU: size-independent properties of the hot path at BASELINE's FULL sizes (196 605 trajectories of 10 000 RK4 steps per
launch -- the bench batch; the oracle would need minutes for them, so nothing here compares with it):

  * the Hamiltonian is a first integral of the state+costate system (autonomous problem, smooth control law): H(X(tf)) = H(X(0))
    up to the integrator's O(dt^4) error, for every trajectory of the batch;
  * a trajectory does not know its batch: the same row gives the same bits in a batch of 196 605, in a batch of 64, first or
    last, and a permutation of the rows permutes the results;
  * semigroup: [0, tf] in N steps == [0, tf/2] in N/2 steps followed by [tf/2, tf] in N/2 steps, bit for bit (same dt; the
    smooth-law dynamics do not read t);
  * the FD-row batch is its own consistency check: row 0 of every problem is the residual of the unperturbed vector, rows whose
    perturbed unknown is pinned by a boundary row differ from row 0 in exactly that row's entry by exactly h.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STARTS = 13107                 # bench default: 15 x 13107 = 196 605 trajectories per launch
N_STEPS = 10000
TF = 0.2640825


def make_ctx(variant):
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(N_STEPS)
    ctx.set_variant(capi.VARIANT_LANE_EXACT if variant == "exact" else capi.VARIANT_LANE_FAST)
    return ctx


def bench_batch():
    """The bench's trajectories: FD-batch rows (z, z + h_j e_j) of 13 107 starts as initial states."""
    from socp_amd import sweep
    Z = sweep.goddard_starts(STARTS, 1e-3)
    X0 = np.repeat(Z, 15, axis=0)
    eps = np.sqrt(1e-15)
    for j in range(14):
        rows = np.arange(STARTS) * 15 + j + 1
        h = eps * np.abs(X0[rows, j])
        h[h == 0] = eps
        X0[rows, j] += h
    return X0


@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_hamiltonian_is_conserved_over_the_full_batch(variant):
    from socp_amd import capi
    ctx = make_ctx(variant)
    X0 = bench_batch()
    assert X0.shape == (196605, 14)
    Xf = ctx.integrate_batch(0.0, TF, X0)
    assert np.all(np.isfinite(Xf))
    H0 = ctx.eval_batch(capi.EVAL_HAMILTONIAN, 0.0, X0)[:, 0]
    Hf = ctx.eval_batch(capi.EVAL_HAMILTONIAN, TF, Xf)[:, 0]
    drift = np.abs(Hf - H0) / np.maximum(1.0, np.abs(H0))
    # RK4 with dt = 2.6e-5 on a right-hand side that is only C^1 across the thrust-off switch (alpha = max(0, .)): observed drift
    # 2e-10 typical, 6e-9 at most over the batch; an error in any costate equation shows up at 1e-3 and above
    assert drift.max() < 5e-8 and np.median(drift) < 2e-9, (drift.max(), np.median(drift))
    # the mass only decreases (dm/dt = -b |u| <= 0) and stays positive on this family
    assert np.all(Xf[:, 6] <= X0[:, 6]) and np.all(Xf[:, 6] > 0)
    ctx.close()


@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_a_trajectory_does_not_know_its_batch(variant):
    ctx = make_ctx(variant)
    X0 = bench_batch()
    Xf = ctx.integrate_batch(0.0, TF, X0)
    rng = np.random.default_rng(11)
    pick = np.concatenate([[0, 1, 63, 64, len(X0) - 65, len(X0) - 1], rng.integers(0, len(X0), 58)])
    small = ctx.integrate_batch(0.0, TF, X0[pick])                       # 64 rows: one wave, one wave per SIMD
    assert np.array_equal(small, Xf[pick])
    perm = rng.permutation(len(X0))
    assert np.array_equal(ctx.integrate_batch(0.0, TF, X0[perm]), Xf[perm])
    ctx.close()


@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_semigroup_bit_for_bit(variant):
    ctx = make_ctx(variant)
    X0 = bench_batch()[:4096]
    whole = ctx.integrate_batch(0.0, TF, X0)
    ctx.set_step_number(N_STEPS // 2)
    mid = ctx.integrate_batch(0.0, TF / 2, X0)
    end = ctx.integrate_batch(TF / 2, TF, mid)
    assert np.array_equal(end, whole)
    ctx.close()


@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_fd_row_batch_is_self_consistent_at_full_size(variant):
    from socp_amd import sweep
    ctx = make_ctx(variant)
    sweep.goddard_single_shooting_problem(ctx)
    Z = sweep.goddard_starts(STARTS, 1e-3)
    rows = ctx.fd_rows(Z)                                                # [13107][15][14]: the bench step
    assert rows.shape == (STARTS, 15, 14) and np.all(np.isfinite(rows))
    # row 0 = F(z): the residual kernel on the same vectors gives the same bits
    F = ctx.residual_batch(Z)
    assert np.array_equal(rows[:, 0, :], F)
    # unknowns 0..6 are the initial STATE, pinned by rows 0..6 (F_j = z_j - x0_j) and ALSO the start of the trajectory;
    # entry j of row j + 1 is exactly (z_j + h_j) - x0_j
    eps = np.sqrt(1e-15)
    for j in range(7):
        h = eps * np.abs(Z[:, j])
        h[h == 0] = eps
        assert np.array_equal(rows[:, j + 1, j], (Z[:, j] + h) - sweep.X0_STATE[j])
        others = [k for k in range(7) if k != j]
        assert np.array_equal(rows[:, j + 1, others], rows[:, 0, others])   # the other initial rows do not move
    # the Jacobian formed from the rows equals the fused FD-column kernel's (dedup on and off), for a slice of the batch
    Fz = rows[:64, 0, :]
    for p in range(0, 64, 21):
        J = ctx.fd_jacobian(Z[p], Fz[p], dedup=True)
        hs = np.array([eps * abs(v) or eps for v in Z[p]])
        assert np.array_equal(J, ((rows[p, 1:] - rows[p, 0]) / hs[:, None]).T)
        assert np.array_equal(J, ctx.fd_jacobian(Z[p], Fz[p], dedup=False))
    ctx.close()
