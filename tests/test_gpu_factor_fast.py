"""GPU: the THROUGHPUT flavour of the device solver's Jacobian refresh (kernels_factor_fast.hip: blocked Householder QR, compact-WY
panels of 16, trailing updates on the FP64 matrix cores) against the order-preserving kernel (solver_dev.hpp: factor, bit-equal to
MINPACK's qrfac / qform as the reference drives hybrd, shooting.cpp:803-826).

Bar: the same factorisation -- same sign convention, same Q = H_0 ... H_{n-1} -- to ROUNDING, not bit for bit:
Q, R, Q^T b, diag(R) and the column norms within 1e-11 of the order-preserving kernel's (relative to the matrix norm), Q orthogonal
and Q R = J to 1e-13; in the engine (SOCP_SOLVER_DEVICE_FAST) every chain ends with the same `info` as with the bit-equal solvers
and its converged unknowns within north_star's 1e-8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problems(n, count, seed):
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((count, n, n))
    J[:, np.arange(n), np.arange(n)] += 0.5 * np.sqrt(n)              # well conditioned, both diagonal signs occur after the reflectors
    J[0] *= rng.choice([-1.0, 1.0], size=(n, 1))                        # ... and mixed signs on the way in
    b = rng.standard_normal((count, n))
    return J, b


@pytest.mark.parametrize("n", [39, 48, 64, 65, 85, 96, 97, 127, 128, 129, 192, 193, 200, 253, 256])
def test_fast_factor_equals_the_order_preserving_one_to_rounding(n):
    from socp_amd import capi
    J, b = _problems(n, 3, 100 + n)
    ex = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_EXACT)
    fa = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_FAST)
    scale = np.linalg.norm(J, axis=(1, 2))[:, None, None]
    eye = np.eye(n)[None]
    # what a QR factorisation is
    assert np.max(np.abs(np.transpose(fa["Q"], (0, 2, 1)) @ fa["Q"] - eye)) <= 1e-13 * n
    assert np.max(np.abs(fa["Q"] @ fa["R"] - J) / scale) <= 1e-13
    assert np.max(np.abs(np.einsum("kij,ki->kj", fa["Q"], b) - fa["qtb"])) <= 1e-12 * np.max(np.abs(b)) * np.sqrt(n)
    # ... and MINPACK's: same signs, same numbers to rounding
    assert np.array_equal(np.sign(fa["rdiag"]), np.sign(ex["rdiag"]))
    assert np.max(np.abs(fa["Q"] - ex["Q"])) <= 1e-11
    assert np.max(np.abs(fa["R"] - ex["R"]) / scale) <= 1e-11
    assert np.max(np.abs(fa["qtb"] - ex["qtb"])) <= 1e-11 * np.max(np.abs(b)) * np.sqrt(n)
    assert np.max(np.abs(fa["rdiag"] - ex["rdiag"]) / scale[:, :, 0]) <= 1e-12
    assert np.max(np.abs(fa["acnorm"] - ex["acnorm"]) / scale[:, :, 0]) <= 1e-14
    assert np.array_equal(fa["rdiag"], fa["R"][:, np.arange(n), np.arange(n)])
    assert not fa["sing"].any() and not ex["sing"].any()


def test_fast_factor_special_columns():
    """A zero column (no reflector, 'singular'), a column that is zero from the diagonal down after the earlier reflectors, entries
    far outside the range in which a plain sum of squares is safe, and a NaN: the flags and the pattern of the order-preserving kernel."""
    from socp_amd import capi
    n = 85
    J, b = _problems(n, 5, 7)
    J[0][:, 17] = 0.0                                                  # zero column inside the second panel
    J[1][:, 40] = J[1][:, 3] * 2.0                                     # dependent column: zero below the diagonal after reflector 3 ... to rounding
    J[2] *= 1e-170                                                     # squares underflow
    J[3] *= 1e170                                                      # squares overflow
    J[4][20, 30] = np.nan
    ex = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_EXACT)
    fa = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_FAST)
    assert fa["sing"][0] == 1 and ex["sing"][0] == 1 and fa["rdiag"][0, 17] == 0.0
    assert np.max(np.abs(fa["Q"][0] @ fa["R"][0] - J[0])) <= 1e-12 * np.linalg.norm(J[0])
    for k in (2, 3):
        s = np.max(np.abs(J[k])) * n                                   # (np.linalg.norm squares the entries: 0 or inf here)
        assert np.all(np.isfinite(fa["R"][k])) and np.all(np.isfinite(fa["Q"][k]))
        assert np.max(np.abs(fa["Q"][k] @ fa["R"][k] - J[k])) <= 1e-12 * s
        assert np.max(np.abs(fa["R"][k] - ex["R"][k])) <= 1e-11 * s and np.max(np.abs(fa["acnorm"][k] - ex["acnorm"][k])) <= 1e-13 * s
    assert np.isnan(fa["R"][4]).any() and np.isnan(ex["R"][4]).any()
    # the problems next to the special ones are untouched by them
    assert np.max(np.abs(fa["Q"][1] @ fa["R"][1] - J[1])) <= 1e-12 * np.linalg.norm(J[1])


_BOTH_FORMS = r"""
import numpy as np
from socp_amd import capi
for n in (39, 64, 85, 127, 200, 253, 256):
    rng = np.random.default_rng(100 + n)
    J = rng.standard_normal((3, n, n)); J[:, np.arange(n), np.arange(n)] += 0.5 * np.sqrt(n)
    J[1][:, 3] = 0.0; J[2][:, n // 2] *= 1e-170                      # a zero column, a tiny one: the rescaling path of the panel
    b = rng.standard_normal((3, n))
    ex = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_EXACT)
    fa = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_FAST)
    scale = np.linalg.norm(J, axis=(1, 2))[:, None, None]
    assert np.max(np.abs(fa["Q"] @ fa["R"] - J) / scale) <= 1e-13, n
    assert np.max(np.abs(np.transpose(fa["Q"], (0, 2, 1)) @ fa["Q"] - np.eye(n)[None])) <= 1e-13 * n, n
    assert np.max(np.abs(fa["R"] - ex["R"]) / scale) <= 1e-11 and np.max(np.abs(fa["qtb"] - ex["qtb"])) <= 1e-11 * np.max(np.abs(b)) * np.sqrt(n), n
    assert np.max(np.abs(fa["acnorm"] - ex["acnorm"]) / np.maximum(ex["acnorm"], 1e-300)) <= 1e-12, n
    assert fa["sing"][1] == 1 and ex["sing"][1] == 1 and fa["sing"][0] == 0 and fa["sing"][2] == 0, n
print("forms ok")
"""


@pytest.mark.parametrize("chain_min", ["0", "99999999"], ids=["chain_of_launches", "single_qrfac_launch"])
def test_both_forms_of_qrfac_at_every_strip_height(chain_min):
    """qrfac runs as a chain of launches per pair of panels -- kernels built for the strip height the pair needs (16, 12, 8, 6 or 4 chunks of
    16 rows) -- when a launch holds more problems than the single kernel keeps resident, as ONE launch otherwise (kernels_factor_fast.hip:
    launch_nch; SOCP_FACTOR_CHAIN_MIN moves the boundary and is read once per process: a child process per form).  Each form forced on three
    problems of every strip height, a zero column (no reflector) and a tiny one (the rescaling path of the panel's column step) among them,
    the same bars as above."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _BOTH_FORMS], capture_output=True, text=True, timeout=300, cwd=root,
                         env=dict(os.environ, SOCP_FACTOR_CHAIN_MIN=chain_min))
    assert out.returncode == 0 and "forms ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


def test_more_problems_than_the_chip_holds_and_a_permuted_list():
    """700 problems are past the boundary (640) from which qrfac runs as the chain of launches: the results must not depend on which workgroup
    of which launch met which problem, nor on the form -- 700 matrices that are 100 copies of 7 distinct ones, and the 7 alone (the single launch)."""
    from socp_amd import capi
    n = 85
    J7, b7 = _problems(n, 7, 5)
    J, b = np.tile(J7, (100, 1, 1)), np.tile(b7, (100, 1))
    fa = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_FAST)
    for k in range(7):
        assert np.array_equal(fa["Q"][k::7], np.broadcast_to(fa["Q"][k], (100, n, n)))
        assert np.array_equal(fa["R"][k::7], np.broadcast_to(fa["R"][k], (100, n, n)))
        assert np.array_equal(fa["qtb"][k::7], np.broadcast_to(fa["qtb"][k], (100, n)))
    few = capi.qr_factor_batch(J7, b7, flavour=capi.FACTOR_FAST)          # (7 problems: the single qrfac launch; same arithmetic, same panels)
    scale = np.linalg.norm(J7, axis=(1, 2))[:, None, None]
    assert np.max(np.abs(few["R"] - fa["R"][:7]) / scale) <= 1e-13 and np.max(np.abs(few["Q"] - fa["Q"][:7])) <= 1e-13


def test_fast_factor_refuses_sizes_it_is_not_built_for():
    from socp_amd import capi
    for n in (14, 32, 33, 38, 257):
        J, b = _problems(n, 1, n)
        with pytest.raises(RuntimeError):
            capi.qr_factor_batch(J, b, flavour=capi.FACTOR_FAST)
        capi.qr_factor_batch(J, b, flavour=capi.FACTOR_EXACT)


def _same_solutions(exact, fast, tol=1e-8):
    assert np.array_equal(exact["info"], fast["info"]), (exact["info"], fast["info"])
    ok = exact["info"] == 1
    scale = np.max(np.abs(exact["z"][ok]), axis=1, keepdims=True)
    assert np.max(np.abs(exact["z"][ok] - fast["z"][ok]) / scale) <= tol
    assert np.array_equal(exact["solves"], fast["solves"])


@pytest.mark.parametrize("M,P", [(6, 300), (9, 150)])
def test_engine_multiple_shooting_sweeps(M, P):
    """The testGoddard layout with M segments (n = 85, 127): bit-equal device solvers vs the throughput factorisation."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    n = sweep.goddard_multiple_shooting_problem(ctx, M)
    Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(P, 0.05), M)
    exact = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10, solver=capi.SOLVER_DEVICE)
    fast = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10, solver=capi.SOLVER_DEVICE_FAST)
    assert np.mean(exact["info"] == 1) >= 0.9                          # (5 % costate spread: a few starts of the larger layout end in info 4 / 5)
    _same_solutions(exact, fast)
    # AUTO on a throughput-flavour context takes the throughput factorisation; on a reference-order context it never does
    auto = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10)
    assert np.array_equal(auto["z"], fast["z"]) and np.array_equal(auto["nfev"], fast["nfev"])
    ctx.set_variant(capi.VARIANT_LANE_EXACT)
    auto_x = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10)
    host_x = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-10, solver=capi.SOLVER_HOST)
    assert np.array_equal(auto_x["z"], host_x["z"]) and np.array_equal(auto_x["nfev"], host_x["nfev"])
    ctx.close()


def test_engine_kd_continuation_chains():
    """testGoddard's KD 0 -> 310 continuation as chains (M = 6, n = 85, step 0.1: eleven solves each)."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    P = 256
    Z0, params, goal, KD = sweep.goddard_kd_chains(ctx, P)
    kw = dict(kind=capi.CHAIN_PARAM, param_index=KD, step=0.1, goal=goal, params=params, xtol=1e-10)
    exact = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE, **kw)
    fast = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE_FAST, **kw)
    assert np.all(exact["info"] == 1) and np.all(exact["b_reached"] == 1.0)
    _same_solutions(exact, fast)
    assert np.array_equal(exact["param_final"], fast["param_final"])
    ctx.close()


def test_engine_config5_interceptor_sweep():
    """BASELINE config 5 (interceptor, adaptive Dormand-Prince, M = 21, n = 253; parity unpinned: tests/test_gpu_interceptor.py):
    the sweep whose factor launch VERDICT r3 #3 is about, 64 starts."""
    from socp_amd import capi, sweep
    ctx, Z0, kw = sweep.interceptor_config5_sweep(64, variant="fast")
    exact = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE, **kw)
    fast = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE_FAST, **kw)
    assert np.all(exact["info"] == 1)
    _same_solutions(exact, fast)
    ctx.close()


@pytest.mark.parametrize("M,P,spread", [(6, 200, 0.05), (9, 120, 0.05)])
def test_engine_with_q_kept_as_factorised(M, P, spread, monkeypatch):
    """Config::lazy_q on the device (solver_dev.hpp: Broyden's rotations kept as a list, Q^T f from the Q of the last refresh; what the
    throughput flavour does by itself from n = 192 up -- the config-5 test above runs it -- forced here on n = 85 / 127, where the lists are
    short (10 / 16 updates) and long solves fill them, so the flush to the matrix runs too): info and solve counts equal, converged
    unknowns within 1e-8 of the bit-equal solver -- and of the same flavour with the eager update."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    sweep.goddard_multiple_shooting_problem(ctx, M)
    Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(P, spread), M)
    kw = dict(kind=capi.CHAIN_PLAIN, xtol=1e-10)
    exact = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE, **kw)
    monkeypatch.setenv("SOCP_SOLVER_LAZY_Q", "0")
    eager = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE_FAST, **kw)
    monkeypatch.setenv("SOCP_SOLVER_LAZY_Q", "1")
    lazy = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE_FAST, **kw)
    monkeypatch.delenv("SOCP_SOLVER_LAZY_Q")
    default = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE_FAST, **kw)
    assert np.array_equal(default["z"], eager["z"])                     # below n = 192 the eager update is what runs
    assert not np.array_equal(lazy["z"], eager["z"])                    # (the switch did switch: rounding differs)
    _same_solutions(exact, lazy)
    _same_solutions(eager, lazy)
    assert np.max(lazy["nfev_total"]) > 150                             # long solves are among them: lists of 10 / 16 updates overflow
    ctx.close()


@pytest.mark.parametrize("family", ["M6", "config5"])
def test_engine_shortcuts_of_the_throughput_flavour_are_rounding_level(family, monkeypatch):
    """Round 5's two liberties of the throughput flavour inside the trial step, each behind a switch: the back substitution's products
    summed in parallel instead of as one chain (SOCP_SOLVER_FAST_SUMS), and the predicted reduction of a Gauss-Newton step taken as the
    rounding noise it is instead of recomputed from R p (SOCP_SOLVER_GN_SHORTCUT).  Each one, and both, against MINPACK's forms in the
    same flavour and against the bit-equal solver: same `info`, converged unknowns within north_star's 1e-8 -- and the switches do switch
    (the iterates differ at rounding level)."""
    from socp_amd import capi, sweep
    if family == "M6":
        ctx = capi.Context(capi.MODEL_GODDARD)
        ctx.set_params(sweep.GODDARD_PARAMS)
        ctx.set_step_number(10)
        ctx.set_variant(capi.VARIANT_LANE_FAST)
        sweep.goddard_multiple_shooting_problem(ctx, 6)
        Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(160, 0.05), 6)
        kw = dict(kind=capi.CHAIN_PLAIN, xtol=1e-10)
    else:
        ctx, Z0, kw = sweep.interceptor_config5_sweep(48, variant="fast")
    exact = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE, **kw)
    runs = {}
    for sums, gn in (("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")):
        monkeypatch.setenv("SOCP_SOLVER_FAST_SUMS", sums)
        monkeypatch.setenv("SOCP_SOLVER_GN_SHORTCUT", gn)
        runs[sums + gn] = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE_FAST, **kw)
    monkeypatch.delenv("SOCP_SOLVER_FAST_SUMS")
    monkeypatch.delenv("SOCP_SOLVER_GN_SHORTCUT")
    default = ctx.chains_solve(Z0, solver=capi.SOLVER_DEVICE_FAST, **kw)
    assert np.array_equal(default["z"], runs["11"]["z"])                  # both are on by default
    for key in ("10", "01", "11"):
        _same_solutions(exact, runs[key])
        _same_solutions(runs["00"], runs[key])
        assert not np.array_equal(runs[key]["z"], runs["00"]["z"])       # (the switch did switch)
    ctx.close()
