"""CPU: the DEVICE Powell iteration (socp_amd/csrc/solver_dev.hpp -- what the gfx950 solver kernels run, one workgroup per
problem) compiled for the host with one thread per problem (tests/tools/solver_sim.cpp) against the library's host solver
(minpack.cpp, itself equal to SciPy's MINPACK bit for bit): same iterates, same factors, same counters -- to the last bit.
This pins the ARITHMETIC of the device solver without a GPU; that the many-thread form computes the same is what
tests/test_gpu_devsolver.py checks on the device."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from socp_amd import capi
from test_minpack import CASES, rosen, rosen_jac, broyden_tri

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = C.POINTER(C.c_double)
FCN = C.CFUNCTYPE(C.c_int, C.c_int, _dp, _dp)
JAC = C.CFUNCTYPE(C.c_int, C.c_int, _dp, _dp, _dp)


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("solver_sim") / "libsolver_sim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "tests", "tools", "solver_sim.cpp")])
    L = C.CDLL(so)
    L.sim_solve.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, FCN, JAC,
                            C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, _dp, _dp, _dp, C.c_int]
    return L


def fd_jacobian(f, x, fvec, epsfcn):
    """MINPACK fdjac1, J[row, col]."""
    eps = np.sqrt(max(epsfcn, np.finfo(float).eps))
    n = len(x)
    J = np.empty((n, n))
    for j in range(n):
        h = eps * abs(x[j]) or eps
        xp = x.copy()
        xp[j] = x[j] + h
        J[:, j] = (f(xp) - fvec) / h
    return J


def sim_solve(L, f, jac, x0, xtol, factor, analytic, epsfcn=1e-15, maxfev=10000, abort_after=None, blocked=0):
    n = len(x0)
    x = np.array(x0, dtype=np.float64)
    out = dict(fvec=np.zeros(n), fjac=np.zeros((n, n)), r=np.zeros(n * (n + 1) // 2), qtf=np.zeros(n), diag=np.zeros(n))
    calls = [0]

    def _f(nn, xp, fp):
        calls[0] += 1
        if abort_after is not None and calls[0] > abort_after:
            return -7
        np.ctypeslib.as_array(fp, shape=(nn,))[:] = f(np.ctypeslib.as_array(xp, shape=(nn,)).copy())
        return 0

    def _j(nn, xp, fp, jp):
        xx = np.ctypeslib.as_array(xp, shape=(nn,)).copy()
        J = jac(xx) if analytic else fd_jacobian(f, xx, np.ctypeslib.as_array(fp, shape=(nn,)).copy(), epsfcn)
        np.ctypeslib.as_array(jp, shape=(nn, nn))[:] = np.asarray(J).T          # column-major
        return 0
    d = lambda a: a.ctypes.data_as(_dp)  # noqa: E731
    nfev, njev = C.c_int(0), C.c_int(0)
    info = L.sim_solve(n, d(x), d(out["fvec"]), xtol, maxfev, epsfcn, factor, int(analytic), FCN(_f), JAC(_j), C.byref(nfev), C.byref(njev),
                       d(out["fjac"]), d(out["r"]), d(out["qtf"]), d(out["diag"]), int(blocked))
    out.update(x=x, info=info, nfev=nfev.value, njev=njev.value)
    return out


def same(a, b):
    for k in ("x", "fvec", "r", "qtf", "diag", "fjac"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert a["info"] == b["info"] and a["nfev"] == b["nfev"]


@pytest.mark.parametrize("name,f,x0", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("factor", [1.0, 100.0])
def test_device_iteration_equals_host_hybrd(sim, name, f, x0, factor):
    host = capi.hybrd(f, x0.copy(), xtol=1e-8, epsfcn=1e-15, factor=factor,
                      fdjac=lambda x, fv, e: fd_jacobian(f, x, fv, e))
    dev = sim_solve(sim, f, None, x0, 1e-8, factor, analytic=False)
    same(dev, host)


def test_device_iteration_equals_host_hybrj(sim):
    x0 = np.array([-1.2, 1.0, -1.2, 1.0])
    host = capi.hybrj(rosen, rosen_jac, x0.copy(), xtol=1e-8, factor=1.0)
    dev = sim_solve(sim, rosen, rosen_jac, x0, 1e-8, 1.0, analytic=True)
    same(dev, host)
    assert dev["njev"] == host["njev"] >= 1


@pytest.mark.parametrize("blocked", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 33, 63, 64, 65, 130])
def test_sizes_and_random_systems(sim, n, blocked):
    """Mildly nonlinear random systems across the vector-length boundaries of the device layout (ld = n + 1 rounded up to 8; panels
    of 8 reflectors and blocks of 64 columns in the blocked factorisation): many Jacobian refreshes, Broyden updates with zero
    and non-zero rotations, singular trial factors.  blocked = 1: the refreshes go through factor_blocked."""
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)) + 3 * np.eye(n)
    b = rng.normal(size=n)

    def f(x):
        return A @ x + 0.3 * np.sin(x) * np.roll(x, 1) - b
    x0 = rng.normal(size=n)
    for xtol in (1e-8, 1e-13):
        host = capi.hybrd(f, x0.copy(), xtol=xtol, epsfcn=1e-15, fdjac=lambda x, fv, e: fd_jacobian(f, x, fv, e))
        dev = sim_solve(sim, f, None, x0, xtol, 1.0, analytic=False, blocked=blocked)
        same(dev, host)


def test_failure_paths_are_the_host_ones(sim):
    """info 2 (maxfev), info 4 / 5 (no progress), a singular Jacobian (zero column -> identity reflector, zero pivot in the back
    substitution), NaN residuals and a negative callback return: the state machine ends as the host one does."""
    x0 = -np.ones(10)
    host = capi.hybrd(broyden_tri, x0.copy(), xtol=1e-8, maxfev=15, epsfcn=1e-15)
    dev = sim_solve(sim, broyden_tri, None, x0, 1e-8, 1.0, analytic=False, maxfev=15)
    same(dev, host)
    assert dev["info"] == 2

    def flat(x):                       # the last unknown does not enter: zero Jacobian column
        f = broyden_tri(x)
        f[-1] = 1.0 + 0 * x[-1]
        return f
    host = capi.hybrd(flat, x0.copy(), xtol=1e-8, epsfcn=1e-15, fdjac=lambda x, fv, e: fd_jacobian(flat, x, fv, e))
    dev = sim_solve(sim, flat, None, x0, 1e-8, 1.0, analytic=False)
    same(dev, host)
    assert dev["info"] in (4, 5)
    same(sim_solve(sim, flat, None, x0, 1e-8, 1.0, analytic=False, blocked=1), host)          # identity reflectors in the blocked form

    # zero columns anywhere: first, in the middle, two in a row, next to last (the fused factor sweeps have a path for "no
    # reflector in hand" and one for "no next reflector")
    for dead in ((0,), (4,), (4, 5), (8,), (0, 1, 9)):
        def holes(x, dead=dead):
            y = x.copy()
            y[list(dead)] = -1.0                                 # those unknowns do not enter
            f = broyden_tri(y)
            return f
        jac = lambda x, fv, e, g=holes: fd_jacobian(g, x, fv, e)  # noqa: E731
        host = capi.hybrd(holes, x0.copy(), xtol=1e-8, epsfcn=1e-15, fdjac=jac)
        same(sim_solve(sim, holes, None, x0, 1e-8, 1.0, analytic=False), host)
        same(sim_solve(sim, holes, None, x0, 1e-8, 1.0, analytic=False, blocked=1), host)

    def nanny(x):
        f = broyden_tri(x)
        f[3] = np.nan
        return f
    host = capi.hybrd(nanny, x0.copy(), xtol=1e-8, epsfcn=1e-15, fdjac=lambda x, fv, e: fd_jacobian(nanny, x, fv, e))
    dev = sim_solve(sim, nanny, None, x0, 1e-8, 1.0, analytic=False)
    same(dev, host)

    calls = [0]

    def stopper(x):
        calls[0] += 1
        return None if calls[0] > 4 else broyden_tri(x)
    host = capi.hybrd(stopper, x0.copy(), xtol=1e-8, epsfcn=1e-15, fdjac=lambda x, fv, e: fd_jacobian(broyden_tri, x, fv, e))
    dev = sim_solve(sim, broyden_tri, None, x0, 1e-8, 1.0, analytic=False, abort_after=4)
    assert host["info"] == -1 and dev["info"] == -7          # the callback's own negative value becomes info (shooting.cpp:873)
    assert np.array_equal(dev["x"], host["x"]) and np.array_equal(dev["fvec"], host["fvec"])


def _minpack_enorm(x):
    """MINPACK's enorm, statement for statement (SURVEY App. A lists it; minpack.cpp: enorm) in numpy float64 scalars = IEEE doubles
    (division by zero and NaN comparisons as in C)."""
    f = np.float64
    n = len(x)
    rdwarf, rgiant = f(3.834e-20), f(1.304e19)
    s1 = s2 = s3 = x1max = x3max = f(0.0)
    agiant = rgiant / f(n)
    with np.errstate(all="ignore"):
        for v in x:
            xabs = np.abs(f(v))
            if xabs > rdwarf and xabs < agiant:
                s2 = s2 + xabs * xabs
            elif xabs <= rdwarf:
                if xabs > x3max:
                    q = x3max / xabs
                    s3 = f(1.0) + s3 * (q * q)
                    x3max = xabs
                elif xabs != 0:
                    q = xabs / x3max
                    s3 = s3 + q * q
            else:
                if xabs > x1max:
                    q = x1max / xabs
                    s1 = f(1.0) + s1 * (q * q)
                    x1max = xabs
                else:
                    q = xabs / x1max
                    s1 = s1 + q * q
        if s1 != 0:
            return float(x1max * np.sqrt(s1 + (s2 / x1max) / x1max))
        if s2 != 0:
            if s2 >= x3max:
                return float(np.sqrt(s2 * (f(1.0) + (x3max / s2) * (x3max * s3))))
            return float(np.sqrt(x3max * ((s2 / x3max) + (x3max * s3))))
        return float(x3max * np.sqrt(s3))


def test_norm_usual_case_and_general_case_are_minpacks(sim):
    """enorm's branch-free usual case (every entry in the middle range or zero: a plain sum of squares) must be MINPACK's number
    bit for bit, and anything else must fall through to the three-accumulator loop: lengths across the 8-entry batches, zeros,
    dwarfs, giants, mixtures, strided input."""
    sim.sim_enorm.restype = C.c_double
    sim.sim_enorm.argtypes = [C.c_int, _dp, C.c_long]
    rng = np.random.default_rng(7)
    cases = []
    for n in (1, 2, 7, 8, 9, 15, 16, 17, 64, 85, 253):
        x = rng.normal(size=n) * 10.0 ** rng.integers(-6, 7, size=n)
        cases.append(x)
        z = x.copy(); z[rng.integers(0, n, size=max(1, n // 3))] = 0.0
        cases.append(z)
        cases.append(np.zeros(n))
        t = x.copy(); t[rng.integers(0, n)] = 1e-25                     # one dwarf: the general loop
        cases.append(t)
        g = x.copy(); g[rng.integers(0, n)] = 3e19                      # one giant
        cases.append(g)
        cases.append(rng.normal(size=n) * 1e-30)                        # all dwarfs
        cases.append(rng.normal(size=n) * 1e25)                         # all giants
        m = x.copy(); m[0] = 1e-22; m[-1] = -7e20
        cases.append(m)
    for x in cases:
        x = np.ascontiguousarray(x, dtype=np.float64)
        got = sim.sim_enorm(len(x), x.ctypes.data_as(_dp), 1)
        want = _minpack_enorm(x)
        assert got == want or (np.isnan(got) and np.isnan(want)), (len(x), got, want)
    # strided (a matrix column) and non-finite entries
    A = np.ascontiguousarray(rng.normal(size=(9, 12)))
    assert sim.sim_enorm(9, A[:, 5:].ctypes.data_as(_dp), 12) == _minpack_enorm(A[:, 5])
    for bad in (np.nan, np.inf):
        x = rng.normal(size=11); x[4] = bad
        got = sim.sim_enorm(11, x.ctypes.data_as(_dp), 1)
        want = _minpack_enorm(x)
        assert (np.isnan(got) and np.isnan(want)) or got == want, (bad, got, want)


@pytest.mark.parametrize("n", [8, 15, 16, 33, 64, 130])
def test_q_kept_as_factorised_moves_iterates_at_rounding_level_only(sim, n):
    """Config::lazy_q (the throughput flavour of the device solver: Broyden's rotations kept as a list, Q^T f formed from the Q of the last
    refresh): the same mathematics, so the same info and evaluation counts and the same solution to rounding; R and Q^T f -- which are
    updated eagerly either way -- agree to rounding too.  Sizes with list capacities 1 (lazy off), 2, 4, 8, 16: with xtol = 1e-13 the solves
    take more Broyden updates between refreshes than the small lists hold, so the flush to the matrix runs as well."""
    sim.sim_lazy_capacity.argtypes = [C.c_int]
    cap = sim.sim_lazy_capacity(n)
    assert cap == min(16, (n + 1) // 8)
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)) + 3 * np.eye(n)
    b = rng.normal(size=n)

    def f(x):
        return A @ x + 0.3 * np.sin(x) * np.roll(x, 1) - b
    x0 = rng.normal(size=n)
    for xtol in (1e-8, 1e-13):
        eager = sim_solve(sim, f, None, x0, xtol, 1.0, analytic=False)
        lazy = sim_solve(sim, f, None, x0, xtol, 1.0, analytic=False, blocked=2)
        if cap < 2:
            same(lazy, eager)                                   # no room for a list: the eager code runs
            continue
        # (the systems of n = 15 and n = 64 end with info 4 after hundreds of evaluations -- on both sides; R is not compared: once
        # |f| is at rounding level Broyden's update divides noise by the step length and the factors drift apart, eagerly or not)
        assert lazy["info"] == eager["info"] and abs(lazy["nfev"] - eager["nfev"]) <= 0.02 * eager["nfev"]
        scale = np.max(np.abs(eager["x"]))
        if eager["info"] == 1:
            assert np.max(np.abs(lazy["x"] - eager["x"])) <= 1e-12 * scale
            assert np.linalg.norm(lazy["fvec"]) <= 10 * max(np.linalg.norm(eager["fvec"]), 1e-14)
        else:
            assert np.max(np.abs(lazy["x"] - eager["x"])) <= 1e-4 * scale


def test_q_kept_as_factorised_on_the_analytic_cases(sim):
    for name, f, x0 in CASES:
        eager = sim_solve(sim, f, None, x0, 1e-8, 1.0, analytic=False)
        lazy = sim_solve(sim, f, None, x0, 1e-8, 1.0, analytic=False, blocked=2)
        assert lazy["info"] == eager["info"], name
        assert np.allclose(lazy["x"], eager["x"], rtol=1e-9, atol=1e-12), name
