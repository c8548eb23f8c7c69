"""Host solver (libsocp_hip.so: hybrd / hybrj / socp_hybrd_batched / resumable object) against
SciPy's MINPACK (scipy.optimize._minpack) on analytic systems.  CMinPack itself is not available
offline; both are translations of the same Fortran MINPACK, so iterates must agree to
rounding -- in practice bit for bit."""
import numpy as np
import pytest
from scipy.optimize import _minpack, fsolve

from socp_amd import capi


def rosen(x):
    f = np.empty_like(x)
    f[0::2] = 10 * (x[1::2] - x[0::2] ** 2)
    f[1::2] = 1 - x[0::2]
    return f


def rosen_jac(x):
    n = len(x)
    J = np.zeros((n, n))
    for k in range(n // 2):
        J[2 * k, 2 * k] = -20 * x[2 * k]
        J[2 * k, 2 * k + 1] = 10
        J[2 * k + 1, 2 * k] = -1
    return J


def powell(x):
    return np.array([x[0] + 10 * x[1], np.sqrt(5) * (x[2] - x[3]), (x[1] - 2 * x[2]) ** 2, np.sqrt(10) * (x[0] - x[3]) ** 2])


def broyden_tri(x):
    f = (3 - 2 * x) * x + 1
    f[1:] -= x[:-1]
    f[:-1] -= 2 * x[1:]
    return f


def trig(x):
    n = len(x)
    return n - np.sum(np.cos(x)) + np.arange(1, n + 1) * (1 - np.cos(x)) - np.sin(x)


CASES = [("rosen4", rosen, np.array([-1.2, 1.0, -1.2, 1.0])), ("powell", powell, np.array([3.0, -1, 0, 1])),
         ("broyden10", broyden_tri, -np.ones(10)), ("trig8", trig, np.ones(8) / 8), ("broyden50", broyden_tri, -np.ones(50))]


@pytest.mark.parametrize("name,f,x0", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("factor", [1.0, 100.0])
def test_hybrd_matches_scipy_minpack(name, f, x0, factor):
    # the reference's knobs: epsfcn 1e-15, mode 1, ml = mu = n-1 (shooting.cpp:95-105, 789-790)
    xs, info, ier, _ = fsolve(f, x0.copy(), full_output=True, xtol=1e-8, maxfev=10000, epsfcn=1e-15, factor=factor)
    mine = capi.hybrd(f, x0.copy(), xtol=1e-8, epsfcn=1e-15, factor=factor)
    assert mine["info"] == ier
    assert np.array_equal(mine["x"], xs)
    assert np.array_equal(mine["qtf"], info["qtf"])
    assert np.array_equal(mine["r"], info["r"])
    assert np.array_equal(mine["fvec"], info["fvec"])
    # SciPy's python wrapper adds its own shape-check calls; the raw count is MINPACK's
    raw = _minpack._hybrd(f, x0.copy(), (), 1, 1e-8, 10000, -10, -10, 1e-15, factor, None)
    assert mine["nfev"] == raw[1]["nfev"]


def test_nfev_is_number_of_callbacks():
    calls = [0]

    def f(x):
        calls[0] += 1
        return rosen(x)
    out = capi.hybrd(f, np.array([-1.2, 1.0]), epsfcn=1e-15)
    assert out["info"] == 1 and out["nfev"] == calls[0]


def test_batched_fd_stage_gives_identical_iterates():
    """socp_hybrd_batched: the whole forward-difference Jacobian in one call; nfev still += n."""
    jac_calls = [0]

    def fd(x, fvec, epsfcn):
        jac_calls[0] += 1
        eps = np.sqrt(max(epsfcn, np.finfo(float).eps))
        J = np.empty((len(x), len(x)))
        for j in range(len(x)):
            h = eps * abs(x[j]) or eps
            xp = x.copy()
            xp[j] = x[j] + h
            J[:, j] = (broyden_tri(xp) - fvec) / h
        return J
    x0 = -np.ones(12)
    a = capi.hybrd(broyden_tri, x0, epsfcn=1e-15)
    b = capi.hybrd(broyden_tri, x0, epsfcn=1e-15, fdjac=fd)
    assert a["info"] == b["info"] == 1
    assert np.array_equal(a["x"], b["x"]) and a["nfev"] == b["nfev"] and jac_calls[0] >= 1


def test_hybrj_matches_scipy():
    x0 = np.array([-1.2, 1.0, -1.2, 1.0])
    xs, info, ier, _ = fsolve(rosen, x0.copy(), fprime=rosen_jac, full_output=True, xtol=1e-8, factor=1.0)
    mine = capi.hybrj(rosen, rosen_jac, x0.copy(), xtol=1e-8, factor=1.0)
    assert mine["info"] == ier == 1
    assert np.array_equal(mine["x"], xs)
    assert mine["njev"] == info["njev"]


def test_resumable_solver_equals_callback_solver():
    x0 = -np.ones(10)
    ref = capi.hybrd(broyden_tri, x0, epsfcn=1e-15)
    s = capi.HybrSolver(10, xtol=1e-8, epsfcn=1e-15)
    s.start(x0)
    flag, eps = 0, np.sqrt(1e-15)
    while True:
        req, xe, out = s.advance(flag)
        if req == capi.REQ_DONE:
            break
        if req == capi.REQ_FVEC:
            out[:] = broyden_tri(xe.copy())
        else:
            x = xe.copy()
            f0 = s.fvec
            J = np.empty((10, 10))
            for j in range(10):
                h = eps * abs(x[j]) or eps
                xp = x.copy()
                xp[j] = x[j] + h
                J[:, j] = (broyden_tri(xp) - f0) / h
            out[:] = J.T.ravel()       # column-major
    assert s.info == ref["info"] == 1
    assert np.array_equal(s.x, ref["x"]) and s.nfev == ref["nfev"]


def test_negative_callback_return_aborts():
    """shooting.cpp:873: stopFlag < 0 returned from the callback becomes info."""
    calls = [0]

    def f(x):
        calls[0] += 1
        return None if calls[0] > 5 else rosen(x)     # binding maps None -> -1
    out = capi.hybrd(f, np.array([-1.2, 1.0]))
    assert out["info"] == -1


def test_improper_input_is_info_zero():
    out = capi.hybrd(rosen, np.array([-1.2, 1.0]), xtol=-1.0)
    assert out["info"] == 0
    out = capi.hybrd(rosen, np.array([-1.2, 1.0]), factor=0.0)
    assert out["info"] == 0


def test_maxfev_is_info_two():
    out = capi.hybrd(trig, np.ones(8) / 8, maxfev=20)
    assert out["info"] == 2


def test_threaded_factor_work_is_bit_identical():
    """qrfac / qform split their columns over host threads for large n; every column is updated by the serial
    sequence of operations, so iterates do not depend on the thread count (n = 260 > the 192 threshold)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from socp_amd import capi
n = 260
rng = np.random.default_rng(3)
A = rng.normal(size=(n, n)) / np.sqrt(n) + 2 * np.eye(n)
b = rng.normal(size=n)
f = lambda x: A @ x + 0.3 * np.sin(x) - b
r = capi.hybrd(f, np.zeros(n), xtol=1e-12, epsfcn=1e-15)
print(json.dumps({"info": r["info"], "nfev": r["nfev"], "x": [float(v).hex() for v in r["x"]]}))
''' % root
    res = {}
    # SOCP_LINALG_VECTOR=0: MINPACK's scalar column algorithm; default: the same chains with the columns in SIMD lanes
    for vec in ("0", "1"):
        for t in ("1", "3", "8"):
            out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                                 env=dict(os.environ, SOCP_LINALG_THREADS=t, SOCP_LINALG_VECTOR=vec))
            assert out.returncode == 0, out.stderr[-2000:]
            res[vec, t] = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["0", "1"]["info"] == 1
    assert all(r == res["0", "1"] for r in res.values())


def test_simd_column_factor_work_is_bit_identical_to_the_scalar_column_algorithm(tmp_path):
    """colvec::factor (qrfac + qtf + R + qform with eight columns per vector) against qrfac_nopivot / qform: Q, R, rdiag,
    column norms and qtf equal to the last bit -- sizes below, at and across the 32-column block / panel boundaries, with a
    zero column and a zero sub-column (identity reflectors), 1 and several threads (tests/tools/qr_bench.cpp)."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "qr_bench")
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"
    subprocess.check_call([cxx, "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-psabi", "-I" + os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "tests", "tools", "qr_bench.cpp"), "-lpthread"])
    for n, threads in ((5, 1), (15, 1), (16, 1), (17, 1), (24, 1), (31, 1), (32, 1), (33, 1), (64, 2), (85, 1), (97, 3), (127, 1), (260, 4)):
        out = subprocess.run([exe, str(n), str(threads), "1"], capture_output=True, text=True, timeout=600)
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        assert out.returncode == 0 and rec["bit_identical"] is True, (n, threads, out.stdout, out.stderr)
    # the ISA picked at load time changes the vector width, never a rounding: the baseline (SSE2) build and an AVX2-only build
    # of the same kernels are bit-identical to the scalar algorithm too
    import platform
    if platform.machine() == "x86_64":
        for tag, flags in (("sse2", ["-DSOCP_ISA_CLONES="]), ("avx2", ["-DSOCP_ISA_CLONES=", "-mavx2"])):
            if tag == "avx2" and "avx2" not in open("/proc/cpuinfo").read():
                continue
            exe2 = str(tmp_path / ("qr_bench_" + tag))
            subprocess.check_call([cxx, "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-psabi"] + flags +
                                  ["-I" + os.path.join(root, "include"), "-o", exe2, os.path.join(root, "tests", "tools", "qr_bench.cpp"), "-lpthread"])
            for n, threads in ((33, 1), (97, 2), (260, 3)):
                out = subprocess.run([exe2, str(n), str(threads), "1"], capture_output=True, text=True, timeout=600)
                rec = json.loads(out.stdout.strip().splitlines()[-1])
                assert out.returncode == 0 and rec["bit_identical"] is True, (tag, n, threads, out.stdout, out.stderr)
