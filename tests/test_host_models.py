"""CPU: a user model that overrides ONLY the reference's host virtuals (Model / Control / Hamiltonian; no device twin) still
runs through the mirror: odeTools::RK1/RK2/RK4 in both call forms (odeTools.cpp:46-98, interceptor.cpp:117), the host
integrate() loop (odeTools.cpp:128-146) and shooting::SolveOCP with the residual assembled from the model's virtuals
(shooting.cpp:918-993).  VERDICT r1 #4.  This is the plugin surface for classes without device dynamics -- in-tree models
never take it: without a GPU they fail loudly (last test)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "socp_amd", "_build", "bin")


def run(*args):
    out = subprocess.run([os.path.join(BIN, "hostmodel_flow")] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    recs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    return out, recs


@pytest.mark.parametrize("M", [1, 4])
def test_host_virtual_model_solves_through_shooting(M):
    out, recs = run(M)
    assert out.returncode == 0, out.stderr
    r = recs[0]
    assert r["info"] == 1 and r["n"] == 4 * M
    assert r["steps_ok"] == 1 and r["struct_ok"] == 1          # RK1/RK2/RK4 fn-pointer form bit-equal to the hand formulas
    # analytic optimum of the rest-to-rest transfer: p_x = -12, p_v(0) = -6, u(0) = 6 (cubic solution: RK4 is exact)
    assert abs(r["p_x"] + 12) < 1e-9 and abs(r["p_v"] + 6) < 1e-9 and abs(r["u0"] - 6) < 1e-9
    assert r["trajectories"] == r["nfev"] * M                  # n sequential callbacks per FD Jacobian, as the reference
    assert out.stderr.count("no device dynamics") == 1         # the one-line warning, once


@pytest.mark.parametrize("threads", [1, 2, 4])
def test_exception_in_user_model_code_reaches_the_caller(threads):
    """the segment workers of the host shooting path (numThread > 1, csrc/host_pool.hpp) run USER model code.  A throw
    inside it -- on the calling thread or on a worker -- is rethrown to the caller of the evaluation, as it would be from the
    serial path (round 3: std::terminate from a worker); the pool drains, stays usable, and computes the same numbers afterwards."""
    out, recs = run("throws", threads)
    assert out.returncode == 0, (out.returncode, out.stderr[-1000:])
    assert recs[0] == {"threads": threads, "caught": 1, "caught_again": 1, "same_after": 1}


def test_host_residual_rows_against_an_independent_restatement():
    """Every row kind of the host assembly (shooting.cpp:945-990, model.hpp:90-328): FREE interior time ->
    SwitchingTimesFunction row, FREE final time -> H row, FIXED / CONTINUOUS interior state modes, FREE final component."""
    out, recs = run("residual")
    assert out.returncode == 0, out.stderr
    z, F = np.array(recs[0]["z"]), np.array(recs[0]["F"])
    assert len(z) == 14                                         # 3 nodes x 4 + two FREE times

    def rhs(X):
        return np.array([X[1], -X[3], 0.0, -X[2]])

    def H(X):
        u = -X[3]
        return u * u / 2 + X[2] * X[1] + X[3] * u

    def traj(t0, X, tf, N=7):
        dt = (tf - t0) / N
        t, X = t0, X.copy()
        while t < tf - dt / 2:
            h = tf - t if t + dt > tf else dt
            F1 = rhs(X); F2 = rhs(X + (h / 2.0) * F1); F3 = rhs(X + (h / 2.0) * F2); F4 = rhs(X + h * F3)
            X = X + (h / 6.0) * (F1 + (F4 + 2.0 * (F2 + F3)))
            t += dt
        return X

    # layout of the program: mode_t = [FIXED, FREE, CONTINUOUS, FREE]; desired data = its initial guess
    vt = np.array([0.4 * i + 0.01 * i * i for i in range(4)])
    vX = np.array([[0.3 * i, 0.1 + 0.05 * i, -1.0 - 0.1 * i, -0.7 + 0.2 * i] for i in range(4)])
    t1, t3 = z[12], z[13]
    tl = [vt[0], t1, t1 + (t3 - t1) / 2, t3]                  # CONTINUOUS node 2 spaced uniformly between the junctions
    X0, X1, X2 = z[0:4], z[4:8], z[8:12]
    want = np.zeros(14)
    want[0:2] = X0[0:2] - vX[0, 0:2]                            # InitialFunction, both FIXED
    E0 = traj(tl[0], X0, tl[1])
    want[12] = H(E0) - H(X1)                                    # default SwitchingTimesFunction at the FREE interior time
    want[4], want[6] = E0[0] - vX[1, 0], X1[0] - vX[1, 0]       # node 1: position FIXED pins both sides
    want[5], want[7] = E0[1] - X1[1], E0[3] - X1[3]             #         velocity CONTINUOUS: state and costate jump
    E1 = traj(tl[1], X1, tl[2])
    want[8:10] = E1[0:2] - X2[0:2]
    want[10:12] = E1[2:4] - X2[2:4]
    E2 = traj(tl[2], X2, tl[3])
    want[2] = E2[0] - vX[3, 0]                                  # FinalHFunction: position FIXED,
    want[3] = E2[3]                                             #   velocity FREE -> transversality p_v = 0,
    want[13] = H(E2)                                            #   free tf -> H = 0
    assert np.max(np.abs(F - want)) < 1e-14, (F, want)


def test_segment_workers_on_the_host_path_change_nothing():
    """numThread > 1 for a class without device dynamics: the segments of a residual go to a persistent pool of workers in the
    reference's contiguous blocks (shooting.cpp:1223-1231); every number of the solve is the serial one."""
    ref_out, ref = run(4, 1)
    assert ref_out.returncode == 0
    for threads in (2, 3, 4, 7):
        out, recs = run(4, threads)
        assert out.returncode == 0, out.stderr
        assert recs[0]["z"] == ref[0]["z"] and recs[0]["nfev"] == ref[0]["nfev"] and recs[0]["trajectories"] == ref[0]["trajectories"]


def runjac(*args):
    out = subprocess.run([os.path.join(BIN, "hostjac_flow")] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    return out, [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("xtol", ["1e-08", "1e-12"])
def test_host_virtual_model_with_variational_equations_runs_hybrj(xtol):
    """modelOrder = 1 for a class WITHOUT a device twin: a host restatement of the double integrator, written
    against the plugin surface only (Model(t, X, 1), Hamiltonian(t, X, 1)), goes through the reference's hybrj scheme -- host
    Jacobian assembly after shooting.cpp:996-1130 -- and reproduces the Newton histories of tests/testDoubleIntegrator.cpp:
    (nfev, njev) = (32, 4), (14, 1), (127, 2) in the survey's scipy count (= 30 / 12 / 125 raw at xtol 1e-8) and the golden
    unknowns of the oracle / device path bit for bit."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "dint_flow.json")))["basic_order1_xtol" + xtol]
    out, recs = runjac("basic", xtol, 1)
    assert out.returncode == 0, out.stderr
    assert len(recs) == 3
    for r, g in zip(recs, gold):
        assert (r["info"], r["nfev"], r["njev"]) == (g["info"], g["nfev"], g["njev"])
        assert r["z"] == g["z"]
    if xtol == "1e-08":
        assert [(r["nfev"] + 2, r["njev"]) for r in recs] == [(32, 4), (14, 1), (127, 2)]


@pytest.mark.parametrize("threads", [1, 2])
def test_host_hybrj_way_point_program_serial_and_threaded(threads):
    """tests/testDoubleIntegrator_WP.cpp (two segments, FREE interior and final time, mixed FIXED / CONTINUOUS way-point modes,
    numThread = 2 in the reference's program) on the host hybrj path: ier = 4 for the first solve, then (60, 5) and (115, 4)
    in the survey's count, unknowns equal to the golden ones, with one worker and with two."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "dint_flow.json")))["wp_order1_xtol1e-08"]
    out, recs = runjac("wp", "1e-08", threads)
    assert out.returncode == 0, out.stderr
    for r, g in zip(recs, gold):
        assert (r["info"], r["nfev"], r["njev"]) == (g["info"], g["nfev"], g["njev"])
        assert r["z"] == g["z"]
    assert recs[0]["info"] == 4 and [(r["nfev"] + 2, r["njev"]) for r in recs[1:]] == [(60, 5), (115, 4)]


def test_in_tree_models_still_have_no_cpu_path():
    """The host path is for classes WITHOUT device dynamics only.  goddard has a device twin: without a GPU its solve must
    fail loudly (no silent CPU fallback), here and in every CPU-only environment."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the device path runs")
    out = subprocess.run([os.path.join(BIN, "goddard_flow"), "full", "10", "1", "1e-6"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "no HIP device" in out.stderr or "no CPU path" in out.stderr or "device" in out.stderr.lower()
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
