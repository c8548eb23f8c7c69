"""GPU: the reference-order (exact) Goddard path is BIT-IDENTICAL to the CPU path.

Everything on that path is IEEE + - * / sqrt in the reference's association order, except one library call:
exp (air density).  socp_amd/csrc/exp_glibc.hpp reproduces glibc's exp as x86-64 hosts with FMA run it, so on
such a host (the GPU boxes' EPYC, this container) the device results equal the CPU results to the last bit:
evaluations, 10^4-step trajectories, residuals, forward-difference Jacobians, and therefore every Newton
iterate of the reference's test program.  On a host whose libm takes the non-FMA variant (~0.07 % of exp
results differ by one ulp) these tests skip and the tolerance-based tests of test_gpu_parity.py /
test_host_flow.py are what holds."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from conftest import GODDARD_TF, goddard_c1_problem, goddard_costate_batch
from oracle.oracle import Oracle, MODEL_GODDARD

# arguments on which glibc's FMA-selected exp and its generic build differ by one ulp
_PROBES = [("-0x1.d55ed93149a8p-1", "0x1.996a94308dc25p-2"), ("0x1.97fa03a1789ap+1", "0x1.8392dea58a875p+4"),
           ("-0x1.2f067b539524p+0", "0x1.397e60d1b62c7p-2")]
HOST_LIBM_FMA = all(math.exp(float.fromhex(x)) == float.fromhex(y) for x, y in _PROBES)

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not HOST_LIBM_FMA, reason="host libm is not glibc's FMA exp variant: bitwise parity of exp "
                                                           "is not defined on this host (tolerance tests still apply)")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def random_states(B, seed=99):
    rng = np.random.default_rng(seed)
    X = np.empty((B, 14))
    dirs = rng.normal(size=(B, 3))
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    X[:, 0:3] = dirs * rng.uniform(0.98, 1.05, (B, 1))
    X[:, 3:6] = rng.normal(size=(B, 3)) * 10.0 ** rng.uniform(-10, -0.5, (B, 1))
    X[:, 6] = rng.uniform(0.2, 1.0, B)
    X[:, 7:10] = rng.normal(size=(B, 3)) * 5
    X[:, 10:13] = rng.normal(size=(B, 3)) * 10.0 ** rng.uniform(-3, 0.5, (B, 1))
    X[:, 13] = rng.uniform(-0.5, 0.5, B)
    return X, rng.uniform(0.0, 0.12, B)


@pytest.mark.parametrize("mu2", [1.0, 0.0])
def test_model_control_hamiltonian_bitwise(built, mu2):
    from socp_amd import capi
    params = [3.5, 7.0, 310.0, 500.0, 1.0, 1.0, mu2, -1.0]
    o = Oracle(MODEL_GODDARD, params=params)
    o.set_switching([0.02, 0.08])
    c = capi.Context(capi.MODEL_GODDARD)
    c.set_params(params)
    c.set_switching_times([0.02, 0.08])
    X, t = random_states(3000)
    f = c.eval_batch(capi.EVAL_RHS, t, X)
    u = c.eval_batch(capi.EVAL_CONTROL, t, X)
    h = c.eval_batch(capi.EVAL_HAMILTONIAN, t, X)[:, 0]
    for b in range(len(X)):
        assert np.array_equal(f[b], o.rhs(t[b], X[b])), b
        assert np.array_equal(u[b], o.control(t[b], X[b])), b
        assert h[b] == o.hamiltonian(t[b], X[b])[0], b
    c.close()


def test_ten_thousand_step_trajectories_bitwise(built):
    """The BASELINE metric's unit of work: 14-dim state + costate, 10^4 RK4 steps, with drag (KD = 310)."""
    from socp_amd import capi
    params = [3.5, 7.0, 310.0, 500.0, 1.0, 1.0, 1.0, -1.0]
    o = Oracle(MODEL_GODDARD, step_nbr=10000, params=params)
    c = capi.Context(capi.MODEL_GODDARD)
    c.set_params(params)
    c.set_step_number(10000)
    X0 = goddard_costate_batch(24, 1e-3)
    Xg = c.integrate_batch(0.0, GODDARD_TF, X0)
    Xc = o.integrate_batch(0.0, GODDARD_TF, X0)
    assert np.array_equal(Xg, Xc)
    c.close()


def test_residual_and_fd_jacobian_bitwise(built):
    from socp_amd import capi
    o = Oracle(MODEL_GODDARD)
    o.set_param("KD", 310.0)
    o.set_param("mu2", 1.0)
    prob, z = goddard_c1_problem(o)
    c = capi.Context(capi.MODEL_GODDARD)
    c.set_params([3.5, 7.0, 310.0, 500.0, 1.0, 1.0, 1.0, -1.0])
    assert c.problem_set(prob.mode_t, prob.mode_x, prob.time, prob.xnode) == 85
    F = c.residual(z)
    assert np.array_equal(F, o.residual(prob, z))
    J = c.fd_jacobian(z, F, dedup=True)
    assert np.array_equal(J, o.fdjac(prob, z, F))        # FD entries compared bit for bit: same F, same h, same division
    c.close()


def test_reference_test_program_newton_history_bitwise():
    """tests/testGoddard.cpp through the C++ mirror, as shipped (xtol 1e-6, trivial guess, KD and mu2 continuation,
    singular arc): info, evaluation count and the converged unknowns of ALL FOUR solves equal the CPU path's
    (golden: SciPy MINPACK over the oracle) -- the first solve alone is 1184 evaluations whose Newton path changes
    entirely under a one-ulp perturbation."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "goddard_flow.json")))["goddard_N10_M6"]
    exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "goddard_flow")
    out = subprocess.run([exe, "full", "10", "1", "1e-6"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, SOCP_VARIANT="exact"))
    stages = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(stages) == 4, out.stderr
    for s, g in zip(stages, gold):
        assert (s["stage"], s["info"], s["nfev"]) == (g["stage"], 1, g["nfev"])
        assert np.array_equal(np.array(s["z"]), np.array(g["z"])), s["stage"]


def test_unmodified_reference_program_reports_success(tmp_path):
    """The reference's own tests/testGoddard.cpp, compiled unchanged against the mirror in the authoring container
    (scripts/dropin_build.sh; the binary travels, the source does not): OK = 1 for all four solves."""
    exe = os.path.join(ROOT, "oracle", "_ref", "dropin", "bin", "testGoddard")
    if not os.path.exists(exe):
        pytest.skip("drop-in binary not built (needs the reference tree: scripts/dropin_build.sh)")
    work = tmp_path / "a" / "b"
    work.mkdir(parents=True)
    (tmp_path / "trace" / "goddard").mkdir(parents=True)
    out = subprocess.run([exe], cwd=work, input="\n", capture_output=True, text=True, timeout=600)
    oks = [l.split("OK =")[1].strip() for l in out.stdout.splitlines() if "OK =" in l]
    assert oks == ["1", "1", "1", "1"], out.stdout


def _run_dropin(tmp_path, prog):
    exe = os.path.join(ROOT, "oracle", "_ref", "dropin", "bin", prog)
    if not os.path.exists(exe):
        pytest.skip("drop-in binary not built (needs the reference tree: scripts/dropin_build.sh)")
    work = tmp_path / "a" / "b"
    work.mkdir(parents=True)
    for m in ("goddard", "doubleIntegrator", "covid19", "interceptor"):
        (tmp_path / "trace" / m).mkdir(parents=True)
    out = subprocess.run([exe], cwd=work, input="\n", capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    return out.stdout


def _printed_solutions(stdout):
    """The `SOL :` lines of the double-integrator programs: (nCallFunc, nCallJac) when printed, tf and the six costates."""
    import re
    sols = []
    for line in stdout.splitlines():
        if "SOL :" not in line:
            continue
        calls = re.search(r"nCallFunc = (\d+), nCallJac = (\d+)", line)
        vals = [float(v) for v in re.findall(r"(?:tf|p_\w+) = ([-+0-9.eE]+)", line)]
        sols.append((None if calls is None else (int(calls.group(1)), int(calls.group(2))), np.array(vals)))
    return sols


def test_unmodified_double_integrator_programs_print_the_cpu_solutions(tmp_path, built):
    """The reference's tests/testDoubleIntegrator.cpp and testDoubleIntegrator_WP.cpp, compiled unchanged against the mirror
    (hybrj over the device's variational Jacobian): every solve returns 1, the printed call counts are MINPACK's on the CPU path
    ((30, 4) and (58, 5): SURVEY 6's (32,4) / (60,5) are SciPy's count, two more) and the printed tf / costates are the CPU
    path's (the same flows over the oracle, tests/flow_oracle.py) in every digit the programs print."""
    from flow_oracle import dint_basic_flow, dint_wp_flow
    out = _run_dropin(tmp_path, "testDoubleIntegrator")
    assert out.count("Algo returned 1") == 3, out
    sols = _printed_solutions(out)
    cpu = dint_basic_flow("socp", 1)
    assert len(sols) == 3 and sols[0][0] == (cpu[0]["nfev"], cpu[0]["njev"]) == (30, 4)
    for (calls, vals), want in zip(sols, cpu):
        z = want["z"]
        ref = np.concatenate([[z[12]], z[6:12]])
        assert np.array_equal(vals, [float("%.6g" % v) for v in ref]), (vals, ref)       # the digits std::cout prints, all of them
    trace = np.loadtxt(tmp_path / "trace" / "doubleIntegrator" / "trace.dat")
    assert trace.ndim == 2 and trace.shape[0] > 10 and np.all(np.isfinite(trace)) and np.all(np.diff(trace[:, 0]) >= 0)

    out = _run_dropin(tmp_path / "wp", "testDoubleIntegrator_WP")
    assert "Algo returned 1" in out and "OK = 1" in out, out
    sols = _printed_solutions(out)
    cpu = dint_wp_flow("socp", 1)
    assert len(sols) == 1 and sols[0][0] == (cpu[1]["nfev"], cpu[1]["njev"]) == (58, 5)
    z = cpu[1]["z"]
    assert np.array_equal(sols[0][1], [float("%.6g" % v) for v in np.concatenate([[z[24]], z[6:12]])])


@pytest.mark.parametrize("prog,solves,trace_files", [("testCovid19", 3, ["covid19/trace.dat"]),
                                                     ("testInterceptor", 3, ["interceptor/trace_S1.dat", "interceptor/trace_S2.dat",
                                                                             "interceptor/trace_S3.dat"])])
def test_unmodified_covid_and_interceptor_programs_report_success(tmp_path, prog, solves, trace_files):
    """tests/testCovid19.cpp (M = 20, 1000 steps, two data continuations) and tests/testInterceptor.cpp (three scenarios, parameter
    continuation; fixed-step RK4 as the reference ships without _USE_BOOST), compiled unchanged: OK = 1 for every solve and the
    trace files they write are complete (finite, time non-decreasing within the file's segments)."""
    out = _run_dropin(tmp_path, prog)
    oks = [l.split("OK =")[1].split(",")[0].strip() for l in out.splitlines() if "OK =" in l]
    assert oks == ["1"] * solves, out
    for f in trace_files:
        t = np.loadtxt(tmp_path / "trace" / f)
        assert t.ndim == 2 and t.shape[0] > 50 and np.all(np.isfinite(t)), f
        assert t[0, 0] <= t[-1, 0]


def test_degenerate_states_take_the_plain_division_path_and_still_match(built):
    """Shared-denominator division (models_exact.hpp: Den) is used only for denominators within 2^-400..2^400;
    zero speed, zero p_v and states scaled by 10^+-150 go through plain IEEE division -- and must equal the CPU
    path as well (NaN where the CPU has NaN)."""
    from socp_amd import capi
    params = [3.5, 7.0, 310.0, 500.0, 1.0, 1.0, 1.0, -1.0]
    o = Oracle(MODEL_GODDARD, params=params)
    c = capi.Context(capi.MODEL_GODDARD)
    c.set_params(params)
    X, t = random_states(400, seed=5)
    X[:100, 3:6] = 0.0
    X[100:200, 10:13] = 0.0
    X[200:300] *= 1e150
    X[300:400] *= 1e-150
    f = c.eval_batch(capi.EVAL_RHS, t, X)
    with np.errstate(all="ignore"):
        ref = np.array([o.rhs(t[b], X[b]) for b in range(len(X))])
    assert np.array_equal(f, ref, equal_nan=True)
    assert np.isnan(ref).any() and np.isinf(ref).any() and np.isfinite(ref).any()   # the set really contains all kinds
    c.close()


def test_extreme_air_density_arguments_bitwise(built):
    """exp(-kr (r - 1)) over the whole range glibc treats by its main algorithm and by its x <= -512 branch
    (radii 0.05 .. 2.35 => arguments +475 .. -675): the right-hand side still equals the CPU path bit for bit."""
    from socp_amd import capi
    params = [3.5, 7.0, 310.0, 500.0, 1.0, 1.0, 1.0, -1.0]
    o = Oracle(MODEL_GODDARD, params=params)
    c = capi.Context(capi.MODEL_GODDARD)
    c.set_params(params)
    X, t = random_states(4000, seed=11)
    rng = np.random.default_rng(12)
    radius = rng.uniform(0.05, 2.35, len(X))
    X[:, 0:3] *= (radius / np.linalg.norm(X[:, 0:3], axis=1))[:, None]
    f = c.eval_batch(capi.EVAL_RHS, t, X)
    with np.errstate(all="ignore"):
        ref = np.array([o.rhs(t[b], X[b]) for b in range(len(X))])
    assert np.array_equal(f, ref, equal_nan=True)
    assert np.sum(radius > 2.03) > 300                  # the x <= -512 branch is well represented
    c.close()
