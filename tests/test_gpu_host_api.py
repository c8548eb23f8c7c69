"""GPU: error and ownership conventions of the reference's API as the C++ mirror keeps them:
exit(1) with a message on bad constructor counts, return codes instead of exceptions for solver failure,
caller-owned new[] from GetParameters(), the -1 of the watchdog overload, std::out_of_range for unknown
parameter names -- and the two places where the mirror is deliberately loud: a model class without device
dynamics and the one-step host Runge-Kutta helpers (there is no CPU path)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "socp_amd", "_build", "bin", "api_conventions")


@pytest.mark.parametrize("mode,needle", [("bad_multi", "numMulti should be superior or equal to 1"),
                                         ("bad_thread", "numThread should be superior or equal to 1")])
def test_bad_constructor_counts_exit_1(mode, needle):
    out = subprocess.run([EXE, mode], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and needle in out.stderr


def test_conventions():
    out = subprocess.run([EXE, "checks"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    got = [l.split()[1] for l in out.stdout.splitlines() if l.startswith("ok ")]
    assert got == ["unknown_parameter_name", "out_of_range", "solve_returns_1", "get_parameters_new_array",
                   "timeout_returns_minus_1", "no_device_twin_runs_on_host", "no_device_twin_adaptive_runs_on_host", "host_rk_helpers_run"], out.stdout + out.stderr
    assert "does not exist" in out.stdout


def test_independent_pairs_solve_concurrently():
    """SURVEY 8b threading contract: one (model, shooting) pair per thread may solve concurrently; every thread's
    result equals, bit for bit, what the same start gives in a serial run."""
    import json
    exe = os.path.join(os.path.dirname(EXE), "concurrent_solves")

    def run(serial):
        out = subprocess.run([exe, "6", str(serial)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr
        rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        return {r["thread"]: r for r in rows}

    par, ser = run(0), run(1)
    assert sorted(par) == sorted(ser) == list(range(6))
    for k in range(6):
        assert par[k]["info"] == ser[k]["info"] == 1
        assert par[k]["nfev"] == ser[k]["nfev"] and par[k]["z"] == ser[k]["z"]
