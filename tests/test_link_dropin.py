"""CPU: the drop-in boundary at LINK level, exercised with the reference's own caller.

oracle/_ref/link/ (built by `make -C oracle link` in the authoring container, travels as binaries) holds the
reference's unmodified shooting.cpp + odeTools.cpp + goddard.cpp compiled against THIS repository's
include/cminpack.h and linked with libsocp_hip.so -- i.e. `hybrd`/`hybrj`, the only undefined symbols of the
reference's shooting.o (shooting.cpp:16,803-826,830-851), resolved by the product instead of CMinPack:

  testGoddard_linked   the reference's tests/testGoddard.cpp, unmodified
  goddard_flow_ref     tests/cpp/goddard_flow.cpp (reference public API only) -> prints nfev and z per solve

The residual is the reference's CPU code, the Newton iteration is the product's: four `OK = 1`, MINPACK's
evaluation counts 1184/188/638/101 (= 1186/190/640/103 through scipy.fsolve, which adds two calls) and the
converged unknowns equal to the goldens (SciPy MINPACK on the oracle residual) to the last bit.
This is a boundary test, not an oracle pin (the oracle is pinned by libsocp_ref.so, which contains no product code).
No GPU is involved: hybrd/hybrj are host code.
"""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINK = os.path.join(ROOT, "oracle", "_ref", "link")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "goddard_flow.json")))

needs_link = pytest.mark.skipif(not os.path.exists(os.path.join(LINK, "goddard_flow_ref")),
                                reason="oracle/_ref/link not built (needs /root/reference: make -C oracle link)")


@needs_link
def test_reference_shooting_objects_bind_to_our_hybrd():
    for exe in ("testGoddard_linked", "goddard_flow_ref"):
        syms = subprocess.run(["nm", "-u", os.path.join(LINK, exe)], capture_output=True, text=True).stdout.split()
        assert "hybrd" in syms and "hybrj" in syms                     # resolved at load time ...
        ldd = subprocess.run(["ldd", os.path.join(LINK, exe)], capture_output=True, text=True).stdout
        assert "libsocp_hip.so" in ldd and "not found" not in ldd      # ... by the product library


@needs_link
def test_reference_testGoddard_unmodified_converges_four_times(tmp_path):
    out = subprocess.run([os.path.join(LINK, "testGoddard_linked")], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    assert out.stdout.count("OK = 1") == 4, out.stdout


@needs_link
@pytest.mark.parametrize("threads", [1, 2])
def test_call_numbers_and_solutions_of_the_reference_flow(threads):
    """GetCallNumber per solve and the converged unknowns; numThread = 2 goes through the reference's
    ShootingFunctionParallel (shooting.cpp:1133-1158) and must not change a bit."""
    env = dict(os.environ, SOCP_FLOW_THREADS=str(threads))
    out = subprocess.run([os.path.join(LINK, "goddard_flow_ref"), "full", "10", "1", "1e-6"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    stages = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [s["info"] for s in stages] == [1, 1, 1, 1]
    assert [s["nfev"] for s in stages] == [1184, 188, 638, 101]
    assert [s["nfev"] + 2 for s in stages] == [1186, 190, 640, 103]           # SURVEY 6 (counted through scipy.fsolve)
    assert [s["n"] for s in stages] == [85, 85, 85, 87]
    for s, g in zip(stages, GOLD["goddard_N10_M6"]):
        assert s["stage"] == g["stage"]
        assert np.array_equal(np.array(s["z"]), np.array(g["z"])), s["stage"]
