import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the device engine's pinned staging buffers in STRICT mode for the whole suite (socp_amd/csrc/staging.hpp): a host access to a
    # buffer whose asynchronous operation has not been synchronised aborts the process, naming the buffer -- instead of being
    # repaired by a forced synchronise the way a production call would repair it
    os.environ.setdefault("SOCP_STAGING_STRICT", "1")


@pytest.fixture(scope="session", autouse=True)
def _artefacts():
    """Source-only checkout: build the product libraries and the test programs once."""
    need = [os.path.join(ROOT, "socp_amd", "_build", n) for n in
            ("libsocp_hip.so", "libsocp_host.so", "bin/goddard_flow", "bin/dint_flow", "bin/covid_flow", "bin/plugin_flow", "bin/interceptor_flow", "bin/api_conventions", "bin/concurrent_solves", "bin/hostmodel_flow", "bin/hostjac_flow", "bin/sweep_flow",
             "plugins/liblqr1d_plugin.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def built():
    """Make sure the checker library exists (the product library is built by __graft_entry__.build())."""
    from oracle import oracle as orc
    if not os.path.exists(orc.ORACLE_SO):
        orc.build(ref=False)
    return True


# ---- shared problem definitions (workloads of the reference's tests) -------------------------

GODDARD_X0_STATE = np.array([0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0])      # testGoddard.cpp:53-59
# converged initial costates of the KD=310, mu2=1 problem (stage 2 of testGoddard)
GODDARD_PSTAR = np.array([-8.121947733, 7.775439382e-3, 0.7775438809, -0.4779369965,
                          5.715013318e-4, 5.715009222e-2, 9.958404873e-2])
GODDARD_TF = 0.2640825


def goddard_c1_problem(oracle_model):
    """testGoddard.cpp:24-82: M=6, free tf, final velocity and mass free; returns (Problem, z0)."""
    from oracle.oracle import Problem, FIXED, FREE, CONTINUOUS
    M, d, s = 6, 7, 14
    Xi = np.concatenate([GODDARD_X0_STATE, np.full(7, 0.1)])
    Xf = np.zeros(14)
    Xf[0] = 1.01
    ti, tf = 0.0, 0.1
    mode_t = [FIXED] + [CONTINUOUS] * (M - 1) + [FREE]
    mode_x = np.zeros((M + 1, d), dtype=np.int32)
    mode_x[1:M] = CONTINUOUS
    mode_x[M, 3:7] = FREE
    time = np.array([ti + i * (tf - ti) / M for i in range(M + 1)])
    X = np.zeros((M + 1, s))
    X[0], X[M] = Xi, Xf
    for i in range(1, M):
        X[i] = oracle_model.traj(ti, Xi, time[i])      # shooting.cpp:218-222
    z = np.concatenate([X[:M].ravel(), [time[M]]])
    return Problem(d, mode_t, mode_x, time, X), z


def goddard_single_problem(tf=GODDARD_TF):
    """BASELINE config 2: single shooting, fixed tf, n = 14 (SURVEY 8 'C2')."""
    from oracle.oracle import Problem, FIXED, FREE
    d, s = 7, 14
    mode_t = [FIXED, FIXED]
    mode_x = np.zeros((2, d), dtype=np.int32)
    mode_x[1, 3:7] = FREE
    X = np.zeros((2, s))
    X[0, :7] = GODDARD_X0_STATE
    X[1, 0] = 1.01
    z = np.concatenate([GODDARD_X0_STATE, GODDARD_PSTAR])
    return Problem(d, mode_t, mode_x, np.array([0.0, tf]), X), z


def goddard_costate_batch(B, eps, seed=20250905):
    """Synthetic starts p = p*(1 + eps*xi), xi uniform(-1,1) (SURVEY 8d 'Synthetic inputs')."""
    rng = np.random.Generator(np.random.MT19937(seed))
    xi = rng.uniform(-1.0, 1.0, size=(B, 7))
    X0 = np.empty((B, 14))
    X0[:, :7] = GODDARD_X0_STATE
    X0[:, 7:] = GODDARD_PSTAR * (1.0 + eps * xi)
    return X0
