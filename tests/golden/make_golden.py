#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE ITSELF.

Run in the authoring container only (needs /root/reference and `make -C oracle ref`): the values
come from the reference's own goddard / doubleIntegrator / odeTools objects through
oracle/ref_driver.cpp -- never from the oracle restatement.  Only numbers are written (inputs and
the reference's outputs); no reference source travels.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import Ref, have_ref, MODEL_GODDARD, MODEL_DINT, MODEL_COVID  # noqa: E402

X0S = np.array([0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0])
PSTAR = np.array([-8.121947733, 7.775439382e-3, 0.7775438809, -0.4779369965, 5.715013318e-4,
                  5.715009222e-2, 9.958404873e-2])
TF = 0.2640825


def goddard_points(rng, n):
    X = np.tile(np.concatenate([X0S, PSTAR]), (n, 1)) * (1 + 0.3 * rng.uniform(-1, 1, (n, 14)))
    X[:, 3:6] = rng.uniform(-0.1, 0.1, (n, 3))
    t = rng.uniform(0, 0.12, n)          # straddles the default switching times 0.0227 / 0.08
    return t, X


def main():
    assert have_ref(), "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(20250905)
    out = {}

    # G1 -- Goddard Model / Control / Hamiltonian incl. every control branch
    t, X = goddard_points(rng, 32)
    out["g_t"], out["g_X"] = t, X
    for mu2 in (1.0, 0.2, 0.0):
        r = Ref(MODEL_GODDARD)
        r.set_param("mu2", mu2)
        tag = "g_mu2_%s" % str(mu2).replace(".", "p")
        out[tag + "_rhs"] = np.stack([r.rhs(t[i], X[i]) for i in range(32)])
        out[tag + "_ctl"] = np.stack([r.control(t[i], X[i]) for i in range(32)])
        out[tag + "_ham"] = np.array([r.hamiltonian(t[i], X[i])[0] for i in range(32)])
    # saturated control (|alpha| > u_max) and constant singular approximation
    r = Ref(MODEL_GODDARD)
    r.set_param("mu2", 1e-3)
    out["g_sat_rhs"] = np.stack([r.rhs(t[i], X[i]) for i in range(32)])
    r = Ref(MODEL_GODDARD)
    r.set_param("mu2", 0.0)
    r.set_param("singularControl", 0.6)
    out["g_singconst_rhs"] = np.stack([r.rhs(t[i], X[i]) for i in range(32)])

    # G1 -- doubleIntegrator: state RHS, augmented (variational) RHS, control, H and dH/dX
    Xd = rng.uniform(-2, 2, (32, 12))
    Xd[16:, 9:12] *= 0.2                 # unsaturated controls in the second half
    out["d_X"] = Xd
    rd = Ref(MODEL_DINT, model_order=1)
    out["d_rhs"] = np.stack([rd.rhs(0.0, Xd[i]) for i in range(32)])
    out["d_ctl"] = np.stack([rd.control(0.0, Xd[i]) for i in range(32)])
    out["d_ham"] = np.array([rd.hamiltonian(0.0, Xd[i])[0] for i in range(32)])
    out["d_dham"] = np.stack([rd.hamiltonian(0.0, Xd[i], 1) for i in range(32)])
    Xa = np.zeros((8, 156))
    Xa[:, :12] = Xd[:8]
    Xa[:, 12:] = rng.uniform(-1, 1, (8, 144))
    out["d_Xaug"] = Xa
    out["d_rhs_aug"] = np.stack([rd.rhs(0.0, Xa[i], 1) for i in range(8)])

    # G2 -- one RK4 step and whole segments
    r = Ref(MODEL_GODDARD)
    r.set_param("mu2", 1.0)
    starts = np.tile(np.concatenate([X0S, PSTAR]), (6, 1))
    starts[:, 7:] *= 1 + 1e-3 * rng.uniform(-1, 1, (6, 7))
    out["g_rk4_in"] = starts
    out["g_rk4_out"] = np.stack([r.rk4_step(0.01, starts[i], 2.5e-3) for i in range(6)])
    out["g_traj_X0"] = starts
    for N in (10, 1000, 10000):
        rN = Ref(MODEL_GODDARD, step_nbr=N)
        rN.set_param("mu2", 1.0)
        out["g_traj_N%d" % N] = np.stack([rN.traj(0.0, starts[i], TF) for i in range(6 if N < 10000 else 2)])
    # bang / singular / off arcs with the default switching times, N = 10, tf = 0.1
    r0 = Ref(MODEL_GODDARD, step_nbr=10)
    r0.set_param("mu2", 0.0)
    out["g_traj_mu0_N10"] = np.stack([r0.traj(0.0, starts[i], 0.1) for i in range(6)])
    # zero-length and backward segments return the input (odeTools.cpp:136)
    out["g_traj_zero"] = r0.traj(0.05, starts[0], 0.05)
    out["g_traj_back"] = r0.traj(0.08, starts[0], 0.02)
    # doubleIntegrator segments, N = 30 (model ctor), state and variational
    out["d_traj_X0"] = Xd[:6]
    out["d_traj"] = np.stack([rd.traj(0.0, Xd[i], 7.5) for i in range(6)])
    Xi = np.zeros((3, 156))
    Xi[:, :12] = Xd[:3]
    for k in range(12):
        Xi[:, 12 * (k + 1) + k] = 1.0
    out["d_traj_aug_X0"] = Xi
    out["d_traj_aug"] = np.stack([rd.traj(0.0, Xi[i], 7.5, 1) for i in range(3)])

    # covid19 (SEIR): Model / Control / Hamiltonian incl. saturated control and active I-penalty, segments
    Xc = np.tile(np.array([0.93, 0.003, 0.01, 0.057, -0.001, 0.001, 0.0, 0.0]), (32, 1)) * (1 + 0.5 * rng.uniform(-1, 1, (32, 8)))
    Xc[:, 2] = rng.uniform(0, 0.3, 32)            # I on both sides of Imax = 0.1
    Xc[::3, 4:] += rng.uniform(-300, 300, (11, 4))  # large costates: control hits umin / umax
    rc = Ref(MODEL_COVID)
    rc.set_params([3.4, 14, 5, 1, 0.1, 1, -10, 20])
    out["c_X"] = Xc
    out["c_rhs"] = np.stack([rc.rhs(0.0, Xc[i]) for i in range(32)])
    out["c_ctl"] = np.stack([rc.control(0.0, Xc[i]) for i in range(32)])
    out["c_ham"] = np.array([rc.hamiltonian(0.0, Xc[i])[0] for i in range(32)])
    Xs = np.tile(np.array([0.93, 0.003, 0.01, 0.057, -0.001, 0.001, 0.0, 0.0]), (6, 1)) * (1 + 0.05 * rng.uniform(-1, 1, (6, 8)))
    out["c_traj_X0"] = Xs
    out["c_traj"] = np.stack([rc.traj(0.0, Xs[i], 1.5) for i in range(6)])      # 1000 RK4 steps (covid19.cpp:36)

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_vectors.npz"), "with", len(out), "arrays")


if __name__ == "__main__":
    main()
