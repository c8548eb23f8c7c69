#!/usr/bin/env python3
"""Golden vectors of the DEFAULT RESIDUAL BLOCKS of the reference's header-only model.hpp (:90-328),
generated from the REFERENCE ITSELF (oracle/_ref/libsocp_ref.so: its goddard / doubleIntegrator / covid19 objects through
oracle/ref_driver.cpp: ref_model_block) -- InitialFunction, InitialHFunction, FinalFunction, FinalHFunction and
SwitchingTimesFunction, value form (isJac = 0) for all three models and Jacobian form (isJac = 1) for the one model that
has variational equations (doubleIntegrator, modelOrder 1).  Only numbers are written.

    python tests/golden/make_block_golden.py        (authoring container: needs /root/reference, make -C oracle ref)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import Ref, have_ref, MODEL_GODDARD, MODEL_DINT, MODEL_COVID  # noqa: E402

CASES = 8


def inputs(rng):
    """Per model: states at the node, desired boundary states, states after a switching time, mode vectors (a FIXED /
    FREE mix per case, first case all FIXED, second all FREE), evaluation times."""
    out = {}
    g0 = np.array([0.999949994, 1e-4, 0.01, 0.02, -0.01, 0.03, 0.9, -8.12, 7.8e-3, 0.78, -0.48, 5.7e-4, 5.7e-2, 0.0996])
    c0 = np.array([0.93, 0.003, 0.01, 0.057, -0.001, 0.001, 0.0, 0.0]) + 1e-3
    for tag, d, base in (("g", 7, g0), ("d", 6, None), ("c", 4, c0)):
        s = 2 * d
        X = rng.uniform(-2, 2, (CASES, s)) if base is None else base * (1 + 0.2 * rng.uniform(-1, 1, (CASES, s)))
        Xp = X * (1 + 0.05 * rng.uniform(-1, 1, (CASES, s)))
        Xd = rng.uniform(-1, 1, (CASES, s))
        mode = rng.integers(0, 2, (CASES, d)).astype(np.int32)
        mode[0] = 0
        mode[1] = 1
        out[tag + "_X"], out[tag + "_Xp"], out[tag + "_Xd"], out[tag + "_mode"] = X, Xp, Xd, mode
        out[tag + "_t"] = rng.uniform(0, 0.12, CASES)
    # doubleIntegrator augmented states [X ; R], R a full random sensitivity block (not the identity)
    Xa = np.concatenate([out["d_X"], rng.uniform(-1, 1, (CASES, 144))], axis=1)
    Xpa = np.concatenate([out["d_Xp"], rng.uniform(-1, 1, (CASES, 144))], axis=1)
    out["d_Xaug"], out["d_Xpaug"] = Xa, Xpa
    return out


def main():
    assert have_ref(), "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(20251004)
    out = inputs(rng)
    models = {"g": Ref(MODEL_GODDARD), "d": Ref(MODEL_DINT, model_order=1), "c": Ref(MODEL_COVID)}
    models["g"].set_param("mu2", 0.2)
    models["c"].set_params([3.4, 14, 5, 1, 0.1, 1, -10, 20])
    for tag, r in models.items():
        for which in range(5):
            other = out[tag + ("_Xp" if which == 4 else "_Xd")]
            out["%s_block%d" % (tag, which)] = np.stack([
                r.residual_block(which, out[tag + "_t"][k], out[tag + "_X"][k], other[k], out[tag + "_mode"][k], 0)
                for k in range(CASES)])
    r = models["d"]
    for which in range(5):
        other = out["d_Xpaug"] if which == 4 else out["d_Xd"]
        out["d_block%d_jac" % which] = np.stack([
            r.residual_block(which, out["d_t"][k], out["d_Xaug"][k], other[k], out["d_mode"][k], 1) for k in range(CASES)])
    np.savez_compressed(os.path.join(HERE, "reference_blocks.npz"), **out)
    print("wrote reference_blocks.npz with", len(out), "arrays;", {k: v.shape for k, v in out.items() if "block" in k})


if __name__ == "__main__":
    main()
