#!/usr/bin/env python3
"""The matrix-core Jacobian refresh at EVERY size it is built for (39 <= n <= 256; the test suite samples 15 of them) against the
order-preserving kernel, on the GPU box:  python tests/tools/factor_every_size.py  ->  one JSON line with the worst deviations.
Bars as tests/test_gpu_factor_fast.py: Q^T Q = I to 1e-13 n, Q R = J to 1e-13 |J|, R / Q^T b / column norms within 1e-11 of the
order-preserving kernel's, same signs on diag(R)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from socp_amd import capi  # noqa: E402

worst = {"orth": 0.0, "qr": 0.0, "r_vs_exact": 0.0, "qtb_vs_exact": 0.0, "acnorm_vs_exact": 0.0}
at = dict.fromkeys(worst, 0)
bad = []
sizes = [n for n in range(39, 257) if capi.fast_factor_applies(n)] if hasattr(capi, "fast_factor_applies") else list(range(39, 257))
for n in sizes:
    rng = np.random.default_rng(7000 + n)
    J = rng.standard_normal((2, n, n))
    J[:, np.arange(n), np.arange(n)] += 0.5 * np.sqrt(n)
    J[1] *= rng.choice([-1.0, 1.0], size=(n, 1))
    b = rng.standard_normal((2, n))
    ex = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_EXACT)
    fa = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_FAST)
    scale = np.linalg.norm(J, axis=(1, 2))[:, None, None]
    v = {"orth": float(np.max(np.abs(np.transpose(fa["Q"], (0, 2, 1)) @ fa["Q"] - np.eye(n)[None])) / n),
         "qr": float(np.max(np.abs(fa["Q"] @ fa["R"] - J) / scale)),
         "r_vs_exact": float(np.max(np.abs(fa["R"] - ex["R"]) / scale)),
         "qtb_vs_exact": float(np.max(np.abs(fa["qtb"] - ex["qtb"])) / (np.max(np.abs(b)) * np.sqrt(n))),
         "acnorm_vs_exact": float(np.max(np.abs(fa["acnorm"] - ex["acnorm"]) / ex["acnorm"]))}
    for k in worst:
        if v[k] > worst[k]:
            worst[k], at[k] = v[k], n
    ok = (v["orth"] <= 1e-13 and v["qr"] <= 1e-13 and v["r_vs_exact"] <= 1e-11 and v["qtb_vs_exact"] <= 1e-11 and v["acnorm_vs_exact"] <= 1e-12
          and np.array_equal(np.sign(fa["rdiag"]), np.sign(ex["rdiag"])) and np.array_equal(fa["sing"], ex["sing"]))
    if not ok:
        bad.append(n)
print(json.dumps({"sizes": len(sizes), "first": sizes[0], "last": sizes[-1], "worst": worst, "worst_at_n": at, "sizes_outside_the_bars": bad}))
sys.exit(1 if bad else 0)
