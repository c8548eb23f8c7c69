"""Time the Jacobian refresh of the device solvers on its own (socp_qr_factor_batch): `count` problems of n unknowns, both flavours.
    python scripts/measure_factor.py [n = 253] [count = 2048] [reps = 3]
Prints one JSON line; algorithmic bytes = read J + write Q, R (n^2 + n^2 + n(n+1)/2 doubles), flops = 8/3 n^3 (qrfac + qform)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socp_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 253
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(1)
J = rng.standard_normal((count, n, n))
J[:, np.arange(n), np.arange(n)] += 0.5 * np.sqrt(n)
b = rng.standard_normal((count, n))
out = {"n": n, "count": count, "reps": reps, "flop": 8.0 / 3.0 * n ** 3 * count, "algorithmic_bytes": 8.0 * count * (2 * n * n + n * (n + 1) / 2)}
flavours = [("exact", capi.FACTOR_EXACT)] + ([("fast", capi.FACTOR_FAST)] if 39 <= n <= 256 else [])
if os.environ.get("SOCP_MEASURE_ONLY"):
    flavours = [f for f in flavours if f[0] == os.environ["SOCP_MEASURE_ONLY"]]
for name, fl in flavours:
    capi.qr_factor_batch(J[:8], b[:8], flavour=fl, outputs=False)             # code-object load
    if count >= 640:
        # (from 640 problems up the throughput flavour's qrfac is a chain of launches of OTHER kernels than the 8-problem call has run:
        # their first launches belong to the warm-up too)
        capi.qr_factor_batch(J[:640], b[:640], flavour=fl, outputs=False)
    ms = capi.qr_factor_batch(J, b, flavour=fl, reps=reps, outputs=False)["kernel_ms"]
    out[name] = {"kernel_ms": ms, "tflops": out["flop"] / (ms * 1e-3) / 1e12, "frac_of_fp64_peak_78.6": out["flop"] / (ms * 1e-3) / 78.6e12}
print(json.dumps(out))
