#!/usr/bin/env python3
"""Application-level timing: one full Newton solve of the testGoddard problem (M = 6, n = 85, KD
continuation stage) at the benchmark's 1e4 RK4 steps per segment -- GPU through the C++ mirror vs the CPU
restatement (oracle + the same hybrd) on one host core.  Run on the GPU box."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flow_oracle import goddard_single_stage  # noqa: E402

G = json.load(open(os.path.join(ROOT, "tests", "golden", "goddard_flow.json")))
start = [g for g in G["goddard_single_stage"] if g["stage"] == 2 and g["xtol"] == 1e-6][0]["init_z"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
out = {"rk4_steps": N}
with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
    f.write(" ".join(repr(v) for v in start))
exe = os.path.join(ROOT, "socp_amd", "_build", "bin", "goddard_flow")
for variant in ("fast", "exact"):
    r = subprocess.run([exe, "stage", "2", str(N), "1", "1e-8", f.name], capture_output=True, text=True,
                       env=dict(os.environ, SOCP_VARIANT=variant))
    s = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    out["gpu_" + variant] = {"seconds": s["seconds"], "info": s["info"], "nfev": s["nfev"], "trajectories": s["trajectories"]}
t = time.perf_counter()
c = goddard_single_stage(2, np.array(start), 1e-8, solver="socp", step_nbr=N)
out["cpu_oracle_1core"] = {"seconds": time.perf_counter() - t, "info": c["info"], "nfev": c["nfev"]}
print(json.dumps(out, indent=1))
