"""GPU: bench.py prints exactly one JSON line with the fields the driver reads, and the numbers hang together."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--starts", "4096",
                          "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    per_step = d["config"]["trajectories_per_step_per_gpu"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1.2
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.05                 # the kernel is (nearly all of) the step
    assert r["hbm"]["bound"] == "hbm" and r["hbm"]["peak"] == 8000.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
