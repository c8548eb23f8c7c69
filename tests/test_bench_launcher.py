"""CPU: `bench.py --gpus N` without a launcher starts N ranks itself (torch.distributed.run child, started before
anything touches the GPU) and fails loudly when they do not all report.  Without a GPU the ranks stop with "bench.py
needs a GPU" -- the product has no CPU path -- so what is checked here is the launcher: N ranks were started, their failure
is the parent's failure, no JSON line is printed.  The N = 2 run itself is tests/test_gpu_bench_contract.py (GPU box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_launches_ranks_and_propagates_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by tests/test_gpu_bench_contract.py")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device0",
                          "--steps", "1", "--warmup", "0", "--starts", "64", "--rk4-steps", "10", "--cpu-seconds", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]          # no result line on failure
    assert "2-rank child job failed" in out.stderr
    assert out.stderr.count("bench.py needs a GPU") >= 1 or "needs a GPU" in out.stderr or "ChildFailedError" in out.stderr


def test_world_size_in_env_means_already_launched():
    """Under a launcher (WORLD_SIZE set) bench.py must not start another job: it goes straight to the rank code,
    which stops for want of a GPU here."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--cpu-seconds", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)
    assert "child job" not in out.stderr
