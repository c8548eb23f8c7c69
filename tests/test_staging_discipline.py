"""CPU: the pinned-buffer rule of the lock-step device engine (socp_amd/csrc/staging.hpp).

Round 4 had a real host/device ordering bug in batchsolve_dev.cpp (commit e52cc58): the advance loop refilled a shared pinned list
before the asynchronous copy of its previous contents had run, and continuation chains restarted from the wrong list.  It was
found by a test and fixed by review; since round 5 every pinned staging buffer of the engine is a `Staged`: its memory can only be
reached by saying what is about to happen -- the host touches it, or an asynchronous operation on a stream is enqueued with it -- and
a host access while the last operation's stream has not been synchronised is caught: repaired by a synchronise in production
(and counted: the engine prints a warning), abort() in strict mode (the whole test suite runs strict, tests/conftest.py).
tests/cpp/staging_discipline.cpp drives the same header with a recording fake backend, so the rule itself is tested here without a
GPU -- including the e52cc58 sequence exactly as it was written."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("staging") / "staging_discipline")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "staging_discipline.cpp"), "-o", out])
    return out


def _env(strict):
    e = dict(os.environ)
    e["SOCP_STAGING_STRICT"] = "1" if strict else "0"
    return e


@pytest.mark.parametrize("strict", [False, True])
def test_the_engines_own_sequences_need_no_forced_synchronise(exe, strict):
    r = subprocess.run([exe, "ok"], capture_output=True, text=True, env=_env(strict))
    assert r.returncode == 0 and "none forced" in r.stdout, r.stdout + r.stderr


def test_the_round4_ordering_bug_is_repaired_in_production_and_fatal_in_strict_mode(exe):
    r = subprocess.run([exe, "e52cc58"], capture_output=True, text=True, env=_env(False))
    assert r.returncode == 0 and "forced_syncs = 1" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, "e52cc58"], capture_output=True, text=True, env=_env(True))
    assert r.returncode != 0 and "hList" in r.stderr and "must synchronise" in r.stderr, r.stdout + r.stderr


def test_every_pinned_buffer_of_the_engine_is_a_staged_one():
    """No raw pointer to pinned memory is left in the engine: the `Pinned` type has no public `.p` / `.d()` / `.i()` any more, every
    stream synchronise inside the rounds goes through a StreamClock, and hipMemcpyAsync is only ever called with a staged source or
    target (or device-to-device)."""
    src = open(os.path.join(ROOT, "socp_amd", "csrc", "batchsolve_dev.cpp")).read()
    body = src[src.index("int socp_chains_solve_device("):src.index("extern \"C\" int socp_qr_factor_batch") if "extern \"C\" int socp_qr_factor_batch" in src else len(src)]
    loop = body[body.index("while (rc == SOCP_OK) {"):body.index("const clk::time_point t_loop_end")]
    assert "hipStreamSynchronize" not in loop
    import re
    for m in re.finditer(r"hipMemcpyAsync\(([^;]*);", loop):
        args = m.group(1)
        assert ".source(clk_" in args or ".target(clk_" in args, args
    assert not re.search(r"\bh(Status|List|ListS|Flags|ListF|ListJ|X|Res|PF|TF|XF|PJ|TJ|XJ|IdxA|IdxB)\.(p\b|d\(\)|i\(\))", src)
