"""CPU (compile-only, no GPU): the ISA store audit finds round 4's mis-compiled store and nothing in today's plugin kernels.

VERDICT r4 #1.  Commit 08f2e7a wrote the interior-node rows of segment_residual as a three-way divergent branch with a pair of
stores in every arm; hipcc 7.2's backend sank one store to the join and left its ADDRESS register undefined on the all-CONTINUOUS
path (profiles/r05_fault_08f2e7a_isa.txt) -- a GPU fault for a continuous iterate, a silent stray write for a discontinuous one.
scripts/isa_store_audit.py compiles a translation unit to ISA and walks every kernel's control-flow graph for that shape.  Here:
the example plugin against the headers of 08f2e7a (through `git archive`: nothing of the working tree is touched) gives exactly the
faulting store, in every residual_lane_kernel instantiation; the plugin against today's headers gives none.  (The whole tree --
four minutes of compilation -- is audited in profiles/r05_isa_store_audit.txt: no candidate in any translation unit.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "isa_store_audit.py")

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles without a GPU)")


def test_todays_plugin_kernels_have_no_store_through_a_possibly_undefined_address():
    r = subprocess.run([sys.executable, SCRIPT, "--only", "plugin", "--expect-clean"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "total candidates: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_the_audit_finds_the_store_that_faulted_in_round_4():
    have = subprocess.run(["git", "-C", ROOT, "cat-file", "-e", "08f2e7a^{commit}"], capture_output=True)
    if have.returncode != 0:
        pytest.skip("commit 08f2e7a is not in this checkout's history")
    r = subprocess.run([sys.executable, SCRIPT, "--rev", "08f2e7a", "--only", "plugin", "--expect-clean"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1, r.stdout[-2000:] + r.stderr[-2000:]
    hits = [l for l in r.stdout.splitlines() if l.strip().startswith("CANDIDATE")]
    stores = [l for l in r.stdout.splitlines() if "may be undefined" in l]
    assert len(hits) == 8 and all("residual_lane_kernelI5Lqr1D" in l for l in hits)
    assert all("flat_store_dwordx2" in l and "offset:16" in l for l in stores)      # the sunk `emit(row + D, .)` of component 1
