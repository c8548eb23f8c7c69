"""CPU: bench.py's in-run PMC measurement (live_traffic) against a stand-in for rocprofv3 -- what it asks the profiler for, how it reads
the counter file (the headline kernel: per kernel name, the launches over the whole workload, median; the matrix-core refresh: a chain of
launches under several names, every dispatch summed and divided by the refreshes the child ran; the guide's 2 x FETCH_SIZE KiB +
WRITE_SIZE KiB), and that a profiler that fails or hangs costs the bench line nothing but the live figure."""
import os
import stat
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAKE = r"""#!/usr/bin/env python3
import os, sys, time
a = sys.argv[1:]
counter, out = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
child = a[a.index("--") + 1:]
open(os.environ["FAKE_LOG"], "a").write(counter + " | " + " ".join(child) + "\n")
mode = os.environ.get("FAKE_MODE", "ok")
if mode == "hang":
    time.sleep(60)
if mode == "fail":
    sys.exit(3)
os.makedirs(os.path.join(out, "host", "1"), exist_ok=True)
rows = ["Kernel_Name,Grid_Size,Counter_Name,Counter_Value"]
val = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 300.0}[counter]
if mode == "empty":
    rows.append('"some_other_kernel",65536,%s,5' % counter)
else:
    # a warm-up launch (small grid), three launches of kernel A (median counts), two of kernel B, an unrelated kernel
    rows.append('"void socp::KERNEL<A>(int)",512,%s,7' % counter)
    for v in (val, val + 2, val - 50):
        rows.append('"void socp::KERNEL<A>(int)",65536,%s,%g' % (counter, v))
    for v in (val / 2, val / 2):
        rows.append('"void socp::KERNEL<B>(int)",65536,%s,%g' % (counter, v))
    rows.append('"copy_kernel",65536,%s,999999' % counter)
open(os.path.join(out, "host", "1", "123_counter_collection.csv"), "w").write("\n".join(rows) + "\n")
"""


@pytest.fixture
def fake_profiler(tmp_path, monkeypatch):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE.replace("#!/usr/bin/env python3", "#!" + sys.executable))
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR)
    log = tmp_path / "calls.log"
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("FAKE_LOG", str(log))
    return log


def _args():
    import bench
    return bench, bench.parse_args(["--starts", "4096", "--rk4-steps", "100"])


def test_live_traffic_reads_the_counters_as_the_guide_prescribes(fake_profiler, monkeypatch):
    bench, args = _args()
    monkeypatch.setenv("FAKE_MODE", "ok")
    got, why = bench.live_traffic(args, kernel="KERNEL<A>", child="headline", limit_s=30)
    # median of the whole-workload launches of the one kernel: FETCH 1000, WRITE 300 (KiB); reads doubled
    assert got == 2.0 * 1000.0 * 1024.0 + 300.0 * 1024.0 and "MEASURED IN THIS RUN" in why
    # two kernels under one name pattern: their medians summed
    got2, _ = bench.live_traffic(args, kernel="KERNEL", child="factor", limit_s=30)
    assert got2 == 2.0 * 1500.0 * 1024.0 + 450.0 * 1024.0
    # a refresh as a CHAIN of launches under several names (round 6): every matching dispatch -- whatever its grid -- summed and divided by
    # the number of refreshes the child ran.  FETCH: 7 + 1000 + 1002 + 950 + 500 + 500 = 3959 KiB, WRITE: 7 + 300 + 302 + 250 + 150 + 150 = 1159
    got3, why3 = bench.live_traffic(args, kernel=("KERNEL<A>", "KERNEL<B>"), child="factor", limit_s=30, refreshes=2)
    assert got3 == (2.0 * 3959.0 * 1024.0 + 1159.0 * 1024.0) / 2.0 and "MEASURED IN THIS RUN" in why3
    calls = fake_profiler.read_text().splitlines()
    assert [c.split(" | ")[0] for c in calls] == ["FETCH_SIZE", "WRITE_SIZE"] * 3          # a pass each, never combined
    # the program after `--` is the interpreter itself running this file in its child mode, at the headline size
    prog = calls[0].split(" | ")[1].split()
    assert prog[0] == sys.executable and prog[1].endswith("bench.py") and prog[2:4] == ["--traffic-child", "headline"]
    assert prog[prog.index("--starts") + 1] == "4096" and prog[prog.index("--rk4-steps") + 1] == "100"
    assert calls[2].split(" | ")[1].split()[2:4] == ["--traffic-child", "factor"]


@pytest.mark.parametrize("mode,needle", [("fail", "exit code 3"), ("empty", "no dispatch"), ("hang", "no result")])
def test_a_profiler_that_fails_or_hangs_costs_only_the_live_figure(fake_profiler, monkeypatch, mode, needle):
    bench, args = _args()
    monkeypatch.setenv("FAKE_MODE", mode)
    t = time.time()
    got, why = bench.live_traffic(args, kernel="KERNEL<A>", limit_s=2.0)
    assert got is None and needle in why
    assert time.time() - t < 20                                            # (the hung profiler is killed at the limit, not waited for)


def test_no_profiler_on_the_path_is_reported(monkeypatch, tmp_path):
    bench, args = _args()
    monkeypatch.setenv("PATH", str(tmp_path))
    real_exists = os.path.exists
    monkeypatch.setattr(bench.os.path, "exists", lambda p: False if "rocprofv3" in p else real_exists(p))
    got, why = bench.live_traffic(args)
    assert got is None and "not found" in why
