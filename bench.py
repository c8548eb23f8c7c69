#!/usr/bin/env python3
"""bench.py -- throughput of the SOCP hot path on MI355X: trajectories integrated per second.

Workload (BASELINE.json configs[1], on synthetic random-init costate batches as north_star asks):
every rank holds `--starts` independent Goddard single-shooting problems (n = 14 unknowns, fixed
tf, KD = 310, mu2 = 1; initial costates p = p*(1 + 1e-3 xi)).  ONE STEP = the
forward-difference-Jacobian batch of every start: base residual + 14 perturbed residuals =
15 trajectories per start, each 10 000 RK4 steps of the 14-dim state+costate system, in one launch
(socp_fd_rows_dev), followed by the small difference kernel that forms the Jacobians.  Inputs are
resident in HBM before the timed region.  value = trajectories of ALL ranks / max-over-ranks time.

Launch forms (both end in one process per GPU over RCCL, no data-path collective: "replicas of
independent problems", the only exchange is the gather of per-rank result records after the timed region):
  python bench.py --gpus N ...                                  N > 1 without WORLD_SIZE: this process starts
                                                                `python -m torch.distributed.run --nproc-per-node N
                                                                bench.py ...` as a child BEFORE touching the GPU,
                                                                relays its JSON line and fails if fewer than N ranks report
  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...     the driver's form

Objects in the JSON line beside the contract's fields (N = 1; `--lean` drops the extra timed legs):
  roofline        dominant kernel (fdrows_lane_kernel).  This path has no dense contraction and moves 224 B per
                  trajectory, so neither "mfma" nor "hbm" bounds it: the binding resource is FP64 vector issue.
                  `bound` is therefore "valu_fp64" (peak = 256 CU x 4 SIMD x 16 FP64 lanes x 2 flop x 2.4 GHz =
                  78.6 TFLOP/s, half the FP32 vector peak of MI355X_MICROARCH.md); the HBM view the contract asks for
                  is in roofline.hbm.  `traffic` IS measured in this run (N = 1, not --lean): this file runs twice more as a
                  child process under `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE; the guide's recipe and corrections) around
                  two launches of the same kernel at the same size (live_traffic); `traffic_recorded` keeps the figure committed
                  under profiles/ beside it, and is what `traffic` falls back to if the profiler cannot run
                  (`traffic_measured_in_this_run`, `traffic_live_measurement` say which).
  exact           the same workload on the bit-identical (reference operation order) flavour -- what a drop-in user gets
                  by default (SOCP_VARIANT_AUTO) -- from a second timed region: value, kernel_ms, roofline frac.
  parity          after the timed regions: FD-batch rows of the first starts recomputed by the CPU oracle (checker, never
                  timed here) and compared with what the two flavours left in HBM: max relative error, bitwise flag.
  single_problem  ONE problem (15 trajectories) per launch: the latency-bound case, stated not hidden.
  north_star_128  north_star's target size: Goddard, M = 9 segments, n = 128 unknowns: one FD Jacobian (1152 trajectories
                  of 1e4 steps as the reference integrates them; fewer with the segment dedup), ms, the reference's
                  trajectory count over that time (reference_trajectories_per_s: a time-to-same-Jacobian rate) and its
                  ratio to cpu_baseline (B1) -- north_star asks for >= 10x -- and to 16 perfectly scaling cores at the measured
                  one-core rate (x_over_16xP1: the ratio that does not swing with the host's other tenants).
  sweep, sweep_large, sweep_xl   STRONG scaling: fixed totals of 65 536 / 524 288 / 4 194 304 single-shooting starts (BASELINE
                  configs[3] class: full Newton solves in lock-step, 1e4 RK4 steps, 40-round budget) sharded over the ranks;
                  wall time barrier to barrier, max over ranks.  A sweep's wall time is rounds x max(one trajectory latency,
                  the block's trajectories / the kernel rate): only a block of >= ~0.5 M starts is throughput-bound, so only
                  sweep_xl (524 288 per rank at N = 8) CAN scale near-linearly to 8 GPUs; the two smaller legs flatten at the
                  latency floor.  At N = 1 every leg carries `predicted` (wall / speedup / efficiency at 2, 4, 8 GPUs read off
                  this run's own one-GPU curve, `sweep_curve_one_gpu`); at N > 1 `expected` (from the curve recorded under
                  profiles/) and measured_over_expected.
  sweep_config5   STRONG scaling of BASELINE configs[4]: 16 384 starts of the 253-unknown interceptor problem (21 segments, adaptive
                  Dormand-Prince, tol 1e-8; Newton solvers on the device with the matrix-core Jacobian refresh) sharded the same
                  way -- 2048 per rank at N = 8.  PARITY UNPINNED (Boost and Eigen are absent: integrator and model are restated
                  from the published algorithm / the reference's text).  `predicted` / `expected` from its own one-GPU curve.
  solver_kernels  the sweeps' second kernel with a roofline of its own: factor_fast = 2048 Jacobian refreshes of n = 253 (the factor
                  launch of a config-5 round) on the FP64 matrix cores, HIP-event time, fraction of the FP64 peak; traffic = the
                  counters of this run as above (a refresh is a chain of launches -- qrfac's panel and trailing launches per pair of panels,
                  then qform: every dispatch of the child's two refreshes summed, halved), the recorded figure
                  (profiles/r06_factor_pmc.json) beside it.
  cpu_baseline    B1: the reference's own model::ComputeTraj (oracle/_ref, kind "reference") or the C oracle (kind
                  "port") on this box's host cores, bounded sample: `value` = MEDIAN of five samples, `spread`, `best`,
                  `worst`, `samples` beside it (all cores and one core).  cpu_baseline.b0 = "as shipped": the reference's
                  shooting.cpp + its per-call std::threads, bound to this library's hybrd (oracle/_ref/link), one
                  continuation solve at 1e4 steps per segment, trajectories/s = nfev x M / wall.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_RK4_STEP = 1170.0    # ~250 FP64 ops / RHS x 4 + 168 (RK4 combine); x 1e4 steps = 1.17e7 per trajectory
N_UNKNOWN = 14
ROWS = N_UNKNOWN + 1          # residual rows (= trajectories, M = 1) per start and step
BYTES_PER_TRAJ = 16 * N_UNKNOWN   # read z (8n) + write F (8n) per residual evaluation
PEAK_FP64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
X0_STATE = np.array([0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0])
PSTAR = np.array([-8.121947733, 7.775439382e-3, 0.7775438809, -0.4779369965, 5.715013318e-4,
                  5.715009222e-2, 9.958404873e-2])
TF = 0.2640825
GODDARD_PARAMS = [3.5, 7.0, 310.0, 500.0, 1.0, 1.0, 1.0, -1.0]
EPSFCN = 1e-15


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--starts", type=int, default=13107,
                    help="independent shooting problems per GPU; default: 15 x 13107 = 196 605 trajectories = 3072 "
                         "wavefronts = 3 per SIMD (an exactly full chip); any size >= 4096 runs within 10 %% of it")
    ap.add_argument("--rk4-steps", type=int, default=10000)
    ap.add_argument("--variant", choices=["exact", "fast"], default="fast",
                    help="flavour of the headline `value`.  fast: restructured arithmetic (<= 1e-8 vs the reference order "
                         "after 1e4 steps, converged solutions within 1e-8: tests/test_gpu_parity.py, test_host_flow.py); "
                         "exact: reference operation order, bit-identical.  The other flavour is reported beside it.")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the CPU baseline legs")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + --share-device0: rehearse the multi-rank path on a one-GPU box (collectives on CPU tensors)")
    ap.add_argument("--share-device0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal only: initialise torch.distributed and run the barrier / all_reduce / all_gather of the N > 1 path "
                         "even with ONE rank -- exercises the RCCL (backend nccl) code path on a one-GPU box")
    ap.add_argument("--lean", action="store_true",
                    help="N = 1: only the headline timed region, roofline and cpu_baseline (no exact / parity / "
                         "single_problem / north_star_128 legs)")
    ap.add_argument("--traffic-child", nargs="?", const="headline", default=None, choices=["headline", "factor"],
                    help="internal (live_traffic): two launches of the headline kernel at the headline size and nothing else -- the program "
                         "rocprofv3 --pmc runs")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched N > 1 job (0: pick a free one)")
    ap.add_argument("--sweep-starts", type=int, default=None,
                    help="the `sweep` leg: a multi-start sweep of this many single-shooting starts IN TOTAL, sharded over the ranks "
                         "(strong scaling: what north_star's \"near-linear to 8 GPUs on the multi-start sweep\" is about); default 65536 "
                         "(0 = skip; --lean skips it unless a count is given)")
    ap.add_argument("--sweep-max-rounds", type=int, default=40, help="socp_chain_options.max_rounds of the sweep leg")
    ap.add_argument("--sweep-large-starts", type=int, default=None,
                    help="the `sweep_large` leg: the same sweep with this many starts in total (default 8 x --sweep-starts): enough "
                         "that one GPU is throughput-bound, so the strong-scaling curve is not flattened by the per-round trajectory "
                         "latency the 65 536-start leg runs into; 0 skips it")
    ap.add_argument("--sweep-xl-starts", type=int, default=None,
                    help="the `sweep_xl` leg: the same sweep with this many starts in total (default 64 x --sweep-starts = 4 194 304): "
                         "a rank's share at N = 8 is still 524 288 starts, i.e. throughput-bound -- the leg on which near-linear strong "
                         "scaling to 8 GPUs can show at all; 0 skips it")
    ap.add_argument("--sweep-c5-starts", type=int, default=None,
                    help="the `sweep_config5` leg: this many starts IN TOTAL of the 253-unknown interceptor problem under the adaptive "
                         "integrator (BASELINE configs[4]; parity unpinned), sharded over the ranks; default 16384 (0 = skip; --lean skips it)")
    args = ap.parse_args(argv)
    if args.sweep_c5_starts is None:
        args.sweep_c5_starts = 0 if args.lean else 16384
    if args.sweep_starts is None:
        args.sweep_starts = 0 if args.lean else 65536
    if args.sweep_large_starts is None:
        args.sweep_large_starts = 0 if args.lean else 8 * args.sweep_starts
    if args.sweep_xl_starts is None:
        args.sweep_xl_starts = 0 if args.lean else 64 * args.sweep_starts
    return args


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start one process per GPU before anything here touches the GPU
# ------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    import socket
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in child.stdout.splitlines():
        if l.startswith("{"):
            try:
                line = json.loads(l)
            except ValueError:
                pass
    if child.returncode != 0 or line is None:
        sys.stderr.write("bench.py: the %d-rank child job failed (exit %d)\n%s\n" % (args.gpus, child.returncode, child.stdout[-2000:]))
        return child.returncode or 1
    if line.get("n_gpus") != args.gpus or line.get("ranks_reported") != args.gpus:
        sys.stderr.write("bench.py: %d ranks were asked for, %s reported\n" % (args.gpus, line.get("ranks_reported")))
        return 1
    print(json.dumps(line), flush=True)
    return 0


def make_starts(P, seed):
    """Costates p*(1 + 1e-3 xi), xi from raw std::mt19937_64(seed) draws exactly as SURVEY 8d prescribes."""
    from socp_amd.sweep import goddard_starts
    return goddard_starts(P, 1e-3, seed)


def setup_context(device, steps_rk4, variant):
    from socp_amd import capi
    ctx = capi.Context(capi.MODEL_GODDARD, device=device)
    ctx.set_params(GODDARD_PARAMS)
    ctx.set_step_number(steps_rk4)
    ctx.set_variant({"exact": capi.VARIANT_LANE_EXACT, "fast": capi.VARIANT_LANE_FAST}[variant])
    mode_t = [capi.FIXED, capi.FIXED]
    mode_x = np.zeros((2, 7), dtype=np.int32)
    mode_x[1, 3:7] = capi.FREE
    X = np.zeros((2, 14))
    X[0, :7] = X0_STATE
    X[1, 0] = 1.01
    n = ctx.problem_set(mode_t, mode_x, np.array([0.0, TF]), X)
    assert n == N_UNKNOWN
    return ctx


def fd_rows_inputs(Z, count):
    """The first `count` trajectories of the step's batch as initial states: row 0 of a start is z, row j + 1 is
    z + h_j e_j with MINPACK's h."""
    eps = np.sqrt(EPSFCN)
    X0 = np.empty((count, 14))
    for k in range(count):
        p, row = divmod(k, ROWS)
        X0[k] = Z[p % len(Z)]
        if row > 0:
            j = row - 1
            h = eps * abs(X0[k, j]) or eps
            X0[k, j] += h
    return X0


def host_cpus():
    """(CPU model string, one logical CPU per PHYSICAL core among the CPUs this process may run on).  SURVEY 8d asks for the
    baseline on all physical cores, pinned: hyper-thread siblings are dropped, and a box that grants a share of its host
    (the GPU boxes show all 256 logical CPUs of the host in the affinity mask and grant 16 of them through the cgroup CPU quota)
    is taken as granted: the thread count is the quota.  Returns (model, cpus to pin to, quota or None)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    # a container's CPU share: the scheduler quota, not the affinity mask (the GPU boxes show 256 logical CPUs and grant 16)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                      # cgroup v2
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                    # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    seen, cores = set(), []
    for cpu in allowed:
        key = cpu
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu).read().strip()
            key = sib
        except OSError:
            pass
        if key not in seen:
            seen.add(key)
            cores.append(cpu)
    if quota is not None and len(cores) > int(quota):
        # Which of the host's cores to pin to: a GPU box is one tenant of a shared host, and every tenant that pinned to "the first
        # 16" would share them (seen: 1111 against 1853 trajectories/s on otherwise identical boxes).  Take the physical cores that
        # were least busy over the last 0.2 s.
        def busy():
            out = {}
            for line in open("/proc/stat"):
                if line.startswith("cpu") and line[3].isdigit():
                    f = line.split()
                    v = [int(x) for x in f[1:]]
                    out[int(f[0][3:])] = (sum(v), v[3] + (v[4] if len(v) > 4 else 0))       # total, idle + iowait
            return out
        try:
            a = busy()
            time.sleep(0.2)
            b = busy()
            load = {c: 1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in cores if c in a and c in b}
            cores = sorted(cores, key=lambda c: (round(load.get(c, 1.0), 2), c))
        except (OSError, ValueError, IndexError):
            pass
        cores = sorted(cores[:max(1, int(quota))])
    return model, cores, quota


def cpu_baseline(steps_rk4, Z, target_seconds):
    """B1: reference (or port) on the host cores, bounded sample of the SAME trajectories (rows of the FD batch of the first
    starts).  As SURVEY 8d prescribes: P = all physical cores of this process's CPU share AND P = 1, threads pinned one per
    core, the MEDIAN of 5 shorter samples with their spread (a box's first sample is regularly 20-40 % low: frequency ramp, cold caches;
    the all-core figure swings with the host's other tenants), CPU named."""
    from oracle import oracle as orc
    model, cores, quota = host_cpus()

    if orc.have_ref():
        ref = orc.Ref(orc.MODEL_GODDARD, step_nbr=steps_rk4)

        def five_samples(threads, seconds):
            cpus = cores[:threads]
            probe = fd_rows_inputs(Z, threads)
            _, sec = ref.goddard_traj_batch(threads, steps_rk4, GODDARD_PARAMS, 0.0, TF, probe, cpus=cpus)
            count = int(max(threads, min(65536, threads * round(seconds / 5 / max(sec, 1e-3)))))
            X0 = fd_rows_inputs(Z, count)
            runs = []
            for _ in range(5):
                _, s = ref.goddard_traj_batch(threads, steps_rk4, GODDARD_PARAMS, 0.0, TF, X0, cpus=cpus)
                runs.append(count / s)
            return count, runs

        def summary(runs):
            med = float(np.median(runs))
            return {"value": med, "median": med, "best": max(runs), "worst": min(runs), "spread": (max(runs) - min(runs)) / med, "samples": runs}
        P = len(cores)
        count, runs = five_samples(P, 0.7 * target_seconds)
        count1, runs1 = five_samples(1, 0.3 * target_seconds)
        all_cores, one_core = summary(runs), summary(runs1)
        # `value` is the MEDIAN of five samples (round 4 quoted the best of three: on a box that is one tenant of a shared host the
        # all-core figure swings 2 x between samples, and with it every ratio built on it); the spread is in the record
        return {**all_cores, "unit": "trajectories/s", "cores": P, "kind": "reference", "per_core": all_cores["value"] / P,
                "pinned": True, "cpu_model": model, "cpu_quota": quota, "logical_cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                "p1": {**one_core, "cores": 1, "trajectories_per_sample": count1},
                "cores_x_p1": P * one_core["value"],
                "parallel_efficiency": all_cores["value"] / (P * one_core["value"]),
                "sample": "B1: median of 5 x %d trajectories (FD-batch rows of the first %d starts), %d RK4 steps each, reference "
                          "model::ComputeTraj, one goddard object per std::thread, %d threads pinned one per physical core (%s); "
                          "P = 1 beside it" % (count, (count + ROWS - 1) // ROWS, steps_rk4, P, model)}
    o = orc.Oracle(orc.MODEL_GODDARD, step_nbr=steps_rk4, params=GODDARD_PARAMS)
    probe = fd_rows_inputs(Z, 2)
    t = time.perf_counter()
    o.integrate_batch(0.0, TF, probe)
    per = (time.perf_counter() - t) / 2
    count = int(max(2, min(4096, round(target_seconds / max(per, 1e-4)))))
    X0 = fd_rows_inputs(Z, count)
    t = time.perf_counter()
    o.integrate_batch(0.0, TF, X0)
    sec = time.perf_counter() - t
    return {"value": count / sec, "median": count / sec, "best": count / sec, "worst": count / sec, "spread": 0.0, "samples": [count / sec],
            "unit": "trajectories/s", "cores": 1, "kind": "port", "pinned": False, "cpu_model": model,
            "p1": {"value": count / sec, "median": count / sec, "spread": 0.0, "samples": [count / sec], "cores": 1}, "cores_x_p1": count / sec,
            "sample": "%d trajectories, %d RK4 steps each, C oracle single thread, %.1f s" % (count, steps_rk4, sec)}


def cpu_baseline_b0(steps_rk4, target_seconds):
    """B0 "as shipped" (BASELINE.md 3): the reference's shooting.cpp (residual callbacks, a std::thread per segment block
    created and joined on every call, shooting.cpp:1142-1157) bound to this library's hybrd at link level, numThread =
    min(M, cores) = 6: the KD 0 -> 310 continuation solve of testGoddard (M = 6, n = 85) from the test's own state before
    that call, with `steps_rk4` RK4 steps per segment.  trajectories/s = nfev x M / wall."""
    exe = os.path.join(ROOT, "oracle", "_ref", "link", "goddard_flow_ref")
    gold = os.path.join(ROOT, "tests", "golden", "goddard_flow.json")
    if not (os.path.exists(exe) and os.path.exists(gold)):
        return None
    import tempfile
    M = 6
    threads = min(M, os.cpu_count() or 1)
    init = [g for g in json.load(open(gold))["goddard_single_stage"] if g["stage"] == 2 and g["xtol"] == 1e-6][0]["init_z"]
    # bound the sample: one evaluation at 1e4 steps is ~6 x 8 ms / threads; a continuation solve is ~190-400 evaluations
    steps = steps_rk4 if target_seconds >= 4 else max(10, steps_rk4 // 10)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(" ".join(repr(v) for v in init))
        zfile = f.name
    try:
        t = time.perf_counter()
        out = subprocess.run([exe, "stage", "2", str(steps), "1", "1e-6", zfile], capture_output=True, text=True,
                             timeout=600, env=dict(os.environ, SOCP_FLOW_THREADS=str(threads)))
        wall = time.perf_counter() - t
    finally:
        os.unlink(zfile)
    recs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    if not recs:
        return None
    r = recs[0]
    traj = r["nfev"] * M
    return {"value": traj / r["seconds"] * (steps / steps_rk4), "unit": "trajectories/s", "cores": threads, "kind": "reference",
            "info": r["info"], "nfev": r["nfev"], "segments": M, "rk4_steps_sampled": steps,
            "sample": "B0: reference shooting.cpp + per-call std::threads (numThread = %d) + this library's hybrd; testGoddard's "
                      "KD 0 -> 310 continuation solve, n = 85, %d evaluations x %d segments of %d RK4 steps in %.2f s (process %.2f s)%s"
                      % (threads, r["nfev"], M, steps, r["seconds"], wall,
                         "" if steps == steps_rk4 else "; rate scaled to %d steps" % steps_rk4)}


def timed_region(torch, dist, ctx, stream, dev, cdev, world, P, d_Z, d_rows, d_J, steps, warmup, collective=None):
    """W untimed steps, then exactly K steps between fences (sync + barrier + sync); returns (max-over-ranks seconds,
    mean HIP-event ms of the dominant kernel on the stream it is launched on)."""
    def step(events=None):
        if events is not None:
            events[0].record(stream)
        ctx.fd_rows_dev(P, d_Z.data_ptr(), EPSFCN, d_rows.data_ptr())
        if events is not None:
            events[1].record(stream)
        ctx.fd_diff_dev(P, d_Z.data_ptr(), EPSFCN, d_rows.data_ptr(), d_J.data_ptr())

    collective = (world > 1) if collective is None else collective

    def fence():
        torch.cuda.synchronize(dev)          # this rank's own launches are done ...
        if collective:
            dist.barrier()                   # ... and so are everybody else's
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step()
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        step(evs[k])
    fence()
    elapsed = time.perf_counter() - t0
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if collective:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    return float(t_max.item()), float(np.mean([a.elapsed_time(b) for a, b in evs]))


def roofline_of(traj_per_launch, kernel_ms, rk4_steps):
    """Algorithmic rates of one launch: flops scale with the step count of a trajectory (1170 per RK4 step), the bytes do not
    (a trajectory reads z and writes F whatever its length)."""
    tflops = FLOP_PER_RK4_STEP * rk4_steps * traj_per_launch / (kernel_ms * 1e-3) / 1e12
    gbs = BYTES_PER_TRAJ * traj_per_launch / (kernel_ms * 1e-3) / 1e9
    return tflops, gbs


def recorded_traffic(P, variant, rk4_steps):
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                entries = json.load(f)
            for tj in (entries if isinstance(entries, list) else [entries]):
                if tj.get("starts") == P and tj.get("variant") == variant and tj.get("rk4_steps") == rk4_steps:
                    return tj.get("hbm_bytes_per_launch"), "profiles/traffic_latest.json <- " + str(tj.get("source", "rocprofv3 --pmc passes"))
        except Exception:
            pass
    return None, None


def traffic_child(args):
    """What `rocprofv3 --pmc <counter> -- python3 bench.py --traffic-child` runs: the headline kernel, same flavour, same size, twice."""
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if args.traffic_child == "factor":
        from socp_amd import capi
        J, b = factor_workload()
        capi.qr_factor_batch(J, b, flavour=capi.FACTOR_FAST, reps=2, outputs=False, device=0)
        return 0
    ctx = setup_context(0, args.rk4_steps, args.variant)
    P = args.starts
    d_Z = torch.from_numpy(make_starts(P, seed=20250905)).to(dev)
    d_rows = torch.empty((P, ROWS, N_UNKNOWN), dtype=torch.float64, device=dev)
    for _ in range(2):
        ctx.fd_rows_dev(P, d_Z.data_ptr(), EPSFCN, d_rows.data_ptr())
    torch.cuda.synchronize(dev)
    return 0


def live_traffic(args, kernel="fdrows_lane_kernel", child="headline", limit_s=90.0, refreshes=None):
    """HBM bytes of one launch of the headline kernel from the PMC counters, measured NOW on this box: this file run twice as a child
    process under `rocprofv3 --pmc` -- FETCH_SIZE, then WRITE_SIZE, a pass each, no trace domain beside them (the guide's recipe:
    /opt/skills/guides/MI355X_MICROARCH.md, HBM section; FETCH_SIZE / WRITE_SIZE in KiB, reads doubled on gfx950).  The children are
    ordinary child processes with a time limit (killed by the process group THIS call started); any failure returns (None, why) and
    the line keeps the recorded figure.  Not under a profiler (the profiling scripts run bench.py --lean) and at N = 1 only.
    `kernel`: a name fragment, or a tuple of them.  `refreshes`: None -- per kernel name the median over its launches of the largest grid
    (the headline kernel: two equal launches); an integer -- the child ran that many whole refreshes and a refresh is a CHAIN of launches
    (round 6: panel and trailing launches per pair of panels, then qform): every matching dispatch summed, divided by it."""
    import csv
    import glob
    import shutil
    import signal
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="socp_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--traffic-child", child,
               "--starts", str(args.starts), "--rk4-steps", str(args.rk4_steps), "--variant", args.variant]
        try:
            proc = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d, start_new_session=True,
                                    env=dict(os.environ, TMPDIR="/tmp"))
            try:
                rc = proc.wait(timeout=limit_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                return None, "rocprofv3 --pmc %s: no result in %.0f s" % (counter, limit_s)
            if rc != 0:
                return None, "rocprofv3 --pmc %s: exit code %d" % (counter, rc)
            rows = {}                                                        # per kernel NAME (the refresh is two kernels: their sum)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if any(k in r.get("Kernel_Name", "") for k in ((kernel,) if isinstance(kernel, str) else kernel)) and r.get("Counter_Name") == counter:
                            rows.setdefault(r["Kernel_Name"], []).append((float(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
            if not rows:
                return None, "rocprofv3 --pmc %s: no dispatch of %s in its output" % (counter, kernel)
            if refreshes:
                vals[counter] = sum(c for v in rows.values() for _, c in v) / float(refreshes)
            else:
                top = max(g for v in rows.values() for g, _ in v)            # (the launches over the whole workload, not a warm-up's)
                total = 0.0
                for v in rows.values():
                    x = sorted(c for g, c in v if g == top)
                    if x:
                        total += x[len(x) // 2]
                vals[counter] = total
        except Exception as exc:                                             # (a measurement aid must never take the bench line down)
            return None, "rocprofv3 --pmc %s: %s" % (counter, exc)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"] * 1024.0 + vals["WRITE_SIZE"] * 1024.0, (
        "MEASURED IN THIS RUN: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, a pass each, around two launches of the kernel in a child process "
        "(bench.py --traffic-child); 2 x FETCH_SIZE KiB + WRITE_SIZE KiB, median of the launches")


def parity_leg(Z_host, rows_by_variant, rk4_steps, K):
    """FD-batch rows of the first K starts against the CPU oracle (the checker; never inside a timed region)."""
    from oracle import oracle as orc
    o = orc.Oracle(orc.MODEL_GODDARD, step_nbr=rk4_steps, params=GODDARD_PARAMS)
    mode_x = np.zeros((2, 7), dtype=np.int32)
    mode_x[1, 3:7] = orc.FREE
    X = np.zeros((2, 14))
    X[0, :7] = X0_STATE
    X[1, 0] = 1.01
    prob = orc.Problem(7, [orc.FIXED, orc.FIXED], mode_x, np.array([0.0, TF]), X)
    eps = np.sqrt(EPSFCN)
    Zp = np.repeat(Z_host[:K], ROWS, axis=0)
    for k in range(K):
        for j in range(N_UNKNOWN):
            h = eps * abs(Zp[k * ROWS + j + 1, j]) or eps
            Zp[k * ROWS + j + 1, j] += h
    t = time.perf_counter()
    want = o.residual_batch(prob, Zp).reshape(K, ROWS, N_UNKNOWN)
    sec = time.perf_counter() - t
    out = {"rows_checked": K * ROWS, "starts_checked": K, "oracle_seconds": sec,
           "metric": "max |row_gpu - row_cpu| / max(1, |row_cpu|_inf) over the FD-batch residual rows (1e4 RK4 steps each)",
           "tolerance": {"exact": 0.0, "fast": 1e-8}}
    for name, d_rows in rows_by_variant.items():
        got = d_rows[:K].cpu().numpy()
        scale = np.maximum(1.0, np.max(np.abs(want), axis=2, keepdims=True))
        err = float(np.max(np.abs(got - want) / scale))
        out[name] = {"max_rel_err": err, "bitwise_equal": bool(np.array_equal(got, want))}
    out["pass"] = bool(out.get("fast", {"max_rel_err": 0})["max_rel_err"] <= 1e-8 and
                       out.get("exact", {"max_rel_err": 0})["max_rel_err"] <= 1e-10)
    return out


def north_star_128(capi, device, rk4_steps, cpu_traj_per_s, p1_traj_per_s=None):
    """Goddard, M = 9, FREE tf + one FREE interior time -> n = 128: one FD Jacobian at a fixed z through the
    host-pointer entry point (PCIe and launch included), nodes along the p* trajectory integrated on the device."""
    from socp_amd import sweep
    out = {"unknowns": 128, "segments": 9, "rk4_steps": rk4_steps,
           "trajectories_as_reference": 128 * 9, "note": "host-pointer socp_fd_jacobian, wall incl. PCIe + launch; "
           "trajectories_as_reference = n x M integrations the reference's fdjac1 performs per Jacobian"}
    for tag, variant in (("fast", capi.VARIANT_LANE_FAST), ("exact", capi.VARIANT_LANE_EXACT)):
        ctx = capi.Context(capi.MODEL_GODDARD, device=device)
        ctx.set_params(GODDARD_PARAMS)
        ctx.set_step_number(rk4_steps)
        ctx.set_variant(variant)
        n, z, _mt, _mx, _tn, _X = sweep.goddard_north_star_128_problem(ctx)
        assert n == 128 and len(z) == n
        F0 = ctx.residual(z)
        res = {}
        for dd, key in ((False, "full"), (True, "dedup")):
            ctx.fd_jacobian(z, F0, dedup=dd)
            c0 = ctx.counters()[0]
            reps = 3
            t = time.perf_counter()
            for _ in range(reps):
                J = ctx.fd_jacobian(z, F0, dedup=dd)
            sec = (time.perf_counter() - t) / reps
            res[key] = {"ms": 1e3 * sec, "trajectories_integrated": int((ctx.counters()[0] - c0) // reps), "finite": bool(np.isfinite(J).all())}
        best = min(res["full"]["ms"], res["dedup"]["ms"])
        res["jacobian_ms"] = best
        # time-to-same-Jacobian rate: the REFERENCE's count of integrations per Jacobian (n x M = 1152) over this path's time for
        # that Jacobian -- not this path's throughput (with the dedup it integrates 252 of them)
        res["reference_trajectories_per_s"] = 128 * 9 / (best * 1e-3)
        res["integrated_trajectories_per_s"] = res["dedup" if res["dedup"]["ms"] <= res["full"]["ms"] else "full"]["trajectories_integrated"] / (best * 1e-3)
        if cpu_traj_per_s:
            res["x_over_cpu_baseline"] = res["reference_trajectories_per_s"] / cpu_traj_per_s
        if p1_traj_per_s:
            # the ratio that survives a throttled host: against 16 perfectly scaling cores at the measured ONE-core rate (the all-core
            # figure of a shared host swung 2 x between boxes in round 4: 96.8 x in one line, 54.8 x in another, same GPU time)
            res["x_over_16xP1"] = res["reference_trajectories_per_s"] / (16.0 * p1_traj_per_s)
        out[tag] = res
        ctx.close()
    return out


def factor_workload(n=253, count=2048):
    """2048 well-conditioned random Jacobians of n = 253 and their right-hand sides: the factor launch of BASELINE config 5"""
    rng = np.random.default_rng(1)
    J = rng.standard_normal((count, n, n))
    J[:, np.arange(n), np.arange(n)] += 0.5 * np.sqrt(n)
    b = rng.standard_normal((count, n))
    return J, b


FACTOR_KERNELS = ("qrfac_panel_kernel", "qrfac_trail_kernel", "factor_fast_kernel")     # the launches of one matrix-core refresh


def solver_kernel_rooflines(capi, device, args=None, live=False):
    """The roofline of the sweeps' second-largest kernel beside the headline one: the Jacobian refresh of the device solvers in the
    throughput flavour (kernels_factor_fast.hip, blocked Householder QR on the FP64 matrix cores), 2048 problems of n = 253 -- the
    factor launch of BASELINE config 5 -- timed with HIP events inside socp_qr_factor_batch.  flop = 8/3 n^3 per problem (qrfac + qform),
    algorithmic bytes = J in, Q and R out; traffic = the PMC counters of this run (live_traffic, child mode `factor`) or, failing that, the figure
    recorded under profiles/ (separate --pmc passes)."""
    n, count = 253, 2048
    J, b = factor_workload()
    capi.qr_factor_batch(J[:8], b[:8], flavour=capi.FACTOR_FAST, outputs=False, device=device)
    capi.qr_factor_batch(J[:640], b[:640], flavour=capi.FACTOR_FAST, outputs=False, device=device)      # (the chain's kernels: from 640 problems up)
    ms = capi.qr_factor_batch(J, b, flavour=capi.FACTOR_FAST, reps=3, outputs=False, device=device)["kernel_ms"]
    flop = 8.0 / 3.0 * n ** 3 * count
    alg = 8.0 * count * (2 * n * n + n * (n + 1) / 2)
    traffic, source = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r06_factor_pmc.json")) as f:
            rec = json.load(f)
        if rec.get("n") == n and rec.get("count") == count:
            traffic, source = rec["hbm_bytes"], "profiles/r06_factor_pmc.json (RECORDED: 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
    except Exception:
        pass
    recorded = {"bytes": traffic, "source": source}
    measured, note = False, "not attempted"
    tflops = flop / (ms * 1e-3) / 1e12
    rec = {"factor_fast": {"workload": "2048 Jacobian refreshes of n = 253 (qrfac + Q^T f + R + qform), throughput flavour", "kernel": "qrfac_panel_kernel / qrfac_trail_kernel per pair of panels (15 launches) + factor_fast_kernel<16, 2, 2> (qform)",
                           "kernel_ms": ms, "roofline": {"bound": "mfma_fp64", "achieved": tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_FP64_TFLOPS,
                                                         "algorithmic_bytes": alg, "traffic": traffic, "traffic_over_algorithmic": traffic / alg if traffic else None,
                                                         "traffic_source": source, "traffic_measured_in_this_run": measured, "traffic_live_measurement": note,
                                                         "traffic_recorded": recorded}}}
    if live and args is not None:
        solver_kernel_live_traffic(rec, args)
    return rec


def solver_kernel_live_traffic(rec, args):
    """The refresh's HBM traffic measured in this run (two more child runs under rocprofv3 --pmc) written into the record
    solver_kernel_rooflines made.  Called after EVERY timed leg of the line: a profiler child takes the card for up to
    90 s and may leave clocks or power state different, so no timed figure may come after one."""
    r = rec["factor_fast"]["roofline"]
    got, why = live_traffic(args, kernel=FACTOR_KERNELS, child="factor", refreshes=2)
    if got is not None:
        r["traffic"], r["traffic_measured_in_this_run"], r["traffic_live_measurement"] = got, True, "ok"
        r["traffic_source"] = why.replace("two launches of the kernel", "two refreshes (qrfac's chain of panel and trailing launches + the qform launch: "
                                                                          "every dispatch summed, halved)").replace(", median of the launches", "")
        r["traffic_over_algorithmic"] = got / r["algorithmic_bytes"]
    else:
        r["traffic_live_measurement"] = why


def predict_wall(curve, starts):
    """Wall time of a `starts`-start sweep on ONE GPU read off a measured one-GPU curve [(starts, wall_s), ...]: log-log
    interpolation between the measured sizes; below the smallest one its wall time (a sweep cannot take less than its rounds x one
    trajectory latency, whatever its size), above the largest one proportional to the size (throughput-bound)."""
    pts = sorted((float(a), float(b)) for a, b in curve if a > 0 and b > 0)
    if not pts:
        return None
    if starts <= pts[0][0]:
        return pts[0][1]
    if starts >= pts[-1][0]:
        return pts[-1][1] * starts / pts[-1][0]
    for (a0, w0), (a1, w1) in zip(pts, pts[1:]):
        if a0 <= starts <= a1:
            f = (np.log(starts) - np.log(a0)) / (np.log(a1) - np.log(a0))
            return float(np.exp(np.log(w0) + f * (np.log(w1) - np.log(w0))))
    return None


def predictions(curve, total, wall_1=None):
    """What the strong-scaling leg of `total` starts should take on 2 / 4 / 8 GPUs if a rank's block of ceil(total / N) starts takes
    what a sweep of that size takes on one GPU (blocks are independent; the gather is one small all_gather): the expectation the
    measured SCALE curve is to be checked against."""
    out = {}
    for N in (2, 4, 8):
        w = predict_wall(curve, -(-total // N))
        if w is not None:
            out[str(N)] = {"wall_s": w}
            if wall_1:
                out[str(N)]["speedup"] = wall_1 / w
                out[str(N)]["efficiency"] = wall_1 / w / N
    return out


def annotate_legs(legs, curve, source, world, skip):
    """`predicted` (N = 1) / `expected` (N > 1) of every strong-scaling leg from the one-GPU curve of its family."""
    for name, r in legs.items():
        if name == skip:
            r["note"] = "curve point only: the block a rank of 8 gets of the full leg"
            continue
        if curve:
            if world == 1:
                r["predicted"] = predictions(curve, r["total_starts"], r["wall_s"])
            else:
                w = predict_wall(curve, r["starts_per_gpu"])
                w1 = predict_wall(curve, r["total_starts"])
                r["expected"] = {"wall_s": w, "one_gpu_wall_s": w1, "speedup": w1 / w if w and w1 else None,
                                 "measured_over_expected": r["wall_s"] / w if w else None}
            r["prediction_source"] = source


def recorded_sweep_curve(rk4_steps, max_rounds, key="curve"):
    """The one-GPU sweep curve of the last profiled run (profiles/sweep_curve_latest.json, written from a bench line by
    scripts/profile_bench.sh): what an N > 1 run -- which has no one-GPU leg of its own -- states as its expectation."""
    path = os.path.join(ROOT, "profiles", "sweep_curve_latest.json")
    try:
        with open(path) as f:
            c = json.load(f)
        if key != "curve" or (c.get("rk4_steps") == rk4_steps and c.get("max_rounds") == max_rounds):
            return [(int(a), float(b)) for a, b in c[key]], "profiles/sweep_curve_latest.json (RECORDED on one GPU: %s)" % c.get("source", "?")
    except Exception:
        pass
    return None, None


def sweep_leg(torch, dist, capi, args, world, rank, local_rank, dev, use_dist, total=None, family="goddard"):
    """BASELINE config 4 at N GPUs as STRONG scaling: a fixed total of `total` independent starts of the n = 14 single-shooting
    problem (1e4 RK4 steps, full Newton solves in lock-step, throughput flavour, round budget --sweep-max-rounds), sharded in
    contiguous blocks (socp_amd/sweep.py), no data-path exchange, one all_gather of the result records.  Wall time = barrier to
    barrier, max over ranks.  Returned on rank 0.  family = "config5": BASELINE config 5 instead -- the 253-unknown interceptor
    problem under the adaptive integrator (parity unpinned), no round budget."""
    from socp_amd import sweep
    stats = {}
    if family == "config5":
        ctx, Z0, kw = sweep.interceptor_config5_sweep(total, variant="fast", device=local_rank)
        workload = ("interceptor_M21_n%d multi-start sweep, adaptive Dormand-Prince tol 1e-8, full Newton solves on the device "
                    "(BASELINE configs[4] class; PARITY UNPINNED: Boost / Eigen absent)" % Z0.shape[1])
        rk4_steps, max_rounds = None, 0
    else:
        ctx = capi.Context(capi.MODEL_GODDARD, device=local_rank)
        ctx.set_params(GODDARD_PARAMS)
        ctx.set_step_number(args.rk4_steps)
        ctx.set_variant(capi.VARIANT_LANE_FAST)
        sweep.goddard_single_shooting_problem(ctx)
        total = args.sweep_starts if total is None else total
        base = min(total, max(args.sweep_starts, 1))
        Z0 = sweep.goddard_starts(base, 1e-3)
        if total > base:
            # the large leg: the SURVEY 8d table repeated, block b with its costates moved by 1e-7 b (distinct starts, same basin)
            reps = -(-total // base)
            Z0 = np.concatenate([Z0 * np.concatenate([np.ones(7), np.full(7, 1.0 + 1e-7 * b)])[None, :] for b in range(reps)])[:total]
        kw = dict(kind=capi.CHAIN_PLAIN, xtol=1e-8, max_rounds=args.sweep_max_rounds)
        workload = "goddard_single_shooting_n14 multi-start sweep, full Newton solves (BASELINE configs[3] class)"
        rk4_steps, max_rounds = args.rk4_steps, args.sweep_max_rounds

    def solve_block(Zb):
        r = ctx.chains_solve(Zb, **kw)
        stats.update(r["stats"])
        r["rounds"] = r["stats"]["rounds"]
        return r
    # warm-up, untimed: what a process pays once (the context's second stream, the copy engines' start-up: socp_ctx_warm_up) and
    # one small sweep (kernel modules)
    ctx.warm_up()
    ctx.chains_solve(Z0[:64], **dict(kw, max_rounds=4))
    c0 = ctx.counters()[0]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    table, _local = sweep.run_sweep(Z0, solve_block, dist if (use_dist and world > 1) else None, dev if args.backend == "nccl" else None)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    wall = torch.tensor([time.perf_counter() - t0, float(ctx.counters()[0] - c0)], dtype=torch.float64,
                        device=dev if args.backend == "nccl" else torch.device("cpu"))
    if use_dist:
        dist.all_reduce(wall[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(wall[1:], op=dist.ReduceOp.SUM)
    ctx.close()
    info = table[:, -2].astype(int)
    return {"workload": workload, "scaling": "strong",
            "total_starts": total, "n_gpus": world, "starts_per_gpu": -(-total // world), "rk4_steps": rk4_steps, "xtol": 1e-8,
            "max_rounds": max_rounds, "wall_s": float(wall[0]), "solves_per_s": total / float(wall[0]),
            "trajectories": int(wall[1]), "trajectories_per_s": float(wall[1]) / float(wall[0]),
            "converged": int(np.sum(info == 1)), "stopped_by_round_limit": int(np.sum(info == -3)), "rounds_rank0": int(stats.get("rounds", 0)),
            "note": "a sweep's wall time is (rounds of its slowest start) x (one trajectory latency + host work per round) while a "
                    "round's launches fit the chip; sharding shortens it only where a rank's share of a round exceeds one latency"}


def main():
    args = parse_args()
    if args.traffic_child:
        sys.exit(traffic_child(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))            # nothing above has touched the GPU (torch is not even imported yet)

    # stdout carries exactly ONE line, the JSON record: native libraries write there too (RCCL prints a version banner on
    # process-group creation), so file descriptor 1 is pointed at stderr for the duration of the run and the record goes to the
    # saved descriptor at the end
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from socp_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d\n" % (args.gpus, world, world))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_device0:
            local_rank = 0
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.share_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where collective buffers live

    ctx = setup_context(local_rank, args.rk4_steps, args.variant)
    stream = torch.cuda.Stream(device=dev)   # a real (non-default) HIP stream owned by torch
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)       # kernels are enqueued on it: torch events on it bracket them

    P = args.starts
    Z_host = make_starts(P, seed=20250905 + rank)
    d_Z = torch.from_numpy(Z_host).to(dev)
    d_rows = torch.empty((P, ROWS, N_UNKNOWN), dtype=torch.float64, device=dev)
    d_J = torch.empty((P, N_UNKNOWN, N_UNKNOWN), dtype=torch.float64, device=dev)

    elapsed_max, kernel_ms = timed_region(torch, dist, ctx, stream, dev, cdev, world, P, d_Z, d_rows, d_J, args.steps, args.warmup,
                                          collective=use_dist)
    traj_per_step_rank = P * ROWS

    # result record of this rank (checksum of the Jacobians, finite count): the only exchange of the
    # multi-start sweep is this small gather of per-rank records -- after the timed region.
    finite = int(torch.isfinite(d_J).all(dim=(1, 2)).sum().item())
    rec = torch.tensor([float(rank), float(finite), float(torch.nan_to_num(d_J).abs().sum().item())],
                       dtype=torch.float64, device=cdev)
    if use_dist:
        gathered = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(gathered, rec)
        recs = [g.tolist() for g in gathered]
    else:
        recs = [rec.tolist()]

    # strong-scaling legs: fixed totals of starts sharded over the ranks.  At N = 1 (not --lean) one more, small sweep (an eighth
    # of the `sweep` leg: what a rank of 8 gets of it) completes the one-GPU curve the predictions are read from.
    legs = {}
    leg_sizes = [("sweep", args.sweep_starts), ("sweep_large", args.sweep_large_starts), ("sweep_xl", args.sweep_xl_starts)]
    if world == 1 and not args.lean and args.sweep_starts >= 8:
        leg_sizes.insert(0, ("sweep_eighth", args.sweep_starts // 8))
    for name, total in leg_sizes:
        if total > 0 and args.sweep_starts > 0:
            legs[name] = sweep_leg(torch, dist, capi, args, world, rank, local_rank, dev, use_dist, total=total)
    # BASELINE config 5 the same way (its own one-GPU curve: the full leg and, at N = 1, the block a rank of 8 gets)
    legs_c5 = {}
    c5_sizes = [("sweep_config5", args.sweep_c5_starts)]
    if world == 1 and not args.lean and args.sweep_c5_starts >= 8:
        c5_sizes.insert(0, ("sweep_config5_eighth", args.sweep_c5_starts // 8))
    for name, total in c5_sizes:
        if total > 0:
            legs_c5[name] = sweep_leg(torch, dist, capi, args, world, rank, local_rank, dev, use_dist, total=total, family="config5")

    status = 0
    if rank == 0:
        ranks_seen = sorted(int(r[0]) for r in recs)
        total_traj = traj_per_step_rank * world * args.steps
        value = total_traj / elapsed_max
        tflops, gbs = roofline_of(traj_per_step_rank, kernel_ms, args.rk4_steps)
        rec_traffic, rec_source = recorded_traffic(P, args.variant, args.rk4_steps)
        # (the line is built with the figure recorded under profiles/; the in-run PMC measurement -- child processes under rocprofv3 --
        # replaces it at the very end, after every timed leg: ADVICE r5)
        traffic, traffic_source, traffic_live = rec_traffic, rec_source, False
        live_note = "not attempted (N > 1, --lean, under a profiler, or SOCP_BENCH_LIVE_PMC=0)"
        want_live = world == 1 and not args.lean and os.environ.get("SOCP_BENCH_LIVE_PMC", "1") != "0" and "rocprof" not in os.environ.get("LD_PRELOAD", "")
        smooth = GODDARD_PARAMS[6] > 0
        # VGPR budget -> waves per SIMD the launcher may use: fast smooth law 156 VGPRs (3), fast general law 252-254 (2),
        # exact 248 (2) -- socp_amd/csrc/launch.hpp picks min(that, ceil(waves / 1024))
        wpe_max = 3 if (args.variant == "fast" and smooth) else 2
        waves = (traj_per_step_rank + 63) // 64
        out = {
            "metric": "trajectories integrated/sec (Goddard, 14-dim state+costate, 1e4 RK4 steps)",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "goddard_single_shooting_n14_fd_jacobian_batch (BASELINE configs[1])",
                       "starts_per_gpu": P, "trajectories_per_step_per_gpu": traj_per_step_rank,
                       "rk4_steps": args.rk4_steps, "unknowns": N_UNKNOWN, "variant": args.variant,
                       "costate_eps": 1e-3,
                       "parallelism": "independent problems per GPU (replicas of the workload with rank-seeded starts); "
                                      "no data-path collective, one gather of result records after the timed region"},
            "ranks_reported": len(ranks_seen),
            "roofline": {"bound": "valu_fp64", "achieved": tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "fdrows_lane_kernel", "kernel_ms": kernel_ms,
                         "flop_per_trajectory": FLOP_PER_RK4_STEP * args.rk4_steps, "flop_per_rk4_step": FLOP_PER_RK4_STEP,
                         "traffic_measured_in_this_run": traffic_live, "traffic_live_measurement": live_note,
                         "traffic_recorded": {"bytes": rec_traffic, "source": rec_source},
                         "hbm": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": gbs / PEAK_HBM_GBS, "bytes_per_trajectory": BYTES_PER_TRAJ}},
            # Newton-level rates: with M = 1 a residual evaluation is one trajectory, and a forward-difference
            # Jacobian is the n + 1 = 15 evaluations of one start (single shooting has nothing to dedup)
            "jacobians_per_s": P * world * args.steps / elapsed_max,
            "residual_evaluations_per_s": value,
            "occupancy": {"waves_per_launch": waves, "waves_per_simd_cap": min(wpe_max, max(1, -(-waves // 1024))), "simds": 1024},
            "finite_jacobians": [int(r[1]) for r in recs],
        }
        if legs:
            if world == 1:
                curve = sorted((r["total_starts"], r["wall_s"]) for r in legs.values())
                source = "this run"
                out["sweep_curve_one_gpu"] = {"curve": curve, "rk4_steps": args.rk4_steps, "max_rounds": args.sweep_max_rounds,
                                              "model": "wall(N) = one-GPU wall of a sweep of ceil(total / N) starts: log-log interpolation of the "
                                                       "measured sizes, flat below the smallest (rounds x one trajectory latency), "
                                                       "proportional above the largest"}
            else:
                curve, source = recorded_sweep_curve(args.rk4_steps, args.sweep_max_rounds)
            annotate_legs(legs, curve, source, world, skip="sweep_eighth")
            out.update(legs)
        if legs_c5:
            if world == 1:
                curve5, source5 = sorted((r["total_starts"], r["wall_s"]) for r in legs_c5.values()), "this run"
                out.setdefault("sweep_curve_one_gpu", {})["curve_config5"] = curve5
            else:
                curve5, source5 = recorded_sweep_curve(None, None, key="curve_config5")
            annotate_legs(legs_c5, curve5, source5, world, skip="sweep_config5_eighth")
            out.update(legs_c5)
        if ranks_seen != list(range(world)):
            sys.stderr.write("bench.py: records of ranks %s, expected 0..%d\n" % (ranks_seen, world - 1))
            status = 1
        if world == 1 and not args.lean:
            rows_by_variant = {args.variant: d_rows}
            # ---- the other flavour, same workload, its own timed region --------------------------------------------
            other = "exact" if args.variant == "fast" else "fast"
            ctx2 = setup_context(local_rank, args.rk4_steps, other)
            ctx2.set_stream(stream.cuda_stream)
            d_rows2 = torch.empty_like(d_rows)
            k2 = max(1, min(args.steps, 5))
            e2, kms2 = timed_region(torch, dist, ctx2, stream, dev, cdev, 1, P, d_Z, d_rows2, d_J, k2, 1)
            tf2, _ = roofline_of(traj_per_step_rank, kms2, args.rk4_steps)
            out[other] = {"value": traj_per_step_rank * k2 / e2, "unit": "trajectories/s", "steps": k2, "warmup": 1,
                          "ms_per_step": 1e3 * e2 / k2, "kernel_ms": kms2,
                          "roofline_frac": tf2 / PEAK_FP64_TFLOPS, "achieved_tflops": tf2,
                          "note": ("reference operation order, bit-identical to the CPU path; the default of the C-ABI and of the "
                                   "C++ mirror (SOCP_VARIANT_AUTO)") if other == "exact" else "restructured arithmetic, <= 1e-8"}
            rows_by_variant[other] = d_rows2
            # ---- in-run parity of both flavours against the oracle -------------------------------------------------
            try:
                out["parity"] = parity_leg(Z_host, rows_by_variant, args.rk4_steps, K=min(P, 16))
                if not out["parity"]["pass"]:
                    status = 1
            except Exception as exc:       # the checker is absent: say so, do not guess
                out["parity"] = {"error": "oracle unavailable: %s" % exc}
            # ---- latency-bound case: one problem (15 trajectories) per launch --------------------------------------
            single = {}
            for name, c in ((args.variant, ctx), (other, ctx2)):
                one = torch.from_numpy(Z_host[:1].copy()).to(dev)
                c.fd_rows_dev(1, one.data_ptr(), EPSFCN, d_rows.data_ptr())
                torch.cuda.synchronize(dev)
                reps = 3
                t1 = time.perf_counter()
                for _ in range(reps):
                    c.fd_rows_dev(1, one.data_ptr(), EPSFCN, d_rows.data_ptr())
                torch.cuda.synchronize(dev)
                ms = 1e3 * (time.perf_counter() - t1) / reps
                single[name] = {"ms": ms, "value": ROWS / (ms * 1e-3)}
            out["single_problem"] = {"trajectories": ROWS, "unit": "trajectories/s", **single,
                                     "note": "latency-bound: one wave on one SIMD, 4e4 serially dependent RHS evaluations"}
            ctx2.close()
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.rk4_steps, Z_host, args.cpu_seconds)
            b0 = cpu_baseline_b0(args.rk4_steps, args.cpu_seconds)
            if b0:
                out["cpu_baseline"]["b0"] = b0
            # the GPU figures as multiples of both CPU baselines (reported, not the target: the roofline fraction is)
            # B1's all-core figure swings with the host's other tenants (samples of one run differ up to 1.8 x); the per-core figure
            # does not, so the ratio to cores x P1 -- a perfectly scaling CPU -- is the stable one
            cb = out["cpu_baseline"]
            ratios = {"headline_over_b1": value / cb["value"], "headline_over_b1_median": value / cb["median"],
                      "headline_over_%dxP1" % cb["cores"]: value / cb["cores_x_p1"], "headline_over_16xP1": value / (16 * cb["p1"]["value"])}
            if "exact" in out:
                ratios["exact_over_b1"] = out["exact"]["value"] / cb["value"]
                ratios["exact_over_16xP1"] = out["exact"]["value"] / (16 * cb["p1"]["value"])
            if b0:
                ratios["headline_over_b0"] = value / b0["value"]
                if "exact" in out:
                    ratios["exact_over_b0"] = out["exact"]["value"] / b0["value"]
            out["cpu_baseline"]["gpu_ratios"] = ratios
        if world == 1 and not args.lean:
            cpu_v = out.get("cpu_baseline", {}).get("value")
            p1_v = out.get("cpu_baseline", {}).get("p1", {}).get("value")
            out["north_star_128"] = north_star_128(capi, local_rank, args.rk4_steps, cpu_v, p1_v)
            if cpu_v:
                out["north_star_128"]["cpu_baseline_value"] = cpu_v
            out["solver_kernels"] = solver_kernel_rooflines(capi, local_rank, args, live=False)
        if want_live:
            # every timed figure of the line exists now: the profiler children may have the card
            got, why = live_traffic(args)
            rl = out["roofline"]
            if got is not None:
                rl["traffic"], rl["traffic_source"], rl["traffic_measured_in_this_run"], rl["traffic_live_measurement"] = got, why, True, "ok"
                if "solver_kernels" in out:
                    solver_kernel_live_traffic(out["solver_kernels"], args)      # (live where the headline's measurement worked)
            else:
                rl["traffic_live_measurement"] = why
        sys.stdout.flush()
        os.write(record_fd, (json.dumps(out) + "\n").encode())

    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(status)


if __name__ == "__main__":
    main()
