#!/usr/bin/env python3
"""bench.py -- throughput of the SOCP hot path on MI355X: trajectories integrated per second.

Workload (BASELINE.json configs[1], on synthetic random-init costate batches as north_star asks):
every rank holds `--starts` independent Goddard single-shooting problems (n = 14 unknowns, fixed
tf, KD = 310, mu2 = 1; initial costates p = p*(1 + 1e-3 xi)).  ONE STEP = the
forward-difference-Jacobian batch of every start: base residual + 14 perturbed residuals =
15 trajectories per start, each 10 000 RK4 steps of the 14-dim state+costate system, in one launch
(socp_fd_rows_dev), followed by the small difference kernel that forms the Jacobians.  Inputs are
resident in HBM before the timed region.  value = trajectories of ALL ranks / max-over-ranks time.

Extra objects in the JSON line:
  roofline      dominant kernel (fdrows_lane_kernel).  This path has no dense contraction and moves
                224 B per trajectory, so neither "mfma" nor "hbm" bounds it: the binding resource is
                FP64 vector issue.  `bound` is therefore "valu_fp64" (peak = 256 CU x 4 SIMD x 16
                FP64 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s, half the FP32 vector peak of
                MI355X_MICROARCH.md); the HBM view the contract asks for is in roofline.hbm.
  cpu_baseline  the reference's own model::ComputeTraj (oracle/_ref, kind "reference") or the C
                oracle (kind "port") timed on this box's host cores on a bounded sample.
  single_problem  (--single-problem) the latency-bound case: ONE problem (15 trajectories) per launch.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TRAJ = 1.17e7        # ~250 FP64 ops / RHS x 4 + 168 (RK4 combine), x 1e4 steps
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


def cpu_baseline(steps_rk4, Z, target_seconds):
    """Reference (or port) on the host cores, bounded sample of the SAME trajectories (rows of the
    FD batch of the first starts)."""
    from oracle import oracle as orc
    threads = min(16, os.cpu_count() or 1)      # the GPU box's CPU share for one GPU
    eps = np.sqrt(1e-15)

    def sample_rows(count):
        X0 = np.empty((count, 14))
        for k in range(count):
            p, row = divmod(k, ROWS)
            X0[k] = Z[p % len(Z)]
            if row > 0:
                j = row - 1
                h = eps * abs(X0[k, j]) or eps
                X0[k, j] += h
        return X0

    if orc.have_ref():
        ref = orc.Ref(orc.MODEL_GODDARD, step_nbr=steps_rk4)
        probe = sample_rows(threads)
        _, sec = ref.goddard_traj_batch(threads, steps_rk4, GODDARD_PARAMS, 0.0, TF, probe)
        count = int(max(threads, min(65536, threads * round(target_seconds / max(sec, 1e-3)))))
        X0 = sample_rows(count)
        _, sec = ref.goddard_traj_batch(threads, steps_rk4, GODDARD_PARAMS, 0.0, TF, X0)
        return {"value": count / sec, "unit": "trajectories/s", "cores": threads, "kind": "reference",
                "per_core": count / sec / threads,
                "sample": "%d trajectories (FD-batch rows of the first %d starts), %d RK4 steps each, "
                          "reference model::ComputeTraj, one goddard object per std::thread, %.1f s"
                          % (count, (count + ROWS - 1) // ROWS, steps_rk4, sec)}
    o = orc.Oracle(orc.MODEL_GODDARD, step_nbr=steps_rk4, params=GODDARD_PARAMS)
    probe = sample_rows(2)
    t = time.perf_counter()
    o.integrate_batch(0.0, TF, probe)
    per = (time.perf_counter() - t) / 2
    count = int(max(2, min(4096, round(target_seconds / max(per, 1e-4)))))
    X0 = sample_rows(count)
    t = time.perf_counter()
    o.integrate_batch(0.0, TF, X0)
    sec = time.perf_counter() - t
    return {"value": count / sec, "unit": "trajectories/s", "cores": 1, "kind": "port",
            "sample": "%d trajectories, %d RK4 steps each, C oracle single thread, %.1f s" % (count, steps_rk4, sec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--starts", type=int, default=13107,
                    help="independent shooting problems per GPU; default: 15 x 13107 = 196 605 trajectories = 3072 "
                         "wavefronts = 3 per SIMD (an exactly full chip); any size >= 4096 runs within 10 %% of it")
    ap.add_argument("--rk4-steps", type=int, default=10000)
    ap.add_argument("--variant", choices=["exact", "fast"], default="fast",
                    help="fast: restructured arithmetic (<= 1e-8 vs the reference order after 1e4 steps, converged "
                         "solutions within 1e-8: tests/test_gpu_parity.py, test_host_flow.py); exact: reference operation order")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + --share-device0: rehearse the multi-rank path on a one-GPU box (collectives on CPU tensors)")
    ap.add_argument("--share-device0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--single-problem", action="store_true",
                    help="also time ONE problem (15 trajectories) per launch: the latency-bound case")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    epsfcn = 1e-15

    def step(events=None):
        if events is not None:
            events[0].record(stream)
        ctx.fd_rows_dev(P, d_Z.data_ptr(), epsfcn, d_rows.data_ptr())
        if events is not None:
            events[1].record(stream)
        ctx.fd_diff_dev(P, d_Z.data_ptr(), epsfcn, d_rows.data_ptr(), d_J.data_ptr())

    def fence():
        torch.cuda.synchronize(dev)          # this rank's own launches are done ...
        if world > 1:
            dist.barrier()                   # ... and so are everybody else's
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    fence()
    elapsed = time.perf_counter() - t0

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed_max = float(t_max.item())

    traj_per_step_rank = P * ROWS
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    # result record of this rank (checksum of the Jacobians, finite count): the only exchange of the
    # multi-start sweep is this small gather of per-rank records -- after the timed region.
    finite = int(torch.isfinite(d_J).all(dim=(1, 2)).sum().item())
    rec = torch.tensor([float(rank), float(finite), float(torch.nan_to_num(d_J).abs().sum().item())],
                       dtype=torch.float64, device=cdev)
    if world > 1:
        gathered = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(gathered, rec)
        recs = [g.tolist() for g in gathered]
    else:
        recs = [rec.tolist()]

    if rank == 0:
        total_traj = traj_per_step_rank * world * args.steps
        value = total_traj / elapsed_max
        tflops = FLOP_PER_TRAJ * traj_per_step_rank / (kernel_ms * 1e-3) / 1e12
        gbs = BYTES_PER_TRAJ * traj_per_step_rank / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    entries = json.load(f)
                for tj in (entries if isinstance(entries, list) else [entries]):
                    if tj.get("starts") == P and tj.get("variant") == args.variant and tj.get("rk4_steps") == args.rk4_steps:
                        traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "trajectories integrated/sec (Goddard, 14-dim state+costate, 1e4 RK4 steps)",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "goddard_single_shooting_n14_fd_jacobian_batch (BASELINE configs[1])",
                       "starts_per_gpu": P, "trajectories_per_step_per_gpu": traj_per_step_rank,
                       "rk4_steps": args.rk4_steps, "unknowns": N_UNKNOWN, "variant": args.variant,
                       "costate_eps": 1e-3},
            "roofline": {"bound": "valu_fp64", "achieved": tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / PEAK_FP64_TFLOPS, "traffic": traffic,
                         "kernel": "fdrows_lane_kernel", "kernel_ms": kernel_ms,
                         "flop_per_trajectory": FLOP_PER_TRAJ,
                         "hbm": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": gbs / PEAK_HBM_GBS, "bytes_per_trajectory": BYTES_PER_TRAJ}},
            # Newton-level rate and wave occupancy beside the trajectory rate
            # Newton-level rates: with M = 1 a residual evaluation is one trajectory, and a forward-difference
            # Jacobian is the n + 1 = 15 evaluations of one start (single shooting has nothing to dedup)
            "jacobians_per_s": P * world * args.steps / elapsed_max,
            "residual_evaluations_per_s": value,
            "occupancy": {"waves_per_launch": (traj_per_step_rank + 63) // 64,
                          "waves_per_simd_cap": min(3 if args.variant == "fast" else 2, max(1, -(-((traj_per_step_rank + 63) // 64) // 1024))),
                          "simds": 1024},
            "finite_jacobians": [int(r[1]) for r in recs],
        }
        if world == 1 and args.single_problem:
            # latency-bound case: one problem (15 trajectories) per launch
            one = torch.from_numpy(Z_host[:1].copy()).to(dev)
            ctx.fd_rows_dev(1, one.data_ptr(), epsfcn, d_rows.data_ptr())
            torch.cuda.synchronize(dev)
            reps = 3
            t1 = time.perf_counter()
            for _ in range(reps):
                ctx.fd_rows_dev(1, one.data_ptr(), epsfcn, d_rows.data_ptr())
            torch.cuda.synchronize(dev)
            ms = 1e3 * (time.perf_counter() - t1) / reps
            out["single_problem"] = {"trajectories": ROWS, "ms": ms, "value": ROWS / (ms * 1e-3), "unit": "trajectories/s"}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.rk4_steps, Z_host, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
