"""Multi-start sweep (BASELINE config 4): P independent initial-costate starts of one shooting problem,
sharded over the ranks of a torch.distributed job -- one process per GPU -- solved locally in lock-step
(socp_multistart_solve) with NO data-path collective, then ONE small all_gather of the per-start
records {z*[n], |F|, info, nfev} (RCCL over xGMI when the backend is nccl; n = 14 -> 136 B per start).

    python -m torch.distributed.run --nproc-per-node N -m socp_amd.sweep --starts 4096 --rk4-steps 10000

Rank r owns the contiguous block of starts [r*P/W, (r+1)*P/W): the gathered table is in start order.
"""
import argparse
import json
import os
import time

import numpy as np

X0_STATE = np.array([0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0])          # testGoddard.cpp:53-59
PSTAR = np.array([-8.121947733, 7.775439382e-3, 0.7775438809, -0.4779369965, 5.715013318e-4,
                  5.715009222e-2, 9.958404873e-2])
TF = 0.2640825
GODDARD_PARAMS = [3.5, 7.0, 310.0, 500.0, 1.0, 1.0, 1.0, -1.0]


def mt19937_64(seed, count):
    """`count` raw outputs of std::mt19937_64(seed) (the generator SURVEY 8d prescribes for the synthetic
    starts), as uint64."""
    NN, MM = 312, 156
    M64 = (1 << 64) - 1
    mt = [0] * NN
    mt[0] = seed & M64
    for i in range(1, NN):
        mt[i] = (6364136223846793005 * (mt[i - 1] ^ (mt[i - 1] >> 62)) + i) & M64
    out = np.empty(count, dtype=np.uint64)
    idx, k = NN, 0
    while k < count:
        if idx >= NN:
            for i in range(NN):
                x = (mt[i] & 0xFFFFFFFF80000000) | (mt[(i + 1) % NN] & 0x7FFFFFFF)
                mt[i] = mt[(i + MM) % NN] ^ (x >> 1) ^ (0xB5026F5AA96619E9 if x & 1 else 0)
            idx = 0
        x = mt[idx]
        idx += 1
        x ^= (x >> 29) & 0x5555555555555555
        x ^= (x << 17) & 0x71D67FFFEDA60000
        x ^= (x << 37) & 0xFFF7EEE000000000
        x ^= x >> 43
        out[k] = x & M64
        k += 1
    return out


def goddard_starts(P, eps, seed=20250905):
    """p = p*(1 + eps*xi), xi = (x >> 11) * 2^-53 * 2 - 1 from raw std::mt19937_64(seed) draws (SURVEY 8d
    "Synthetic inputs"); the SAME table on every rank, sliced by shard()."""
    raw = mt19937_64(seed, 7 * P)
    xi = ((raw >> np.uint64(11)).astype(np.float64) * 2.0 ** -53 * 2.0 - 1.0).reshape(P, 7)
    Z = np.empty((P, 14))
    Z[:, :7] = X0_STATE
    Z[:, 7:] = PSTAR * (1.0 + eps * xi)
    return Z


def shard(P, rank, world):
    """Contiguous block of rank `rank`: sizes differ by at most one."""
    base, rem = divmod(P, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def goddard_single_shooting_problem(ctx, tf=TF):
    """BASELINE config 2/4 problem: n = 14, fixed tf, final velocity and mass free."""
    from . import capi
    mode_x = np.zeros((2, 7), dtype=np.int32)
    mode_x[1, 3:7] = capi.FREE
    X = np.zeros((2, 14))
    X[0, :7] = X0_STATE
    X[1, 0] = 1.01
    return ctx.problem_set([capi.FIXED, capi.FIXED], mode_x, np.array([0.0, tf]), X)


def goddard_multiple_shooting_problem(ctx, M, tf=TF):
    """The testGoddard layout (testGoddard.cpp:24-82) with M segments: initial state fixed, interior nodes
    CONTINUOUS, final altitude pinned at 1.01, final velocity and mass free, final time FREE: n = 14 M + 1."""
    from . import capi
    mode_t = [capi.FIXED] + [capi.CONTINUOUS] * (M - 1) + [capi.FREE]
    mode_x = np.zeros((M + 1, 7), dtype=np.int32)
    mode_x[1:M] = capi.CONTINUOUS
    mode_x[M, 3:7] = capi.FREE
    X = np.zeros((M + 1, 14))
    X[0, :7] = X0_STATE
    X[M, 0] = 1.01
    return ctx.problem_set(mode_t, mode_x, np.linspace(0.0, tf, M + 1), X)


def goddard_multiple_shooting_starts(ctx, Z14, M, tf=TF):
    """Unknown vectors [node states | tf] of the M-segment problem: node 0 = the start's perturbed initial state,
    nodes 1..M-1 = the converged trajectory re-gridded to M uniform segments (the set-up of the survey's
    convergence-basin probe), integrated on the device."""
    P = Z14.shape[0]
    Z = np.empty((P, 14 * M + 1))
    Z[:, :14] = Z14
    Z[:, -1] = tf
    if M > 1:
        star = np.concatenate([X0_STATE, PSTAR])[None, :]
        tk = np.linspace(0.0, tf, M + 1)[1:M]
        nodes = ctx.integrate_batch(np.zeros(M - 1), tk, np.repeat(star, M - 1, axis=0))
        Z[:, 14:14 * M] = nodes.reshape(1, 14 * (M - 1))
    return Z


def goddard_north_star_128_problem(ctx, M=9):
    """north_star's "128-unknown" Goddard layout: M = 9 segments, FREE final time plus ONE FREE interior time (node
    M // 2) -> n = 14 * 9 + 2 = 128.  Node states along the p* trajectory, integrated on the device.  The interior free time adds
    the row H(X-) = 0 (goddard.cpp:343-370, shooting.cpp:961-973) and spaces the CONTINUOUS nodes on either side of it uniformly
    (shooting.cpp:1592-1609); under the smooth law (mu2 > 0) that row is redundant with the free-tf row, so this layout is for
    Jacobian batches at a fixed z, not for solves.  Returns (n, z, mode_t, mode_x, time, X)."""
    from . import capi
    d, s = 7, 14
    mode_t = [capi.FIXED] + [capi.CONTINUOUS] * (M - 1) + [capi.FREE]
    mode_t[M // 2] = capi.FREE
    mode_x = np.full((M + 1, d), capi.CONTINUOUS, dtype=np.int32)
    mode_x[0] = capi.FIXED
    mode_x[M] = capi.FIXED
    mode_x[M, 3:7] = capi.FREE
    tn = np.linspace(0.0, TF, M + 1)
    X = np.zeros((M + 1, s))
    X[0] = np.concatenate([X0_STATE, PSTAR])
    X[M, 0] = 1.01
    X[1:M] = ctx.integrate_batch(np.zeros(M - 1), tn[1:M], np.repeat(X[0][None, :], M - 1, axis=0))
    n = ctx.problem_set(mode_t, mode_x, tn, X)
    z = np.concatenate([X[:M].ravel(), [tn[j] for j in range(M + 1) if mode_t[j] == capi.FREE]])
    return n, z, mode_t, mode_x, tn, X


def torch_at_least(major, minor):
    """The running torch build is at least major.minor (local version tags such as +rocm7.0 ignored; unparsable: False)."""
    import re
    import torch
    m = re.match(r"(\d+)\.(\d+)", getattr(torch, "__version__", ""))
    return bool(m) and (int(m.group(1)), int(m.group(2))) >= (major, minor)


def single_tensor_gather(dist, device_type="cpu"):
    """Whether the gather runs as all_gather_into_tensor.  Decided from facts every rank shares -- the torch build, the process group's
    backend, where the collective's buffers live -- so the answer is the same on every rank:
    nccl (= RCCL) implements it; gloo does from torch 2.1 on and for CPU tensors only (older builds HAVE the function but
    ProcessGroupGloo raises "no support for _allgather_base"), anything else takes the list form.
    SOCP_SWEEP_GATHER=list forces the list form, SOCP_SWEEP_GATHER=tensor the single-tensor one."""
    forced = os.environ.get("SOCP_SWEEP_GATHER")
    if forced == "list":
        return False
    if not hasattr(dist, "all_gather_into_tensor"):
        return False
    if forced == "tensor":
        return True
    backend = dist.get_backend()
    if backend == "nccl":
        return True
    return backend == "gloo" and device_type == "cpu" and torch_at_least(2, 1)


def run_sweep(Z0, solve_local, dist=None, device=None):
    """Shard the rows of Z0 over the ranks, solve, gather.  `solve_local(Zblock)` returns a dict with
    z [k][n], info [k], nfev [k], fnorm [k].  Returns (table [P][n+3] in start order, local dict)."""
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    P, n = Z0.shape
    lo, hi = shard(P, rank, world)
    local = solve_local(Z0[lo:hi])
    rec = np.concatenate([local["z"], local["fnorm"][:, None], local["info"][:, None].astype(float),
                          local["nfev"][:, None].astype(float)], axis=1)
    if world == 1:
        return rec, local
    # equal-size records for all_gather: pad the short shards by one row
    kmax = max(shard(P, r, world)[1] - shard(P, r, world)[0] for r in range(world))
    buf = torch.zeros((kmax, n + 3), dtype=torch.float64, device=device or "cpu")
    buf[:hi - lo] = torch.from_numpy(rec).to(buf.device)
    # ONE collective into ONE tensor and one copy back (a list of per-rank tensors costs a device-to-host copy and a concatenation
    # per rank: 0.2 s of a 3 s sweep at 4 M starts on 8 ranks)
    # The FORM of the collective is chosen before anything is communicated, from facts every rank shares (the torch build and the
    # process group's backend) -- never by catching an error of the collective itself: an error raised on one rank only (a timeout, an
    # out-of-memory condition) would send that rank into a different collective than its peers, and the job would hang.
    # Communication errors propagate.
    full = torch.empty((world * kmax, n + 3), dtype=torch.float64, device=buf.device)
    if single_tensor_gather(dist, buf.device.type):
        try:
            dist.all_gather_into_tensor(full, buf)
        except RuntimeError as exc:
            # NOT a fallback (see above: a rank must never change the collective's form on its own) -- only the way out named
            raise RuntimeError("%s  [socp_amd.sweep: the single-tensor gather failed on backend %s; SOCP_SWEEP_GATHER=list on every "
                               "rank selects the list form]" % (exc, dist.get_backend())) from exc
    else:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        full = torch.cat(parts)
    host = full.cpu().numpy()
    if P == world * kmax:
        return host, local                                            # equal blocks: the gathered tensor IS the table
    table = np.concatenate([host[r * kmax:r * kmax + shard(P, r, world)[1] - shard(P, r, world)[0]] for r in range(world)])
    return table, local


def interceptor_config5_problem(ctx, M=21):
    """BASELINE config 5 class: the interceptor's scenario 1 (tests/testInterceptor.cpp) as an M-segment multiple-shooting problem,
    final time and final velocity free: n = 12 M + 1 (253 at M = 21).  Node states along the CONVERGED trajectory of the test program
    (tests/golden/interceptor_flow.json), integrated on the device with the reference's fixed-step scheme.  Returns the unknown vector."""
    from . import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = json.load(open(os.path.join(root, "tests", "golden", "interceptor_flow.json")))["scenario1_xtol1e-12"][-1]["z"]
    RE = 6378145.0
    X0, tf = np.array(gold[:12]), gold[12]
    Xf = np.zeros(12)
    Xf[:6] = [12000, 1000, 0.0, np.pi / 8, 5475000 / RE, 42000 / RE]
    mode_t = [capi.FIXED] + [capi.CONTINUOUS] * (M - 1) + [capi.FREE]
    mode_x = np.zeros((M + 1, 6), dtype=np.int32)
    mode_x[1:M] = capi.CONTINUOUS
    mode_x[M, 1] = capi.FREE
    time = np.array([tf * i / M for i in range(M + 1)])
    X = np.zeros((M + 1, 12))
    X[0], X[M] = X0, Xf
    ctx.set_integrator(capi.INT_RK4)
    X[1:M] = ctx.integrate_batch(np.zeros(M - 1), time[1:M], np.repeat(X0[None, :], M - 1, axis=0))
    n = ctx.problem_set(mode_t, mode_x, time, X)
    return n, np.concatenate([X[:M].ravel(), [tf]])


def interceptor_config5_sweep(P, variant="fast", eps=1e-3, ode_tol=1e-8, fixed_step=False, device=0, xtol=1e-8):
    """The config-5 solve sweep: a context with the n = 253 problem set and the adaptive integrator chosen, P starts around the
    converged trajectory (node-0 costates perturbed by eps xi, SURVEY 8d's generator) and the keyword arguments of chains_solve.
    Returns (ctx, Z0, kw)."""
    from . import capi
    ctx = capi.Context(capi.MODEL_INTERCEPTOR, device=device)
    ctx.set_variant(capi.VARIANT_LANE_FAST if variant == "fast" else capi.VARIANT_LANE_EXACT)
    n, z = interceptor_config5_problem(ctx)
    if not fixed_step:
        ctx.set_integrator(capi.INT_DOPRI5, ode_tol)
    raw = mt19937_64(20250905, 6 * P)
    xi = ((raw >> np.uint64(11)).astype(np.float64) * 2.0 ** -53 * 2.0 - 1.0).reshape(P, 6)
    Z0 = np.tile(z, (P, 1))
    Z0[:, 6:12] *= 1.0 + eps * xi
    return ctx, Z0, dict(kind=capi.CHAIN_PLAIN, xtol=xtol)


def goddard_kd_chains(ctx, P, eps=0.05, kd_goal=310.0, kd_spread=0.0):
    """testGoddard's state before its KD continuation as P chains: M = 6, free tf, KD = 0, the converged no-drag unknowns
    (tests/golden/goddard_flow.json) with node-0 costates perturbed by eps xi; chain p's goal = kd_goal (1 + kd_spread xi_p).
    Sets the problem on ctx.  Returns (Z0, params, goals, index of KD in the parameter block)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = json.load(open(os.path.join(root, "tests", "golden", "goddard_flow.json")))
    z_nd = np.array([g for g in gold["goddard_single_stage"] if g["stage"] == 2 and g["xtol"] == 1e-6][0]["init_z"])
    goddard_multiple_shooting_problem(ctx, 6, tf=z_nd[-1])
    xi = (goddard_starts(P, eps)[:, 7:] / PSTAR - 1.0) / eps              # the sweep's own uniform(-1, 1) draws
    Z0 = np.tile(z_nd, (P, 1))
    Z0[:, 7:14] *= 1.0 + eps * xi
    params = np.tile(np.array(GODDARD_PARAMS), (P, 1))
    params[:, 2] = 0.0
    goals = kd_goal * (1.0 + kd_spread * xi[:, 0])
    return Z0, params, goals, 2


SOLVERS = {"auto": 0, "host": 1, "device": 2, "device_fast": 3}         # socp_chain_options.solver (include/socp_solver.h)
SOLVER_HELP = ("socp_chain_options.solver (include/socp_solver.h has the rule): where the chains' hybrd state machines run.  auto: on the "
               "device where the host side is the bottleneck (P n^2 >= 1.6e6 -- 4e5 for n <= 32 -- and 20 P >= n: 2048 starts of n = 14, 222 of "
               "n = 85, 25 of n = 253) and the state fits HBM, with the matrix-core factorisation on a throughput-flavour context (39 <= n <= 256); "
               "device: the bit-equal device solvers; device_fast: the matrix-core factorisation asked for explicitly")


def warm_up(ctx, args, solve_first):
    """What a process pays ONCE, kept out of the timed solve like the context and the problem set-up: the context's second stream
    (~6 ms), the copy engines' start-up at the first pinned copy above 16 KB (~8 ms; both: socp_ctx_warm_up) and the first
    launch of every kernel the engine uses (code-object load, ~9 ms) -- the last through a solve of the first --warmup starts
    (0: no warm-up at all; the timed call then includes all three).  Returns the trajectory counter to subtract."""
    if args.warmup > 0:
        ctx.warm_up()
        solve_first(min(args.warmup, args.starts))
    return ctx.counters()[0]


def interceptor_sweep(args, torch, dist, capi, world, rank, local_rank, dev, record_fd):
    eps = args.eps if args.eps is not None else 1e-3
    ctx, Z0, _kw = interceptor_config5_sweep(args.starts, variant=args.variant, eps=eps, ode_tol=args.ode_tol, fixed_step=args.fixed_step,
                                             device=local_rank)
    n = Z0.shape[1]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    stats = {}

    def solve_block(Zb):
        r = ctx.chains_solve(Zb, kind=capi.CHAIN_PLAIN, xtol=args.xtol, speculate=args.speculate, max_rounds=args.max_rounds,
                             solver=SOLVERS[args.solver])
        stats.update(r["stats"])
        r["rounds"] = r["stats"]["rounds"]
        return r
    c0 = warm_up(ctx, args, lambda k: solve_block(Z0[:k]))
    stats.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    table, local = run_sweep(Z0, solve_block, dist if world > 1 else None, dev)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    traj = torch.tensor([float(ctx.counters()[0] - c0)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(traj)
    if rank == 0:
        info = table[:, -2].astype(int)
        conv = table[info == 1, :n]
        record = json.dumps({"sweep": "interceptor_config5_M21_n%d_%s" % (n, "rk4" if args.fixed_step else "dopri5_tol%g" % args.ode_tol),
                             "starts": args.starts, "eps": eps, "n_gpus": world, "variant": args.variant, "solver": args.solver, "xtol": args.xtol, "wall_s": wall,
                             "warmup_starts": args.warmup,
                             "converged": int(np.sum(info == 1)), "info_histogram": {str(k): int(np.sum(info == k)) for k in np.unique(info)},
                             "solution_spread_rel": float(np.max(np.abs(conv - np.median(conv, axis=0))) / np.max(np.abs(conv))) if len(conv) else None,
                             "solves_per_s": args.starts / wall, "trajectories": int(traj.item()), "trajectories_per_s": traj.item() / wall,
                             "rounds_rank0": int(local["rounds"]), "mean_nfev": float(np.mean(table[:, -1])), "engine_rank0": stats})
        os.write(record_fd, (record + "\n").encode())
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--starts", type=int, default=4096)
    ap.add_argument("--eps", type=float, default=None,
                    help="relative costate perturbation; default 1e-3 for single shooting, 0.05 for --segments >= 2 "
                         "(single shooting diverges beyond ~0.3 %%, SURVEY 7 hard part 3)")
    ap.add_argument("--segments", type=int, default=1, help="1: single shooting, fixed tf (n = 14); M >= 2: the "
                    "testGoddard layout with M segments and free tf (n = 14 M + 1)")
    ap.add_argument("--rk4-steps", type=int, default=10000)
    ap.add_argument("--xtol", type=float, default=1e-8)
    ap.add_argument("--variant", choices=["exact", "fast"], default="fast")
    ap.add_argument("--continuation", choices=["none", "kd"], default="none",
                    help="kd: every start is a CHAIN of testGoddard's drag continuation, SolveOCP(step, \"KD\", goal) "
                         "(shooting.cpp:695-778), started from the no-drag solution of the test (tests/golden) with node-0 "
                         "costates perturbed by --eps; all chains advance in lock-step (socp_chains_solve)")
    ap.add_argument("--model", choices=["goddard", "interceptor"], default="goddard",
                    help="interceptor: BASELINE config 5 class -- M = 21 segments, n = 253 unknowns, nodes along the converged "
                         "scenario-1 trajectory of the reference's test program (tests/golden), node-0 costates perturbed by --eps "
                         "(default 1e-3), adaptive Dormand-Prince unless --fixed-step; every start is a full Newton solve")
    ap.add_argument("--fixed-step", action="store_true", help="interceptor: the reference's 50 fixed RK4 steps per stage instead of Dormand-Prince")
    ap.add_argument("--ode-tol", type=float, default=1e-8, help="interceptor: tolerance of the adaptive integrator")
    ap.add_argument("--kd-goal", type=float, default=310.0)
    ap.add_argument("--kd-spread", type=float, default=0.0, help="chain p's goal = kd-goal * (1 + spread * xi_p), xi uniform(-1, 1)")
    ap.add_argument("--step", type=float, default=1.0, help="continuationStep of the chains")
    ap.add_argument("--max-rounds", type=int, default=0, help="socp_chain_options.max_rounds: stop chains still solving after "
                    "that many launch rounds (info = -3); 0 = no limit")
    ap.add_argument("--speculate", type=int, default=-1, help="socp_chain_options.speculate: -1 auto, 0 never, 1 always")
    ap.add_argument("--warmup", type=int, default=8, help="starts of one untimed solve before the timed one (what a process pays once: the "
                    "context's second stream ~6 ms, first-launch code-object loads ~9 ms); 0: the timed call includes them")
    ap.add_argument("--solver", choices=sorted(SOLVERS), default="auto", help=SOLVER_HELP)
    args = ap.parse_args()

    # stdout carries only the JSON record: RCCL prints a version banner to file descriptor 1 when a process group is created
    import sys
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from . import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.model == "interceptor":
        return interceptor_sweep(args, torch, dist, capi, world, rank, local_rank, dev, record_fd)
    ctx = capi.Context(capi.MODEL_GODDARD, device=local_rank)
    ctx.set_params(GODDARD_PARAMS)
    ctx.set_step_number(args.rk4_steps)
    ctx.set_variant(capi.VARIANT_LANE_FAST if args.variant == "fast" else capi.VARIANT_LANE_EXACT)
    eps = args.eps if args.eps is not None else (1e-3 if (args.segments == 1 and args.continuation == "none") else 0.05)
    Z0 = goddard_starts(args.starts, eps)
    chain_kw = None
    if args.continuation == "kd":
        args.segments = 6
        Z0, params, goals, _kd = goddard_kd_chains(ctx, args.starts, eps, args.kd_goal, args.kd_spread)
        chain_kw = dict(kind=capi.CHAIN_PARAM, param_index=2, step=args.step, speculate=args.speculate)
    elif args.segments == 1:
        goddard_single_shooting_problem(ctx)
    else:
        goddard_multiple_shooting_problem(ctx, args.segments)
        lo, hi = shard(args.starts, rank, world)
        Zfull = np.zeros((args.starts, 14 * args.segments + 1))
        Zfull[lo:hi] = goddard_multiple_shooting_starts(ctx, Z0[lo:hi], args.segments)     # each rank builds its own rows
        Z0 = Zfull
    n_unknown = Z0.shape[1]

    stats = {}

    solver = SOLVERS[args.solver]

    def solve_block(Zb):
        lo, hi = shard(args.starts, rank, world)
        if chain_kw is not None:
            r = ctx.chains_solve(Zb, goal=goals[lo:hi], params=params[lo:hi], xtol=args.xtol, max_rounds=args.max_rounds, solver=solver, **chain_kw)
        else:
            r = ctx.chains_solve(Zb, kind=capi.CHAIN_PLAIN, xtol=args.xtol, speculate=args.speculate, max_rounds=args.max_rounds, solver=solver)
        stats.update(r["stats"])
        stats["solves"] = int(np.sum(r["solves"]))
        r["rounds"] = r["stats"]["rounds"]
        return r
    lo_w, hi_w = shard(args.starts, rank, world)

    def warm(k):
        # the first k starts of this rank's block through the same call (continuation chains: with their own goals / parameters)
        if chain_kw is not None:
            return ctx.chains_solve(Z0[lo_w:lo_w + k], goal=goals[lo_w:lo_w + k], params=params[lo_w:lo_w + k], xtol=args.xtol, solver=solver, **chain_kw)
        return ctx.chains_solve(Z0[lo_w:lo_w + k], kind=capi.CHAIN_PLAIN, xtol=args.xtol, speculate=args.speculate, max_rounds=args.max_rounds, solver=solver)
    c0 = warm_up(ctx, args, warm)
    stats.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    table, local = run_sweep(Z0, solve_block, dist if world > 1 else None, dev)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    traj = torch.tensor([float(ctx.counters()[0] - c0)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(traj)
    if rank == 0:
        info = table[:, -2].astype(int)
        conv = table[info == 1, :n_unknown]
        spread = float(np.max(np.abs(conv - np.median(conv, axis=0))) / np.max(np.abs(conv))) if len(conv) else None
        record = json.dumps({"solution_spread_rel": spread, "max_fnorm_converged": float(np.max(table[info == 1, -3])) if len(conv) else None,
                          "sweep": ("goddard_kd_continuation_chains_M6_n85" if chain_kw is not None else
                                    "goddard_single_shooting_n14" if args.segments == 1 else "goddard_multiple_shooting_M%d_n%d" % (args.segments, n_unknown)),
                          "chains_per_s": (args.starts / wall) if chain_kw is not None else None,
                          "continuation": None if chain_kw is None else {"parameter": "KD", "from": 0.0, "goal": args.kd_goal, "goal_spread": args.kd_spread, "step": args.step},
                          "engine_rank0": stats,
                          "starts": args.starts, "eps": eps, "n_gpus": world, "solver": args.solver,
                          "rk4_steps": args.rk4_steps, "variant": args.variant, "xtol": args.xtol, "wall_s": wall, "warmup_starts": args.warmup, "max_rounds": args.max_rounds,
                          "converged": int(np.sum(info == 1)), "info_histogram": {str(k): int(np.sum(info == k)) for k in np.unique(info)},
                          "trajectories": int(traj.item()), "trajectories_per_s": traj.item() / wall,
                          "solves_per_s": args.starts / wall, "rounds_rank0": int(local["rounds"]),
                          "mean_nfev": float(np.mean(table[:, -1]))})
        os.write(record_fd, (record + "\n").encode())
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
