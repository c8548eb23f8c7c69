/*
 * socp_solver.h -- Powell-hybrid Newton solver of the shooting problem, in two extra forms
 * besides the CMinPack-compatible hybrd/hybrj of cminpack.h:
 *
 *  (1) socp_hybrd_batched: hybrd whose finite-difference stage (MINPACK fdjac1: n sequential
 *      callbacks in the reference, 85-93 % of all residual evaluations of a testGoddard solve,
 *      SURVEY 8a row a2) is ONE call that returns the whole forward-difference Jacobian -- the
 *      hook through which the n perturbed residuals become one GPU batch.
 *
 *  (2) a resumable ("reverse communication") solver object: the caller evaluates what the
 *      solver asks for.  Many independent problems (multi-start / continuation sweeps) can
 *      then advance in lock-step with every request of a round evaluated in one launch.
 *
 * Same algorithm, constants and `info` codes as hybrd/hybrj; nfev is accounted as MINPACK does
 * (a forward-difference Jacobian adds min(ml+mu+1, n) = n) so shooting::GetCallNumber keeps
 * its meaning (shooting.cpp:491-493).
 */
#ifndef SOCP_SOLVER_H_
#define SOCP_SOLVER_H_

#include "cminpack.h"

#ifdef __cplusplus
extern "C" {
#endif

/* whole forward-difference Jacobian at x (fvec = F(x) already known), column-major, ld = ldfjac.
 * Return < 0 to abort. */
typedef int (*socp_fdjac_fn)(void *p, int n, const double *x, const double *fvec, double epsfcn,
                             double *fjac, int ldfjac);

int socp_hybrd_batched(cminpack_func_nn fcn, socp_fdjac_fn fdjac, void *p, int n, double *x,
                       double *fvec, double xtol, int maxfev, int ml, int mu, double epsfcn,
                       double *diag, int mode, double factor, int nprint, int *nfev,
                       double *fjac, int ldfjac, double *r, int lr, double *qtf,
                       double *wa1, double *wa2, double *wa3, double *wa4);

/* ---- resumable solver ---- */
#define SOCP_REQ_DONE 0   /* finished: read info / x / fvec                        */
#define SOCP_REQ_FVEC 1   /* evaluate F at *x_eval, write n values to *out         */
#define SOCP_REQ_JAC  2   /* evaluate J at *x_eval, write n*n (column-major) to *out */

typedef struct socp_hybr socp_hybr;

/* analytic_jac = 0: Jacobians are forward differences (hybrd accounting: nfev += n each);
 * analytic_jac = 1: Jacobians are exact (hybrj accounting: njev += 1 each). */
socp_hybr *socp_hybr_create(int n, double xtol, int maxfev, double epsfcn, int mode, double factor,
                            int analytic_jac);
void socp_hybr_destroy(socp_hybr *s);
/* `count` solvers of one size in one arena (huge pages where the system grants them): what the lock-step engine uses.  The
 * solvers belong to the pool: never pass them to socp_hybr_destroy. */
typedef struct socp_hybr_pool socp_hybr_pool;
socp_hybr_pool *socp_hybr_pool_create(int count, int n, double xtol, int maxfev, double epsfcn, int mode, double factor,
                                      int analytic_jac);
socp_hybr *socp_hybr_pool_get(socp_hybr_pool *pool, int i);
void socp_hybr_pool_destroy(socp_hybr_pool *pool);
/* (re)start from x0; diag may be NULL (mode 1) */
int socp_hybr_start(socp_hybr *s, const double *x0, const double *diag);
/* user_flag: value the caller's evaluation returned for the PREVIOUS request (< 0 aborts). */
int socp_hybr_advance(socp_hybr *s, int user_flag, const double **x_eval, double **out);
/* host threads for this solver's O(n^3) factor work (qrfac / qform); default 1.  The factorisation is bit-identical
 * for any count (columns are dealt out to threads, each updated by the serial sequence of operations).  The
 * blocking entry points (hybrd, hybrj, socp_hybrd_batched) pick up to 16 threads by themselves when n >= 192;
 * the environment variable SOCP_LINALG_THREADS overrides that choice. */
int socp_hybr_set_threads(socp_hybr *s, int threads);
int socp_hybr_info(const socp_hybr *s);
int socp_hybr_nfev(const socp_hybr *s);
int socp_hybr_njev(const socp_hybr *s);
const double *socp_hybr_x(const socp_hybr *s);
const double *socp_hybr_fvec(const socp_hybr *s);
double socp_hybr_epsfcn(const socp_hybr *s);
/* the trust-region radius, |diag x| and |F| of the current iterate: what MINPACK's convergence test reads -- info = 1 means
 * delta <= xtol * xnorm (or |F| = 0), whatever |F| is */
void socp_hybr_trust_region(const socp_hybr *s, double *delta, double *xnorm, double *fnorm);

/* ---- lock-step multi-start = socp_chains_solve with SOCP_CHAIN_PLAIN (BASELINE config 4; the sequential continuation loops of
 * shooting.cpp:598-778 solve one problem at a time -- a sweep over P independent starts does not have to).
 * P starts Z0[P][n] of the problem currently set on ctx (socp_problem_set); each start runs its own
 * hybrd state machine with the reference's knobs; per round all residual requests are one
 * socp_residual_batch_dev launch and all Jacobian requests one socp_fd_jacobian_multi_dev launch.
 * Outputs per start: final iterate Zout[P][n], MINPACK info, nfev, |F| at the final iterate;
 * rounds = number of launch rounds.  Host pointers. */
struct socp_ctx;
int socp_multistart_solve(struct socp_ctx *ctx, int P, const double *Z0, double xtol, int maxfev, double epsfcn,
                          double factor, int dedup, double *Zout, int *info, int *nfev, double *fnorm,
                          long long *rounds);

/* ---- lock-step continuation chains.  The reference's two continuation loops are sequential: homotopy
 * on the boundary data, (1-b) previous + b desired (shooting.cpp:598-692), and on one model parameter reached through a
 * real& (shooting.cpp:695-778, shooting.hpp:120-128), each iteration one full Newton solve, with step bisection on failure
 * (:627-648, b - b_prec below continuationStepMin ends the loop) and b += step on success.  Here P chains of ONE problem
 * structure run those loops at once: every chain has its own homotopy state, hybrd state machine, packed parameters and
 * boundary data; per round all residual requests are one launch and all Jacobian requests one launch.  Every chain's solves
 * are, bit for bit, the ones shooting::SolveOCP(step[, Rdata, Rgoal]) of the host mirror performs for it alone. */
#define SOCP_CHAIN_PLAIN 0   /* one Newton solve per chain: the multi-start sweep (SolveOCP(0)) */
#define SOCP_CHAIN_PARAM 1   /* SolveOCP(step, Rdata, Rgoal): packed parameter `param_index` moves from its value to goal[p] */
#define SOCP_CHAIN_DATA  2   /* SolveOCP(step): boundary data move from (time_prev, x_prev) to (time_goal, x_goal) */

typedef struct socp_chain_options {
    int kind;                 /* SOCP_CHAIN_* */
    int param_index;          /* SOCP_CHAIN_PARAM: slot of the packed parameter block (SOCP_*_NPARAMS order) */
    double step;              /* continuationStep > 0 (the mirror maps SolveOCP(<= 0, Rdata, Rgoal) to 1.0) */
    double step_min;          /* continuationStepMin, shooting.cpp:89: 1e-12 */
    double xtol;              /* hybrd knobs, shooting.cpp:95-105 */
    int maxfev;
    double epsfcn;
    double factor;
    int dedup;                /* launched FD Jacobians integrate only the segments a column can change */
    int speculate;            /* (both engines) residual requests evaluated as whole FD batches so that later Jacobian requests at an accepted
                                 point need no launch: -1 = when the chip has idle SIMDs (default), 0 = never (also what a zeroed
                                 struct means), 1 = always.
                                 No iterate depends on it.  Environment SOCP_CHAINS_SPECULATE overrides. */
    int max_rounds;           /* 0 = no limit.  > 0: chains still solving after that many launch rounds are stopped with
                                 info = SOCP_INFO_ROUND_LIMIT, like a callback returning < 0 (shooting.cpp:873): a sweep's wall
                                 time is rounds x one trajectory latency and the round count is set by its slowest chain
                                 (typically one that ends in info 4/5 anyway).  The other chains' iterates do not change. */
    int analytic_jac;         /* 0: hybrd, forward-difference Jacobians (modelOrder 0).  1: hybrj with the variational Jacobian
                                 (modelOrder 1, shooting.cpp:828-852,996-1130; models with variational equations only): the
                                 Jacobian requests of a round are one batched variational integration, one wavefront per
                                 (chain, segment); nfev / njev are accounted as hybrj does. */
    int solver;               /* where the chains' hybrd / hybrj state machines run: SOCP_SOLVER_AUTO (0; also what a zeroed struct
                                 means), SOCP_SOLVER_HOST, SOCP_SOLVER_DEVICE.  DEVICE: one workgroup per chain runs MINPACK's qrfac /
                                 qform / dogleg / r1updt / r1mpyq in HBM with the per-column operation order of the host code, so
                                 every iterate, nfev and info is the host solver's, bit for bit; Jacobians never cross PCIe.  AUTO
                                 picks DEVICE where the host side is the bottleneck (P n^2 >= 1.6e6 -- 4e5 for n <= 32 -- and 20 P >= n:
                                 2048 chains of n = 14, 222 of n = 85, 25 of n = 253, 42 of n = 832) and the solver state fits HBM (if the device engine cannot
                                 allocate it after all, AUTO runs the host engine; an explicit DEVICE returns SOCP_ERR_HIP).
                                 SOCP_SOLVER_DEVICE_FAST: the device solvers with the Jacobian refresh (qrfac + qform, 70 % of the solver
                                 time at n = 253) in its THROUGHPUT flavour -- blocked Householder, compact-WY panels of 16, trailing
                                 updates on the FP64 matrix cores, free summation order (kernels_factor_fast.hip): iterates equal the
                                 host solver's to rounding, NOT bit for bit; converged solutions within north_star's 1e-8.  Sizes
                                 39 <= n <= 256, otherwise it is DEVICE.  From n = 192 up this flavour also leaves Q as factorised
                                 between refreshes: Broyden's rotations (r1mpyq) are kept as a list and applied to the vector
                                 Q0^T f instead of to the matrix (rounding-level again; SOCP_SOLVER_LAZY_Q=0|1: never / always).
                                 AUTO picks it instead of DEVICE on a context whose arithmetic
                                 flavour is the throughput one (SOCP_VARIANT_LANE_FAST: its trajectories differ from the reference
                                 order's at rounding level already); on a reference-order context AUTO never does.
                                 Environment SOCP_CHAINS_SOLVER=host|device|device_fast overrides. */
} socp_chain_options;
#define SOCP_SOLVER_AUTO   0
#define SOCP_SOLVER_HOST   1
#define SOCP_SOLVER_DEVICE 2
#define SOCP_SOLVER_DEVICE_FAST 3
#define SOCP_INFO_ROUND_LIMIT (-3)

typedef struct socp_chain_stats {
    long long rounds;                 /* launch rounds */
    long long jacobians_launched;     /* Jacobian requests that needed trajectories */
    long long jacobians_from_cache;   /* Jacobian requests formed from rows already in HBM */
    long long speculative_rounds;     /* rounds in which residual requests ran as FD batches */
    long long restarts;               /* Newton solves started after the first one of each chain (homotopy steps) */
    double wall_ms;
} socp_chain_stats;

/* Host pointers.  Z0[P][n]: start of every chain (shooting's tab_param).  params: NULL (every chain uses the context's
 * parameters) or [P][nparams].  goal[P]: SOCP_CHAIN_PARAM only.  time_prev[P][M+1], x_prev[P][(M+1)*2d], time_goal, x_goal:
 * SOCP_CHAIN_DATA (only the first d entries of each node row are blended and read, as in shooting.cpp:609-611); for the
 * other kinds time_goal / x_goal may give every chain its own fixed boundary data (NULL: the problem's).
 * Outputs per chain (any but Zout / info may be NULL): Zout = the unknowns of the last CONVERGED solve (tab_param; the start
 * if none converged) -- for SOCP_CHAIN_PLAIN the final iterate whatever info says, as socp_multistart_solve returns it;
 * info / nfev_last of the last solve; nfev_total and solves over the chain; b_reached = largest homotopy value solved
 * (1 = goal reached); param_final = the moving parameter at exit (the reference leaves Rdata there); fnorm = |F| of the last
 * solve's final iterate.
 * Return value SOCP_OK: every output array is written for all P chains.  ANY OTHER return value: the contents of the output arrays
 * are UNDEFINED (a call whose chains run as groups may have written the rows of the groups that finished before another failed --
 * e.g. an explicit SOCP_SOLVER_DEVICE whose second group could not allocate returns SOCP_ERR_HIP with the first group's rows
 * written) and the context's trajectory / launch counters include whatever ran before the failure. */
int socp_chains_solve(struct socp_ctx *ctx, int P, const socp_chain_options *opt, const double *Z0, const double *params,
                      const double *goal, const double *time_prev, const double *x_prev, const double *time_goal,
                      const double *x_goal, double *Zout, int *info, int *nfev_last, int *nfev_total, int *solves,
                      double *b_reached, double *param_final, double *fnorm, socp_chain_stats *stats);
/* the same with njev_last[P] (Jacobian evaluations of the last solve: what GetCallNumber()[1] reports on the hybrj path) */
int socp_chains_solve_ex(struct socp_ctx *ctx, int P, const socp_chain_options *opt, const double *Z0, const double *params,
                         const double *goal, const double *time_prev, const double *x_prev, const double *time_goal,
                         const double *x_goal, double *Zout, int *info, int *nfev_last, int *nfev_total, int *njev_last,
                         int *solves, double *b_reached, double *param_final, double *fnorm, socp_chain_stats *stats);

/* ---- the device engine's workspace.  socp_chains_solve with the solvers on the device takes ONE device allocation and ONE pinned host
 * allocation per call (32 GB + 1.8 GB for 4 M chains of n = 14; 26 GB for 16 384 of n = 253).  They are KEPT for the next call on that
 * device instead of being returned (up to FOUR pairs per device -- one per workspace slot: chain groups of one call run side by side,
 * each on its own pair --, grown when a call needs more, never shrunk; a call can only take the pair of its own slot, so when an
 * allocation fails the idle blocks OF THE SAME KIND -- device memory or pinned memory, whichever could not be had -- of the device's
 * slots are released and it is tried once more): returning and taking tens of GB
 * costs 0.7-1 s per call on this platform -- freed device memory is cleared before it is handed out again, and an allocation that
 * follows a large free waits for that (measured: a 4.3 GB arena 756 ms after a 32 GB one was freed, 0.4 ms otherwise).  A second
 * engine call on the same device while the first is running takes private allocations.  SOCP_WORKSPACE_CACHE=0 turns the keeping
 * off.  No reference counterpart (the reference allocates std::vectors per callback, shooting.cpp:784-799).
 *
 * socp_workspace_release: frees what is kept for `device` (< 0: every device) and not in use; returns the number of bytes freed.
 * socp_workspace_cached_bytes: device + pinned bytes currently kept for `device` (< 0: all). */
double socp_workspace_release(int device);
double socp_workspace_cached_bytes(int device);

/* ---- multi-GPU sweep from C++ (SURVEY 8e level 1; north_star: "a continuation/multi-start outer loop shards independent
 * shooting problems across the 8 GPUs of one node with a trivial gather").  The reference has no counterpart: its continuation
 * loops (shooting.cpp:598-778) solve one problem at a time on one thread; what is kept is its plugin surface -- the problem is
 * described once, through a context (or, from the C++ mirror, through model + shooting: shooting.hpp:29-219).
 *
 * Problems are independent, so the sharding is contiguous blocks with NO data-path exchange; the only communication is the
 * gather of fixed-size result records at the end.
 *
 * socp_sweep_shard: the block of rank `rank` of `world` -- sizes differ by at most one (the reference's partition of segments
 * over threads, shooting.cpp:1223-1231, applied to problems). */
void socp_sweep_shard(int P, int rank, int world, int *lo, int *hi);

/* A copy of `proto` on another device: same model, packed parameters, switching times, step number, integrator, arithmetic
 * flavour and shooting problem (its tables are rebuilt there).  device < 0: the calling thread's current device. */
int socp_ctx_clone(const struct socp_ctx *proto, int device, struct socp_ctx **out);

typedef struct socp_sweep_stats {
    int ndev;
    double wall_ms;                   /* whole call */
    double device_wall_ms[16];        /* per device: its block's lock-step solve */
    long long device_rounds[16];
    long long trajectories;           /* integrated on all devices */
} socp_sweep_stats;

/* ONE process, `ndev` GPUs: one host thread and one context (socp_ctx_clone of `proto`) per device, device k solving block k
 * of the P starts Z0[P][n] with socp_chains_solve (opt->kind = SOCP_CHAIN_PLAIN: a multi-start sweep; the per-chain arrays
 * params / goal / time_* / x_* of socp_chains_solve may be given for continuation chains, NULL otherwise) and writing its
 * rows of the outputs -- the gather is the shared host memory.  devices[k]: HIP device indices (ndev <= 16; NULL = 0 .. ndev-1).
 * Outputs as socp_chains_solve.  Every chain's result is the one it has on a single GPU (blocks are independent). */
int socp_sweep_solve(const struct socp_ctx *proto, const int *devices, int ndev, int P, const socp_chain_options *opt, const double *Z0,
                     const double *params, const double *goal, const double *time_prev, const double *x_prev, const double *time_goal,
                     const double *x_goal, double *Zout, int *info, int *nfev_last, int *nfev_total, int *solves, double *b_reached,
                     double *param_final, double *fnorm, socp_sweep_stats *stats);

/* ONE process PER GPU (the layout of an MPI / RCCL job): rank `rank` of `world` solves its block on `ctx` and the records
 * {Zout[n], fnorm, info, nfev_last, nfev_total, solves} (n + 5 doubles per start, + one status double per rank: a rank whose
 * solve failed still enters the collective and every rank returns that failure) of all ranks are gathered with the caller's
 * collective: gather(user, send, count, recv) must behave like an all-gather of `count` doubles per rank into
 * recv[world][count] -- ncclAllGather(send, recv, count, ncclDouble, comm, stream) + a stream synchronise, MPI_Allgather, or
 * a copy when world = 1 (INTEGRATION.md shows the RCCL form; this library does not link a communication library itself).
 * send / recv are HOST buffers unless gather_on_device != 0, in which case they are device-visible buffers: memory of ctx's
 * device, or -- when that staging cannot be allocated or written on some rank -- pinned host memory (hipHostMalloc), which a
 * device collective reads and writes like device memory.
 * The collective is entered EXACTLY ONCE by every rank whatever fails locally (the device switch, the staging allocation,
 * the copy of the message, the solve): the failure travels in the rank's status slot and every rank returns it.  The one
 * exception: a rank that can obtain neither device nor pinned memory for the message returns SOCP_ERR_HIP without having
 * called gather -- the caller must then abort the communicator (ncclCommAbort / MPI_Abort), the other ranks are waiting.
 * Argument errors (SOCP_ERR_ARG / SOCP_ERR_UNSUPPORTED before any work) are returned without the collective: the ranks of
 * a job make the same call, so they all take that exit.
 * Outputs: the full tables of all P starts, on every rank. */
typedef int (*socp_allgather_fn)(void *user, const double *send, long count, double *recv);
int socp_sweep_solve_rank(struct socp_ctx *ctx, int rank, int world, int P, const socp_chain_options *opt, const double *Z0,
                          socp_allgather_fn gather, void *user, int gather_on_device, double *Zout, int *info, int *nfev_last,
                          int *nfev_total, int *solves, double *fnorm, socp_chain_stats *stats);

/* ---- the Jacobian refresh of the device solvers on its own: `count` QR factorisations of n x n matrices in MINPACK's convention
 * (qrfac without pivoting: Householder vectors a_j / |a_j| + e_j, diag(R) = -|a_j| sign(a_jj); qform: Q = H_0 ... H_{n-1}; Q^T b),
 * what hybrd does with every fresh Jacobian (call site shooting.cpp:803-826) -- for tests and for measuring the
 * kernel.  flavour SOCP_FACTOR_EXACT: MINPACK's per-column operation order (bit-equal to the host solver's factors);
 * SOCP_FACTOR_FAST: the blocked matrix-core form (39 <= n <= 256).  Host pointers; J[count][n * n] COLUMN-major (as the
 * forward-difference kernels write Jacobians), b[count][n].  Outputs (any may be NULL): Q[count][n][n] row-major,
 * R[count][n (n + 1) / 2] packed by rows (row i: diag, then (i, i + 1 .. n - 1)), qtb[count][n], rdiag[count][n],
 * acnorm[count][n] (column norms of J), sing[count].  kernel_ms: mean time of the factor launch over `reps` >= 1 runs
 * (the matrices are restored before each run, untimed). */
#define SOCP_FACTOR_EXACT 0
#define SOCP_FACTOR_FAST  1
int socp_qr_factor_batch(int device, int n, int count, const double *J, const double *b, int flavour, int reps, double *Q, double *R,
                         double *qtb, double *rdiag, double *acnorm, int *sing, double *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* SOCP_SOLVER_H_ */
