// batchsolve.cpp -- lock-step Newton solves of many independent shooting problems of ONE structure: multi-start sweeps
// (BASELINE config 4) and CONTINUATION CHAINS: the reference's two discrete continuation loops --
// homotopy on the boundary data (shooting.cpp:598-692) and on one model parameter reached through a real&
// (shooting.cpp:695-778) -- run for P chains at once.  Every chain owns
//     * a resumable hybrd state machine (minpack.cpp),
//     * its homotopy state (b, b_prec, running) with the reference's bisection rules,
//     * its own packed model parameters and boundary tables (per-problem blocks, dev_common.hpp).
// Per round all pending residual requests are ONE launch and all pending Jacobian requests ONE launch; chains are not
// synchronised with each other (a chain starts its next homotopy step as soon as its solve ends).  Each chain follows,
// bit for bit, the iterates the sequential loops of the host mirror follow for it alone (tests/test_gpu_chains.py).
//
// Speculative Jacobians: a residual-only round of 4096 single-shooting starts puts 64 waves on 1024 SIMDs.
// When the chip has idle SIMDs, a residual request is therefore evaluated as the WHOLE forward-difference batch of its
// point (fdrows kernel: row 0 = F(x), rows j + 1 = F(x + h_j e_j)) at no extra latency; the rows of a chain's last
// accepted point stay in HBM, and the Jacobian hybrd asks for after two failed trial steps (always at the last accepted
// point) or at the start of a solve is formed from them without another round of trajectories.  Values are the ones the
// fdjac kernel would produce ((F(x + h e) - F(x)) / h from the same arithmetic), so no iterate changes.
#include "../../include/socp_hip.h"
#include "../../include/socp_solver.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "chains_common.hpp"
#include "host_pool.hpp"

// batchsolve_dev.cpp: the same engine with the solvers on the device
int socp_chains_solve_device(socp_ctx *ctx, int P, const socp_chain_options *opt, const double *Z0, const double *params,
                             const double *goal, const double *time_prev, const double *x_prev, const double *time_goal,
                             const double *x_goal, double *Zout, int *info, int *nfev_last, int *nfev_total, int *njev_last,
                             int *solves, double *b_reached, double *param_final, double *fnorm, socp_chain_stats *stats, int fast_factor,
                             int workspace_slot);
// ... what it allocates on the device for P chains of this context's problem (one plan, shared with the allocation itself), whether the
// throughput factorisation exists for size n, and the code with which it reports "could not allocate, nothing has run yet"
double socp_chains_device_bytes(const socp_ctx *ctx, int P, const socp_chain_options *opt, bool per_chain_params, bool per_chain_bounds);
bool socp_chains_fast_factor_applies(int n);
double socp_workspace_reusable_device_bytes(int device, int slot);
constexpr int kDeviceEngineAllocFailed = -1000;
using clk_t = std::chrono::steady_clock;

namespace {

struct Pinned {
    void *p = nullptr;
    size_t cap = 0;
    bool reserve(size_t bytes)
    {
        if (bytes <= cap) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) return false;
        cap = bytes;
        return true;
    }
    ~Pinned() { if (p) (void)hipHostFree(p); }
    double *d() const { return static_cast<double *>(p); }
    int *i() const { return static_cast<int *>(p); }
};
struct Dev {
    void *p = nullptr;
    bool alloc(size_t bytes) { release(); return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess; }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
    ~Dev() { release(); }
    double *d() const { return static_cast<double *>(p); }
    int *i() const { return static_cast<int *>(p); }
};

// dst[dst_idx[k]][0..len) = src[src_idx[k]][0..len) for k < count: moves (n+1) x n row blocks between the round's staging
// area and the chains' cache slots in one launch (thousands of separate hipMemcpy calls would cost more than a round)
__global__ void copy_blocks_kernel(const double *__restrict__ src, const int *__restrict__ src_idx, double *__restrict__ dst,
                                   const int *__restrict__ dst_idx, int count, int len)
{
    const int k = blockIdx.y;
    if (k >= count) return;
    const double *s = src + (size_t)src_idx[k] * len;
    double *d = dst + (size_t)dst_idx[k] * len;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < len; e += gridDim.x * blockDim.x) d[e] = s[e];
}

bool copy_blocks(hipStream_t st, const double *src, const int *d_src_idx, double *dst, const int *d_dst_idx, int count, int len)
{
    const unsigned gx = (unsigned)std::min(8, (len + 255) / 256);
    for (int k0 = 0; k0 < count; k0 += 32768) {                  // grid.y is limited to 65535
        const int kc = std::min(32768, count - k0);
        hipLaunchKernelGGL(copy_blocks_kernel, dim3(gx, (unsigned)kc), dim3(256), 0, st, src, d_src_idx + k0, dst, d_dst_idx + k0, kc, len);
    }
    return hipGetLastError() == hipSuccess;
}

struct Chain : socp::chains::ChainCore {
    socp_hybr *solver = nullptr;
    // request state
    int flag = 0, req = SOCP_REQ_DONE;
    const double *xin = nullptr;
    double *xout = nullptr;
    bool need_advance = true;
    // speculative-Jacobian cache
    std::vector<double> eval_x;         // the point whose FD rows sit in the staging area (this round's request)
    int stage_idx = -1;                 // its block in the staging area, -1: none
    std::vector<double> slot_x;         // the point whose rows sit in this chain's slot
    bool slot_valid = false;
};

}  // namespace

extern "C" int socp_chains_solve(socp_ctx *ctx, int P, const socp_chain_options *opt, const double *Z0, const double *params,
                                 const double *goal, const double *time_prev, const double *x_prev, const double *time_goal,
                                 const double *x_goal, double *Zout, int *info, int *nfev_last, int *nfev_total, int *solves,
                                 double *b_reached, double *param_final, double *fnorm, socp_chain_stats *stats)
{
    return socp_chains_solve_ex(ctx, P, opt, Z0, params, goal, time_prev, x_prev, time_goal, x_goal, Zout, info, nfev_last, nfev_total,
                                nullptr, solves, b_reached, param_final, fnorm, stats);
}

// how many groups of chains a device-solver call runs side by side (see the call site below)
static int device_groups(int P, const socp_chain_options *opt)
{
    int G = (P >= 262144 && !opt->analytic_jac) ? 2 : 1;
    if (const char *e = std::getenv("SOCP_CHAINS_DEVICE_GROUPS")) G = std::max(1, std::min(4, std::atoi(e)));
    if (opt->analytic_jac || G > P) G = 1;
    return G;
}

extern "C" int socp_chains_solve_ex(socp_ctx *ctx, int P, const socp_chain_options *opt, const double *Z0, const double *params,
                                    const double *goal, const double *time_prev, const double *x_prev, const double *time_goal,
                                    const double *x_goal, double *Zout, int *info, int *nfev_last, int *nfev_total, int *njev_last,
                                    int *solves, double *b_reached, double *param_final, double *fnorm, socp_chain_stats *stats)
{
    if (!ctx || !opt || P < 0 || (P > 0 && (!Z0 || !Zout || !info))) return SOCP_ERR_ARG;
    const int n = socp_problem_num_param(ctx);
    if (n <= 0) return SOCP_ERR_ARG;
    int dim = 0, S = 0;
    socp_ctx_dims(ctx, &dim, &S, nullptr);
    const int nparams = socp_ctx_num_params(ctx);
    double shared_params[SOCP_MAX_NPARAMS + 2] = {0};
    if (nparams < 0 || nparams > SOCP_MAX_NPARAMS || socp_ctx_get_params(ctx, shared_params, nparams) != SOCP_OK) return SOCP_ERR_ARG;
    const int kind = opt->kind;
    if (int bad = socp::chains::validate(opt, nparams, goal, time_prev, x_prev, time_goal, x_goal)) return bad;
    // the variational Jacobian exists for models with variational equations only (modelOrder 1: the double integrator, plugins with the trait)
    if (opt->analytic_jac && socp_ctx_has_variational(ctx) != 1) return SOCP_ERR_UNSUPPORTED;
    if (P == 0) return SOCP_OK;

    // ---- where the state machines run (socp_chain_options.solver) -----------------------------------------------------------
    {
        int solver = opt->solver;
        if (const char *e = std::getenv("SOCP_CHAINS_SOLVER")) {
            if (std::strcmp(e, "device") == 0) solver = SOCP_SOLVER_DEVICE;
            else if (std::strcmp(e, "device_fast") == 0) solver = SOCP_SOLVER_DEVICE_FAST;
            else if (std::strcmp(e, "host") == 0) solver = SOCP_SOLVER_HOST;
        }
        if (solver != SOCP_SOLVER_AUTO && solver != SOCP_SOLVER_HOST && solver != SOCP_SOLVER_DEVICE && solver != SOCP_SOLVER_DEVICE_FAST) return SOCP_ERR_ARG;
        const bool automatic = solver == SOCP_SOLVER_AUTO;
        if (automatic) {
            // The host side (P state machines advanced on <= 16 threads, P n^2 doubles over PCIe per Jacobian refresh) is the
            // bottleneck from P n^2 ~ 1.6e6 up: 222 chains of n = 85, 25 of n = 253, 8192 of n = 14.  Measured at n = 14 (10^4
            // steps, 40-round budget): 8192 starts 0.53 -> 0.49 s, 65 536 starts 0.91 -> 0.70 s.  Below that the rounds are kernel
            // latency and the host engine's speculative FD rows save rounds (4096 starts: 65 rounds against 68).
            // FEW LARGE problems stay on the host too (20 P >= n): a problem is one workgroup = one CU, so a handful of 832-unknown
            // factorisations stream their matrices at the bandwidth of a handful of CUs (0.33-0.36 s for 1 ... 48 chains of n = 832)
            // while the host gives each its share of 16 threads (0.04 s for one chain, 0.28 s for 32, 0.39 s for 48); measured
            // crossovers: 44 chains at n = 832, 10 at n = 253 (scripts/probes/large_n_chains.py).
            // Round 4: the device engine has the speculative FD rows too, so small problems go there earlier -- n = 14, 10^4 steps,
            // host / device: 1024 starts 0.264 / 0.263 s, 2048: 0.361 / 0.358, 4096: 0.747 / 0.719 (65 rounds each), 16 384: 0.566 / 0.466.
            const double floor_pn2 = n <= 32 ? 4e5 : 1.6e6;
            solver = (n <= 2048 && (double)P * n * n >= floor_pn2 && 20L * P >= n) ? SOCP_SOLVER_DEVICE : SOCP_SOLVER_HOST;
            if (solver == SOCP_SOLVER_DEVICE) {
                // does the device engine's state fit?  The figure is the engine's own allocation plan (an estimate of its
                // own had drifted below what the arena really takes)
                size_t free_b = 0, total_b = 0;
                int prev = -1;
                // (chain groups -- below -- each allocate their own plan, Jacobian buffer of up to 8 GiB included: the estimate is the
                // SUM of the groups' plans, not the plan of one engine of P chains)
                const int Gest = device_groups(P, opt);
                double need = 0;
                for (int g = 0; g < Gest; g++) {
                    const int Pg = (int)((long long)P * (g + 1) / Gest) - (int)((long long)P * g / Gest);
                    need += socp_chains_device_bytes(ctx, Pg, opt, params != nullptr || kind == SOCP_CHAIN_PARAM,
                                                     kind == SOCP_CHAIN_DATA || time_goal != nullptr || x_goal != nullptr);
                }
                if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(socp_ctx_device(ctx)) != hipSuccess ||
                    hipMemGetInfo(&free_b, &total_b) != hipSuccess ||
                    need > 0.9 * ((double)free_b + socp_workspace_reusable_device_bytes(socp_ctx_device(ctx), 0)))
                    solver = SOCP_SOLVER_HOST;
                if (prev >= 0) (void)hipSetDevice(prev);
            }
            // a throughput-flavour context gets the throughput factorisation (its trajectories are rounding-level away from the
            // reference order's already); a reference-order context keeps the bit-equal solver
            if (solver == SOCP_SOLVER_DEVICE && socp_ctx_get_variant(ctx) == SOCP_VARIANT_LANE_FAST && socp_chains_fast_factor_applies(n))
                solver = SOCP_SOLVER_DEVICE_FAST;
        }
        if (solver == SOCP_SOLVER_DEVICE || solver == SOCP_SOLVER_DEVICE_FAST) {
            const int fast = (solver == SOCP_SOLVER_DEVICE_FAST && socp_chains_fast_factor_applies(n)) ? 1 : 0;
            // LARGE sweeps run as two groups of chains side by side -- two host threads, the second on a clone of the context, each
            // with its own engine call on its half of the chains: while one group's trajectory launches keep the chip busy, the other's
            // host logic, state read-back and solver kernels proceed.  Measured on 1 M starts of n = 14, 10^4 steps
            // (scripts/probes/two_groups_probe.py): 5.36 -> 5.03 s; four groups no better.  A chain only ever meets its own group, so
            // its iterates are those of one call (with a round budget the ROUND in which a chain is served can differ, as it does
            // between any two batch sizes).  SOCP_CHAINS_DEVICE_GROUPS=1|2 overrides.
            const int G = device_groups(P, opt);
            int rc = SOCP_OK;
            if (G == 1) {
                rc = socp_chains_solve_device(ctx, P, opt, Z0, params, goal, time_prev, x_prev, time_goal, x_goal, Zout, info, nfev_last, nfev_total,
                                              njev_last, solves, b_reached, param_final, fnorm, stats, fast, 0);
            } else {
                const int nodes = socp_problem_num_nodes(ctx);
                const clk_t::time_point t_groups = clk_t::now();
                std::vector<int> grc(G, SOCP_OK);
                std::vector<socp_chain_stats> gst(G);
                std::vector<socp_ctx *> gctx(G, nullptr);
                gctx[0] = ctx;
                // (group 0 runs on `ctx` itself: what it counts is taken back below when AUTO repeats every chain on the host engine)
                long long traj_before = 0, launches_before = 0;
                socp_ctx_counters(ctx, &traj_before, &launches_before);
                for (int g = 1; g < G && rc == SOCP_OK; g++) rc = socp_ctx_clone(ctx, socp_ctx_device(ctx), &gctx[g]);
                if (rc == SOCP_OK) {
                    auto work = [&](int g) {
                        const int lo = (int)((long long)P * g / G), hi = (int)((long long)P * (g + 1) / G);
                        auto at = [&](const double *a, size_t width) { return a ? a + (size_t)lo * width : nullptr; };
                        auto ati = [&](int *a) { return a ? a + lo : nullptr; };
                        auto atd = [&](double *a) { return a ? a + lo : nullptr; };
                        std::memset(&gst[g], 0, sizeof(gst[g]));
                        try {                                // (a group runs on its own thread: nothing may leave it but a status)
                        grc[g] = socp_chains_solve_device(gctx[g], hi - lo, opt, Z0 + (size_t)lo * n, at(params, nparams), at(goal, 1), at(time_prev, nodes),
                                                          at(x_prev, (size_t)nodes * S), at(time_goal, nodes), at(x_goal, (size_t)nodes * S),
                                                          Zout + (size_t)lo * n, info + lo, ati(nfev_last), ati(nfev_total), ati(njev_last), ati(solves),
                                                          atd(b_reached), atd(param_final), atd(fnorm), &gst[g], fast, g);
                        } catch (...) { grc[g] = SOCP_ERR_ARG; }   // (host memory for the chains' tables: std::bad_alloc)
                    };
                    std::vector<std::thread> th;
                    for (int g = 1; g < G; g++) th.emplace_back(work, g);
                    work(0);
                    for (std::thread &t : th) t.join();
                    for (int g = 0; g < G; g++) if (grc[g] != SOCP_OK && (rc == SOCP_OK || rc == kDeviceEngineAllocFailed)) rc = grc[g];
                }
                for (int g = 1; g < G; g++) {
                    if (!gctx[g]) continue;
                    // (a group that ran while another could not allocate: AUTO repeats ALL chains on the host engine -- below -- and the
                    // finished group's trajectories are then not part of the result: not counted)
                    if (rc != kDeviceEngineAllocFailed) {
                        long long traj = 0, launches = 0;
                        socp_ctx_counters(gctx[g], &traj, &launches);
                        socp_ctx_add_counters(ctx, traj, launches);
                    }
                    socp_ctx_destroy(gctx[g]);
                }
                if (rc == kDeviceEngineAllocFailed) {
                    // ... and neither are group 0's, which ran on `ctx`: its counters go back to where the call found them
                    long long traj_now = 0, launches_now = 0;
                    socp_ctx_counters(ctx, &traj_now, &launches_now);
                    socp_ctx_add_counters(ctx, traj_before - traj_now, launches_before - launches_now);
                }
                if (stats && rc == SOCP_OK) {
                    std::memset(stats, 0, sizeof(*stats));
                    for (int g = 0; g < G; g++) {
                        stats->rounds = std::max(stats->rounds, gst[g].rounds);
                        stats->speculative_rounds = std::max(stats->speculative_rounds, gst[g].speculative_rounds);
                        stats->jacobians_launched += gst[g].jacobians_launched;
                        stats->jacobians_from_cache += gst[g].jacobians_from_cache;
                        stats->restarts += gst[g].restarts;
                    }
                    stats->wall_ms = std::chrono::duration<double, std::milli>(clk_t::now() - t_groups).count();
                }
            }
            if (rc != kDeviceEngineAllocFailed) return rc;
            // the arena did not fit after all (another process took the memory since the estimate).  With one group nothing has run;
            // with two, one group may have run to completion while the other could not allocate -- its results are overwritten by the
            // host engine below, which repeats every chain.  AUTO made the choice, so AUTO takes the other engine; a caller who asked
            // for the device solvers gets the error.
            (void)hipGetLastError();
            if (!automatic) return SOCP_ERR_HIP;
        }
    }

    // every allocation, copy and stream below lives on the context's device, whatever the calling thread's current device
    struct DeviceGuard {
        int prev = -1;
        bool ok = false;
        explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } device_guard(socp_ctx_device(ctx));
    if (!device_guard.ok) return SOCP_ERR_HIP;

    const int nodes = socp_problem_num_nodes(ctx);          // M + 1
    if (nodes < 2) return SOCP_ERR_ARG;
    const int segs = nodes - 1;
    double shared_sw[2] = {0, 0};
    socp_ctx_get_switching_times(ctx, shared_sw);
    socp::chains::Blocks blk;
    blk.init(P, *opt, nparams, nodes, S, dim, params, shared_params, shared_sw, goal, time_prev, x_prev, time_goal, x_goal);
    const bool pp_params = blk.pp_params, pp_bound = blk.pp_bound;
    const int stride = blk.stride;
    // Goddard chains with their own parameters: when every chain is on the smooth control law for the whole call (mu2 > 0 at its
    // start and, if mu2 is the parameter being moved, at its goal), the launches may take the smooth-law kernels
    if (pp_params && socp_ctx_model_id(ctx) == SOCP_MODEL_GODDARD) {
        bool smooth = true;
        for (int p = 0; p < P && smooth; p++) {
            smooth = blk.pblock[(size_t)p * stride + 6] > 0;
            if (smooth && kind == SOCP_CHAIN_PARAM && opt->param_index == 6) smooth = goal[p] > 0 && blk.rstart[p] > 0;
        }
        socp_problem_blocks_all_smooth(ctx, smooth ? 1 : 0);
    }
    // ... a promise that ends with this call, whichever way it returns
    struct SmoothPromise {
        socp_ctx *c;
        ~SmoothPromise() { socp_problem_blocks_all_smooth(c, 0); }
    } smooth_promise{ctx};

    using clk = std::chrono::steady_clock;
    const clk::time_point t_begin = clk::now();
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = ((long)P * n * n < 200000) ? 1 : (int)std::max(1u, std::min(16u, hw ? hw : 1u));
    // one pool of host workers for the whole call (the reference starts and joins its threads on every residual call,
    // shooting.cpp:1152-1157; an engine round calls this several times): chains are dealt out in blocks, results never depend on
    // which thread advances a chain
    socp::Pool workers(nthreads);
    auto parallel_for = [&](int count, auto &&body) {
        if (nthreads <= 1 || count < 2) { for (int k = 0; k < count; k++) body(k); return; }
        const int blocks = std::min(count, 8 * nthreads), per = (count + blocks - 1) / blocks;
        workers.run(blocks, [&](int b) { for (int k = b * per, e = std::min(count, (b + 1) * per); k < e; k++) body(k); });
    };

    // ---- per-chain state --------------------------------------------------------------------------------------------
    std::vector<Chain> ch(P);
    bool alloc_ok = true;
    socp_hybr_pool *pool = socp_hybr_pool_create(P, n, opt->xtol, opt->maxfev, opt->epsfcn, 1, opt->factor, opt->analytic_jac ? 1 : 0);
    if (!pool) return SOCP_ERR_ARG;
    parallel_for(P, [&](int p) {
        Chain &c = ch[p];
        c.solver = socp_hybr_pool_get(pool, p);
        c.committed.assign(Z0 + (size_t)p * n, Z0 + (size_t)(p + 1) * n);
        c.eval_x.resize(n); c.slot_x.resize(n);
        if (kind != SOCP_CHAIN_PLAIN) { c.b = std::min(opt->step, 1.0); c.b_prec = 0; }
        blk.set(p, c.b);
        // FEW LARGE problems (fewer chains than workers, n >= 192): the workers left over go INTO the solvers, for the O(n^3)
        // factor work of a refresh (columns in SIMD lanes over a pool of their own, minpack.cpp; same numbers for any count) --
        // 16 chains of n = 832 took 1.54 s one after the other on one thread each, 4 chains 0.39 s
        if (c.solver && n >= 192 && P < nthreads) socp_hybr_set_threads(c.solver, std::max(1, nthreads / P));
        if (c.solver) socp_hybr_start(c.solver, c.committed.data(), nullptr);
    });
    auto cleanup = [&]() { for (Chain &c : ch) c.solver = nullptr; socp_hybr_pool_destroy(pool); pool = nullptr; };
    for (int p = 0; p < P; p++) if (!ch[p].solver) alloc_ok = false;
    if (!alloc_ok) { cleanup(); return SOCP_ERR_ARG; }

    // ---- groups and buffers -------------------------------------------------------------------------------------------
    // The chains can be split into G groups that take turns: while the launches of one group are in flight the host advances
    // the state machines of the other and stages its requests.  Every group has its own staging buffers and streams; a chain
    // only ever meets its own group, so no iterate depends on G (tests/test_gpu_chains.py).  What it buys is half of the host
    // work of a round (a group's own advance still precedes its launch), which only shows where the host work is a sizeable
    // part of a round: measured 6 % on the 4096-start n = 85 sweep, nothing at n = 14 (4096 and 65 536 starts: the rounds are
    // kernel time).  Hence G = 2 from 2048 chains of n >= 32 up, else 1; SOCP_CHAINS_GROUPS overrides.
    const size_t rowB = sizeof(double) * n, jacB = rowB * n, rowsLen = (size_t)(n + 1) * n, rowsB = sizeof(double) * rowsLen;
    int speculate = opt->speculate;
    if (const char *e = std::getenv("SOCP_CHAINS_SPECULATE")) speculate = std::atoi(e);
    if ((double)P * rowsB * 2 > 16e9) speculate = 0;              // slots + staging would not be "free"
    if (opt->analytic_jac) speculate = 0;                          // no finite differences on the hybrj path
    bool spec_on = speculate != 0;
    int G = (P >= 2048 && n >= 32 && !opt->analytic_jac) ? 2 : 1;  // the batched variational Jacobian keeps one scratch area per context
    if (const char *e = std::getenv("SOCP_CHAINS_GROUPS")) G = std::max(1, std::min(4, std::atoi(e)));
    if (opt->analytic_jac || G > P) G = 1;

    int num_simd = 1024;                                            // MI355X: 256 CUs x 4 SIMDs; read from the device the context lives on
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, socp_ctx_device(ctx)) == hipSuccess && prop.multiProcessorCount > 0) num_simd = 4 * prop.multiProcessorCount;
    }
    static const bool trace = std::getenv("SOCP_MULTISTART_TRACE") != nullptr;
    static const bool overlap = [] { const char *e = std::getenv("SOCP_MULTISTART_OVERLAP"); return !(e && e[0] == '0'); }();
    void *main_stream_v = nullptr;
    if (socp_ctx_synchronize(ctx) != SOCP_OK || socp_ctx_get_stream(ctx, &main_stream_v) != SOCP_OK) { cleanup(); return SOCP_ERR_HIP; }
    hipStream_t main_stream = static_cast<hipStream_t>(main_stream_v);

    struct Group {
        int lo = 0, hi = 0;                      // chains [lo, hi)
        int jchunk = 1;                          // Jacobians per pass through the pinned staging buffer (<= 64 MiB)
        int jlaunch = 1;                         // Jacobians per LAUNCH: all of a round's requests when they fit the device buffer
        Pinned hX, hF, hJx, hJf, hJ, hPF, hTF, hXF, hPJ, hTJ, hXJ, hIdx;
        Dev dX, dF, dJx, dJf, dJ, dPF, dTF, dXF, dPJ, dTJ, dXJ, dStage, dIdx;
        hipStream_t fs = nullptr, js = nullptr;  // residual-type work (and everything speculative) / launched Jacobians
        bool own_fs = false, own_js = false;
        std::vector<int> reqF, reqJ, reqJc, accepted;
        int kS = 0, j_next = 0;                  // speculative prefix of reqF; next Jacobian chunk to launch
        bool pending = false;                    // launches in flight, results not yet handed to the solvers
        long long rounds = 0;
    };
    std::vector<Group> grp(G);
    Dev dSlots;
    // device and pinned buffers of every group.  full = false: the Jacobians of a round are launched in passes of the pinned
    // staging size instead of all at once (what is left to try when the device buffers do not fit)
    auto allocate = [&](bool full) {
        bool ok = !spec_on || dSlots.alloc(rowsB * P);
        for (int g = 0; g < G && ok; g++) {
            Group &q = grp[g];
            q.lo = (int)((long)P * g / G); q.hi = (int)((long)P * (g + 1) / G);
            const size_t C = (size_t)(q.hi - q.lo);
            q.jchunk = (int)std::max<size_t>(1, std::min<size_t>(C, ((size_t)64 << 20) / jacB / G));
            // The launch is not cut to the staging size: a chunk of a few hundred problems leaves the chip half empty (measured: 93 k
            // trajectories per chunk ran at 3.7 M traj/s against 6.3 M for the whole round in one launch).  HBM is plentiful: up to
            // 8 GiB of Jacobians per group stay on the device and come back through the pinned buffer in passes.
            q.jlaunch = full ? (int)std::max<size_t>((size_t)q.jchunk, std::min<size_t>(C, ((size_t)8 << 30) / jacB / G)) : q.jchunk;
            ok = q.hX.reserve(rowB * C) && q.hF.reserve(rowB * C) && q.hJx.reserve(rowB * C) && q.hJf.reserve(rowB * C) && q.hJ.reserve(jacB * q.jchunk) &&
                 q.dX.alloc(rowB * C) && q.dF.alloc(rowB * C) && q.dJx.alloc(rowB * C) && q.dJf.alloc(rowB * C) && q.dJ.alloc(jacB * q.jlaunch) &&
                 q.hIdx.reserve(sizeof(int) * 2 * C) && q.dIdx.alloc(sizeof(int) * 2 * C);
            if (ok && pp_params) ok = q.hPF.reserve(sizeof(double) * stride * C) && q.hPJ.reserve(sizeof(double) * stride * C) &&
                                      q.dPF.alloc(sizeof(double) * stride * C) && q.dPJ.alloc(sizeof(double) * stride * C);
            if (ok && pp_bound) ok = q.hTF.reserve(sizeof(double) * nodes * C) && q.hTJ.reserve(sizeof(double) * nodes * C) &&
                                     q.hXF.reserve(sizeof(double) * nodes * S * C) && q.hXJ.reserve(sizeof(double) * nodes * S * C) &&
                                     q.dTF.alloc(sizeof(double) * nodes * C) && q.dTJ.alloc(sizeof(double) * nodes * C) &&
                                     q.dXF.alloc(sizeof(double) * nodes * S * C) && q.dXJ.alloc(sizeof(double) * nodes * S * C);
            if (ok && spec_on) ok = q.dStage.alloc(rowsB * C);
        }
        if (!ok) (void)hipGetLastError();                          // an out-of-memory hipMalloc leaves a sticky error behind
        return ok;
    };
    bool ok = allocate(true);
    if (!ok) {
        // not enough HBM for the comfortable set-up: drop what is optional -- the speculation cache (2 x P x (n+1) x n doubles)
        // and the whole-round Jacobian buffer -- and try once more before giving up.  No iterate depends on either.
        dSlots.release();
        for (Group &q : grp) { q.dStage.release(); q.dJ.release(); }
        spec_on = false; speculate = 0;
        ok = allocate(false);
    }
    for (int g = 0; g < G && ok; g++) {
        Group &q = grp[g];
        // group 0 launches its Jacobians on the context's stream (as the engine always did); residual-type work on a second one
        if (g == 0) q.js = main_stream;
        else { ok = hipStreamCreateWithFlags(&q.js, hipStreamNonBlocking) == hipSuccess; q.own_js = ok; }
        if (ok && overlap && g == 0) {
            void *aux = nullptr;                             // the context's own second stream (kept across calls)
            ok = socp_ctx_aux_stream(ctx, &aux) == SOCP_OK;
            q.fs = static_cast<hipStream_t>(aux);
        } else if (ok && overlap) { ok = socp::chains::create_residual_stream(&q.fs) == hipSuccess; q.own_fs = ok; }
        else if (ok) q.fs = q.js;
    }
    auto release_streams = [&]() {
        for (Group &q : grp) {
            if (q.own_fs && q.fs) { (void)hipStreamSynchronize(q.fs); (void)hipStreamDestroy(q.fs); }
            else if (q.fs && q.fs != q.js) (void)hipStreamSynchronize(q.fs);      // the context's second stream: left idle, not destroyed
            if (q.own_js && q.js) { (void)hipStreamSynchronize(q.js); (void)hipStreamDestroy(q.js); }
            q.fs = q.js = nullptr;
        }
    };
    if (!ok) { release_streams(); cleanup(); return SOCP_ERR_HIP; }

    bool round_limit_hit = false;
    long long spec_rows_rounds = 0, jac_from_cache = 0, jac_launched = 0, restarts = 0;
    double t_adv = 0, t_wait = 0, t_copy = 0;
    const double t_setup = ms_since(t_begin);
    int rc = SOCP_OK;

    // chain logic at the end of one Newton solve (chains_common.hpp: the bisection rules of shooting.cpp:627-660 / 724-760)
    auto solve_finished = [&](int p) {
        Chain &c = ch[p];
        std::vector<double> next;                                  // tab_param_temp for the next solve
        const double *f = socp_hybr_fvec(c.solver);
        double ss = 0;
        for (int i = 0; i < n; i++) ss += f[i] * f[i];
        c.fnorm = std::sqrt(ss);
        if (!socp::chains::after_solve(*opt, blk, p, c, socp_hybr_x(c.solver), n, socp_hybr_info(c.solver), socp_hybr_nfev(c.solver),
                                       socp_hybr_njev(c.solver), next)) return;
        socp_hybr_start(c.solver, next.data(), nullptr);
        c.slot_valid = false;                                      // another problem now: cached rows are not its rows
        c.stage_idx = -1;
        c.flag = 0;
        c.need_advance = true;
    };

    // ---- phase 1 of a group's turn: advance its chains, serve cached Jacobians, stage and enqueue its requests ------------
    auto advance_and_launch = [&](Group &q) -> int {
        const clk::time_point ta = clk::now();
        const int C = q.hi - q.lo;
        // advance every chain that received what it asked for; chains whose solve ended restart (or retire) and advance again,
        // so that afterwards every live chain has exactly one pending request.  Jacobian requests whose point is cached are
        // served here, without a round of trajectories.
        for (;;) {
            parallel_for(C, [&](int k) {
                const int p = q.lo + k;
                Chain &c = ch[p];
                while (!c.finished && c.need_advance) {
                    c.req = socp_hybr_advance(c.solver, c.flag, &c.xin, &c.xout);
                    c.flag = 0;
                    c.need_advance = false;
                    if (c.req == SOCP_REQ_DONE) solve_finished(p);          // may set need_advance again (next homotopy step)
                }
            });
            if (!spec_on) break;
            // the rows evaluated in the group's last round belong to a chain's slot if that point is now its solver's x
            q.accepted.clear(); q.reqJc.clear();
            for (int p = q.lo; p < q.hi; p++) {
                Chain &c = ch[p];
                if (c.finished || c.stage_idx < 0) continue;
                if (std::memcmp(socp_hybr_x(c.solver), c.eval_x.data(), rowB) == 0) q.accepted.push_back(p);
            }
            if (!q.accepted.empty()) {
                int *idx = q.hIdx.i();
                for (size_t k = 0; k < q.accepted.size(); k++) { idx[k] = ch[q.accepted[k]].stage_idx; idx[C + k] = q.accepted[k]; }
                if (hipMemcpyAsync(q.dIdx.p, idx, sizeof(int) * 2 * (size_t)C, hipMemcpyHostToDevice, q.fs) != hipSuccess ||
                    !copy_blocks(q.fs, q.dStage.d(), q.dIdx.i(), dSlots.d(), q.dIdx.i() + C, (int)q.accepted.size(), (int)rowsLen) ||
                    hipStreamSynchronize(q.fs) != hipSuccess) return SOCP_ERR_HIP;
                for (int p : q.accepted) { ch[p].slot_x = ch[p].eval_x; ch[p].slot_valid = true; }
            }
            for (int p = q.lo; p < q.hi; p++) ch[p].stage_idx = -1;      // the staging area is about to be reused
            for (int p = q.lo; p < q.hi; p++) {
                Chain &c = ch[p];
                if (!c.finished && c.req == SOCP_REQ_JAC && c.slot_valid && std::memcmp(c.xin, c.slot_x.data(), rowB) == 0) q.reqJc.push_back(p);
            }
            if (q.reqJc.empty()) break;
            // Jacobians from cached rows: slot -> staging (gather), fd_diff, read back, scatter; then those chains advance again
            for (size_t j0 = 0; j0 < q.reqJc.size(); j0 += q.jchunk) {
                const int kc = (int)std::min<size_t>(q.jchunk, q.reqJc.size() - j0);
                int *idx = q.hIdx.i();
                for (int k = 0; k < kc; k++) {
                    idx[k] = q.reqJc[j0 + k]; idx[C + k] = k;
                    std::memcpy(q.hJx.d() + (size_t)k * n, ch[q.reqJc[j0 + k]].xin, rowB);
                }
                if (hipMemcpyAsync(q.dIdx.p, idx, sizeof(int) * 2 * (size_t)C, hipMemcpyHostToDevice, q.fs) != hipSuccess ||
                    hipMemcpyAsync(q.dJx.p, q.hJx.p, rowB * kc, hipMemcpyHostToDevice, q.fs) != hipSuccess ||
                    !copy_blocks(q.fs, dSlots.d(), q.dIdx.i(), q.dStage.d(), q.dIdx.i() + C, kc, (int)rowsLen)) return SOCP_ERR_HIP;
                socp_ctx_set_stream(ctx, q.fs, 0);
                const int r = socp_fd_diff_dev(ctx, kc, q.dJx.d(), opt->epsfcn, q.dStage.d(), q.dJ.d());
                socp_ctx_set_stream(ctx, main_stream, 0);
                if (r != SOCP_OK) return r;
                if (hipMemcpyAsync(q.hJ.p, q.dJ.p, jacB * kc, hipMemcpyDeviceToHost, q.fs) != hipSuccess ||
                    hipStreamSynchronize(q.fs) != hipSuccess) return SOCP_ERR_HIP;
                parallel_for(kc, [&](int k) { std::memcpy(ch[q.reqJc[j0 + k]].xout, q.hJ.d() + (size_t)k * n * n, jacB); });
                for (int k = 0; k < kc; k++) ch[q.reqJc[j0 + k]].need_advance = true;
                jac_from_cache += kc;
            }
        }
        // gather the requests in chain order (keeps the batches deterministic)
        q.reqF.clear(); q.reqJ.clear();
        for (int p = q.lo; p < q.hi; p++) {
            Chain &c = ch[p];
            if (c.finished) continue;
            if (c.req == SOCP_REQ_FVEC) q.reqF.push_back(p);
            else if (c.req == SOCP_REQ_JAC) q.reqJ.push_back(p);
        }
        q.pending = false;
        if (q.reqF.empty() && q.reqJ.empty()) { t_adv += ms_since(ta); return SOCP_OK; }
        if (opt->max_rounds > 0 && q.rounds >= opt->max_rounds) {
            // round budget spent: the chains still solving stop the way a negative callback return stops hybrd
            for (int p : q.reqF) { ch[p].flag = SOCP_INFO_ROUND_LIMIT; ch[p].need_advance = true; }
            for (int p : q.reqJ) { ch[p].flag = SOCP_INFO_ROUND_LIMIT; ch[p].need_advance = true; }
            round_limit_hit = true;
            parallel_for(C, [&](int k) {
                Chain &c = ch[q.lo + k];
                if (!c.finished && c.need_advance) { c.req = socp_hybr_advance(c.solver, c.flag, &c.xin, &c.xout); c.need_advance = false; if (c.req == SOCP_REQ_DONE) solve_finished(q.lo + k); }
            });
            q.reqF.clear(); q.reqJ.clear();
            t_adv += ms_since(ta);
            return SOCP_OK;
        }
        q.rounds++;
        const int kF = (int)q.reqF.size(), kJ = (int)q.reqJ.size();
        // how many of the residual requests are evaluated as whole FD batches: all of them when they fit the group's share of
        // the idle SIMDs (one wave per SIMD keeps the round at one trajectory latency), else none -- a round that is part FD
        // batches, part plain residuals would be two launches on one stream, i.e. two latencies.  speculate = 1 forces all.
        int kS = 0;
        if (spec_on && kF) {
            if (speculate > 0) kS = kF;
            else {
                const long lanes = (num_simd / G - ((long)kJ * n * segs + 63) / 64) * 64;
                kS = ((long)kF * (n + 1) * segs <= lanes) ? kF : 0;
            }
        }
        q.kS = kS;
        if (trace) std::fprintf(stderr, "[socp_chains] group %d round %lld: %d residual requests (%d as FD batches), %d Jacobian requests\n",
                                (int)(&q - grp.data()), q.rounds, kF, kS, kJ);
        for (int k = 0; k < kF; k++) {
            const int p = q.reqF[k];
            std::memcpy(q.hX.d() + (size_t)k * n, ch[p].xin, rowB);
            blk.stage(p, k, q.hPF.d(), q.hTF.d(), q.hXF.d());
            if (k < kS) { std::memcpy(ch[p].eval_x.data(), ch[p].xin, rowB); ch[p].stage_idx = k; }
        }
        for (int k = 0; k < kJ; k++) {
            const int p = q.reqJ[k];
            std::memcpy(q.hJx.d() + (size_t)k * n, ch[p].xin, rowB);
            std::memcpy(q.hJf.d() + (size_t)k * n, socp_hybr_fvec(ch[p].solver), rowB);
            blk.stage(p, k, q.hPJ.d(), q.hTJ.d(), q.hXJ.d());
        }
        t_adv += ms_since(ta);
        int r = SOCP_OK;
        // residual-type launches (stream fs): all requests through the FD-row kernel, or all through the residual kernel
        if (kF) {
            bool h2d = hipMemcpyAsync(q.dX.p, q.hX.p, rowB * kF, hipMemcpyHostToDevice, q.fs) == hipSuccess;
            if (h2d && pp_params) h2d = hipMemcpyAsync(q.dPF.p, q.hPF.p, sizeof(double) * stride * kF, hipMemcpyHostToDevice, q.fs) == hipSuccess;
            if (h2d && pp_bound) h2d = hipMemcpyAsync(q.dTF.p, q.hTF.p, sizeof(double) * nodes * kF, hipMemcpyHostToDevice, q.fs) == hipSuccess &&
                                       hipMemcpyAsync(q.dXF.p, q.hXF.p, sizeof(double) * nodes * S * kF, hipMemcpyHostToDevice, q.fs) == hipSuccess;
            if (!h2d) return SOCP_ERR_HIP;
            socp_ctx_set_stream(ctx, q.fs, 0);
            if (kS) {
                socp_problem_set_blocks_dev(ctx, pp_params ? q.dPF.d() : nullptr, stride, pp_bound ? q.dTF.d() : nullptr, pp_bound ? q.dXF.d() : nullptr);
                r = socp_fd_rows_dev(ctx, kS, q.dX.d(), opt->epsfcn, q.dStage.d());
                spec_rows_rounds++;
            }
            if (r == SOCP_OK && kF > kS) {
                socp_problem_set_blocks_dev(ctx, pp_params ? q.dPF.d() + (size_t)kS * stride : nullptr, stride,
                                            pp_bound ? q.dTF.d() + (size_t)kS * nodes : nullptr, pp_bound ? q.dXF.d() + (size_t)kS * nodes * S : nullptr);
                r = socp_residual_batch_dev(ctx, kF - kS, q.dX.d() + (size_t)kS * n, q.dF.d() + (size_t)kS * n);
            }
            // F of the speculative requests is row 0 of their (n+1) x n block
            bool d2h = r == SOCP_OK;
            if (d2h && kS) d2h = hipMemcpy2DAsync(q.hF.p, rowB, q.dStage.p, rowsB, rowB, (size_t)kS, hipMemcpyDeviceToHost, q.fs) == hipSuccess;
            if (d2h && kF > kS) d2h = hipMemcpyAsync(q.hF.d() + (size_t)kS * n, q.dF.d() + (size_t)kS * n, rowB * (kF - kS), hipMemcpyDeviceToHost, q.fs) == hipSuccess;
            socp_ctx_set_stream(ctx, main_stream, 0);
            if (r != SOCP_OK) return r;
            if (!d2h) return SOCP_ERR_HIP;
        }
        q.j_next = 0;
        if (kJ) {
            bool h2d = hipMemcpyAsync(q.dJx.p, q.hJx.p, rowB * kJ, hipMemcpyHostToDevice, q.js) == hipSuccess &&
                       hipMemcpyAsync(q.dJf.p, q.hJf.p, rowB * kJ, hipMemcpyHostToDevice, q.js) == hipSuccess;
            if (h2d && pp_params) h2d = hipMemcpyAsync(q.dPJ.p, q.hPJ.p, sizeof(double) * stride * kJ, hipMemcpyHostToDevice, q.js) == hipSuccess;
            if (h2d && pp_bound) h2d = hipMemcpyAsync(q.dTJ.p, q.hTJ.p, sizeof(double) * nodes * kJ, hipMemcpyHostToDevice, q.js) == hipSuccess &&
                                       hipMemcpyAsync(q.dXJ.p, q.hXJ.p, sizeof(double) * nodes * S * kJ, hipMemcpyHostToDevice, q.js) == hipSuccess;
            if (!h2d) return SOCP_ERR_HIP;
            jac_launched += kJ;
        }
        q.pending = true;
        return SOCP_OK;
    };

    // one launch of a group's Jacobian requests (all of them when they fit the device buffer): enqueue on its Jacobian stream
    auto launch_jac_chunk = [&](Group &q) -> int {
        const int kJ = (int)q.reqJ.size(), j0 = q.j_next;
        if (j0 >= kJ) return SOCP_OK;
        const int kc = std::min(q.jlaunch, kJ - j0);
        socp_ctx_set_stream(ctx, q.js, 0);
        socp_problem_set_blocks_dev(ctx, pp_params ? q.dPJ.d() + (size_t)j0 * stride : nullptr, stride,
                                    pp_bound ? q.dTJ.d() + (size_t)j0 * nodes : nullptr, pp_bound ? q.dXJ.d() + (size_t)j0 * nodes * S : nullptr);
        const int r = opt->analytic_jac
                          ? socp_var_jacobian_multi_dev(ctx, kc, q.dJx.d() + (size_t)j0 * n, q.dJ.d())
                          : socp_fd_jacobian_multi_dev(ctx, kc, q.dJx.d() + (size_t)j0 * n, q.dJf.d() + (size_t)j0 * n, opt->epsfcn, q.dJ.d(), opt->dedup);
        socp_ctx_set_stream(ctx, main_stream, 0);
        return r;
    };

    // ---- phase 2 of a group's turn: wait for its launches and hand the results to the solvers -------------------------------
    auto collect = [&](Group &q) -> int {
        if (!q.pending) return SOCP_OK;
        const clk::time_point tw = clk::now();
        const int kF = (int)q.reqF.size(), kJ = (int)q.reqJ.size();
        if (kF) {
            if (hipStreamSynchronize(q.fs) != hipSuccess) return SOCP_ERR_HIP;
            for (int k = 0; k < kF; k++) std::memcpy(ch[q.reqF[k]].xout, q.hF.d() + (size_t)k * n, rowB);
        }
        while (q.j_next < kJ) {
            const int j0 = q.j_next, kl = std::min(q.jlaunch, kJ - j0);
            if (hipStreamSynchronize(q.js) != hipSuccess) return SOCP_ERR_HIP;
            t_wait += ms_since(tw);
            const clk::time_point tc = clk::now();
            // hundreds of MB per round at n ~ 100: back through the pinned buffer in passes, and from there into the solvers'
            // own buffers spread over the host threads
            for (int s0 = 0; s0 < kl; s0 += q.jchunk) {
                const int kc = std::min(q.jchunk, kl - s0);
                if (hipMemcpy(q.hJ.p, q.dJ.d() + (size_t)s0 * n * n, jacB * kc, hipMemcpyDeviceToHost) != hipSuccess) return SOCP_ERR_HIP;
                parallel_for(kc, [&](int k) { std::memcpy(ch[q.reqJ[j0 + s0 + k]].xout, q.hJ.d() + (size_t)k * n * n, jacB); });
            }
            t_copy += ms_since(tc);
            q.j_next += kl;
            if (q.j_next < kJ) { const int r = launch_jac_chunk(q); if (r != SOCP_OK) return r; }
        }
        if (!kJ) t_wait += ms_since(tw);
        for (int p : q.reqF) ch[p].need_advance = true;
        for (int p : q.reqJ) ch[p].need_advance = true;
        q.pending = false;
        return SOCP_OK;
    };

    // ---- the turns: a group's results are collected only after the other groups have been advanced and launched --------------
    for (Group &q : grp) {
        if ((rc = advance_and_launch(q)) != SOCP_OK) break;
        if (q.pending && (rc = launch_jac_chunk(q)) != SOCP_OK) break;
    }
    while (rc == SOCP_OK) {
        bool any = false;
        for (Group &q : grp) {
            if (!q.pending) continue;
            any = true;
            if ((rc = collect(q)) != SOCP_OK) break;
            if ((rc = advance_and_launch(q)) != SOCP_OK) break;
            if (q.pending && (rc = launch_jac_chunk(q)) != SOCP_OK) break;
        }
        if (!any) break;
    }
    long long rounds = 0;
    for (const Group &q : grp) rounds = std::max(rounds, q.rounds);
    socp_problem_set_blocks_dev(ctx, nullptr, 0, nullptr, nullptr);
    if (rc == SOCP_OK) {
        for (int p = 0; p < P; p++) {
            const Chain &c = ch[p];
            std::memcpy(Zout + (size_t)p * n, c.committed.data(), rowB);
            info[p] = c.info;
            if (nfev_last) nfev_last[p] = c.nfev_last;
            if (nfev_total) nfev_total[p] = c.nfev_total;
            if (njev_last) njev_last[p] = c.njev_last;
            if (solves) solves[p] = c.solves;
            if (b_reached) b_reached[p] = kind == SOCP_CHAIN_PLAIN ? 1.0 : (c.info == 1 ? c.b : c.b_prec);
            if (param_final) param_final[p] = kind == SOCP_CHAIN_PARAM ? blk.pblock[(size_t)p * stride + opt->param_index] : 0.0;
            if (fnorm) fnorm[p] = c.fnorm;
        }
    }
    for (int p = 0; p < P; p++) restarts += std::max(0, ch[p].solves - 1);
    if (trace && round_limit_hit) std::fprintf(stderr, "[socp_chains] round limit %d reached: the chains still solving were stopped\n", opt->max_rounds);
    if (trace)
        std::fprintf(stderr, "[socp_chains] set-up %.1f ms, host advance + staging %.1f ms, waiting for launches %.1f ms, Jacobian scatter %.1f ms, "
                             "total %.1f ms; %d group(s), %lld rounds, %lld Jacobians launched, %lld from cached rows, %lld solver restarts\n",
                     t_setup, t_adv, t_wait, t_copy, ms_since(t_begin), G, rounds, jac_launched, jac_from_cache, restarts);
    if (stats) {
        stats->rounds = rounds; stats->jacobians_launched = jac_launched; stats->jacobians_from_cache = jac_from_cache;
        stats->speculative_rounds = spec_rows_rounds; stats->restarts = restarts; stats->wall_ms = ms_since(t_begin);
    }
    release_streams();
    cleanup();
    return rc;
}

// The multi-start sweep is the chain engine with one plain Newton solve per chain.
extern "C" int socp_multistart_solve(socp_ctx *ctx, int P, const double *Z0, double xtol, int maxfev, double epsfcn,
                                     double factor, int dedup, double *Zout, int *info, int *nfev, double *fnorm,
                                     long long *rounds_out)
{
    socp_chain_options opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.kind = SOCP_CHAIN_PLAIN;
    opt.xtol = xtol; opt.maxfev = maxfev; opt.epsfcn = epsfcn; opt.factor = factor; opt.dedup = dedup;
    opt.speculate = -1;
    socp_chain_stats st;
    std::memset(&st, 0, sizeof(st));
    const int rc = socp_chains_solve(ctx, P, &opt, Z0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, Zout, info, nfev,
                                     nullptr, nullptr, nullptr, nullptr, fnorm, &st);
    if (rounds_out) *rounds_out = st.rounds;
    return rc;
}
