// batchsolve_dev.cpp -- the lock-step engine with the Newton solvers ON THE DEVICE (VERDICT r2 "Next" #2).
//
// batchsolve.cpp advances P resumable hybrd state machines on the host: for sweeps of n = 85 .. 253 that is 75-90 % of the wall
// time (P factorisations per Jacobian refresh on <= 16 host threads) plus P n^2 doubles over PCIe per refresh (237 MB at
// n = 85, 1 GB at n = 253).  Here the state machines live in HBM (solver_dev.hpp: one workgroup per problem, MINPACK's
// per-column / per-row operation order kept, so every chain follows the host engine's iterates bit for bit):
//
//   per round:  advance kernel (all chains that received what they asked for)
//               -> the advanced chains' status back to the host (24 B each: which request is pending, counters)
//               -> host: request lists in chain order, homotopy logic of chains whose solve ended (restart / retire)
//               -> gather the evaluation points on the device, ONE residual launch, ONE Jacobian launch (two streams),
//                  scatter the results into the problems' workspaces (the Jacobian transposed into the solver's row-major
//                  matrix on the way).  No Jacobian, factor or iterate ever crosses PCIe.
//
// The chain logic (continuation homotopy, bisection, per-chain parameter / boundary blocks) is the host engine's
// (chains_common.hpp).  Speculative FD rows as in the host engine (round 4): when a round's residual requests fit the idle SIMDs
// as whole forward-difference batches they are evaluated that way, the rows of a chain's last accepted point stay in HBM, and a
// Jacobian asked for at that point is formed from them without trajectories -- no iterate changes.  Not here: chain groups.
#include "../../include/socp_hip.h"
#include "../../include/socp_solver.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "chains_common.hpp"
#include "solver_launch.hpp"
#include "staging.hpp"

namespace {

using socp::devsolver::PoolDev;
using socp::devsolver::State;
using socp::devsolver::Status;

// What is kept between calls (include/socp_solver.h "the device engine's workspace"): per device one block of device memory and
// one of pinned host memory.  Never destroyed (a static destructor would call into a HIP runtime that may be gone already).
struct KeptBlock {
    void *p = nullptr;
    size_t cap = 0;
    bool busy = false;
};
constexpr int kSlotsPerDevice = 4;                   // concurrent engine calls per device that find a kept block (chain groups)
struct Workspaces {
    std::mutex m;
    std::map<int, KeptBlock> dev, host;              // key: device * kSlotsPerDevice + slot
};
Workspaces &workspaces()
{
    static Workspaces *w = new Workspaces;
    return *w;
}
bool keep_workspaces()
{
    static const bool on = [] { const char *e = std::getenv("SOCP_WORKSPACE_CACHE"); return !(e && e[0] == '0'); }();
    return on;
}
hipError_t raw_alloc_once(void **p, size_t bytes, bool host) { return host ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes); }
void raw_free(void *p, bool host) { if (p) (void)(host ? hipHostFree(p) : hipFree(p)); }
// the idle kept blocks of `device` (every slot nobody is using) of ONE kind -- device memory or pinned host memory, the kind whose
// allocation has just failed: freed.  Returns the bytes given back.  (Freeing the other kind gives the failed allocation nothing: a
// pinned block returns no device memory, and this call's own idle pinned block would be re-pinned a moment later -- GBs, ~1 s.)
double release_idle_blocks(int device, bool host)
{
    Workspaces &w = workspaces();
    std::lock_guard<std::mutex> lock(w.m);
    double freed = 0;
    for (auto &kv : host ? w.host : w.dev) {
        KeptBlock &b = kv.second;
        if (kv.first / kSlotsPerDevice != device || b.busy || !b.p) continue;
        raw_free(b.p, host);
        freed += (double)b.cap;
        b.p = nullptr; b.cap = 0;
    }
    return freed;
}
// An allocation for the engine on the CURRENT device.  A call can only take the kept block of its own slot (Arena::alloc), so the
// idle blocks of the device's OTHER slots -- tens of GB after a two-group sweep of millions of starts -- are memory it cannot use but
// that stands in its way: when the allocation fails the idle blocks of the SAME kind are released and it is tried once more.
hipError_t raw_alloc(void **p, size_t bytes, bool host, int device)
{
    hipError_t e = raw_alloc_once(p, bytes, host);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    if (release_idle_blocks(device, host) <= 0) return e;
    e = raw_alloc_once(p, bytes, host);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}

// The engine's buffers: ONE device allocation and ONE pinned host allocation, carved into aligned pieces (thirty-odd separate
// hipMalloc / hipHostMalloc calls were 5 of the 6-8 ms a call spent before its first launch).  The allocation comes from -- and goes
// back to -- the kept block of its device when that is free.
struct Arena {
    char *base = nullptr;
    size_t used = 0, cap = 0;
    bool pinned = false, kept = false;
    int key = 0;
    static size_t pad(size_t bytes) { return (bytes + 255) / 256 * 256; }
    size_t plan(size_t bytes) { const size_t at = used; used += pad(bytes ? bytes : 8); return at; }
    bool alloc(bool host, int dev, int slot)
    {
        pinned = host;
        key = dev * kSlotsPerDevice + (slot < 0 ? 0 : slot % kSlotsPerDevice);
        cap = used ? used : 256;
        void *p = nullptr;
        if (keep_workspaces()) {
            Workspaces &w = workspaces();
            KeptBlock *b = nullptr;
            {
                std::lock_guard<std::mutex> lock(w.m);
                KeptBlock &kb = (host ? w.host : w.dev)[key];            // (std::map: the reference stays valid)
                if (!kb.busy) { kb.busy = true; b = &kb; }
            }
            if (b) {
                // the block is this call's now; growing it happens OUTSIDE the lock (tens of GB take up to a second, and the threads
                // of socp_sweep_solve -- one per device -- all arrive here at once)
                kept = true;
                if (b->cap < cap) {
                    // the new block first (the old one is cleared in the background once freed; an allocation right behind a large
                    // free waits for that), the other order only when both do not fit
                    void *const old = b->p;                  // (nobody else touches a busy slot; its fields change under the lock only,
                                                             // socp_workspace_cached_bytes reads them)
                    bool ok = raw_alloc_once(&p, cap, host) == hipSuccess;
                    if (!ok) (void)hipGetLastError();
                    raw_free(old, host);
                    if (!ok) { p = nullptr; ok = raw_alloc(&p, cap, host, dev) == hipSuccess; }        // (with the other slots' idle blocks released if need be)
                    std::lock_guard<std::mutex> lock(w.m);
                    b->p = ok ? p : nullptr;
                    b->cap = ok ? cap : 0;
                    if (!ok) return false;                   // (~Arena hands the empty slot back)
                }
                base = static_cast<char *>(b->p);
                return true;
            }
        }
        const hipError_t e = raw_alloc(&p, cap, host, dev);
        base = static_cast<char *>(p);
        return e == hipSuccess;
    }
    ~Arena()
    {
        if (kept) {
            Workspaces &w = workspaces();
            std::lock_guard<std::mutex> lock(w.m);
            (pinned ? w.host : w.dev)[key].busy = false;
        } else {
            raw_free(base, pinned);
        }
    }
};
struct Piece {
    size_t at = 0;
    void *p = nullptr;
    void plan(Arena &a, size_t bytes) { at = a.plan(bytes); }
    void bind(const Arena &a) { p = a.base + at; }
    double *d() const { return static_cast<double *>(p); }
    int *i() const { return static_cast<int *>(p); }
};
using Dev = Piece;
// A pinned staging buffer: its memory is reached through socp::staging::Staged only (staging.hpp) -- host() when the host is about
// to read or write it, async_source() / async_target() when an asynchronous operation is about to be enqueued with it -- so that "the
// host rewrote a list before the copy of its last contents had run" (round 4, commit e52cc58) cannot be written down again.
struct HipBackend {
    using stream_type = hipStream_t;
    static bool synchronize(hipStream_t s) { return hipStreamSynchronize(s) == hipSuccess; }
};
using StreamClock = socp::staging::StreamClock<HipBackend>;
struct Pinned {
    Piece piece;
    socp::staging::Staged<HipBackend> st;
    explicit Pinned(const char *name) : st(name) {}
    void bind(const Arena &a) { piece.bind(a); st.set_memory(piece.p); }
    void *host() { return st.host(); }
    double *hd() { return st.host_as<double>(); }
    int *hi() { return st.host_as<int>(); }
    const void *source(StreamClock &c) { return st.async_source(c); }    // an asynchronous READ of the buffer is being enqueued on c
    void *target(StreamClock &c) { return st.async_target(c); }          // ... an asynchronous WRITE
};

// Chain lists into ascending order.  A round's request lists are read off lists that were ascending themselves, so they are a few
// ascending runs (one per inner pass): merged in O(n) instead of sorted (4 M starts: two sorts per round were 1 s of a 22 s sweep).
void sort_runs(std::vector<int> &v)
{
    size_t runs = 1;
    for (size_t i = 1; i < v.size(); i++) runs += v[i] < v[i - 1];
    if (runs == 1) return;
    if (runs > 4) { std::sort(v.begin(), v.end()); return; }
    auto mid = std::is_sorted_until(v.begin(), v.end());
    while (mid != v.end()) {
        auto next = std::is_sorted_until(mid, v.end());
        std::inplace_merge(v.begin(), mid, next);
        mid = next;
    }
}

}  // namespace

constexpr int kDeviceEngineAllocFailed = -1000;      // to socp_chains_solve (batchsolve.cpp): the arenas did not fit, nothing has run

namespace {
// The device engine's allocation plan for P chains: per chain the solver workspace, its state, five index lists, six rows of n and
// (per-chain parameters / boundary data) two staged copies of those; the Jacobians of a round in one buffer of at most 8 GiB.
struct EnginePlan {
    size_t ws_stride = 0, rowB = 0, jacB = 0, intsB = 0, parB = 0, timeB = 0, nodeB = 0;
    int jlaunch = 1;
    EnginePlan(int n, int P, int nodes, int S, int stride)
    {
        ws_stride = (size_t)socp::devsolver::ws_doubles(n, socp::devsolver::ld_for(n));
        rowB = sizeof(double) * (size_t)n; jacB = rowB * n; intsB = sizeof(int) * (size_t)P;
        parB = sizeof(double) * (size_t)stride * P; timeB = sizeof(double) * (size_t)nodes * P; nodeB = timeB * S;
        // Jacobians of a round: one launch whenever they fit 8 GiB (batchsolve.cpp has the measurement), else passes
        jlaunch = (int)std::max<size_t>(1, std::min<size_t>((size_t)P, ((size_t)8 << 30) / jacB));
    }
    double device_bytes(int P, bool pp_params, bool pp_bound) const
    {
        double b = (double)P * (sizeof(double) * (double)ws_stride + sizeof(State) + sizeof(Status) + 6.0 * rowB) + 6.0 * intsB + (double)jacB * jlaunch;
        if (pp_params) b += 2.0 * parB;
        if (pp_bound) b += 2.0 * (timeB + nodeB);
        b += 2.0 * (double)P * rowB * (rowB / sizeof(double) + 1);          // (upper bound: a cache slot per chain and as much staging)
        return b + 64.0 * 256;                                               // every piece is rounded up to 256 B
    }
};
}  // namespace

double socp_chains_device_bytes(const socp_ctx *ctx, int P, const socp_chain_options *, bool per_chain_params, bool per_chain_bounds)
{
    const int n = socp_problem_num_param(ctx), nodes = socp_problem_num_nodes(ctx), nparams = socp_ctx_num_params(ctx);
    int S = 0;
    socp_ctx_dims(ctx, nullptr, &S, nullptr);
    return EnginePlan(n, P, nodes, S, nparams + 2).device_bytes(P, per_chain_params, per_chain_bounds);
}

extern "C" double socp_workspace_cached_bytes(int device)
{
    Workspaces &w = workspaces();
    std::lock_guard<std::mutex> lock(w.m);
    double total = 0;
    for (auto *table : {&w.dev, &w.host})
        for (auto &kv : *table)
            if (device < 0 || kv.first / kSlotsPerDevice == device) total += (double)kv.second.cap;
    return total;
}

extern "C" double socp_workspace_release(int device)
{
    Workspaces &w = workspaces();
    std::lock_guard<std::mutex> lock(w.m);
    double freed = 0;
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int host = 0; host < 2; host++)
        for (auto &kv : host ? w.host : w.dev) {
            KeptBlock &b = kv.second;
            if ((device >= 0 && kv.first / kSlotsPerDevice != device) || b.busy || !b.p) continue;
            if (hipSetDevice(kv.first / kSlotsPerDevice) != hipSuccess) { (void)hipGetLastError(); continue; }
            raw_free(b.p, host != 0);
            freed += (double)b.cap;
            b.p = nullptr; b.cap = 0;
        }
    if (prev >= 0) (void)hipSetDevice(prev);
    return freed;
}

// Device memory a call with workspace slot `slot` could take on `device` beyond what hipMemGetInfo calls free: the kept block of
// ITS slot when not in use (Arena::alloc takes no other), plus -- because a failed allocation releases them (raw_alloc) -- the idle
// blocks of the device's other slots.  Both are memory the call can really get; blocks in use by a running call are not.
double socp_workspace_reusable_device_bytes(int device, int slot)
{
    Workspaces &w = workspaces();
    std::lock_guard<std::mutex> lock(w.m);
    (void)slot;                                      // (every idle block counts, whichever slot: the retry after a release reaches them all)
    double total = 0;
    for (auto &kv : w.dev)
        if (kv.first / kSlotsPerDevice == device && !kv.second.busy) total += (double)kv.second.cap;
    return total;
}

bool socp_chains_fast_factor_applies(int n) { return socp::devsolver::fast_factor_applies(n); }

// Same contract as socp_chains_solve_ex (include/socp_solver.h); called from there when the device solvers are chosen.
// fast_factor: the Jacobian refreshes go through the throughput factorisation (kernels_factor_fast.hip).  workspace_slot: which of the
// device's kept blocks this call may take (chain groups run side by side: batchsolve.cpp).
int socp_chains_solve_device(socp_ctx *ctx, int P, const socp_chain_options *opt, const double *Z0, const double *params,
                             const double *goal, const double *time_prev, const double *x_prev, const double *time_goal,
                             const double *x_goal, double *Zout, int *info, int *nfev_last, int *nfev_total, int *njev_last,
                             int *solves, double *b_reached, double *param_final, double *fnorm, socp_chain_stats *stats, int fast_factor, int workspace_slot)
{
    using clk = std::chrono::steady_clock;
    const clk::time_point t_begin = clk::now();
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    static const bool trace = std::getenv("SOCP_MULTISTART_TRACE") != nullptr;

    const int n = socp_problem_num_param(ctx), nodes = socp_problem_num_nodes(ctx);
    int dim = 0, S = 0;
    socp_ctx_dims(ctx, &dim, &S, nullptr);
    const int nparams = socp_ctx_num_params(ctx), kind = opt->kind;
    double shared_params[SOCP_MAX_NPARAMS + 2] = {0}, shared_sw[2] = {0, 0};
    if (socp_ctx_get_params(ctx, shared_params, nparams) != SOCP_OK) return SOCP_ERR_ARG;
    socp_ctx_get_switching_times(ctx, shared_sw);

    struct DeviceGuard {
        int prev = -1;
        bool ok = false;
        explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } device_guard(socp_ctx_device(ctx));
    if (!device_guard.ok) return SOCP_ERR_HIP;

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

    std::vector<socp::chains::ChainCore> ch(P);
    for (int p = 0; p < P; p++) {
        ch[p].committed.assign(Z0 + (size_t)p * n, Z0 + (size_t)(p + 1) * n);
        if (kind != SOCP_CHAIN_PLAIN) { ch[p].b = std::min(opt->step, 1.0); ch[p].b_prec = 0; }
        blk.set(p, ch[p].b);
    }

    // ---- the pool of device solvers and the round's buffers -----------------------------------------------------------
    PoolDev pool;
    pool.cfg.n = n; pool.cfg.ld = socp::devsolver::ld_for(n); pool.cfg.maxfev = opt->maxfev; pool.cfg.mode = 1;
    pool.cfg.analytic = opt->analytic_jac ? 1 : 0; pool.cfg.xtol = opt->xtol; pool.cfg.epsfcn = opt->epsfcn; pool.cfg.factor = opt->factor;
    // the throughput flavour of LARGE problems also keeps Q as factorised between refreshes (solver_dev.hpp: lazy_capacity): it saves
    // a quarter of a trial step's memory traffic -- measured (profiles/r04_lazy_q_ab.txt): n = 253 -9 ... -11 % of a sweep's wall
    // time, n = 85 / 127 nothing (their Q is small: r1mpyq is not what they wait for).  SOCP_SOLVER_LAZY_Q=0 / 1: never / always.
    if (fast_factor) {
        const char *e = std::getenv("SOCP_SOLVER_LAZY_Q");
        pool.cfg.lazy_q = e ? (e[0] == '0' ? 0 : 1) : (n >= 192 ? 1 : 0);
        // ... and forms the back substitution's row sums in parallel (solver_dev.hpp: dogleg).  SOCP_SOLVER_FAST_SUMS=0: the serial chains.
        const char *f = std::getenv("SOCP_SOLVER_FAST_SUMS");
        // (every size: KD chains and M = 6 sweeps, n = 85, -3 ... -4 %, M = 9, n = 127, -11 %; profiles/r05_fast_sums_ab.txt.  An earlier
        // A/B of this round said +10 % at n = 85 -- with the row rings still in those launches, which was what cost the time)
        pool.cfg.fast_sums = (f && f[0] == '0') ? 0 : 1;
        // ... and takes the predicted reduction of a Gauss-Newton step as exact instead of forming qtf + R p (solver_dev.hpp: after_trial).
        // SOCP_SOLVER_GN_SHORTCUT=0: MINPACK's product.
        const char *gs = std::getenv("SOCP_SOLVER_GN_SHORTCUT");
        pool.cfg.gn_shortcut = (gs && gs[0] == '0') ? 0 : 1;
    }
    const EnginePlan plan_sizes(n, P, nodes, S, stride);
    pool.ws_stride = (long)plan_sizes.ws_stride;
    pool.P = P;
    const size_t rowB = plan_sizes.rowB, jacB = plan_sizes.jacB;
    const int jlaunch = plan_sizes.jlaunch;

    void *main_stream_v = nullptr;
    if (socp_ctx_synchronize(ctx) != SOCP_OK || socp_ctx_get_stream(ctx, &main_stream_v) != SOCP_OK) return SOCP_ERR_HIP;
    hipStream_t main_stream = static_cast<hipStream_t>(main_stream_v), fs = nullptr;

    // ---- speculative FD rows (DESIGN section 6 "Fewer rounds"; the host engine's rule): off for hybrj chains and when a slot per chain
    // ((n + 1) n doubles) would take more than 4 GiB
    int num_simd = 1024;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, socp_ctx_device(ctx)) == hipSuccess && prop.multiProcessorCount > 0) num_simd = 4 * prop.multiProcessorCount;
    }
    const int segs = nodes - 1;
    const long rowsLen = (long)(n + 1) * n;
    const size_t rowsB = sizeof(double) * (size_t)rowsLen;
    int speculate = opt->speculate;
    if (const char *e = std::getenv("SOCP_CHAINS_SPECULATE")) speculate = std::atoi(e);
    if (opt->analytic_jac || (double)P * rowsB > 4.0 * 1024 * 1024 * 1024) speculate = 0;
    const bool spec_on = speculate != 0;
    // the most residual requests a round evaluates as FD batches: all P when forced, else what one wavefront per SIMD holds
    const int capS = !spec_on ? 0 : (speculate > 0 ? P : (int)std::min<long>(P, std::max<long>(1, (long)num_simd * 64 / ((long)(n + 1) * segs))));
    Dev dWs, dStates, dStatus, dList, dListS, dFlags, dListF, dListJ, dX, dF, dJx, dJf, dJ, dRes, dPF, dTF, dXF, dPJ, dTJ, dXJ, dSlots, dStage, dIdxA, dIdxB;
    Pinned hIdxA{"hIdxA"}, hIdxB{"hIdxB"};
    Pinned hStatus{"hStatus"}, hList{"hList"}, hListS{"hListS"}, hFlags{"hFlags"}, hListF{"hListF"}, hListJ{"hListJ"}, hX{"hX"}, hRes{"hRes"}, hPF{"hPF"},
        hTF{"hTF"}, hXF{"hXF"}, hPJ{"hPJ"}, hTJ{"hTJ"}, hXJ{"hXJ"};
    Arena dev_arena, host_arena;
    {
        const size_t intsB = plan_sizes.intsB, parB = plan_sizes.parB, timeB = plan_sizes.timeB, nodeB = plan_sizes.nodeB;
        struct { Piece *piece; size_t bytes; bool host; } plan[] = {
            {&dWs, sizeof(double) * pool.ws_stride * P, false}, {&dStates, sizeof(State) * P, false}, {&dStatus, sizeof(Status) * P, false}, {&dList, intsB, false}, {&dListS, intsB, false},
            {&dFlags, intsB, false}, {&dListF, intsB, false}, {&dListJ, intsB, false}, {&dX, rowB * P, false}, {&dF, rowB * P, false},
            {&dJx, rowB * P, false}, {&dJf, rowB * P, false}, {&dJ, jacB * jlaunch, false}, {&dRes, 2 * rowB * P, false},
            {&dPF, pp_params ? parB : 0, false}, {&dPJ, pp_params ? parB : 0, false}, {&dTF, pp_bound ? timeB : 0, false},
            {&dTJ, pp_bound ? timeB : 0, false}, {&dXF, pp_bound ? nodeB : 0, false}, {&dXJ, pp_bound ? nodeB : 0, false},
            {&hStatus.piece, sizeof(Status) * P, true}, {&hList.piece, intsB, true}, {&hListS.piece, intsB, true}, {&hFlags.piece, intsB, true}, {&hListF.piece, intsB, true},
            {&hListJ.piece, intsB, true}, {&hX.piece, rowB * P, true}, {&hRes.piece, 2 * rowB * P, true}, {&hPF.piece, pp_params ? parB : 0, true},
            {&hPJ.piece, pp_params ? parB : 0, true}, {&hTF.piece, pp_bound ? timeB : 0, true}, {&hTJ.piece, pp_bound ? timeB : 0, true},
            {&hXF.piece, pp_bound ? nodeB : 0, true}, {&hXJ.piece, pp_bound ? nodeB : 0, true},
            {&dSlots, spec_on ? rowsB * P : 0, false}, {&dStage, spec_on ? rowsB * capS : 0, false}, {&dIdxA, spec_on ? intsB : 0, false},
            {&dIdxB, spec_on ? intsB : 0, false}, {&hIdxA.piece, spec_on ? intsB : 0, true}, {&hIdxB.piece, spec_on ? intsB : 0, true}};
        for (auto &e : plan) e.piece->plan(e.host ? host_arena : dev_arena, e.bytes);
        const double t0 = ms_since(t_begin);
        bool ok = dev_arena.alloc(false, socp_ctx_device(ctx), workspace_slot);
        const double t1 = ms_since(t_begin);
        ok = ok && host_arena.alloc(true, socp_ctx_device(ctx), workspace_slot);
        const double t2 = ms_since(t_begin);
        void *aux = nullptr;
        if (ok) ok = socp_ctx_aux_stream(ctx, &aux) == SOCP_OK;              // the context's second stream (created on its first use: ~6 ms)
        fs = static_cast<hipStream_t>(aux);
        if (trace) std::fprintf(stderr, "[socp_chains/device] set-up: host tables %.2f ms, device arena (%.1f MB) %.2f ms, pinned arena (%.1f MB) %.2f ms, stream %.2f ms\n",
                                t0, 1e-6 * dev_arena.cap, t1 - t0, 1e-6 * host_arena.cap, t2 - t1, ms_since(t_begin) - t2);
        if (!ok) { (void)hipGetLastError(); return (dev_arena.base && host_arena.base) ? SOCP_ERR_HIP : kDeviceEngineAllocFailed; }
        for (auto &e : plan) e.piece->bind(e.host ? host_arena : dev_arena);
        for (Pinned *h : {&hStatus, &hList, &hListS, &hFlags, &hListF, &hListJ, &hX, &hRes, &hPF, &hPJ, &hTF, &hTJ, &hXF, &hXJ, &hIdxA, &hIdxB}) h->bind(host_arena);
    }
    // what the host knows about the two streams' progress (staging.hpp): every synchronise below goes through these
    StreamClock clk_main(main_stream), clk_fs(fs);
    auto sync_main = [&] { return clk_main.synchronize() ? hipSuccess : hipErrorUnknown; };
    auto sync_fs = [&] { return clk_fs.synchronize() ? hipSuccess : hipErrorUnknown; };
    pool.states = static_cast<State *>(dStates.p);
    pool.ws = dWs.d();
    // what the host knows of every chain's state machine: refreshed after each advance for the chains that were advanced (the others
    // have not changed), from a compact record gathered on the device
    std::vector<Status> hS(P, Status{0, 0, 0, 0, 0, 0});

    // per chain: where its request's FD rows sit in the staging area (-1: none), what the solver's iteration counter and the kind of
    // the request were when they were staged, and for which iteration counter the rows in its slot are the rows at x (-1: none).
    // The solver's x changes exactly when `iter` grows (an accepted trial step) or a new solve starts, so the counter stands for x.
    std::vector<int> stage_idx(spec_on ? P : 0, -1), stage_iter(spec_on ? P : 0, 0), stage_sel(spec_on ? P : 0, 0), slot_iter(spec_on ? P : 0, -1);
    std::vector<int> accepted, reqJc;
    int rc = SOCP_OK;
    long long rounds = 0, jac_launched = 0, restarts = 0, jac_from_cache = 0, spec_rounds = 0;
    bool round_limit_hit = false;
    double t_adv = 0, t_eval = 0, t_host = 0;
    const double t_setup = ms_since(t_begin);
    auto hip_ok = [&](hipError_t e) { if (e != hipSuccess && rc == SOCP_OK) rc = SOCP_ERR_HIP; return e == hipSuccess; };

    // (re)start the chains in `list` from their rows of hX (list order) -- they then need an advance
    std::vector<int> adv, advflag, reqF, reqJ, done, restart;
    const bool blocked_factor = socp::devsolver::blocked_factor_applies(n);
    int adv_jac = 0;                     // the first adv_jac entries of `adv` are chains whose pending request was a Jacobian
    auto start_chains = [&](const std::vector<int> &list) {
        if (list.empty()) return;
        // its OWN list buffers: the copy below is asynchronous and the advance loop refills hList right after this returns -- with
        // the same chains when nothing else is waiting for an advance, but with cached-Jacobian chains in front of them when the
        // speculative rows are on (found in round 4: continuation chains restarted from the wrong list, iterates changed)
        std::memcpy(hListS.host(), list.data(), sizeof(int) * list.size());
        hip_ok(hipMemcpyAsync(dListS.p, hListS.source(clk_main), sizeof(int) * list.size(), hipMemcpyHostToDevice, main_stream));
        // the start kernel reads the rows straight from the pinned buffer (it is mapped into the device's address space): a
        // hipMemcpyAsync of these 4 MB held the calling thread for 7 ms at the first start of 4096 chains.  hX is not written
        // again before the next stream synchronise (Staged: a host access before that would be caught).
        hip_ok(socp::devsolver::launch_start(main_stream, pool, dListS.i(), (int)list.size(), static_cast<const double *>(hX.source(clk_main))));
    };

    {
        std::vector<int> all(P);
        double *const x0 = hX.hd();
        for (int p = 0; p < P; p++) { all[p] = p; std::memcpy(x0 + (size_t)p * n, ch[p].committed.data(), rowB); }
        hip_ok(hipMemsetAsync(dStates.p, 0, sizeof(State) * P, main_stream));
        start_chains(all);
        adv = all;
        advflag.assign(P, 0);
    }

    const double t_pre = ms_since(t_begin) - t_setup;
    const clk::time_point t_loop_begin = clk::now();
    while (rc == SOCP_OK) {
        // ---- advance until every live chain has one pending evaluation request -------------------------------------------
        reqF.clear(); reqJ.clear();
        while (!adv.empty() && rc == SOCP_OK) {
            const clk::time_point ta = clk::now();
            const int count = (int)adv.size();
            std::memcpy(hList.host(), adv.data(), sizeof(int) * count);
            std::memcpy(hFlags.host(), advflag.data(), sizeof(int) * count);
            hip_ok(hipMemcpyAsync(dList.p, hList.source(clk_main), sizeof(int) * count, hipMemcpyHostToDevice, main_stream));
            hip_ok(hipMemcpyAsync(dFlags.p, hFlags.source(clk_main), sizeof(int) * count, hipMemcpyHostToDevice, main_stream));
            // the chains that have just received a Jacobian come first in the list (adv_jac of them) and go in their own launch
            if (adv_jac > 0 && fast_factor) hip_ok(socp::devsolver::launch_factor_fast(main_stream, pool, dList.i(), adv_jac));
            else if (adv_jac > 0 && blocked_factor) hip_ok(socp::devsolver::launch_factor(main_stream, pool, dList.i(), adv_jac));
            // (the chains whose factor work a factor KERNEL has just done advance like the others: what is left for them is the
            // refresh's tail and a dogleg step -- a trial launch's work, not a factorisation's; as a "factor phase" launch, a thread per
            // column and three wavefronts per SIMD, that tail was the longest launch of a KD-chain round: 1.6 ms against 0.7 ms for a
            // whole trial step of the same 4096 problems)
            hip_ok(socp::devsolver::launch_advance(main_stream, pool, dList.i(), adv_jac, dFlags.i(), !(fast_factor || blocked_factor)));
            hip_ok(socp::devsolver::launch_advance(main_stream, pool, dList.i() + adv_jac, count - adv_jac, dFlags.i() + adv_jac, false));
            adv_jac = 0;
            hip_ok(socp::devsolver::launch_gather_status(main_stream, pool, dList.i(), count, static_cast<Status *>(dStatus.p)));
            hip_ok(hipMemcpyAsync(hStatus.target(clk_main), dStatus.p, sizeof(Status) * count, hipMemcpyDeviceToHost, main_stream));
            hip_ok(sync_main());
            t_adv += ms_since(ta);
            if (rc != SOCP_OK) break;
            const Status *const hNew = static_cast<const Status *>(hStatus.host());
            for (int k = 0; k < count; k++) hS[adv[k]] = hNew[k];
            const clk::time_point th = clk::now();
            done.clear();
            if (spec_on) {
                // the rows staged for a chain's last request are the rows at its x now if that request was F(x) itself, or a trial
                // point that has just been accepted: staging -> the chain's slot, one gather launch
                accepted.clear();
                for (int p : adv) {
                    if (stage_idx[p] < 0) continue;
                    const bool live = hS[p].req != socp::devsolver::RQ_DONE;
                    if (live && (stage_sel[p] == 0 || hS[p].iter == stage_iter[p] + 1)) accepted.push_back(p);
                    else if (!live) slot_iter[p] = -1;
                }
                if (!accepted.empty()) {
                    int *const ia = hIdxA.hi(), *const ib = hIdxB.hi();
                    for (size_t k = 0; k < accepted.size(); k++) { ia[k] = stage_idx[accepted[k]]; ib[k] = accepted[k]; }
                    hip_ok(hipMemcpyAsync(dIdxA.p, hIdxA.source(clk_main), sizeof(int) * accepted.size(), hipMemcpyHostToDevice, main_stream));
                    hip_ok(hipMemcpyAsync(dIdxB.p, hIdxB.source(clk_main), sizeof(int) * accepted.size(), hipMemcpyHostToDevice, main_stream));
                    hip_ok(socp::devsolver::launch_copy_blocks(main_stream, dStage.d(), dIdxA.i(), dSlots.d(), dIdxB.i(), (int)accepted.size(), rowsLen));
                    hip_ok(sync_main());                             // (the index buffers are reused below)
                    for (int p : accepted) slot_iter[p] = hS[p].iter;
                }
                for (int p : adv) stage_idx[p] = -1;                 // the staging area is about to be reused
            }
            reqJc.clear();
            for (int p : adv) {
                const int rq = hS[p].req;
                if (rq == socp::devsolver::RQ_FVEC) reqF.push_back(p);
                else if (rq == socp::devsolver::RQ_JAC) {
                    if (spec_on && slot_iter[p] >= 0 && slot_iter[p] == hS[p].iter) reqJc.push_back(p);
                    else reqJ.push_back(p);
                } else done.push_back(p);
            }
            adv.clear(); advflag.clear();
            if (!reqJc.empty() && rc == SOCP_OK) {
                // Jacobians from cached rows, no trajectories: slots -> staging (gather), differences, into the solvers' matrices;
                // those chains go straight back into the advance loop (their factorisation)
                sort_runs(reqJc);
                const int chunk = std::max(1, std::min(capS, jlaunch));
                for (size_t j0 = 0; j0 < reqJc.size() && rc == SOCP_OK; j0 += (size_t)chunk) {
                    const int kc = (int)std::min<size_t>((size_t)chunk, reqJc.size() - j0);
                    std::memcpy(hIdxA.host(), reqJc.data() + j0, sizeof(int) * kc);
                    hip_ok(hipMemcpyAsync(dIdxA.p, hIdxA.source(clk_main), sizeof(int) * kc, hipMemcpyHostToDevice, main_stream));
                    hip_ok(socp::devsolver::launch_copy_blocks(main_stream, dSlots.d(), dIdxA.i(), dStage.d(), nullptr, kc, rowsLen));
                    hip_ok(socp::devsolver::launch_gather_jac(main_stream, pool, dIdxA.i(), kc, dJx.d(), dJf.d()));
                    const int r = socp_fd_diff_dev(ctx, kc, dJx.d(), opt->epsfcn, dStage.d(), dJ.d());
                    if (r != SOCP_OK) { rc = r; break; }
                    hip_ok(socp::devsolver::launch_scatter_jac(main_stream, pool, dIdxA.i(), kc, dJ.d()));
                    hip_ok(sync_main());                             // (dIdxA is rewritten by the next chunk / the next pass)
                }
                jac_from_cache += (long long)reqJc.size();
                adv = reqJc;
                adv_jac = (int)reqJc.size();
                advflag.assign(adv.size(), 0);
            }
            if (!done.empty()) {
                // results of the solves that ended: x and fvec of those chains, then the homotopy logic on the host
                std::memcpy(hList.host(), done.data(), sizeof(int) * done.size());
                hip_ok(hipMemcpyAsync(dList.p, hList.source(clk_main), sizeof(int) * done.size(), hipMemcpyHostToDevice, main_stream));
                hip_ok(socp::devsolver::launch_gather_result(main_stream, pool, dList.i(), (int)done.size(), dRes.d()));
                hip_ok(hipMemcpyAsync(hRes.target(clk_main), dRes.p, 2 * rowB * done.size(), hipMemcpyDeviceToHost, main_stream));
                hip_ok(sync_main());
                if (rc != SOCP_OK) break;
                restart.clear();
                std::vector<double> next;
                const double *const res = hRes.hd();
                double *const x0 = hX.hd();
                for (size_t k = 0; k < done.size(); k++) {
                    const int p = done[k];
                    const double *x = res + 2 * (size_t)n * k, *f = x + n;
                    double ss = 0;
                    for (int i = 0; i < n; i++) ss += f[i] * f[i];
                    ch[p].fnorm = std::sqrt(ss);
                    if (socp::chains::after_solve(*opt, blk, p, ch[p], x, n, hS[p].info, hS[p].nfev, hS[p].njev, next)) {
                        std::memcpy(x0 + (size_t)restart.size() * n, next.data(), rowB);
                        restart.push_back(p);
                    }
                }
                restarts += (long long)restart.size();
                start_chains(restart);
                if (spec_on) for (int p : restart) slot_iter[p] = -1;         // another problem now: cached rows are not its rows
                adv.insert(adv.end(), restart.begin(), restart.end());       // (after the chains that have just received a cached Jacobian)
                advflag.assign(adv.size(), 0);
            }
            t_host += ms_since(th);
        }
        if (rc != SOCP_OK) break;
        // chain order keeps the batches those of the host engine (a restarted chain's request arrives in a later inner pass)
        sort_runs(reqF);
        sort_runs(reqJ);
        const int kF = (int)reqF.size(), kJ = (int)reqJ.size();
        if (kF == 0 && kJ == 0) break;
        if (opt->max_rounds > 0 && rounds >= opt->max_rounds) {
            // round budget spent: the chains still solving stop the way a negative callback return stops hybrd
            round_limit_hit = true;
            adv = reqF;
            adv.insert(adv.end(), reqJ.begin(), reqJ.end());
            adv_jac = 0;
            advflag.assign(adv.size(), SOCP_INFO_ROUND_LIMIT);
            continue;
        }
        rounds++;
        if (trace) std::fprintf(stderr, "[socp_chains/device] round %lld: %d residual requests, %d Jacobian requests\n", rounds, kF, kJ);
        const clk::time_point te = clk::now();
        // ---- residual requests: stream fs ------------------------------------------------------------------------------------
        if (kF) {
            std::memcpy(hListF.host(), reqF.data(), sizeof(int) * kF);
            {
                double *const pf = hPF.hd(), *const tf = hTF.hd(), *const xf = hXF.hd();
                for (int k = 0; k < kF; k++) blk.stage(reqF[k], k, pf, tf, xf);
            }
            hip_ok(hipMemcpyAsync(dListF.p, hListF.source(clk_fs), sizeof(int) * kF, hipMemcpyHostToDevice, fs));
            if (pp_params) hip_ok(hipMemcpyAsync(dPF.p, hPF.source(clk_fs), sizeof(double) * stride * kF, hipMemcpyHostToDevice, fs));
            if (pp_bound) {
                hip_ok(hipMemcpyAsync(dTF.p, hTF.source(clk_fs), sizeof(double) * nodes * kF, hipMemcpyHostToDevice, fs));
                hip_ok(hipMemcpyAsync(dXF.p, hXF.source(clk_fs), sizeof(double) * nodes * S * kF, hipMemcpyHostToDevice, fs));
            }
            hip_ok(socp::devsolver::launch_gather_eval(fs, pool, dListF.i(), kF, dX.d()));
            // all of them as whole forward-difference batches when they fit the idle SIMDs (one wavefront per SIMD keeps the round at
            // one trajectory latency), else none: a round that is part FD batches, part plain residuals would be two launches on
            // one stream, i.e. two latencies.  speculate = 1 forces all.
            bool as_batches = false;
            if (spec_on && kF <= capS) {
                const long lanes = ((long)num_simd - ((long)kJ * n * segs + 63) / 64) * 64;
                as_batches = speculate > 0 || (long)kF * (n + 1) * segs <= lanes;
            }
            socp_ctx_set_stream(ctx, fs, 0);
            socp_problem_set_blocks_dev(ctx, pp_params ? dPF.d() : nullptr, stride, pp_bound ? dTF.d() : nullptr, pp_bound ? dXF.d() : nullptr);
            const int r = as_batches ? socp_fd_rows_dev(ctx, kF, dX.d(), opt->epsfcn, dStage.d()) : socp_residual_batch_dev(ctx, kF, dX.d(), dF.d());
            socp_ctx_set_stream(ctx, main_stream, 0);
            if (r != SOCP_OK) { rc = r; break; }
            if (as_batches) {
                // F is row 0 of a request's (n + 1) x n block; the block waits in the staging area for the solver's verdict on the point
                hip_ok(socp::devsolver::launch_scatter_fvec(fs, pool, dListF.i(), kF, dStage.d(), rowsLen));
                for (int k = 0; k < kF; k++) { const int p = reqF[k]; stage_idx[p] = k; stage_iter[p] = hS[p].iter; stage_sel[p] = hS[p].eval_sel; }
                spec_rounds++;
            } else {
                hip_ok(socp::devsolver::launch_scatter_fvec(fs, pool, dListF.i(), kF, dF.d()));
            }
            if (trace && as_batches) std::fprintf(stderr, "[socp_chains/device]   (the %d residual requests as FD batches)\n", kF);
        }
        // ---- Jacobian requests: the context's stream, in passes of jlaunch -----------------------------------------------------
        if (kJ) {
            std::memcpy(hListJ.host(), reqJ.data(), sizeof(int) * kJ);
            {
                double *const pj = hPJ.hd(), *const tj = hTJ.hd(), *const xj = hXJ.hd();
                for (int k = 0; k < kJ; k++) blk.stage(reqJ[k], k, pj, tj, xj);
            }
            hip_ok(hipMemcpyAsync(dListJ.p, hListJ.source(clk_main), sizeof(int) * kJ, hipMemcpyHostToDevice, main_stream));
            if (pp_params) hip_ok(hipMemcpyAsync(dPJ.p, hPJ.source(clk_main), sizeof(double) * stride * kJ, hipMemcpyHostToDevice, main_stream));
            if (pp_bound) {
                hip_ok(hipMemcpyAsync(dTJ.p, hTJ.source(clk_main), sizeof(double) * nodes * kJ, hipMemcpyHostToDevice, main_stream));
                hip_ok(hipMemcpyAsync(dXJ.p, hXJ.source(clk_main), sizeof(double) * nodes * S * kJ, hipMemcpyHostToDevice, main_stream));
            }
            hip_ok(socp::devsolver::launch_gather_jac(main_stream, pool, dListJ.i(), kJ, dJx.d(), dJf.d()));
            for (int j0 = 0; j0 < kJ && rc == SOCP_OK; j0 += jlaunch) {
                const int kc = std::min(jlaunch, kJ - j0);
                socp_problem_set_blocks_dev(ctx, pp_params ? dPJ.d() + (size_t)j0 * stride : nullptr, stride,
                                            pp_bound ? dTJ.d() + (size_t)j0 * nodes : nullptr, pp_bound ? dXJ.d() + (size_t)j0 * nodes * S : nullptr);
                const int r = opt->analytic_jac
                                  ? socp_var_jacobian_multi_dev(ctx, kc, dJx.d() + (size_t)j0 * n, dJ.d())
                                  : socp_fd_jacobian_multi_dev(ctx, kc, dJx.d() + (size_t)j0 * n, dJf.d() + (size_t)j0 * n, opt->epsfcn, dJ.d(), opt->dedup);
                if (r != SOCP_OK) { rc = r; break; }
                hip_ok(socp::devsolver::launch_scatter_jac(main_stream, pool, dListJ.i() + j0, kc, dJ.d()));
            }
            jac_launched += kJ;
        }
        if (rc != SOCP_OK) break;
        hip_ok(sync_fs());
        hip_ok(sync_main());
        t_eval += ms_since(te);
        adv = reqJ;
        adv.insert(adv.end(), reqF.begin(), reqF.end());
        adv_jac = kJ;
        advflag.assign(adv.size(), 0);
    }
    const clk::time_point t_loop_end = clk::now();
    socp_problem_set_blocks_dev(ctx, nullptr, 0, nullptr, nullptr);
    socp_ctx_set_stream(ctx, main_stream, 0);
    (void)hipStreamSynchronize(fs);
    (void)hipStreamSynchronize(main_stream);

    if (rc == SOCP_OK) {
        for (int p = 0; p < P; p++) {
            const socp::chains::ChainCore &c = ch[p];
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
    if (clk_main.forced_syncs + clk_fs.forced_syncs)
        std::fprintf(stderr, "[socp_chains/device] WARNING: %llu host accesses to pinned staging buffers had to synchronise a stream first (staging.hpp): an ordering the "
                             "engine should have by construction is missing\n", clk_main.forced_syncs + clk_fs.forced_syncs);
    if (trace && round_limit_hit) std::fprintf(stderr, "[socp_chains/device] round limit %d reached: the chains still solving were stopped\n", opt->max_rounds);
    if (trace) std::fprintf(stderr, "[socp_chains/device] first start of the chains %.2f ms, rounds %.2f ms (of which outside the three timers %.2f ms), after the last round %.2f ms\n",
                            t_pre, std::chrono::duration<double, std::milli>(t_loop_end - t_loop_begin).count(),
                            std::chrono::duration<double, std::milli>(t_loop_end - t_loop_begin).count() - t_adv - t_eval - t_host, ms_since(t_loop_end));
    if (trace)
        std::fprintf(stderr, "[socp_chains/device] set-up %.1f ms, solver kernels + state read-back %.1f ms, evaluation launches %.1f ms, host chain logic %.1f ms, "
                             "total %.1f ms; %lld rounds, %lld Jacobians launched, %lld from cached rows, %lld solver restarts; %d threads per problem, %s factorisation, %.1f MB of solver state\n",
                     t_setup, t_adv, t_eval, t_host, ms_since(t_begin), rounds, jac_launched, jac_from_cache, restarts, socp::devsolver::threads_for(n),
                     fast_factor ? "matrix-core (throughput)" : "order-preserving",
                     1e-6 * sizeof(double) * pool.ws_stride * P);
    if (trace) {
        unsigned long long pf[16];
        if (socp::devsolver::read_profile(pf, true) == hipSuccess && (pf[0] | pf[4] | pf[6]))
            std::fprintf(stderr, "[socp_chains/device] solver phases, clock ticks of thread 0 summed over problems (a -DSOCP_SOLVER_PROFILE build): trial head %llu, "
                                 "Q^T w %llu, r1updt %llu, r1mpyq %llu, dogleg %llu, step tail %llu, factor %llu, Jacobian tail %llu\n",
                         pf[0], pf[1], pf[2], pf[3], pf[4], pf[5], pf[6], pf[7]);
        if (pf[8] && !pf[11])
            std::fprintf(stderr, "[socp_chains/device] inside the trial step: dogleg = back substitution %llu + gradient %llu + the rest (above); r1updt = first sweep %llu + second sweep (above)\n",
                         pf[8], pf[9], pf[10]);
        if (pf[9] | pf[11])
            std::fprintf(stderr, "[socp_chains/device] inside the factor work: set-up %llu; qrfac: column ahead %llu, its norm and scaling %llu, sweep %llu; "
                                 "R, clearing %llu; qform %llu\n", pf[8], pf[9], pf[10], pf[11], pf[12], pf[13]);
    }
    if (stats) {
        stats->rounds = rounds; stats->jacobians_launched = jac_launched; stats->jacobians_from_cache = jac_from_cache;
        stats->speculative_rounds = spec_rounds; stats->restarts = restarts; stats->wall_ms = ms_since(t_begin);
    }
    return rc;
}

// include/socp_solver.h: the Jacobian refresh alone, `count` problems at once
extern "C" int socp_qr_factor_batch(int device, int n, int count, const double *J, const double *b, int flavour, int reps, double *Q, double *R,
                                    double *qtb, double *rdiag, double *acnorm, int *sing, double *kernel_ms)
{
    using namespace socp::devsolver;
    if (n < 1 || count < 0 || !J || !b || reps < 1 || (flavour != SOCP_FACTOR_EXACT && flavour != SOCP_FACTOR_FAST)) return SOCP_ERR_ARG;
    if (flavour == SOCP_FACTOR_FAST && !fast_factor_applies(n)) return SOCP_ERR_UNSUPPORTED;
    if (count == 0) return SOCP_OK;
    struct DeviceGuard {
        int prev = -1;
        bool ok = false;
        explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && (dev < 0 || hipSetDevice(dev) == hipSuccess); }
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard(device);
    if (!guard.ok) return SOCP_ERR_HIP;
    PoolDev pool;
    pool.cfg.n = n; pool.cfg.ld = ld_for(n); pool.cfg.maxfev = 1; pool.cfg.mode = 1; pool.cfg.analytic = 0;
    pool.cfg.xtol = 0; pool.cfg.epsfcn = 0; pool.cfg.factor = 1;
    pool.ws_stride = ws_doubles(n, pool.cfg.ld);
    pool.P = count;
    const size_t nn = (size_t)n * n, wsB = sizeof(double) * (size_t)pool.ws_stride * count;
    double *dJ = nullptr, *dB = nullptr;
    int *dList = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<int> list(count);
    for (int k = 0; k < count; k++) list[k] = k;
    int rc = SOCP_OK;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == SOCP_OK) rc = SOCP_ERR_HIP; return e == hipSuccess; };
    ok(hipMalloc(&pool.ws, wsB));
    ok(hipMalloc(&pool.states, sizeof(State) * count));
    ok(hipMalloc(&dJ, sizeof(double) * nn * count));
    ok(hipMalloc(&dB, sizeof(double) * (size_t)n * count));
    ok(hipMalloc(&dList, sizeof(int) * count));
    ok(hipEventCreate(&e0));
    ok(hipEventCreate(&e1));
    if (rc == SOCP_OK) {
        ok(hipMemset(pool.ws, 0, wsB));
        ok(hipMemset(pool.states, 0, sizeof(State) * count));                 // eval_sel = 0: scatter_fvec writes fvec
        ok(hipMemcpy(dJ, J, sizeof(double) * nn * count, hipMemcpyHostToDevice));
        ok(hipMemcpy(dB, b, sizeof(double) * (size_t)n * count, hipMemcpyHostToDevice));
        ok(hipMemcpy(dList, list.data(), sizeof(int) * count, hipMemcpyHostToDevice));
    }
    double total_ms = 0;
    for (int rep = 0; rep < reps && rc == SOCP_OK; rep++) {
        ok(launch_scatter_jac(nullptr, pool, dList, count, dJ));
        ok(launch_scatter_fvec(nullptr, pool, dList, count, dB));
        ok(hipEventRecord(e0, nullptr));
        ok(flavour == SOCP_FACTOR_FAST ? launch_factor_fast(nullptr, pool, dList, count) : launch_factor_exact(nullptr, pool, dList, count));
        ok(hipEventRecord(e1, nullptr));
        ok(hipEventSynchronize(e1));
        float ms = 0;
        if (rc == SOCP_OK && ok(hipEventElapsedTime(&ms, e0, e1))) total_ms += ms;
    }
    if (kernel_ms) *kernel_ms = total_ms / reps;
    if (flavour == SOCP_FACTOR_FAST && std::getenv("SOCP_MULTISTART_TRACE")) {
        unsigned long long pf[16];
        if (read_factor_profile(pf, true) == hipSuccess && (pf[1] | pf[3]))
            std::fprintf(stderr, "[socp_qr_factor_batch] clock ticks of wave 0, summed over problems and runs (a -DSOCP_FACTOR_PROFILE build): norms %llu, panel %llu, "
                                 "waiting for the panel %llu, trailing strips %llu, waiting after them %llu, R / qtf %llu, qform: panel load %llu, strips %llu, waiting %llu\n",
                         pf[0], pf[1], pf[2], pf[3], pf[4], pf[5], pf[6], pf[7], pf[8]);
        if (pf[11])
            std::fprintf(stderr, "[socp_qr_factor_batch] inside wave 0's panels: load (+ norms, to LDS) %llu, to the row layout %llu, the 16 columns %llu, "
                                 "V to LDS + store %llu, Gram + T %llu\n", pf[9], pf[10], pf[11], pf[12], pf[13]);
    }
    if (rc == SOCP_OK && (Q || R || qtb || rdiag || acnorm || sing)) {
        // the workspaces come back whole, in slices of at most 256 MB, and are taken apart here
        const int per = (int)std::max<size_t>(1, ((size_t)256 << 20) / (sizeof(double) * (size_t)pool.ws_stride));
        std::vector<double> h((size_t)pool.ws_stride * std::min(per, count));
        std::vector<State> hs(count);
        ok(hipMemcpy(hs.data(), pool.states, sizeof(State) * count, hipMemcpyDeviceToHost));
        for (int k0 = 0; k0 < count && rc == SOCP_OK; k0 += per) {
            const int kc = std::min(per, count - k0);
            ok(hipMemcpy(h.data(), pool.ws + (size_t)k0 * pool.ws_stride, sizeof(double) * (size_t)pool.ws_stride * kc, hipMemcpyDeviceToHost));
            for (int k = 0; k < kc && rc == SOCP_OK; k++) {
                Work w(h.data() + (size_t)k * pool.ws_stride, n, pool.cfg.ld);
                const size_t p = (size_t)(k0 + k);
                if (Q) for (int i = 0; i < n; i++) std::memcpy(Q + p * nn + (size_t)i * n, w.A + (size_t)i * pool.cfg.ld, sizeof(double) * n);
                if (R) std::memcpy(R + p * ((size_t)n * (n + 1) / 2), w.r, sizeof(double) * ((size_t)n * (n + 1) / 2));
                if (qtb) std::memcpy(qtb + p * n, w.qtf, sizeof(double) * n);
                if (rdiag) std::memcpy(rdiag + p * n, w.wa1, sizeof(double) * n);
                if (acnorm) std::memcpy(acnorm + p * n, w.wa2, sizeof(double) * n);
                if (sing) sing[p] = hs[p].sing;
            }
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (dList) (void)hipFree(dList);
    if (dB) (void)hipFree(dB);
    if (dJ) (void)hipFree(dJ);
    if (pool.states) (void)hipFree(pool.states);
    if (pool.ws) (void)hipFree(pool.ws);
    if (rc != SOCP_OK) (void)hipGetLastError();
    return rc;
}
