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
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess; }
    ~Dev() { if (p) (void)hipFree(p); }
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

struct Chain {
    socp_hybr *solver = nullptr;
    // homotopy state (shooting.cpp:598-692 / 695-778; the host mirror's `Homotopy`)
    double b = 1, b_prec = 0;
    bool finished = false;
    int info = 0, nfev_last = 0, nfev_total = 0, solves = 0;
    std::vector<double> committed;      // tab_param: the unknowns of the last converged solve (or the start)
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

double blend(double b, double a0, double a1) { return (1 - b) * a0 + b * a1; }

}  // namespace

extern "C" int socp_chains_solve(socp_ctx *ctx, int P, const socp_chain_options *opt, const double *Z0, const double *params,
                                 const double *goal, const double *time_prev, const double *x_prev, const double *time_goal,
                                 const double *x_goal, double *Zout, int *info, int *nfev_last, int *nfev_total, int *solves,
                                 double *b_reached, double *param_final, double *fnorm, socp_chain_stats *stats)
{
    return socp_chains_solve_ex(ctx, P, opt, Z0, params, goal, time_prev, x_prev, time_goal, x_goal, Zout, info, nfev_last, nfev_total,
                                nullptr, solves, b_reached, param_final, fnorm, stats);
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
    if (kind != SOCP_CHAIN_PLAIN && kind != SOCP_CHAIN_PARAM && kind != SOCP_CHAIN_DATA) return SOCP_ERR_ARG;
    if (kind == SOCP_CHAIN_PARAM && (!goal || opt->param_index < 0 || opt->param_index >= nparams)) return SOCP_ERR_ARG;
    if (kind == SOCP_CHAIN_DATA && (!time_prev || !x_prev || !time_goal || !x_goal)) return SOCP_ERR_ARG;
    if (kind != SOCP_CHAIN_PLAIN && !(opt->step > 0)) return SOCP_ERR_ARG;
    if (P == 0) return SOCP_OK;

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
    const bool pp_params = params != nullptr || kind == SOCP_CHAIN_PARAM;
    const bool pp_bound = kind == SOCP_CHAIN_DATA || (time_goal && x_goal);
    const int stride = nparams + 2;
    double shared_sw[2] = {0, 0};
    socp_ctx_get_switching_times(ctx, shared_sw);

    using clk = std::chrono::steady_clock;
    const clk::time_point t_begin = clk::now();
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = ((long)P * n * n < 200000) ? 1 : (int)std::max(1u, std::min(16u, hw ? hw : 1u));
    auto parallel_for = [&](int count, auto &&body) {
        if (nthreads <= 1 || count < 2 * nthreads) { for (int k = 0; k < count; k++) body(k); return; }
        std::vector<std::thread> pool;
        pool.reserve(nthreads);
        for (int t = 0; t < nthreads; t++)
            pool.emplace_back([&, t]() { for (int k = t; k < count; k += nthreads) body(k); });
        for (std::thread &th : pool) th.join();
    };

    // ---- per-chain state --------------------------------------------------------------------------------------------
    std::vector<Chain> ch(P);
    std::vector<double> pblock(pp_params ? (size_t)P * stride : 0), rstart(P, 0.0);
    std::vector<double> tblock(pp_bound ? (size_t)P * nodes : 0), xblock(pp_bound ? (size_t)P * nodes * S : 0);
    auto set_blocks = [&](int p) {
        Chain &c = ch[p];
        if (kind == SOCP_CHAIN_PARAM) pblock[(size_t)p * stride + opt->param_index] = blend(c.b, rstart[p], goal[p]);
        if (kind == SOCP_CHAIN_DATA) {
            for (int i = 0; i < nodes; i++) {
                tblock[(size_t)p * nodes + i] = blend(c.b, time_prev[(size_t)p * nodes + i], time_goal[(size_t)p * nodes + i]);
                for (int j = 0; j < dim; j++) {
                    const size_t e = ((size_t)p * nodes + i) * S + j;
                    xblock[e] = blend(c.b, x_prev[e], x_goal[e]);
                }
            }
        }
    };
    bool alloc_ok = true;
    parallel_for(P, [&](int p) {
        Chain &c = ch[p];
        c.solver = socp_hybr_create(n, opt->xtol, opt->maxfev, opt->epsfcn, 1, opt->factor, opt->analytic_jac ? 1 : 0);
        c.committed.assign(Z0 + (size_t)p * n, Z0 + (size_t)(p + 1) * n);
        c.eval_x.resize(n); c.slot_x.resize(n);
        if (pp_params) {
            double *blk = &pblock[(size_t)p * stride];
            std::memcpy(blk, params ? params + (size_t)p * nparams : shared_params, sizeof(double) * nparams);
            blk[nparams] = shared_sw[0]; blk[nparams + 1] = shared_sw[1];
            rstart[p] = kind == SOCP_CHAIN_PARAM ? blk[opt->param_index] : 0.0;
        }
        if (pp_bound && kind != SOCP_CHAIN_DATA) {
            std::memcpy(&tblock[(size_t)p * nodes], time_goal + (size_t)p * nodes, sizeof(double) * nodes);
            std::memcpy(&xblock[(size_t)p * nodes * S], x_goal + (size_t)p * nodes * S, sizeof(double) * nodes * S);
        }
        if (kind != SOCP_CHAIN_PLAIN) { c.b = std::min(opt->step, 1.0); c.b_prec = 0; }
        set_blocks(p);
        if (c.solver) socp_hybr_start(c.solver, c.committed.data(), nullptr);
    });
    auto cleanup = [&]() { parallel_for(P, [&](int p) { socp_hybr_destroy(ch[p].solver); ch[p].solver = nullptr; }); };
    for (int p = 0; p < P; p++) if (!ch[p].solver) alloc_ok = false;
    if (!alloc_ok) { cleanup(); return SOCP_ERR_ARG; }

    // ---- buffers ----------------------------------------------------------------------------------------------------
    const size_t rowB = sizeof(double) * n, jacB = rowB * n, rowsLen = (size_t)(n + 1) * n, rowsB = sizeof(double) * rowsLen;
    const int jchunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)P, ((size_t)64 << 20) / jacB));
    int speculate = opt->speculate;
    if (const char *e = std::getenv("SOCP_CHAINS_SPECULATE")) speculate = std::atoi(e);
    if ((double)P * rowsB * 2 > 16e9) speculate = 0;              // slots + staging would not be "free"
    if (opt->analytic_jac) speculate = 0;                          // no finite differences on the hybrj path
    const bool spec_on = speculate != 0;
    Pinned hX, hF, hJx, hJf, hJ, hPF, hTF, hXF, hPJ, hTJ, hXJ, hIdx;
    Dev dX, dF, dJx, dJf, dJ, dPF, dTF, dXF, dPJ, dTJ, dXJ, dStage, dSlots, dIdx;
    bool ok = hX.reserve(rowB * P) && hF.reserve(rowB * P) && hJx.reserve(rowB * P) && hJf.reserve(rowB * P) && hJ.reserve(jacB * jchunk) &&
              dX.alloc(rowB * P) && dF.alloc(rowB * P) && dJx.alloc(rowB * P) && dJf.alloc(rowB * P) && dJ.alloc(jacB * jchunk) &&
              hIdx.reserve(sizeof(int) * 2 * (size_t)P) && dIdx.alloc(sizeof(int) * 2 * (size_t)P);
    if (ok && pp_params) ok = hPF.reserve(sizeof(double) * stride * P) && hPJ.reserve(sizeof(double) * stride * P) &&
                              dPF.alloc(sizeof(double) * stride * P) && dPJ.alloc(sizeof(double) * stride * P);
    if (ok && pp_bound) ok = hTF.reserve(sizeof(double) * nodes * P) && hTJ.reserve(sizeof(double) * nodes * P) &&
                             hXF.reserve(sizeof(double) * nodes * S * P) && hXJ.reserve(sizeof(double) * nodes * S * P) &&
                             dTF.alloc(sizeof(double) * nodes * P) && dTJ.alloc(sizeof(double) * nodes * P) &&
                             dXF.alloc(sizeof(double) * nodes * S * P) && dXJ.alloc(sizeof(double) * nodes * S * P);
    if (ok && spec_on) ok = dStage.alloc(rowsB * P) && dSlots.alloc(rowsB * P);
    if (!ok) { cleanup(); return SOCP_ERR_HIP; }

    static const bool trace = std::getenv("SOCP_MULTISTART_TRACE") != nullptr;
    static const bool overlap = [] { const char *e = std::getenv("SOCP_MULTISTART_OVERLAP"); return !(e && e[0] == '0'); }();
    void *main_stream_v = nullptr;
    hipStream_t aux = nullptr;
    if (socp_ctx_synchronize(ctx) != SOCP_OK || socp_ctx_get_stream(ctx, &main_stream_v) != SOCP_OK ||
        hipStreamCreateWithFlags(&aux, hipStreamNonBlocking) != hipSuccess) {
        cleanup();
        return SOCP_ERR_HIP;
    }
    hipStream_t main_stream = static_cast<hipStream_t>(main_stream_v);
    hipStream_t fstream = overlap ? aux : main_stream;      // residual-type work (and everything speculative) goes here

    bool round_limit_hit = false;
    long long rounds = 0, spec_rows_rounds = 0, jac_from_cache = 0, jac_launched = 0, restarts = 0;
    double t_adv = 0, t_gpu = 0, t_copy = 0;
    const double t_setup = ms_since(t_begin);
    int rc = SOCP_OK;
    std::vector<int> reqF, reqJ, reqJc, accepted;

    // chain logic at the end of one Newton solve: the bisection rules of shooting.cpp:627-660 / 724-760
    auto solve_finished = [&](int p) {
        Chain &c = ch[p];
        c.info = socp_hybr_info(c.solver);
        c.nfev_last = socp_hybr_nfev(c.solver);
        c.nfev_total += c.nfev_last;
        c.solves++;
        const double *x = socp_hybr_x(c.solver);
        if (kind == SOCP_CHAIN_PLAIN) {
            c.committed.assign(x, x + n);                         // multi-start: the final iterate, whatever info says
            c.finished = true;
            return;
        }
        if (c.info < 0) { c.finished = true; return; }             // aborted (round limit): no further homotopy step
        bool running = true;
        std::vector<double> next;                                  // tab_param_temp for the next solve
        if (c.info != 1) {
            if (std::fabs(c.b - c.b_prec) < opt->step_min) running = false;
            c.b = c.b_prec + (c.b - c.b_prec) / 2;
            next = c.committed;
        } else if (c.b == 1) {
            running = false;
            c.committed.assign(x, x + n);
        } else {
            c.b_prec = c.b;
            c.b = std::min(c.b + opt->step, 1.0);
            c.committed.assign(x, x + n);
            next = c.committed;
        }
        set_blocks(p);                                             // the reference also moves Rdata / the boundary data on the failing exit
        if (!running) { c.finished = true; return; }
        socp_hybr_start(c.solver, next.data(), nullptr);
        c.slot_valid = false;                                      // another problem now: cached rows are not its rows
        c.stage_idx = -1;
        c.flag = 0;
        c.need_advance = true;
    };

    for (;;) {
        const clk::time_point ta = clk::now();
        // (1) advance every chain that received what it asked for; chains whose solve ended restart (or retire) and advance
        //     again, so that after this loop every live chain has exactly one pending request.  Jacobian requests whose
        //     point is cached are served here, without a round of trajectories.
        for (;;) {
            parallel_for(P, [&](int p) {
                Chain &c = ch[p];
                while (!c.finished && c.need_advance) {
                    c.req = socp_hybr_advance(c.solver, c.flag, &c.xin, &c.xout);
                    c.flag = 0;
                    c.need_advance = false;
                    if (c.req == SOCP_REQ_DONE) solve_finished(p);          // may set need_advance again (next homotopy step)
                }
            });
            if (!spec_on) break;
            // the rows evaluated last round belong to the chain's slot if that point is now the solver's x
            accepted.clear(); reqJc.clear();
            for (int p = 0; p < P; p++) {
                Chain &c = ch[p];
                if (c.finished || c.stage_idx < 0) continue;
                if (std::memcmp(socp_hybr_x(c.solver), c.eval_x.data(), rowB) == 0) accepted.push_back(p);
            }
            if (!accepted.empty()) {
                int *idx = hIdx.i();
                for (size_t k = 0; k < accepted.size(); k++) { idx[k] = ch[accepted[k]].stage_idx; idx[P + k] = accepted[k]; }
                if (hipMemcpyAsync(dIdx.p, idx, sizeof(int) * 2 * (size_t)P, hipMemcpyHostToDevice, fstream) != hipSuccess ||
                    !copy_blocks(fstream, dStage.d(), dIdx.i(), dSlots.d(), dIdx.i() + P, (int)accepted.size(), (int)rowsLen) ||
                    hipStreamSynchronize(fstream) != hipSuccess) { rc = SOCP_ERR_HIP; break; }
                for (int p : accepted) { ch[p].slot_x = ch[p].eval_x; ch[p].slot_valid = true; }
            }
            for (int p = 0; p < P; p++) ch[p].stage_idx = -1;      // the staging area is about to be reused
            for (int p = 0; p < P; p++) {
                Chain &c = ch[p];
                if (!c.finished && c.req == SOCP_REQ_JAC && c.slot_valid && std::memcmp(c.xin, c.slot_x.data(), rowB) == 0) reqJc.push_back(p);
            }
            if (reqJc.empty()) break;
            // Jacobians from cached rows: slot -> staging (gather), fd_diff, read back, scatter; then those chains advance again
            for (size_t j0 = 0; j0 < reqJc.size() && rc == SOCP_OK; j0 += jchunk) {
                const int kc = (int)std::min<size_t>(jchunk, reqJc.size() - j0);
                int *idx = hIdx.i();
                for (int k = 0; k < kc; k++) {
                    idx[k] = reqJc[j0 + k]; idx[P + k] = k;
                    std::memcpy(hJx.d() + (size_t)k * n, ch[reqJc[j0 + k]].xin, rowB);
                }
                if (hipMemcpyAsync(dIdx.p, idx, sizeof(int) * 2 * (size_t)P, hipMemcpyHostToDevice, fstream) != hipSuccess ||
                    hipMemcpyAsync(dJx.p, hJx.p, rowB * kc, hipMemcpyHostToDevice, fstream) != hipSuccess ||
                    !copy_blocks(fstream, dSlots.d(), dIdx.i(), dStage.d(), dIdx.i() + P, kc, (int)rowsLen)) { rc = SOCP_ERR_HIP; break; }
                socp_ctx_set_stream(ctx, fstream, 0);
                rc = socp_fd_diff_dev(ctx, kc, dJx.d(), opt->epsfcn, dStage.d(), dJ.d());
                socp_ctx_set_stream(ctx, main_stream, 0);
                if (rc != SOCP_OK) break;
                if (hipMemcpyAsync(hJ.p, dJ.p, jacB * kc, hipMemcpyDeviceToHost, fstream) != hipSuccess ||
                    hipStreamSynchronize(fstream) != hipSuccess) { rc = SOCP_ERR_HIP; break; }
                parallel_for(kc, [&](int k) { std::memcpy(ch[reqJc[j0 + k]].xout, hJ.d() + (size_t)k * n * n, jacB); });
                for (int k = 0; k < kc; k++) ch[reqJc[j0 + k]].need_advance = true;
                jac_from_cache += kc;
            }
            if (rc != SOCP_OK) break;
        }
        if (rc != SOCP_OK) break;
        // (2) gather the requests in chain order (keeps the batches deterministic)
        reqF.clear(); reqJ.clear();
        for (int p = 0; p < P; p++) {
            Chain &c = ch[p];
            if (c.finished) continue;
            if (c.req == SOCP_REQ_FVEC) reqF.push_back(p);
            else if (c.req == SOCP_REQ_JAC) reqJ.push_back(p);
        }
        t_adv += ms_since(ta);
        if (reqF.empty() && reqJ.empty()) break;
        if (opt->max_rounds > 0 && rounds >= opt->max_rounds) {
            // round budget spent: the chains still solving stop the way a negative callback return stops hybrd
            for (int p : reqF) { ch[p].flag = SOCP_INFO_ROUND_LIMIT; ch[p].need_advance = true; }
            for (int p : reqJ) { ch[p].flag = SOCP_INFO_ROUND_LIMIT; ch[p].need_advance = true; }
            round_limit_hit = true;
            continue;
        }
        rounds++;
        const int kF = (int)reqF.size(), kJ = (int)reqJ.size();
        // how many of the residual requests are evaluated as whole FD batches: as many as fit the idle SIMDs
        // (one wave per SIMD keeps the round at one trajectory latency); speculate = 1 forces all of them
        int kS = 0;
        if (spec_on && kF) {
            if (speculate > 0) kS = kF;
            else {
                // all or nothing: a round that is part FD batches, part plain residuals is two launches on one stream, i.e. two
                // trajectory latencies.  All of them fit when kF (n+1) segs lanes + the Jacobian launch <= 1024 waves x 64.
                const long lanes = (1024 - ((long)kJ * n * segs + 63) / 64) * 64;
                kS = ((long)kF * (n + 1) * segs <= lanes) ? kF : 0;
            }
        }
        if (trace) std::fprintf(stderr, "[socp_chains] round %lld: %d residual requests (%d as FD batches), %d Jacobian requests\n", rounds, kF, kS, kJ);
        // stage the requests' inputs and per-problem blocks: the first kS residual requests are the speculative ones
        for (int k = 0; k < kF; k++) {
            const int p = reqF[k];
            std::memcpy(hX.d() + (size_t)k * n, ch[p].xin, rowB);
            if (pp_params) std::memcpy(hPF.d() + (size_t)k * stride, &pblock[(size_t)p * stride], sizeof(double) * stride);
            if (pp_bound) {
                std::memcpy(hTF.d() + (size_t)k * nodes, &tblock[(size_t)p * nodes], sizeof(double) * nodes);
                std::memcpy(hXF.d() + (size_t)k * nodes * S, &xblock[(size_t)p * nodes * S], sizeof(double) * nodes * S);
            }
            if (k < kS) { std::memcpy(ch[p].eval_x.data(), ch[p].xin, rowB); ch[p].stage_idx = k; }
        }
        for (int k = 0; k < kJ; k++) {
            const int p = reqJ[k];
            std::memcpy(hJx.d() + (size_t)k * n, ch[p].xin, rowB);
            std::memcpy(hJf.d() + (size_t)k * n, socp_hybr_fvec(ch[p].solver), rowB);
            if (pp_params) std::memcpy(hPJ.d() + (size_t)k * stride, &pblock[(size_t)p * stride], sizeof(double) * stride);
            if (pp_bound) {
                std::memcpy(hTJ.d() + (size_t)k * nodes, &tblock[(size_t)p * nodes], sizeof(double) * nodes);
                std::memcpy(hXJ.d() + (size_t)k * nodes * S, &xblock[(size_t)p * nodes * S], sizeof(double) * nodes * S);
            }
        }
        // residual-type launches (stream F): the speculative prefix through the fdrows kernel, the rest through the residual kernel
        if (kF) {
            bool h2d = hipMemcpyAsync(dX.p, hX.p, rowB * kF, hipMemcpyHostToDevice, fstream) == hipSuccess;
            if (h2d && pp_params) h2d = hipMemcpyAsync(dPF.p, hPF.p, sizeof(double) * stride * kF, hipMemcpyHostToDevice, fstream) == hipSuccess;
            if (h2d && pp_bound) h2d = hipMemcpyAsync(dTF.p, hTF.p, sizeof(double) * nodes * kF, hipMemcpyHostToDevice, fstream) == hipSuccess &&
                                       hipMemcpyAsync(dXF.p, hXF.p, sizeof(double) * nodes * S * kF, hipMemcpyHostToDevice, fstream) == hipSuccess;
            if (!h2d) { rc = SOCP_ERR_HIP; break; }
            socp_ctx_set_stream(ctx, fstream, 0);
            if (kS) {
                socp_problem_set_blocks_dev(ctx, pp_params ? dPF.d() : nullptr, stride, pp_bound ? dTF.d() : nullptr, pp_bound ? dXF.d() : nullptr);
                rc = socp_fd_rows_dev(ctx, kS, dX.d(), opt->epsfcn, dStage.d());
                spec_rows_rounds++;
            }
            if (rc == SOCP_OK && kF > kS) {
                socp_problem_set_blocks_dev(ctx, pp_params ? dPF.d() + (size_t)kS * stride : nullptr, stride,
                                            pp_bound ? dTF.d() + (size_t)kS * nodes : nullptr, pp_bound ? dXF.d() + (size_t)kS * nodes * S : nullptr);
                rc = socp_residual_batch_dev(ctx, kF - kS, dX.d() + (size_t)kS * n, dF.d() + (size_t)kS * n);
            }
            socp_ctx_set_stream(ctx, main_stream, 0);
            if (rc != SOCP_OK) break;
        }
        if (kJ) {
            bool h2d = hipMemcpyAsync(dJx.p, hJx.p, rowB * kJ, hipMemcpyHostToDevice, main_stream) == hipSuccess &&
                       hipMemcpyAsync(dJf.p, hJf.p, rowB * kJ, hipMemcpyHostToDevice, main_stream) == hipSuccess;
            if (h2d && pp_params) h2d = hipMemcpyAsync(dPJ.p, hPJ.p, sizeof(double) * stride * kJ, hipMemcpyHostToDevice, main_stream) == hipSuccess;
            if (h2d && pp_bound) h2d = hipMemcpyAsync(dTJ.p, hTJ.p, sizeof(double) * nodes * kJ, hipMemcpyHostToDevice, main_stream) == hipSuccess &&
                                       hipMemcpyAsync(dXJ.p, hXJ.p, sizeof(double) * nodes * S * kJ, hipMemcpyHostToDevice, main_stream) == hipSuccess;
            if (!h2d) { rc = SOCP_ERR_HIP; break; }
            jac_launched += kJ;
        }
        bool f_collected = (kF == 0);
        for (int j0 = 0; j0 < kJ || !f_collected; j0 += jchunk) {
            const int kc = j0 < kJ ? std::min(jchunk, kJ - j0) : 0;
            const clk::time_point t_chunk = clk::now();
            if (kc) {
                socp_problem_set_blocks_dev(ctx, pp_params ? dPJ.d() + (size_t)j0 * stride : nullptr, stride,
                                            pp_bound ? dTJ.d() + (size_t)j0 * nodes : nullptr, pp_bound ? dXJ.d() + (size_t)j0 * nodes * S : nullptr);
                rc = opt->analytic_jac
                         ? socp_var_jacobian_multi_dev(ctx, kc, dJx.d() + (size_t)j0 * n, dJ.d())
                         : socp_fd_jacobian_multi_dev(ctx, kc, dJx.d() + (size_t)j0 * n, dJf.d() + (size_t)j0 * n, opt->epsfcn, dJ.d(), opt->dedup);
                if (rc != SOCP_OK) break;
            }
            if (!f_collected) {
                // F of the speculative requests is row 0 of their (n+1) x n block
                bool d2h = true;
                if (kS) d2h = hipMemcpy2DAsync(hF.p, rowB, dStage.p, rowsB, rowB, (size_t)kS, hipMemcpyDeviceToHost, fstream) == hipSuccess;
                if (d2h && kF > kS) d2h = hipMemcpyAsync(hF.d() + (size_t)kS * n, dF.d() + (size_t)kS * n, rowB * (kF - kS), hipMemcpyDeviceToHost, fstream) == hipSuccess;
                if (!d2h || hipStreamSynchronize(fstream) != hipSuccess) { rc = SOCP_ERR_HIP; break; }
                for (int k = 0; k < kF; k++) std::memcpy(ch[reqF[k]].xout, hF.d() + (size_t)k * n, rowB);
                f_collected = true;
            }
            if (kc) {
                if (hipStreamSynchronize(main_stream) != hipSuccess) { rc = SOCP_ERR_HIP; break; }
                t_gpu += ms_since(t_chunk);
                const clk::time_point tc = clk::now();
                if (hipMemcpy(hJ.p, dJ.p, jacB * kc, hipMemcpyDeviceToHost) != hipSuccess) { rc = SOCP_ERR_HIP; break; }
                parallel_for(kc, [&](int k) { std::memcpy(ch[reqJ[j0 + k]].xout, hJ.d() + (size_t)k * n * n, jacB); });
                t_copy += ms_since(tc);
            }
        }
        if (rc != SOCP_OK) break;
        for (int p : reqF) ch[p].need_advance = true;
        for (int p : reqJ) ch[p].need_advance = true;
    }
    socp_problem_set_blocks_dev(ctx, nullptr, 0, nullptr, nullptr);
    if (rc == SOCP_OK) {
        for (int p = 0; p < P; p++) {
            const Chain &c = ch[p];
            std::memcpy(Zout + (size_t)p * n, c.committed.data(), rowB);
            info[p] = c.info;
            if (nfev_last) nfev_last[p] = c.nfev_last;
            if (nfev_total) nfev_total[p] = c.nfev_total;
            if (njev_last) njev_last[p] = socp_hybr_njev(c.solver);
            if (solves) solves[p] = c.solves;
            if (b_reached) b_reached[p] = kind == SOCP_CHAIN_PLAIN ? 1.0 : (c.info == 1 ? c.b : c.b_prec);
            if (param_final) param_final[p] = kind == SOCP_CHAIN_PARAM ? pblock[(size_t)p * stride + opt->param_index] : 0.0;
            if (fnorm) {
                const double *f = socp_hybr_fvec(c.solver);
                double s = 0;
                for (int i = 0; i < n; i++) s += f[i] * f[i];
                fnorm[p] = std::sqrt(s);
            }
        }
    }
    for (int p = 0; p < P; p++) restarts += std::max(0, ch[p].solves - 1);
    if (trace && round_limit_hit) std::fprintf(stderr, "[socp_chains] round limit %d reached: the chains still solving were stopped\n", opt->max_rounds);
    if (trace)
        std::fprintf(stderr, "[socp_chains] set-up %.1f ms, host advance %.1f ms, launches + wait (Jacobian rounds) %.1f ms, Jacobian read-back + scatter %.1f ms, "
                             "total %.1f ms; %lld rounds, %lld Jacobians launched, %lld from cached rows, %lld solver restarts\n",
                     t_setup, t_adv, t_gpu, t_copy, ms_since(t_begin), rounds, jac_launched, jac_from_cache, restarts);
    if (stats) {
        stats->rounds = rounds; stats->jacobians_launched = jac_launched; stats->jacobians_from_cache = jac_from_cache;
        stats->speculative_rounds = spec_rows_rounds; stats->restarts = restarts; stats->wall_ms = ms_since(t_begin);
    }
    (void)hipStreamSynchronize(aux);
    (void)hipStreamDestroy(aux);
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
