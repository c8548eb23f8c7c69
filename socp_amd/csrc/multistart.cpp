// multistart.cpp -- lock-step Newton solves of many independent starts of ONE shooting problem
// (BASELINE config 4): every start owns a resumable hybrd state machine
// (minpack.cpp); per round all pending residual requests are ONE residual launch and all pending
// Jacobian requests ONE forward-difference launch, instead of P serial host solvers each launching
// 15-trajectory batches.  Each start follows exactly the iterates it would follow alone.
#include "../../include/socp_hip.h"
#include "../../include/socp_solver.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>

namespace {
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool reserve(size_t bytes)
    {
        if (bytes <= cap) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return false;
        cap = bytes;
        return true;
    }
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    double *d() const { return static_cast<double *>(p); }
};
struct DevBuf {
    void *p = nullptr;
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess; }
    ~DevBuf() { if (p) (void)hipFree(p); }
    double *d() const { return static_cast<double *>(p); }
};
}  // namespace

extern "C" int socp_multistart_solve(socp_ctx *ctx, int P, const double *Z0, double xtol, int maxfev, double epsfcn,
                                     double factor, int dedup, double *Zout, int *info, int *nfev, double *fnorm,
                                     long long *rounds_out)
{
    if (!ctx || P < 0 || (P > 0 && (!Z0 || !Zout || !info))) return SOCP_ERR_ARG;
    const int n = socp_problem_num_param(ctx);
    if (n <= 0) return SOCP_ERR_ARG;
    if (P == 0) return SOCP_OK;
    // every allocation, copy and stream below must live on the context's device, whatever the calling thread's
    // current device is (a fresh std::thread starts on device 0); the caller's device is restored on exit
    struct DeviceGuard {
        int prev = -1;
        bool ok = false;
        explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } device_guard(socp_ctx_device(ctx));
    if (!device_guard.ok) return SOCP_ERR_HIP;

    const std::chrono::steady_clock::time_point t_begin = std::chrono::steady_clock::now();
    // host threads for the per-start work (state-machine advances, workspace set-up, Jacobian scatter): all cores
    // up to 16, one thread for small sweeps (thread start-up is ~50 us each)
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = ((long)P * n * n < 200000) ? 1 : (int)std::max(1u, std::min(16u, hw ? hw : 1u));
    auto parallel_for = [&](int count, auto &&body) {
        if (nthreads <= 1) { for (int k = 0; k < count; k++) body(k); return; }
        std::vector<std::thread> pool;
        pool.reserve(nthreads);
        for (int t = 0; t < nthreads; t++)
            pool.emplace_back([&, t]() { for (int k = t; k < count; k += nthreads) body(k); });
        for (std::thread &th : pool) th.join();
    };

    // one solver workspace per start (~n^2 doubles each: 400 MB of first-touch pages at P = 4096, n = 85)
    std::vector<socp_hybr *> solver(P, nullptr);
    parallel_for(P, [&](int p) {
        solver[p] = socp_hybr_create(n, xtol, maxfev, epsfcn, 1, factor, 0);
        if (solver[p]) socp_hybr_start(solver[p], Z0 + (size_t)p * n, nullptr);
    });
    auto cleanup = [&]() { parallel_for(P, [&](int p) { socp_hybr_destroy(solver[p]); solver[p] = nullptr; }); };
    for (int p = 0; p < P; p++)
        if (!solver[p]) { cleanup(); return SOCP_ERR_ARG; }

    const size_t rowB = sizeof(double) * n, jacB = rowB * n;
    // Jacobian blocks are P*n*n doubles: stage them in chunks of at most ~64 MiB (pinned + device)
    const int jchunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)P, ((size_t)64 << 20) / jacB));
    PinnedBuf hX, hF, hJx, hJf, hJ;
    DevBuf dX, dF, dJx, dJf, dJ;
    if (!hX.reserve(rowB * P) || !hF.reserve(rowB * P) || !hJx.reserve(rowB * P) || !hJf.reserve(rowB * P) || !hJ.reserve(jacB * jchunk) ||
        !dX.alloc(rowB * P) || !dF.alloc(rowB * P) || !dJx.alloc(rowB * P) || !dJf.alloc(rowB * P) || !dJ.alloc(jacB * jchunk)) {
        cleanup();
        return SOCP_ERR_HIP;
    }

    static const bool trace = std::getenv("SOCP_MULTISTART_TRACE") != nullptr;
    static const bool overlap = [] { const char *e = std::getenv("SOCP_MULTISTART_OVERLAP"); return !(e && e[0] == '0'); }();   // =0: one stream (A/B)
    // Residual requests and Jacobian requests of one round are independent launches, each one trajectory latency
    // long: the residual batch goes to a second stream so the two overlap (rounds with both kinds are about half
    // of a sweep's rounds once the starts fall out of step).
    void *main_stream = nullptr;
    hipStream_t aux = nullptr;
    if (socp_ctx_synchronize(ctx) != SOCP_OK || socp_ctx_get_stream(ctx, &main_stream) != SOCP_OK ||
        hipStreamCreateWithFlags(&aux, hipStreamNonBlocking) != hipSuccess) {
        cleanup();
        return SOCP_ERR_HIP;
    }
    std::vector<int> flag(P, 0), reqF, reqJ, req(P, SOCP_REQ_DONE);
    std::vector<const double *> xin(P, nullptr);
    std::vector<double *> xout(P, nullptr);
    std::vector<double *> outF(P), outJ(P);
    std::vector<char> active(P, 1);
    long long rounds = 0;
    int rc = SOCP_OK;
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    double t_adv = 0, t_gpu = 0, t_copy = 0;
    const double t_setup = ms_since(t_begin);
    for (;;) {
        const clk::time_point ta = clk::now();
        reqF.clear(); reqJ.clear();
        // (1) advance every active state machine -- QR, dogleg, Broyden update: O(n^2)..O(n^3) host work per start,
        //     independent between starts, so it is spread over the host cores
        parallel_for(P, [&](int p) {
            if (!active[p]) return;
            req[p] = socp_hybr_advance(solver[p], flag[p], &xin[p], &xout[p]);
            flag[p] = 0;
        });
        // (2) gather the requests into the staging buffers in start order (keeps the batches deterministic)
        for (int p = 0; p < P; p++) {
            if (!active[p]) continue;
            if (req[p] == SOCP_REQ_DONE) { active[p] = 0; continue; }
            if (req[p] == SOCP_REQ_FVEC) {
                std::memcpy(hX.d() + (size_t)reqF.size() * n, xin[p], rowB);
                outF[reqF.size()] = xout[p];
                reqF.push_back(p);
            } else {
                std::memcpy(hJx.d() + (size_t)reqJ.size() * n, xin[p], rowB);
                std::memcpy(hJf.d() + (size_t)reqJ.size() * n, socp_hybr_fvec(solver[p]), rowB);
                outJ[reqJ.size()] = xout[p];
                reqJ.push_back(p);
            }
        }
        t_adv += ms_since(ta);
        if (reqF.empty() && reqJ.empty()) break;
        rounds++;
        const clk::time_point tg = clk::now();
        const int kF = (int)reqF.size(), kJ = (int)reqJ.size();
        if (trace) std::fprintf(stderr, "[socp_multistart] round %lld: %d residual requests, %d Jacobian requests\n", rounds, kF, kJ);
        // the residual launch and the first Jacobian chunk are enqueued before either result is awaited
        if (kF) {
            if (hipMemcpy(dX.p, hX.p, rowB * kF, hipMemcpyHostToDevice) != hipSuccess) { rc = SOCP_ERR_HIP; break; }
            if (overlap) socp_ctx_set_stream(ctx, aux, 0);
            rc = socp_residual_batch_dev(ctx, kF, dX.d(), dF.d());
            if (overlap) socp_ctx_set_stream(ctx, main_stream, 0);
            if (rc != SOCP_OK) break;
        }
        if (kJ) {
            if (hipMemcpy(dJx.p, hJx.p, rowB * kJ, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dJf.p, hJf.p, rowB * kJ, hipMemcpyHostToDevice) != hipSuccess) { rc = SOCP_ERR_HIP; break; }
        }
        bool f_collected = (kF == 0);
        for (int j0 = 0; j0 < kJ || !f_collected; j0 += jchunk) {
            const int kc = j0 < kJ ? std::min(jchunk, kJ - j0) : 0;
            if (kc && (rc = socp_fd_jacobian_multi_dev(ctx, kc, dJx.d() + (size_t)j0 * n, dJf.d() + (size_t)j0 * n, epsfcn,
                                                        dJ.d(), dedup)) != SOCP_OK) break;
            if (!f_collected) {
                if (hipStreamSynchronize(overlap ? aux : static_cast<hipStream_t>(main_stream)) != hipSuccess ||
                    hipMemcpy(hF.p, dF.p, rowB * kF, hipMemcpyDeviceToHost) != hipSuccess) { rc = SOCP_ERR_HIP; break; }
                for (int k = 0; k < kF; k++) std::memcpy(outF[k], hF.d() + (size_t)k * n, rowB);
                f_collected = true;
            }
            if (kc) {
                if ((rc = socp_ctx_synchronize(ctx)) != SOCP_OK) break;
                t_gpu += ms_since(tg);
                const clk::time_point tc = clk::now();
                if (hipMemcpy(hJ.p, dJ.p, jacB * kc, hipMemcpyDeviceToHost) != hipSuccess) { rc = SOCP_ERR_HIP; break; }
                // hundreds of MB per round at n ~ 100: spread the copies into the solvers' own buffers over the host threads
                parallel_for(kc, [&](int k) { std::memcpy(outJ[j0 + k], hJ.d() + (size_t)k * n * n, jacB); });
                t_copy += ms_since(tc);
            }
        }
        if (rc != SOCP_OK) break;
    }
    if (rc == SOCP_OK) {
        for (int p = 0; p < P; p++) {
            std::memcpy(Zout + (size_t)p * n, socp_hybr_x(solver[p]), rowB);
            info[p] = socp_hybr_info(solver[p]);
            if (nfev) nfev[p] = socp_hybr_nfev(solver[p]);
            if (fnorm) {
                const double *f = socp_hybr_fvec(solver[p]);
                double s = 0;
                for (int i = 0; i < n; i++) s += f[i] * f[i];
                fnorm[p] = std::sqrt(s);
            }
        }
    }
    if (trace) std::fprintf(stderr, "[socp_multistart] set-up %.1f ms, host advance %.1f ms, launches + wait (Jacobian rounds) %.1f ms, Jacobian read-back + scatter %.1f ms, total %.1f ms\n", t_setup, t_adv, t_gpu, t_copy, ms_since(t_begin));
    if (rounds_out) *rounds_out = rounds;
    (void)hipStreamSynchronize(aux);
    (void)hipStreamDestroy(aux);
    cleanup();
    return rc;
}
