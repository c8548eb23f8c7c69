// kernels_solver.hip -- the Powell hybrid iteration of many small problems on the device (solver_dev.hpp): one workgroup per
// problem.  MUST be compiled with -ffp-contract=off: every iterate has to equal the host solver's (minpack.cpp) bit for bit.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "solver_launch.hpp"

namespace socp {
namespace devsolver {

namespace {

__global__ void start_kernel(Config c, State *states, double *ws, long ws_stride, const int *__restrict__ list, const double *__restrict__ X0)
{
    const int p = list[blockIdx.x];
    BlockExec ex;
    start(ex, c, states[p], ws + (long)p * ws_stride, X0 + (long)blockIdx.x * c.n);
}

// LDS: eight vectors of n doubles every thread reads again and again (solver_dev.hpp: Work::f) and, optionally, the matrix of
// the problem in hand for the factor work (lds_matrix_doubles > 0).  A workgroup takes problems blockIdx.x,
// blockIdx.x + gridDim.x, ... one after the other (the grid may be capped, launch_advance).
// MAXT: the workgroup size the instantiation is built for -- without it the compiler budgets registers for 1024 threads
// (128 VGPRs) and spills the rest of this large function to scratch.
#ifdef SOCP_SOLVER_WAVES          // A/B: cap the registers so that this many wavefronts fit a SIMD (the compiler then spills)
#define SOCP_SOLVER_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SOCP_SOLVER_WAVES + 0 * WPE, SOCP_SOLVER_WAVES)))
#else
// FOUR wavefronts per SIMD (128 registers) for the instantiations that can have them.  Left alone the function takes 163-171
// registers (three wavefronts, or silently two); its trial rounds are serial chains that wait on memory, so what
// a SIMD gets done is proportional to the wavefronts it holds, and the two dozen registers the compiler spills cost less than the
// fourth wavefront brings -- measured (scripts/probes/solver_waves_ab.sh; 3 / 4 / 5 / 6 wavefronts): config 5, 2048 starts 0.0234 /
// 0.0221 / 0.0222 / 0.0222 s; KD chains 4096: 0.0340 / 0.0312 / 0.0316 / 0.0320 s, 16 384: 0.096 / 0.093 / 0.091 / 0.089 s;
// M = 9, 10 steps: 0.227 / 0.2215 / 0.225 / 0.229 s.
// The launches that hold problems with a fresh Jacobian keep three: their order-preserving factor sweeps are bandwidth, and the
// spills cost them 7 % (2048 x (n = 253), bit-equal solver: 0.080 -> 0.086 s with four).  WPE: 4 = trial launches, 3 = factor launches.
#define SOCP_SOLVER_OCCUPANCY __attribute__((amdgpu_waves_per_eu(MAXT <= 256 ? WPE : (MAXT <= 512 ? 2 : 4))))
#endif
template <int MAXT, int WPE, bool RING = false, bool FACTOR = true>
__global__ __launch_bounds__(MAXT) SOCP_SOLVER_OCCUPANCY void advance_kernel(Config c, State *states, double *ws, long ws_stride, const int *__restrict__ list,
                                                       const int *__restrict__ flags, int count, int lds_matrix_doubles)
{
    extern __shared__ double lds[];
    // RING: rows of R fetched ahead of the serial chains (solver_dev.hpp: ring_fetch).  For ONE-WAVEFRONT workgroups of LARGE problems
    // only -- launches that were shrunk until every problem is resident (launch_advance) with n > 128: a thread owns three or four
    // entries of a row there and a step's trip to memory is what the problem waits for (config 5, 2048 / 16 384 starts: -18 % / -13 %
    // of the sweep against the round-4 library on one box).  Everything else keeps the plain loops: with one or two entries per
    // thread the ring's bookkeeping -- clamped loads, dump stores, the idle steps of a group -- costs more than the wait it removes
    // (KD chains and M = 6 sweeps, n = 85: +10 ... +16 % with rings of any shape; profiles/r05_r04_vs_now.txt).
    // FACTOR = false: the builds of the TRIAL launches (their fresh Jacobians, if any, come from a factor kernel: launch_advance) carry no
    // factorisation
    using Rows = std::conditional_t<RING, BlockExecRing<4, 4>, BlockExec>;
    using Exec = std::conditional_t<FACTOR, Rows, NoInlineFactor<Rows>>;
    Exec ex;
    for (int b = blockIdx.x; b < count; b += gridDim.x) {
        const int p = list[b];
        Machine<Exec> m(ex, c, states[p], ws + (long)p * ws_stride, lds);
        if (lds_matrix_doubles > 0) m.fast_matrix = lds + 8 * (long)c.n;
        m.advance(flags ? flags[b] : 0);
        __syncthreads();
    }
}

// The factor work of problems that have just received a Jacobian, blocked through LDS (solver_dev.hpp: factor_blocked): ONE
// wavefront per problem, a thread per column of the 64-column block in hand; LDS = (8 + 64) n doubles.  Leaves A = Q, r, qtf,
// diag(R), the column norms and the "singular" flag in the problem's workspace / state; the advance kernel then skips its own
// factorisation (State::pad).
__global__ __launch_bounds__(64) void factor_kernel(Config c, State *states, double *ws, long ws_stride, const int *__restrict__ list, int count)
{
    extern __shared__ double lds[];
    BlockExec ex;
    for (int b = blockIdx.x; b < count; b += gridDim.x) {
        const int p = list[b];
        Work w(ws + (long)p * ws_stride, c.n, c.ld, lds);
        const bool sing = factor_blocked(ex, c.n, c.ld, w, lds, lds + (long)c.n * kPanel);
        __syncthreads();
        if (threadIdx.x == 0) { states[p].sing = sing ? 1 : 0; states[p].pad = 1; }
        __syncthreads();
    }
}

// The order-preserving factor work alone (what advance_kernel does with a fresh Jacobian), for socp_qr_factor_batch
template <int MAXT>
__global__ __launch_bounds__(MAXT) void factor_only_kernel(Config c, State *states, double *ws, long ws_stride, const int *__restrict__ list, int count)
{
    extern __shared__ double lds[];
    BlockExec ex;
    for (int b = blockIdx.x; b < count; b += gridDim.x) {
        const int p = list[b];
        Work w(ws + (long)p * ws_stride, c.n, c.ld, lds);
        const bool sing = factor(ex, c.n, c.ld, w);
        __syncthreads();
        if (threadIdx.x == 0) { states[p].sing = sing ? 1 : 0; states[p].pad = 1; }
        __syncthreads();
    }
}

__global__ void gather_eval_kernel(Config c, const State *states, double *ws, long ws_stride, const int *__restrict__ list, double *__restrict__ dst)
{
    const int p = list[blockIdx.x];
    const Work w(ws + (long)p * ws_stride, c.n, c.ld);
    const double *src = states[p].eval_sel ? w.wa2 : w.x;
    for (int j = threadIdx.x; j < c.n; j += blockDim.x) dst[(long)blockIdx.x * c.n + j] = src[j];
}

// src_stride: doubles between the results of consecutive problems (n for a residual batch; (n + 1) n when the evaluation ran as a
// whole forward-difference batch and F is row 0 of the problem's block)
__global__ void scatter_fvec_kernel(Config c, const State *states, double *ws, long ws_stride, const int *__restrict__ list, const double *__restrict__ src,
                                    long src_stride)
{
    const int p = list[blockIdx.x];
    const Work w(ws + (long)p * ws_stride, c.n, c.ld);
    double *dst = states[p].eval_sel ? w.wa4 : w.fvec;
    for (int j = threadIdx.x; j < c.n; j += blockDim.x) dst[j] = src[(long)blockIdx.x * src_stride + j];
}

// dst[dst_idx[k]][0 .. len) = src[src_idx[k]][0 .. len) (an index list may be null: block k): moves the (n + 1) x n row blocks of
// forward-difference batches between a round's staging area and the chains' cache slots in one launch
__global__ void copy_blocks_kernel(const double *__restrict__ src, const int *__restrict__ src_idx, double *__restrict__ dst,
                                   const int *__restrict__ dst_idx, int count, long len)
{
    const int k = blockIdx.y;
    if (k >= count) return;
    const double *s = src + (long)(src_idx ? src_idx[k] : k) * len;
    double *d = dst + (long)(dst_idx ? dst_idx[k] : k) * len;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += (long)gridDim.x * blockDim.x) d[e] = s[e];
}

__global__ void gather_jac_kernel(Config c, double *ws, long ws_stride, const int *__restrict__ list, double *__restrict__ dX, double *__restrict__ dF)
{
    const int p = list[blockIdx.x];
    const Work w(ws + (long)p * ws_stride, c.n, c.ld);
    for (int j = threadIdx.x; j < c.n; j += blockDim.x) {
        dX[(long)blockIdx.x * c.n + j] = w.x[j];
        dF[(long)blockIdx.x * c.n + j] = w.fvec[j];
    }
}

__global__ void gather_result_kernel(Config c, double *ws, long ws_stride, const int *__restrict__ list, double *__restrict__ out)
{
    const int p = list[blockIdx.x];
    const Work w(ws + (long)p * ws_stride, c.n, c.ld);
    for (int j = threadIdx.x; j < c.n; j += blockDim.x) {
        out[(long)blockIdx.x * 2 * c.n + j] = w.x[j];
        out[(long)blockIdx.x * 2 * c.n + c.n + j] = w.fvec[j];
    }
}

// A[i][j] (row-major, stride ld) = J[i + n j] (column-major): 32 x 32 tiles through LDS so that both sides are accessed along
// their contiguous direction.  grid = (tiles, tiles, count), block = (32, 8)
__global__ void scatter_jac_kernel(Config c, double *ws, long ws_stride, const int *__restrict__ list, const double *__restrict__ J)
{
    __shared__ double tile[32][33];
    const int p = list[blockIdx.z], n = c.n;
    const double *src = J + (long)blockIdx.z * n * n;
    double *A = ws + (long)p * ws_stride;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    for (int jj = threadIdx.y; jj < 32; jj += 8) {           // read: consecutive threads along i (contiguous in J)
        const int i = i0 + threadIdx.x, j = j0 + jj;
        if (i < n && j < n) tile[jj][threadIdx.x] = src[i + (long)n * j];
    }
    __syncthreads();
    for (int ii = threadIdx.y; ii < 32; ii += 8) {           // write: consecutive threads along j (contiguous in A)
        const int i = i0 + ii, j = j0 + threadIdx.x;
        if (i < n && j < n) A[(long)i * c.ld + j] = tile[threadIdx.x][ii];
    }
}

// More than 64 KB of dynamic LDS needs the kernel's limit raised -- on the CURRENT device's copy of the kernel, so the fact is
// remembered per device (socp_sweep_solve drives several devices from one process).
template <auto Kernel>
hipError_t raise_lds_limit()
{
    static std::atomic<unsigned long long> raised{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && ((raised.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && dev < 64) raised.fetch_or(1ull << dev, std::memory_order_release);
    return e;
}

}  // namespace

int threads_for(int n)
{
    const int t = ((n + 1 + 63) / 64) * 64;
    return t > 1024 ? 1024 : t;
}

hipError_t launch_start(hipStream_t st, const PoolDev &pool, const int *d_list, int count, const double *d_X0)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(start_kernel, dim3(count), dim3(64), 0, st, pool.cfg, pool.states, pool.ws, pool.ws_stride, d_list, d_X0);
    return hipGetLastError();
}

hipError_t launch_advance(hipStream_t st, const PoolDev &pool, const int *d_list, int count, const int *d_flags, bool factor_phase)
{
    if (count <= 0) return hipSuccess;
    const int n = pool.cfg.n;
    const size_t matrix_bytes = sizeof(double) * (size_t)n * pool.cfg.ld, vector_bytes = sizeof(double) * 8 * (size_t)n;
    // Launches whose problems all hold a fresh Jacobian (factor_phase) CAN run the factor work on an LDS copy of the matrix when
    // that fits beside the vectors (SOCP_SOLVER_LDS_BYTES=65536: n <= 85) -- a refresh streams its matrix ~2n/3 times.  Off by
    // default: measured on 4096 problems of n = 85 the in-place refresh takes 8.4 ms (at HBM bandwidth, 4.8 TB/s) and the LDS
    // form no less (KD chains 51 -> 58 ms of solver time): 60 KB of LDS per problem leaves 512 problems in flight instead of
    // 4096, and one problem alone is bound by the latency of its serial chains, not by bandwidth.
    static const long lds_limit = [] { const char *e = std::getenv("SOCP_SOLVER_LDS_BYTES"); return e ? std::atol(e) : 0L; }();
    const int lds_matrix = (factor_phase && (long)(matrix_bytes + vector_bytes) <= lds_limit) ? (int)((size_t)n * pool.cfg.ld) : 0;
    // Optional (SOCP_SOLVER_MAX_GROUPS): cap on the problems in flight (a workgroup then takes several in turn).  Measured on
    // the 2048 x (n = 253) sweep: 256 in flight (matrices within the last-level cache) took twice as long as all 2048 at once.
    static const long cap = [] { const char *e = std::getenv("SOCP_SOLVER_MAX_GROUPS"); return e ? std::atol(e) : 0L; }();
    const unsigned grid = (unsigned)((cap > 0 && cap < count) ? cap : count);
    const size_t lds_bytes = vector_bytes + (lds_matrix ? matrix_bytes : 0);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;           // n > 2560: the caller keeps such problems on the host
    int threads = threads_for(n);
    {
        // Workgroup size.  The iteration is the same for any size (a thread then owns several columns; same numbers).  Launches of
        // problems with a fresh Jacobian keep a thread per column: the factor sweeps are bandwidth, and fewer threads per problem
        // measured 20-25 % slower.  The other launches (trial steps: dogleg, Broyden update -- latency chains with a barrier per
        // step) shrink their workgroups when the launch does not fit the chip at full size, so that every problem is resident at
        // once instead of queueing behind a workgroup that mostly waits at barriers: measured 5-9 % of the solver time of
        // 4096 x (n = 85 / 127) and 2048 x (n = 253); a launch that fits keeps the full size (256 x (n = 253) is 5 % slower at 64).
        // SOCP_SOLVER_THREADS_FACTOR / SOCP_SOLVER_THREADS_TRIAL (multiples of 64) override both (A/B, tests).
        static const bool fit_rule = [] { const char *f = std::getenv("SOCP_SOLVER_FIT"); return !(f && f[0] == '0'); }();     // (A/B: 0 = a thread per column)
        const char *e = std::getenv(factor_phase ? "SOCP_SOLVER_THREADS_FACTOR" : "SOCP_SOLVER_THREADS_TRIAL");
        const long want = e ? std::atol(e) : 0L;
        if (want >= 64 && want % 64 == 0) {
            if (want < threads) threads = (int)want;
        } else if (!factor_phase && threads > 64 && fit_rule) {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
                const long slots = (long)cus * 4 * 3;                        // (three per SIMD: the rule was measured with that; with the four the
                                                                             // kernel is now built for, 2048 x (n = 253) would get 128 threads -- 5 % slower)
                const long waves = slots / count;                            // ... per problem, if all are to be resident
                const int fit = (int)(waves < 1 ? 1 : waves) * 64;
                if (fit < threads) threads = fit;
            }
        }
    }
#define SOCP_LAUNCH_ADVANCE_W(MAXT, WPE, FACTOR)                                                                                           \
    do {                                                                                                                                   \
        if (lds_bytes > 65536) {                                                                                                           \
            const hipError_t raised = raise_lds_limit<advance_kernel<MAXT, WPE, false, FACTOR>>();                                         \
            if (raised != hipSuccess) return raised;                                                                                       \
        }                                                                                                                                  \
        hipLaunchKernelGGL((advance_kernel<MAXT, WPE, false, FACTOR>), dim3(grid), dim3(threads), lds_bytes, st, pool.cfg, pool.states,    \
                           pool.ws, pool.ws_stride, d_list, d_flags, count, lds_matrix);                                                   \
    } while (0)
    // The TRIAL launches keep round 4's four wavefronts per SIMD -- except the one-wavefront workgroups of large problems (n > 128):
    // those get the row rings and 256 registers (two per SIMD: a 2048-problem launch has no more wavefronts than that anyway, and
    // the rings are what hides their memory latency now): config 5, 2048 starts 0.0203 -> 0.0185 s against four
    // (profiles/r05_trial_wpe_ab.txt).
#define SOCP_LAUNCH_ADVANCE(MAXT)                                                                                                          \
    do {                                                                                                                                   \
        if (factor_phase) SOCP_LAUNCH_ADVANCE_W(MAXT, 3, true);                                                                            \
        else SOCP_LAUNCH_ADVANCE_W(MAXT, 4, false);                                                                                        \
    } while (0)
    if (!factor_phase && threads <= 64 && n > 128) {
        if (lds_bytes > 65536) {
            const hipError_t raised = raise_lds_limit<advance_kernel<64, 2, true, false>>();
            if (raised != hipSuccess) return raised;
        }
        hipLaunchKernelGGL((advance_kernel<64, 2, true, false>), dim3(grid), dim3(threads), lds_bytes, st, pool.cfg, pool.states, pool.ws, pool.ws_stride, d_list, d_flags,
                           count, lds_matrix);
        return hipGetLastError();
    }
    if (threads <= 64) SOCP_LAUNCH_ADVANCE(64);
    else if (threads <= 128) SOCP_LAUNCH_ADVANCE(128);
    else if (threads <= 256) SOCP_LAUNCH_ADVANCE(256);
    else if (threads <= 512) SOCP_LAUNCH_ADVANCE_W(512, 3, true);
    else SOCP_LAUNCH_ADVANCE_W(1024, 3, true);
#undef SOCP_LAUNCH_ADVANCE
#undef SOCP_LAUNCH_ADVANCE_W
    return hipGetLastError();
}

// Whether the refreshes of this pool go through the blocked factor kernel.  OFF by default (SOCP_SOLVER_BLOCKED_MIN_N=<n> turns
// it on from that size up, where panel + block fit a CU's LDS: n <= 284): it removes the re-reads -- one read and one write of
// the trailing matrix per panel of 8 reflectors instead of three passes per reflector -- but a CU's LDS holds ONE 64-column
// block at n = 253, i.e. one wavefront per CU, and a lone wavefront cannot hide the latency of its own LDS reads and of the
// dependent FP64 adds of an order-preserving dot product.  Measured, 2048 problems of n = 253: 142 ms of solver kernels
// against 102 ms in place (256 problems: 20.3 against 18.8 ms); n = 127: 37 against 29 ms; n = 85: 65 against 58 ms.  The
// in-place form streams 200 x the algorithmic bytes but does it with 8192 wavefronts in flight, at cache + HBM bandwidth.
bool blocked_factor_applies(int n)
{
    const char *e = std::getenv("SOCP_SOLVER_BLOCKED_MIN_N");        // read at every call: tests switch it within one process
    const long min_n = e ? std::atol(e) : 0L;
    return min_n > 0 && n >= min_n && (size_t)blocked_lds_doubles(n) * sizeof(double) <= 160 * 1024;
}

hipError_t launch_factor(hipStream_t st, const PoolDev &pool, const int *d_list, int count)
{
    if (count <= 0) return hipSuccess;
    const size_t lds_bytes = (size_t)blocked_lds_doubles(pool.cfg.n) * sizeof(double);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    if (lds_bytes > 65536) {
        const hipError_t raised = raise_lds_limit<factor_kernel>();
        if (raised != hipSuccess) return raised;
    }
    hipLaunchKernelGGL(factor_kernel, dim3(count), dim3(64), lds_bytes, st, pool.cfg, pool.states, pool.ws, pool.ws_stride, d_list, count);
    return hipGetLastError();
}

hipError_t launch_factor_exact(hipStream_t st, const PoolDev &pool, const int *d_list, int count)
{
    if (count <= 0) return hipSuccess;
    const size_t lds_bytes = sizeof(double) * 8 * (size_t)pool.cfg.n;
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    const int threads = threads_for(pool.cfg.n);
#define SOCP_LAUNCH_FACTOR_ONLY(MAXT)                                                                                                     \
    do {                                                                                                                                   \
        if (lds_bytes > 65536) {                                                                                                           \
            const hipError_t raised = raise_lds_limit<factor_only_kernel<MAXT>>();                                                         \
            if (raised != hipSuccess) return raised;                                                                                       \
        }                                                                                                                                  \
        hipLaunchKernelGGL(factor_only_kernel<MAXT>, dim3((unsigned)count), dim3(threads), lds_bytes, st, pool.cfg, pool.states, pool.ws, \
                           pool.ws_stride, d_list, count);                                                                                 \
    } while (0)
    if (threads <= 64) SOCP_LAUNCH_FACTOR_ONLY(64);
    else if (threads <= 128) SOCP_LAUNCH_FACTOR_ONLY(128);
    else if (threads <= 256) SOCP_LAUNCH_FACTOR_ONLY(256);
    else if (threads <= 512) SOCP_LAUNCH_FACTOR_ONLY(512);
    else SOCP_LAUNCH_FACTOR_ONLY(1024);
#undef SOCP_LAUNCH_FACTOR_ONLY
    return hipGetLastError();
}

// -DSOCP_SOLVER_PROFILE: the per-phase tick totals of solver_dev.hpp's Prof (zeros otherwise); reset = start a new measurement
hipError_t read_profile(unsigned long long out[16], bool reset)
{
    for (int k = 0; k < 16; k++) out[k] = 0;
#ifdef SOCP_SOLVER_PROFILE
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16);
    if (e != hipSuccess) return e;
    if (reset) {
        const unsigned long long zero[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof(zero));
    }
    return e;
#else
    (void)reset;
    return hipSuccess;
#endif
}

hipError_t launch_gather_eval(hipStream_t st, const PoolDev &pool, const int *d_list, int count, double *d_dst)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_eval_kernel, dim3(count), dim3(64), 0, st, pool.cfg, pool.states, pool.ws, pool.ws_stride, d_list, d_dst);
    return hipGetLastError();
}

hipError_t launch_scatter_fvec(hipStream_t st, const PoolDev &pool, const int *d_list, int count, const double *d_src, long src_stride)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_fvec_kernel, dim3(count), dim3(64), 0, st, pool.cfg, pool.states, pool.ws, pool.ws_stride, d_list, d_src,
                       src_stride > 0 ? src_stride : (long)pool.cfg.n);
    return hipGetLastError();
}

hipError_t launch_copy_blocks(hipStream_t st, const double *src, const int *d_src_idx, double *dst, const int *d_dst_idx, int count, long len)
{
    if (count <= 0) return hipSuccess;
    const unsigned gx = (unsigned)std::min<long>(8, (len + 255) / 256);
    for (int k0 = 0; k0 < count; k0 += 32768) {                  // grid.y is limited to 65535
        const int kc = std::min(32768, count - k0);
        hipLaunchKernelGGL(copy_blocks_kernel, dim3(gx, (unsigned)kc), dim3(256), 0, st, src + (d_src_idx ? 0 : (long)k0 * len), d_src_idx ? d_src_idx + k0 : nullptr,
                           dst + (d_dst_idx ? 0 : (long)k0 * len), d_dst_idx ? d_dst_idx + k0 : nullptr, kc, len);
    }
    return hipGetLastError();
}

hipError_t launch_gather_jac(hipStream_t st, const PoolDev &pool, const int *d_list, int count, double *d_X, double *d_F)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_jac_kernel, dim3(count), dim3(64), 0, st, pool.cfg, pool.ws, pool.ws_stride, d_list, d_X, d_F);
    return hipGetLastError();
}

hipError_t launch_scatter_jac(hipStream_t st, const PoolDev &pool, const int *d_list, int count, const double *d_J)
{
    if (count <= 0) return hipSuccess;
    const unsigned tiles = (unsigned)((pool.cfg.n + 31) / 32);
    for (int k0 = 0; k0 < count; k0 += 32768) {              // grid.z is limited to 65535
        const int kc = count - k0 < 32768 ? count - k0 : 32768;
        hipLaunchKernelGGL(scatter_jac_kernel, dim3(tiles, tiles, (unsigned)kc), dim3(32, 8), 0, st, pool.cfg, pool.ws, pool.ws_stride, d_list + k0,
                           d_J + (long)k0 * pool.cfg.n * pool.cfg.n);
    }
    return hipGetLastError();
}

// what the host needs to know about the problems it has just advanced: six integers each, in list order (the whole State array --
// 96 B per problem, advanced or not -- was 400 MB per pass for 4 M chains)
__global__ __launch_bounds__(256) void gather_status_kernel(const State *__restrict__ states, const int *__restrict__ list, int count, Status *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const State &s = states[list[k]];
    Status r;
    r.req = s.req; r.iter = s.iter; r.eval_sel = s.eval_sel; r.info = s.info; r.nfev = s.nfev; r.njev = s.njev;
    out[k] = r;
}

hipError_t launch_gather_status(hipStream_t st, const PoolDev &pool, const int *d_list, int count, Status *d_out)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_status_kernel, dim3((count + 255) / 256), dim3(256), 0, st, pool.states, d_list, count, d_out);
    return hipGetLastError();
}

hipError_t launch_gather_result(hipStream_t st, const PoolDev &pool, const int *d_list, int count, double *d_out)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_result_kernel, dim3(count), dim3(64), 0, st, pool.cfg, pool.ws, pool.ws_stride, d_list, d_out);
    return hipGetLastError();
}

}  // namespace devsolver
}  // namespace socp
