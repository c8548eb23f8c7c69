// solver_dev.hpp -- the Powell hybrid iteration (MINPACK hybrd / hybrj as minpack.cpp states it) for MANY small problems
// on the device: one workgroup per problem, one thread per column / row / vector element.
//
// Why: a lock-step sweep of thousands of n = 85 .. 253 problems is host-bound (75-90 % of the wall time in
// host qrfac / dogleg / r1updt, and P n^2 doubles of Jacobians over PCIe per refresh).  With the state machines in HBM the
// Jacobians never leave the device and the factor work of all problems runs at once.
//
// What is kept: every number is produced by the same IEEE operations in the same order as in minpack.cpp (which equals
// SciPy's MINPACK bit for bit, tests/test_minpack.py), so a chain follows the same iterates whichever side advances it:
//   * MINPACK's algorithms are column / row / element recurrences that never mix (qrfac: reflector j applied to column k is
//     one dot product, one division, one axpy; qform likewise; r1updt / r1mpyq: one Givens rotation per step applied to
//     independent elements; dogleg / prered: one serial sum per row).  A thread owns a column (row, element) and runs ITS
//     recurrence in the serial order; only genuinely serial chains (enorm, back substitution, the rotation scalars) are
//     left serial -- they are computed redundantly by every thread from broadcast loads, which costs no wall time.
//   * no FMA contraction (the translation unit is compiled -ffp-contract=off), IEEE division and square root,
//     std::max / std::min written out as the comparisons they are (NaN behaviour included).
//
// The same source compiles for the host with ONE thread per problem (SOCP_SOLVER_HOST: tests/tools/solver_sim.cpp), which
// is how the arithmetic is checked against minpack.cpp without a GPU; races cannot show there -- the -m gpu tests compare
// the real kernels with the host solvers (tests/test_gpu_devsolver.py).
//
// Storage per problem (doubles): A[n][ld] ROW-major with ld >= n + 1 (thread = column => coalesced; column n carries fvec
// through the reflectors and comes out as Q^T fvec, as colvec::factor does in minpack.cpp), r[n (n+1) / 2] = R by rows,
// fourteen vectors of n.  After a factorisation A holds Q (row-major).
#pragma once
#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#define SOCP_HD __host__ __device__ __forceinline__
#else
#define SOCP_HD inline
#endif
#if defined(__HIPCC__)
#include "wave_reduce.hpp"            // wave_sum: the throughput flavour's parallel sums (Config::fast_sums)
#endif

namespace socp {
namespace devsolver {

enum Phase { PH_INIT = 0, PH_F0 = 1, PH_JAC = 2, PH_TRIAL = 3, PH_DONE = 4 };
enum Request { RQ_DONE = 0, RQ_FVEC = 1, RQ_JAC = 2 };

// configuration shared by every problem of a pool (socp_hybr_create's arguments)
struct Config {
    int n, ld, maxfev, mode, analytic;
    double xtol, epsfcn, factor;
    int lazy_q = 0;    // throughput flavour: Broyden's rotations are kept as a list and Q stays as factorised (see lazy_capacity)
    int fast_sums = 0; // throughput flavour: the back substitution's row sums are formed in parallel (dogleg), summation order free
    int gn_shortcut = 0;   // throughput flavour: the predicted reduction of a Gauss-Newton step is taken as what it is (after_trial)
};

// per-problem iteration state (Core of minpack.cpp)
struct State {
    int phase, iter, ncsuc, ncfail, nslow1, nslow2, nfev, njev, info, jeval, sing;
    int req;           // Request left pending by the last advance
    int eval_sel;      // RQ_FVEC: 0 -> evaluate at x, result to fvec; 1 -> evaluate at wa2 (trial point), result to wa4
    int pad;           // 1: the Jacobian in A has already been factorised by the factor kernel (sing holds its flag)
    int lazy;          // lazy_q: rank-1 updates since A was last brought up to date (their rotations are in the V area)
    int gn;            // the pending trial step is the Gauss-Newton step of a nonsingular R (Config::gn_shortcut reads it)
    double delta, xnorm, fnorm, pnorm;
};

// the part of it the lock-step engine reads back after every advance (kernels_solver.hip: gather_status_kernel)
struct Status {
    int req, iter, eval_sel, info, nfev, njev;
};

constexpr int kVectors = 16;
SOCP_HD long ws_doubles(int n, int ld) { return (((long)n * ld + (long)n * (n + 1) + (long)kVectors * n) + 7) / 8 * 8; }
SOCP_HD int ld_for(int n) { return (n + 1 + 7) / 8 * 8; }

// views into one problem's workspace
struct Work {
    double *A, *r, *x, *fvec, *diag, *qtf, *wa1, *wa2, *wa3, *wa4;
    double *V;       // the Householder vectors of the last factorisation, packed: v_k at V + row_off(n, k), n - k entries (factor_blocked)
    // eight "fast" vectors of n doubles for what every thread reads again and again (a column or row in hand, rotation
    // tables, copies whose norm is taken): LDS on the device, the tail of the workspace otherwise
    double *f[8];
    SOCP_HD Work(double *base, int n, int ld, double *fast = nullptr)
    {
        A = base; r = A + (long)n * ld;
        double *v = r + (long)n * (n + 1) / 2;
        x = v; fvec = v + n; diag = v + 2 * n; qtf = v + 3 * n; wa1 = v + 4 * n; wa2 = v + 5 * n; wa3 = v + 6 * n; wa4 = v + 7 * n;
#if defined(__HIP_DEVICE_COMPILE__)
        V = v + 16 * n;
        double *fb = fast;                                   // always LDS on the device: a select between an LDS and a global pointer
                                                             // would make every access to the fast vectors a FLAT one
#else
        V = v + 16 * n;
        double *fb = fast ? fast : v + 8 * n;
#endif
        for (int k = 0; k < 8; k++) f[k] = fb + (long)k * n;
    }
    // Device only: the 8 n doubles behind wa4 are the fast vectors' home on the HOST (LDS holds them on the device) and free here:
    // where the ringed loops of r1updt send the stores of lanes that own no entry of the row in hand (kRingE doubles per thread; the
    // rings take a problem only if that fits)
    SOCP_HD double *dump_area(int n) const { return x + 8 * (long)n; }
};

// executors: one thread of a workgroup (device) or the whole problem on one host thread (simulation)
#if defined(__HIPCC__)
struct BlockExec {
    int tid, nt;
    __device__ BlockExec() : tid((int)threadIdx.x), nt((int)blockDim.x) {}
    __device__ void sync() const { __syncthreads(); }
    // The first wavefront of the workgroup.  What every thread would compute alike from broadcast loads (a norm, the sum of a
    // back-substitution step) is computed THERE and handed on through LDS: repeated by every wavefront it costs nothing extra only
    // while each wavefront has a SIMD to itself -- with 14 wavefronts per problem (n = 832), or three workgroups sharing a CU,
    // the copies take each other's issue slots and the chain runs 3-4 x slower.
    __device__ bool leader() const { return tid < 64; }
    // A hand-over through LDS alone (the values a step of a serial chain needs from their owner threads): the wavefronts meet, LDS
    // operations are complete -- but the wave's outstanding GLOBAL loads stay in flight.  __syncthreads() also waits for those
    // (s_waitcnt vmcnt(0)), i.e. it drains the rows of R fetched ahead of the chain (row rings, below) at every step.  A workgroup of
    // one wavefront needs no barrier at all: its LDS operations execute in order.
    __device__ void sync_lds() const
    {
        if (nt <= 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // row rings (r1updt, the dogleg's back substitution): kRingE entries of a row per thread, kRingB rows in flight; 0 = none
    static constexpr int kRingE = 0, kRingB = 0;
    // whether Machine::after_jacobian carries the factorisation itself (false: the launch's fresh Jacobians have all been through a
    // factor kernel -- the trial launches' builds leave the code, and the registers it costs them, out)
    static constexpr bool kInlineFactor = true;
};
// the advance kernels' executor: a thread owns at most E entries of a row of R (n <= E * threads; larger problems take the plain loops)
template <int E, int B>
struct BlockExecRing : BlockExec {
    static constexpr int kRingE = E, kRingB = B;
};
template <class Base>
struct NoInlineFactor : Base {
    static constexpr bool kInlineFactor = false;
};
#endif
struct SerialExec {
    int tid = 0, nt = 1;
    SOCP_HD void sync() const {}
    SOCP_HD void sync_lds() const {}
    SOCP_HD bool leader() const { return true; }
    static constexpr int kRingE = 0, kRingB = 0;
    static constexpr bool kInlineFactor = true;
};

// Development aid (-DSOCP_SOLVER_PROFILE, device only): thread 0 of every workgroup adds the clock ticks between marks to
// per-phase totals (kernels_solver.hip: read_profile; printed with SOCP_MULTISTART_TRACE).  Compiled out otherwise.
#if defined(SOCP_SOLVER_PROFILE) && defined(__HIPCC__)
__device__ unsigned long long g_prof[16];
#endif
#if defined(SOCP_SOLVER_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
struct Prof {
    unsigned long long t;
    __device__ Prof() : t(clock64()) {}
    __device__ void mark(int slot, int tid)
    {
        const unsigned long long now = clock64();
        if (tid == 0) atomicAdd(&g_prof[slot], now - t);
        t = now;
    }
};
#else
struct Prof {
    SOCP_HD void mark(int, int) {}
};
#endif
enum { PF_TRIAL_HEAD = 0, PF_QTW = 1, PF_R1UPDT = 2, PF_R1MPYQ = 3, PF_DOGLEG = 4, PF_STEP_TAIL = 5, PF_FACTOR = 6, PF_JAC_TAIL = 7 };

// Elements are dealt out by ABSOLUTE index: element i belongs to thread i mod nt whatever the loop bounds are.  r1updt relies
// on it (a thread keeps "its" w[i] and s(., i) across rotation steps without a barrier), and a thread re-visits the same
// columns of A from reflector to reflector.
SOCP_HD int par_first_plain(int lo, int tid, int nt)
{
    const int m = (lo < nt) ? lo : lo % nt;
    const int d = tid - m;
    return lo + (d < 0 ? d + nt : d);
}
SOCP_HD int par_first(int lo, int tid, int nt)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // The thread index re-defined where a loop starts: every loop's first address is otherwise computed ONCE, at the top of the
    // kernel (the state machine's hundred loops are all invariant in the problem loop), and held -- or spilled -- from there on:
    // the four-wavefront builds kept 24 such addresses in scratch memory and 150 scalar values in the lanes of three more registers.
    asm volatile("" : "+v"(tid));
#endif
    const int m = (lo < nt) ? lo : lo % nt;                  // (one division at most, none in the common case n <= nt)
    const int d = tid - m;
    return lo + (d < 0 ? d + nt : d);
}
#define SOCP_PAR_FOR(i, lo, hi) for (int i = par_first((lo), ex.tid, ex.nt); i < (hi); i += ex.nt)
// (the loops of the order-preserving factorisation, two or three per reflector: there the addresses held from one reflector to the next
// are worth their registers -- config 5 with the bit-equal solver, n = 253: 0.0805 s against 0.0857 s with them re-derived per loop)
#define SOCP_PAR_FOR_KEPT(i, lo, hi) for (int i = par_first_plain((lo), ex.tid, ex.nt); i < (hi); i += ex.nt)

constexpr double kEpsMch = DBL_EPSILON;
constexpr double kGiant = DBL_MAX;

// minpack.cpp: enorm, over x[0], x[stride], ... (n entries).  Serial by definition (three scaled accumulators).
SOCP_HD double enorm(int n, const double *x, long stride = 1)
{
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    // The usual case first, without a branch in the loop: every entry in the middle range or exactly zero.  Then s1 = s3 = 0,
    // x3max = 0, s2 is the plain sum of squares in the order of i (a zero adds +0.0: no change) and the closing formula
    // sqrt(s2 (1 + (x3max / s2) (x3max s3))) is sqrt(s2) exactly.  The general loop below has four bodies with a division each
    // behind data-dependent branches; on the device they cost ~350 cycles per entry, this ~40.
    {
        bool plain = true;
        double sq = 0;
        int i = 0;
        for (; i + 8 <= n; i += 8) {                         // eight entries fetched together: the chain is the additions only
            double xv[8];
#pragma unroll
            for (int q = 0; q < 8; q++) xv[q] = fabs(x[(long)(i + q) * stride]);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                plain = plain & (((xv[q] > rdwarf) & (xv[q] < agiant)) | (xv[q] == 0));
                sq += xv[q] * xv[q];
            }
        }
        for (; i < n; i++) {
            const double xabs = fabs(x[(long)i * stride]);
            plain = plain & (((xabs > rdwarf) & (xabs < agiant)) | (xabs == 0));
            sq += xabs * xabs;
        }
        if (plain) return sqrt(sq);                          // = sqrt(sq (1 + (0 / sq) (0 0))), and 0 sqrt(0) when sq = 0
    }
    for (int i = 0; i < n; i++) {
        const double xabs = fabs(x[(long)i * stride]);
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) {
                const double q = x3max / xabs;
                s3 = 1 + s3 * (q * q);
                x3max = xabs;
            } else if (xabs != 0) {
                const double q = xabs / x3max;
                s3 += q * q;
            }
        } else {
            if (xabs > x1max) {
                const double q = x1max / xabs;
                s1 = 1 + s1 * (q * q);
                x1max = xabs;
            } else {
                const double q = xabs / x1max;
                s1 += q * q;
            }
        }
    }
    if (s1 != 0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0) {
        if (s2 >= x3max) return sqrt(s2 * (1 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

SOCP_HD double max_of(double a, double b) { return (a < b) ? b : a; }      // std::max(a, b)
SOCP_HD double min_of(double a, double b) { return (b < a) ? b : a; }      // std::min(a, b)
SOCP_HD long row_off(int n, int i) { return (long)i * n - (long)i * (i - 1) / 2; }   // start of row i of the packed R

// The inner loops below are serial recurrences over memory (a dot product in a fixed order, an axpy, a chain of rotations).
// The ARITHMETIC must stay in that order; the LOADS need not wait for it: they are issued kBatch at a time into registers and
// the recurrence then runs on the registers.  Written out by hand because the compiler cannot do it: it has to assume that a
// store to the matrix may alias the next load (all pointers are generic -- global or LDS), and one trip to HBM per iteration
// is what a 253-unknown factorisation then costs 128 000 times per thread.
#ifndef SOCP_SOLVER_BATCH
#define SOCP_SOLVER_BATCH 8
#endif
constexpr int kBatch = SOCP_SOLVER_BATCH;
// ... and for the sweeps that hold three operands per entry (rotate_row, axpy_dot_run): with 8 the solver kernel needs 170
// registers, two more than three wavefronts per SIMD leave it
#ifndef SOCP_SOLVER_BATCH3
#define SOCP_SOLVER_BATCH3 6
#endif
constexpr int kBatch3 = SOCP_SOLVER_BATCH3;

// sum + sum_{i = lo}^{hi - 1} v[i] * a[i * stride], added in the order of i
SOCP_HD double dot_run(const double *v, const double *a, long stride, int lo, int hi, double sum)
{
    int i = lo;
    for (; i + kBatch <= hi; i += kBatch) {
        double av[kBatch], vv[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) { av[u] = a[(long)(i + u) * stride]; vv[u] = v[i + u]; }
#pragma unroll
        for (int u = 0; u < kBatch; u++) sum += vv[u] * av[u];
    }
    for (; i < hi; i++) sum += v[i] * a[(long)i * stride];
    return sum;
}

// sum + a[lo] + a[lo + 1] + ... + a[hi - 1], added in that order
SOCP_HD double sum_run(const double *a, int lo, int hi, double sum)
{
    int i = lo;
    for (; i + kBatch <= hi; i += kBatch) {
        double av[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) av[u] = a[i + u];
#pragma unroll
        for (int u = 0; u < kBatch; u++) sum += av[u];
    }
    for (; i < hi; i++) sum += a[i];
    return sum;
}

// sum + v[lo] c + v[lo + 1] c + ... (every product formed: c = 0 still turns a NaN or an infinity in v into a NaN)
SOCP_HD double dot_const_run(const double *v, double c, int lo, int hi, double sum)
{
    int i = lo;
    for (; i + kBatch <= hi; i += kBatch) {
        double vv[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) vv[u] = v[i + u];
#pragma unroll
        for (int u = 0; u < kBatch; u++) sum += vv[u] * c;
    }
    for (; i < hi; i++) sum += v[i] * c;
    return sum;
}

// a[i * stride] -= temp * v[i] for i = lo .. hi - 1
SOCP_HD void axpy_run(double *a, long stride, const double *v, int lo, int hi, double temp)
{
    int i = lo;
    for (; i + kBatch <= hi; i += kBatch) {
        double av[kBatch], vv[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) { av[u] = a[(long)(i + u) * stride]; vv[u] = v[i + u]; }
#pragma unroll
        for (int u = 0; u < kBatch; u++) a[(long)(i + u) * stride] = av[u] - temp * vv[u];
    }
    for (; i < hi; i++) a[(long)i * stride] -= temp * v[i];
}

// the same with the vector v strided too (a column of a panel held row-major in LDS)
SOCP_HD double dot_run2(const double *v, long vs, const double *a, long as, int lo, int hi, double sum)
{
    int i = lo;
    for (; i + kBatch <= hi; i += kBatch) {
        double av[kBatch], vv[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) { av[u] = a[(long)(i + u) * as]; vv[u] = v[(long)(i + u) * vs]; }
#pragma unroll
        for (int u = 0; u < kBatch; u++) sum += vv[u] * av[u];
    }
    for (; i < hi; i++) sum += v[(long)i * vs] * a[(long)i * as];
    return sum;
}
SOCP_HD void axpy_run2(double *a, long as, const double *v, long vs, int lo, int hi, double temp)
{
    int i = lo;
    for (; i + kBatch <= hi; i += kBatch) {
        double av[kBatch], vv[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) { av[u] = a[(long)(i + u) * as]; vv[u] = v[(long)(i + u) * vs]; }
#pragma unroll
        for (int u = 0; u < kBatch; u++) a[(long)(i + u) * as] = av[u] - temp * vv[u];
    }
    for (; i < hi; i++) a[(long)i * as] -= temp * v[(long)i * vs];
}

// dst[i * ds] = src[i * ss] for i = lo .. hi - 1, sixteen loads in flight (a lone wavefront has nothing else to hide them behind)
SOCP_HD void copy_run(double *dst, long ds, const double *src, long ss, int lo, int hi)
{
    constexpr int kCopy = 16;
    int i = lo;
    for (; i + kCopy <= hi; i += kCopy) {
        double t[kCopy];
#pragma unroll
        for (int u = 0; u < kCopy; u++) t[u] = src[(long)(i + u) * ss];
#pragma unroll
        for (int u = 0; u < kCopy; u++) dst[(long)(i + u) * ds] = t[u];
    }
    for (; i < hi; i++) dst[(long)i * ds] = src[(long)i * ss];
}

// One sweep of r1mpyq over a row a[0 .. n - 1] (contiguous): for j = n - 2 .. 0 (first = true: rotations of the first sweep of
// r1updt, temp = c a_j - s a_n, a_n = s a_j + c a_n) or j = 0 .. n - 2 (first = false: temp = c a_j + s a_n, a_n = -s a_j + c a_n)
SOCP_HD double rotate_row(double *a, const double *c, const double *s, int n, double an, bool first)
{
    if (first) {
        int j = n - 2;
        for (; j - kBatch3 + 1 >= 0; j -= kBatch3) {
            double av[kBatch3], cv[kBatch3], sv[kBatch3];
#pragma unroll
            for (int u = 0; u < kBatch3; u++) { av[u] = a[j - u]; cv[u] = c[j - u]; sv[u] = s[j - u]; }
#pragma unroll
            for (int u = 0; u < kBatch3; u++) {
                const double temp = cv[u] * av[u] - sv[u] * an;
                an = sv[u] * av[u] + cv[u] * an;
                av[u] = temp;
            }
#pragma unroll
            for (int u = 0; u < kBatch3; u++) a[j - u] = av[u];
        }
        for (; j >= 0; j--) {
            const double aj = a[j];
            const double temp = c[j] * aj - s[j] * an;
            an = s[j] * aj + c[j] * an;
            a[j] = temp;
        }
    } else {
        int j = 0;
        for (; j + kBatch3 <= n - 1; j += kBatch3) {
            double av[kBatch3], cv[kBatch3], sv[kBatch3];
#pragma unroll
            for (int u = 0; u < kBatch3; u++) { av[u] = a[j + u]; cv[u] = c[j + u]; sv[u] = s[j + u]; }
#pragma unroll
            for (int u = 0; u < kBatch3; u++) {
                const double temp = cv[u] * av[u] + sv[u] * an;
                an = -sv[u] * av[u] + cv[u] * an;
                av[u] = temp;
            }
#pragma unroll
            for (int u = 0; u < kBatch3; u++) a[j + u] = av[u];
        }
        for (; j < n - 1; j++) {
            const double aj = a[j];
            const double temp = c[j] * aj + s[j] * an;
            an = -s[j] * aj + c[j] * an;
            a[j] = temp;
        }
    }
    return an;
}

// One reflector's axpy and the NEXT reflector's dot product in one sweep over a column (a[i * stride]):
//   a[i] -= temp * v[i]            for i = lo  .. hi - 1     (reflector in hand)
//   returns sum_{i = dlo}^{hi - 1} u[i] * a[i] (the new a where both ranges overlap), added in the order of i, from 0.0
// |lo - dlo| <= 1 in both users: qrfac's next reflector starts one row further down (dlo = lo + 1), qform's one row further up
// (dlo = lo - 1).  The entries are produced in the order the dot product consumes them, so every number is the one the two
// separate sweeps (axpy_run, then dot_run) give -- and the column crosses the memory system once instead of twice.
SOCP_HD double axpy_dot_run(double *a, long stride, const double *v, const double *u, int lo, int dlo, int hi, double temp)
{
    double sum = 0.0;
    int i = (lo < dlo) ? lo : dlo;
    const int both = (lo < dlo) ? dlo : lo;
    for (; i < both && i < hi; i++) {
        if (i >= lo) a[(long)i * stride] -= temp * v[i];     // (qrfac) a row of R: no part in the next dot product
        else sum += u[i] * a[(long)i * stride];              // (qform) a row the reflector in hand does not reach
    }
    for (; i + kBatch3 <= hi; i += kBatch3) {
        double av[kBatch3], vv[kBatch3], uv[kBatch3];
#pragma unroll
        for (int q = 0; q < kBatch3; q++) { av[q] = a[(long)(i + q) * stride]; vv[q] = v[i + q]; uv[q] = u[i + q]; }
#pragma unroll
        for (int q = 0; q < kBatch3; q++) av[q] = av[q] - temp * vv[q];
#pragma unroll
        for (int q = 0; q < kBatch3; q++) a[(long)(i + q) * stride] = av[q];
#pragma unroll
        for (int q = 0; q < kBatch3; q++) sum += uv[q] * av[q];
    }
    for (; i < hi; i++) {
        const double t = a[(long)i * stride] - temp * v[i];
        a[(long)i * stride] = t;
        sum += u[i] * t;
    }
    return sum;
}

// One Jacobian refresh's factor work: qrfac (no pivoting) with Q^T fvec riding along as column n, R packed by rows, qform in
// place.  In: A[i][j] = J(i, j), fvec.  Out: A = Q (row-major), r, qtf, rdiag (wa1), acnorm (wa2); returns "singular".
// A column's norm is a serial chain over its entries: the column is first copied to a fast vector by all threads at once (one
// strided load each) and the chain then runs on that copy -- on the matrix itself it would be n dependent trips to memory.
//
// Two sweeps over the trailing matrix per reflector instead of three (dot, then axpy = read + read/write): while a column
// takes reflector j's axpy it already accumulates its dot product with reflector j + 1 (axpy_dot_run).  That needs v_{j+1}
// BEFORE the sweep, so column j + 1 runs one step ahead: all threads together give it reflector j (one entry each), its norm
// and scaling follow as before, and the sweep then covers the columns from j + 2 on.  A column still meets the reflectors in
// the order 0, 1, 2, ..., each as the same dot product and the same axpy over its rows in the order of the rows.
template <class E>
SOCP_HD bool factor(const E &ex, int n, int ld, Work &w, Prof *pf = nullptr)
{
#define SOCP_PF(slot) do { if (pf) pf->mark((slot), ex.tid); } while (0)
    double *A = w.A, *rdiag = w.wa1, *acnorm = w.wa2, *col = w.f[0];
    double *va = w.f[1], *vb = w.f[2];                       // the reflector in hand, the next one (they swap every step)
    double *sums = w.f[3];                                   // sums[k], k = 0 .. n: column k's dot product with the reflector in hand
                                                             // (n + 1 entries: runs one into w.f[4], unused here)
    SOCP_PAR_FOR_KEPT(j, 0, n) acnorm[j] = enorm(n, A + j, ld);
    SOCP_PAR_FOR_KEPT(i, 0, n) A[(long)i * ld + n] = w.fvec[i];
    SOCP_PAR_FOR_KEPT(i, 0, n) col[i] = A[(long)i * ld];
    ex.sync();
    // reflector 0 from column 0 as it stands, and every later column's dot product with it
    bool cur = false;
    double *const shared = w.f[5];                           // shared[0]: a scalar from the leading wavefront to everybody
    {
        if (ex.leader()) { const double nrm = enorm(n, col); if (ex.tid == 0) shared[0] = nrm; }
        ex.sync();
        double ajnorm = shared[0];
        if (ajnorm != 0 && col[0] < 0) ajnorm = -ajnorm;
        cur = ajnorm != 0;
        if (cur) {
            SOCP_PAR_FOR_KEPT(i, 0, n) {
                double t = col[i] / ajnorm;
                if (i == 0) t += 1;
                A[(long)i * ld] = t;
                va[i] = t;
            }
        }
        if (ex.tid == 0) rdiag[0] = -ajnorm;
        ex.sync();
        if (cur) SOCP_PAR_FOR_KEPT(k, 1, n + 1) sums[k] = dot_run(va, A + k, ld, 0, n, 0.0);
        ex.sync();
    }
    SOCP_PF(8);
    for (int j = 0; j < n; j++) {
        // in hand: va = v_j (if cur), sums[k] = v_j . a_k for k > j
        const double piv = cur ? va[j] : 1.0;
        const int c1 = j + 1;                                // the column that runs ahead (c1 = n: fvec's column, never a reflector)
        bool next = false;
        if (cur) {
            const double temp = sums[c1] / piv;
            SOCP_PAR_FOR_KEPT(i, j, n) {
                const double t = A[(long)i * ld + c1] - temp * va[i];
                A[(long)i * ld + c1] = t;
                col[i] = t;
            }
        } else if (c1 < n) {
            SOCP_PAR_FOR_KEPT(i, c1, n) col[i] = A[(long)i * ld + c1];
        }
        ex.sync();
        SOCP_PF(9);
        if (c1 < n) {
            if (ex.leader()) { const double nrm = enorm(n - c1, col + c1); if (ex.tid == 0) shared[0] = nrm; }
            ex.sync();
            double ajnorm = shared[0];
            if (ajnorm != 0 && col[c1] < 0) ajnorm = -ajnorm;
            next = ajnorm != 0;
            if (next) {
                SOCP_PAR_FOR_KEPT(i, c1, n) {
                    double t = col[i] / ajnorm;
                    if (i == c1) t += 1;
                    A[(long)i * ld + c1] = t;
                    vb[i] = t;
                }
            }
            if (ex.tid == 0) rdiag[c1] = -ajnorm;
            ex.sync();
        }
        SOCP_PF(10);
        // the columns from j + 2 on (and fvec's): reflector j's axpy, reflector j + 1's dot product
        if (cur && next) {
            SOCP_PAR_FOR_KEPT(k, j + 2, n + 1) sums[k] = axpy_dot_run(A + k, ld, va, vb, j, c1, n, sums[k] / piv);
        } else if (cur) {
            SOCP_PAR_FOR_KEPT(k, j + 2, n + 1) axpy_run(A + k, ld, va, j, n, sums[k] / piv);
        } else if (next) {
            SOCP_PAR_FOR_KEPT(k, j + 2, n + 1) sums[k] = dot_run(vb, A + k, ld, c1, n, 0.0);
        }
        ex.sync();
        SOCP_PF(11);
        double *const t = va; va = vb; vb = t;
        cur = next;
    }
    SOCP_PAR_FOR_KEPT(i, 0, n) w.qtf[i] = A[(long)i * ld + n];
    // R by rows: row i = [rdiag[i], A(i, i+1 .. n-1)]
    for (int i = 0; i < n; i++) {
        const long off = row_off(n, i);
        SOCP_PAR_FOR_KEPT(k, i, n) w.r[off + (k - i)] = (k == i) ? rdiag[i] : A[(long)i * ld + k];
    }
    bool sing = false;
    SOCP_PAR_FOR_KEPT(j, 0, n) col[j] = rdiag[j];
    ex.sync();
    for (int j = 0; j < n; j++) if (col[j] == 0) sing = true;
    // qform, MINPACK's in-place order: the strict upper triangle is cleared, then for k = n-1 .. 0 column k's Householder
    // vector moves out, the column becomes e_k and the columns j >= k go through reflector k.  Fused the same way: while the
    // columns j >= k take reflector k's axpy they accumulate their dot product with reflector k - 1, whose vector is simply
    // column k - 1 as stored (no reflector of this sweep touches it before its turn).
    for (int i = 0; i < n; i++)
        SOCP_PAR_FOR_KEPT(j, i + 1, n) A[(long)i * ld + j] = 0;
    SOCP_PAR_FOR_KEPT(i, n - 1, n) { va[i] = A[(long)i * ld + n - 1]; A[(long)i * ld + n - 1] = 1.0; }
    ex.sync();
    cur = va[n - 1] != 0;
    if (cur) SOCP_PAR_FOR_KEPT(jc, n - 1, n) sums[jc] = dot_run(va, A + jc, ld, n - 1, n, 0.0);
    ex.sync();
    SOCP_PF(12);
    for (int k = n - 1; k >= 0; k--) {
        // in hand: va = v_k (rows k .. n-1), sums[jc] = v_k . q_jc for jc >= k (if cur)
        const double piv = cur ? va[k] : 1.0;
        bool next = false;
        if (k > 0) {
            SOCP_PAR_FOR_KEPT(i, k - 1, n) { vb[i] = A[(long)i * ld + k - 1]; A[(long)i * ld + k - 1] = (i == k - 1) ? 1.0 : 0.0; }
            ex.sync();
            next = vb[k - 1] != 0;
        }
        if (cur && next) {
            // column k - 1 is the new e_{k-1}, which reflector k does not reach: its dot product with v_{k-1} runs on the vector
            // alone (1.0 and 0.0 as the column's entries, every product formed) by all threads alike -- one thread on a path
            // of its own would hold its whole wavefront back for the length of a second sweep
            double odd = 0.0;
            odd += vb[k - 1] * 1.0;
            odd = dot_const_run(vb, 0.0, k, n, odd);
            if (ex.tid == (k - 1) % ex.nt) sums[k - 1] = odd;
            SOCP_PAR_FOR_KEPT(jc, k, n) sums[jc] = axpy_dot_run(A + jc, ld, va, vb, k, k - 1, n, sums[jc] / piv);
        } else if (cur) {
            SOCP_PAR_FOR_KEPT(jc, k, n) axpy_run(A + jc, ld, va, k, n, sums[jc] / piv);
        } else if (next) {
            double odd = 0.0;                                // (column k - 1 as above: on the vector alone)
            odd += vb[k - 1] * 1.0;
            odd = dot_const_run(vb, 0.0, k, n, odd);
            if (ex.tid == (k - 1) % ex.nt) sums[k - 1] = odd;
            SOCP_PAR_FOR_KEPT(jc, k, n) sums[jc] = dot_run(vb, A + jc, ld, k - 1, n, 0.0);
        }
        ex.sync();
        double *const t = va; va = vb; vb = t;
        cur = next;
    }
    SOCP_PF(13);
#undef SOCP_PF
    return sing;
}

// The same factor work with the matrix taken through LDS in blocks (for problems whose matrix does not stay on chip: the
// in-place form streams it ~2n/3 times -- 0.26 GB per refresh at n = 253 against 1.3 MB of algorithmic bytes).  Right-looking in
// panels of kPanel reflectors: the panel's columns are factorised in a small LDS buffer Pn[n][kPanel]; every later column --
// 64 at a time, each thread its own, in an LDS block Bk[n][kBlock] -- then goes through the panel's reflectors in order and is
// written back once.  A column still meets the reflectors in the order 0, 1, 2, ..., each as one dot product and one axpy over
// its rows in the order of the rows: the numbers are those of factor() (and of MINPACK).  The packed Householder vectors are kept
// (w.V) and qform is done the same way from them: a block of Q's columns starts as the identity in LDS and goes through the
// reflectors k = j, j - 1, ..., 0.  One block read and one write per PANEL instead of three passes over the trailing matrix per
// REFLECTOR.  Meant for a workgroup of ONE wavefront (a thread per column of a block); correct for any executor.
constexpr int kPanel = 8;
constexpr int kBlock = 64;
SOCP_HD long blocked_lds_doubles(int n) { return (long)n * (kPanel + kBlock); }

template <class E>
SOCP_HD bool factor_blocked(const E &ex, int n, int ld, Work &w, double *Pn, double *Bk)
{
    double *A = w.A, *rdiag = w.wa1, *acnorm = w.wa2, *V = w.V;
    SOCP_PAR_FOR(j, 0, n) acnorm[j] = enorm(n, A + j, ld);
    SOCP_PAR_FOR(i, 0, n) A[(long)i * ld + n] = w.fvec[i];
    ex.sync();
    unsigned skip = 0;                                       // bit t: reflector j0 + t of the panel in hand is the identity
    for (int j0 = 0; j0 < n; j0 += kPanel) {
        const int np = (n - j0 < kPanel) ? n - j0 : kPanel, rows = n - j0;
        // ---- the panel: columns j0 .. j0 + np - 1, rows j0 .. n - 1, factorised among themselves in LDS
        SOCP_PAR_FOR(e, 0, rows * np) { const int i = j0 + e / np, t = e - (e / np) * np; Pn[(long)i * kPanel + t] = A[(long)i * ld + j0 + t]; }
        ex.sync();
        for (int t = 0; t < np; t++) {
            const int j = j0 + t;
            double ajnorm = enorm(n - j, Pn + (long)j * kPanel + t, kPanel);
            if (ajnorm != 0 && Pn[(long)j * kPanel + t] < 0) ajnorm = -ajnorm;
            skip = (ajnorm == 0) ? (skip | (1u << t)) : (skip & ~(1u << t));
            ex.sync();                                       // everyone has the norm before the column is scaled
            if (ajnorm != 0) {
                SOCP_PAR_FOR(i, j, n) {
                    double val = Pn[(long)i * kPanel + t] / ajnorm;
                    if (i == j) val += 1;
                    Pn[(long)i * kPanel + t] = val;
                }
            }
            if (ex.tid == 0) rdiag[j] = -ajnorm;
            ex.sync();
            if (ajnorm != 0) {
                const double piv = Pn[(long)j * kPanel + t];
                SOCP_PAR_FOR(t2, t + 1, np) {
                    const double sum = dot_run2(Pn + t, kPanel, Pn + t2, kPanel, j, n, 0.0);
                    axpy_run2(Pn + t2, kPanel, Pn + t, kPanel, j, n, sum / piv);
                }
            }
            ex.sync();
        }
        // the finished panel goes home (its upper part is R's, its lower part the vectors) and the vectors are packed
        SOCP_PAR_FOR(e, 0, rows * np) {
            const int i = j0 + e / np, t = e - (e / np) * np;
            const double val = Pn[(long)i * kPanel + t];
            A[(long)i * ld + j0 + t] = val;
            if (i >= j0 + t) V[row_off(n, j0 + t) + (i - j0 - t)] = val;
        }
        // ---- every later column (and fvec's, column n): through the panel's reflectors, 64 columns at a time, a thread per column
        for (int c0 = j0 + np; c0 <= n; c0 += kBlock) {
            const int bw = (n + 1 - c0 < kBlock) ? n + 1 - c0 : kBlock;
            SOCP_PAR_FOR(c, 0, bw) {
                double *col = Bk + c;                        // this thread's column: rows j0 .. n - 1 at col[(i - j0) * kBlock]
                copy_run(col - (long)j0 * kBlock, kBlock, A + c0 + c, ld, j0, n);
                for (int t = 0; t < np; t++) {
                    if (skip & (1u << t)) continue;
                    const int j = j0 + t;
                    const double sum = dot_run2(Pn + t, kPanel, col - (long)j0 * kBlock, kBlock, j, n, 0.0);
                    axpy_run2(col - (long)j0 * kBlock, kBlock, Pn + t, kPanel, j, n, sum / Pn[(long)j * kPanel + t]);
                }
                copy_run(A + c0 + c, ld, col - (long)j0 * kBlock, kBlock, j0, n);
            }
        }
        ex.sync();                                           // the next panel reads columns other threads have just written
    }
    SOCP_PAR_FOR(i, 0, n) w.qtf[i] = A[(long)i * ld + n];
    for (int i = 0; i < n; i++) {
        const long off = row_off(n, i);
        SOCP_PAR_FOR(k, i, n) w.r[off + (k - i)] = (k == i) ? rdiag[i] : A[(long)i * ld + k];
    }
    bool sing = false;
    for (int j = 0; j < n; j++) if (rdiag[j] == 0) sing = true;
    ex.sync();
    // ---- qform from the packed vectors: column j of Q is e_j pushed through the reflectors j, j - 1, ..., 0
    for (int c0 = 0; c0 < n; c0 += kBlock) {
        const int bw = (n - c0 < kBlock) ? n - c0 : kBlock, jtop = c0 + bw - 1;
        SOCP_PAR_FOR(c, 0, bw) {
            double *col = Bk + c;
            for (int i = 0; i < n; i++) col[(long)i * kBlock] = (i == c0 + c) ? 1.0 : 0.0;
        }
        for (int k1 = jtop; k1 >= 0; k1 -= kPanel) {
            const int np = (k1 + 1 < kPanel) ? k1 + 1 : kPanel, klo = k1 - np + 1;
            ex.sync();                                       // the previous panel of vectors is no longer read
            for (int t = 0; t < np; t++) {
                const int k = klo + t;
                const double *vk = V + row_off(n, k);
                SOCP_PAR_FOR(i, k, n) Pn[(long)i * kPanel + t] = vk[i - k];
            }
            ex.sync();
            SOCP_PAR_FOR(c, 0, bw) {
                double *col = Bk + c;
                const int j = c0 + c;
                for (int k = k1; k >= klo; k--) {
                    if (k > j) continue;                     // column j starts at reflector j
                    const int t = k - klo;
                    const double piv = Pn[(long)k * kPanel + t];
                    if (piv == 0) continue;
                    const double sum = dot_run2(Pn + t, kPanel, col, kBlock, k, n, 0.0);
                    axpy_run2(col, kBlock, Pn + t, kPanel, k, n, sum / piv);
                }
            }
        }
        SOCP_PAR_FOR(c, 0, bw) {
            copy_run(A + c0 + c, ld, Bk + c, kBlock, 0, n);
        }
        ex.sync();
    }
    return sing;
}

// The rows of the packed R a serial chain walks (r1updt's two sweeps, the dogleg's back substitution), fetched AHEAD of the chain.
// Every step of those chains begins with its row of R, and left to itself each load waits for the step before it: the compiler
// must assume that the row just stored aliases the next one, and the per-step barrier of the second sweep / the back substitution
// drained the wave's loads (vmcnt(0)).  ~760 dependent trips to memory per trial step at n = 253, 80 % of a trial round
// (profiles/r05j_solver_phases.txt).  A thread owns the entries i = tid, tid + nt, ... of every row (absolute ownership, above), so
// it can hold its kRingE entries of the next kRingB rows in registers: slot b is refilled with row j -+ kRingB as soon as row j
// has been consumed.  The ARITHMETIC is untouched -- same operations, same order -- so every flavour keeps its bits.
// Every load and store of the ringed loops is UNCONDITIONAL -- a lane without an entry in row j loads the diagonal entry instead
// (and ignores it), and stores to a dump slot of its own: the compiler counts outstanding memory operations per path and, where
// paths differ, waits for the shortest count it can prove -- behind per-lane branches (first version) it drained all but the last
// few operations at every step and the ring bought nothing.
template <int E>
SOCP_HD void ring_fetch(double (&slot)[E], const double *s, int n, int j, int tid, int nt)
{
    const double *row = s + row_off(n, j) - j;               // row[i] = s(j, i), i = j .. n - 1
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int i = tid + e * nt;
        slot[e] = row[(i >= j && i < n) ? i : j];
    }
}

// minpack.cpp: dogleg; the step comes out in w.wa1.  Fast vectors: f0 = the Gauss-Newton step, f1 / f2 = the row of R in hand
// (alternating), f3 = qtb, f4 = scaled vectors whose norm is taken, f5 = the gradient direction.  Every thread returns with
// the step complete (synchronised).
template <class E>
SOCP_HD bool dogleg(const E &ex, int n, Work &w, double delta, Prof *pf = nullptr, bool fast_sums = false)
{
    const double *r = w.r, *diag = w.diag;
    double *x = w.wa1, *xl = w.f[0], *qtb = w.f[3], *sc = w.f[4], *g = w.f[5];
    SOCP_PAR_FOR(j, 0, n) qtb[j] = w.qtf[j];
    // Gauss-Newton direction by back substitution: x[j] needs every later x[i], the last computed first -- one serial chain.
    // Row j of R is brought to a fast vector by all threads at once; every thread then runs the same sum.
    long jj = (long)n * (n + 1) / 2;
    bool ringed = false;
    if constexpr (E::kRingE > 0) {
        if (n <= E::kRingE * ex.nt) {
            // the rows fetched ahead of the chain (ring_fetch, above r1updt): the sum a step runs is a chain of n - j dependent
            // additions -- long enough to cover the next rows' trip to memory, if that trip has been started
            ringed = true;
            constexpr int RE = E::kRingE, RB = E::kRingB;
            const int tid = ex.tid, nt = ex.nt;
            double ring[RB][RE];
#pragma unroll
            for (int b = 0; b < RB; b++) ring_fetch<RE>(ring[b], r, n, (n - 1 - b > 0) ? n - 1 - b : 0, tid, nt);
            for (int jb = n - 1; jb >= 0; jb -= RB) {
#pragma unroll
                for (int b = 0; b < RB; b++) {               // (straight-line groups: see r1updt)
                    const bool live = jb - b >= 0;
                    const int j = live ? jb - b : 0;
                    const int k = n - j;
                    if (live) jj -= k;
                    double *row = (k & 1) ? w.f[2] : w.f[1];
#pragma unroll
                    for (int e = 0; e < RE; e++) {
                        const int i = tid + e * nt;
                        if (live && i >= j && i < n) row[i] = (i >= j + 2) ? ring[b][e] * xl[i] : ring[b][e];
                    }
                    ring_fetch<RE>(ring[b], r, n, (j - RB > 0) ? j - RB : 0, tid, nt);
                    ex.sync_lds();                           // the row, qtb, and the x[j + 1] thread 0 stored before arriving here
                    if (live && ex.leader()) {
                        double sum = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
                        if (fast_sums) {
                            // throughput flavour: the n - j - 2 products are summed by the wavefront's lanes side by side and the
                            // partial sums by a wave-wide reduction -- ~40 instructions a step instead of a chain of n - j dependent
                            // additions fed from LDS (32 000 of them per trial step at n = 253).  The order of the additions changes:
                            // rounding level, this flavour only.
                            double part = 0.0;
                            for (int i = j + 2 + (tid & 63); i < n; i += 64) part += row[i];
                            sum = wave_sum(part);
                            if (j + 1 < n) sum += row[j + 1] * xl[j + 1];
                        } else
#endif
                        {
                            if (j + 1 < n) sum += row[j + 1] * xl[j + 1];
                            sum = sum_run(row, j + 2, n, sum);
                        }
                        double temp = row[j];
                        if (temp == 0) {
                            long l = j;
                            for (int i = 0; i <= j; i++) { temp = max_of(temp, fabs(r[l])); l += n - i - 1; }
                            temp = kEpsMch * temp;
                            if (temp == 0) temp = kEpsMch;
                        }
                        if (ex.tid == 0) xl[j] = (qtb[j] - sum) / temp;
                    }
                }
            }
        }
    }
    for (int k = 1; k <= n && !ringed; k++) {
        const int j = n - k;
        jj -= k;
        double *row = (k & 1) ? w.f[2] : w.f[1];            // (no run-time index into the pointer table: it would go to scratch)
        // the products r(j, i) x[i] are formed by the threads that fetch the row, all at once -- except the one with x[j + 1],
        // which thread 0 stored after the last barrier: that entry travels as it is and is multiplied after this one.  The chain
        // every thread then runs is additions only, in the order of i, of the same rounded products.
        SOCP_PAR_FOR(i, j, n) {
            const double rv = r[jj + (i - j)];
            row[i] = (i >= j + 2) ? rv * xl[i] : rv;
        }
        ex.sync();                                           // the row, qtb, and the x[j + 1] thread 0 stored before arriving here
        if (ex.leader()) {                                   // (the other wavefronts go on to fetch the next row)
            double sum = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
            if (fast_sums) {                                 // (throughput flavour: the products summed in parallel, see the ringed loop)
                double part = 0.0;
                for (int i = j + 2 + (ex.tid & 63); i < n; i += 64) part += row[i];
                sum = wave_sum(part);
                if (j + 1 < n) sum += row[j + 1] * xl[j + 1];
            } else
#endif
            {
                if (j + 1 < n) sum += row[j + 1] * xl[j + 1];
                sum = sum_run(row, j + 2, n, sum);
            }
            double temp = row[j];
            if (temp == 0) {
                long l = j;
                for (int i = 0; i <= j; i++) { temp = max_of(temp, fabs(r[l])); l += n - i - 1; }
                temp = kEpsMch * temp;
                if (temp == 0) temp = kEpsMch;
            }
            if (ex.tid == 0) xl[j] = (qtb[j] - sum) / temp;
        }
    }
    ex.sync();
    if (pf) pf->mark(8, ex.tid);                             // (profile builds: the back substitution)
    SOCP_PAR_FOR(j, 0, n) sc[j] = diag[j] * xl[j];
    ex.sync();
    const double qnorm = enorm(n, sc);
    if (qnorm <= delta) {
        SOCP_PAR_FOR(j, 0, n) x[j] = xl[j];
        ex.sync();
        return true;                                         // the Gauss-Newton step itself
    }
    // scaled gradient direction: element i collects r(j, i) qtb[j] for j = 0 .. i in that order, then is divided
    SOCP_PAR_FOR(i, 0, n) {
        double acc = 0;                                      // wa1[i] starts at 0
        for (int j = 0; j <= i; j++) acc += r[row_off(n, j) + (i - j)] * qtb[j];
        g[i] = acc / diag[i];
    }
    ex.sync();
    if (pf) pf->mark(9, ex.tid);                             // (the gradient)
    const double gnorm = enorm(n, g);
    double sgnorm = 0;
    double alpha = delta / qnorm;
    if (gnorm != 0) {
        ex.sync();                                           // everyone has gnorm (and qnorm) before g and sc change
        SOCP_PAR_FOR(j, 0, n) g[j] = (g[j] / gnorm) / diag[j];
        ex.sync();
        SOCP_PAR_FOR(j, 0, n) {
            sc[j] = dot_run(g, r + row_off(n, j) - j, 1, j, n, 0.0);
        }
        ex.sync();
        double temp = enorm(n, sc);
        sgnorm = (gnorm / temp) / temp;
        alpha = 0;
        if (sgnorm < delta) {
            const double bnorm = enorm(n, qtb);
            const double dq = delta / qnorm, sd = sgnorm / delta;
            temp = (bnorm / gnorm) * (bnorm / qnorm) * sd;
            const double d1 = temp - dq;
            temp = temp - dq * (sd * sd) + sqrt(d1 * d1 + (1 - dq * dq) * (1 - sd * sd));
            alpha = (dq * (1 - sd * sd)) / temp;
        }
    }
    const double temp = (1 - alpha) * min_of(sgnorm, delta);
    SOCP_PAR_FOR(j, 0, n) x[j] = temp * g[j] + alpha * xl[j];
    ex.sync();
    return false;
}

SOCP_HD void givens(double a, double b, double &cs, double &sn, double &tau)
{
    if (fabs(a) < fabs(b)) {
        const double cotan = a / b;
        sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
        cs = sn * cotan;
        tau = 1;
        if (fabs(cs) * kGiant > 1) tau = 1 / cs;
    } else {
        const double tn = b / a;
        cs = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
        sn = cs * tn;
        tau = sn;
    }
}

SOCP_HD void decode_rotation(double t, double &cs, double &sn)
{
    if (fabs(t) > 1) { cs = 1 / t; sn = sqrt(1 - cs * cs); }
    else { sn = t; cs = sqrt(1 - sn * sn); }
}

// minpack.cpp: r1updt on the packed R (m = n).  Fast vectors: v = f4 (in), u = f5 (in), w = f6, the first sweep's rotation
// encodings -> f7, the diagonal of s after the first sweep -> f1, the second sweep's encodings -> f4 (v is dead by then).
// MINPACK leaves the encodings in v and w themselves; here nobody may see a half-updated vector, so they go elsewhere.
// A thread owns element i of w and column-position i of every row of s for the whole routine.
template <class E>
SOCP_HD bool r1updt(const E &ex, int n, Work &wk, Prof *pf = nullptr)
{
    double *s = wk.r, *w = wk.f[6], *rot = wk.f[7], *sd = wk.f[1], *rot2 = wk.f[4];
    const double *u = wk.f[5], *v = wk.f[4];
    long jj = (long)n * (n + 1) / 2 - 1;                     // the last diagonal entry
    if (ex.tid == (n - 1) % ex.nt) w[n - 1] = s[jj];
    double vn = v[n - 1];
    if constexpr (E::kRingE > 0) {
        if (n <= E::kRingE * ex.nt && (long)E::kRingE * ex.nt <= 8L * n) {     // (the second condition: the dump slots fit, Work::dump_area)
            constexpr int RE = E::kRingE, RB = E::kRingB;
            const int tid = ex.tid, nt = ex.nt;
            double *dump = wk.dump_area(n) + (long)tid * RE;       // a lane's own dump slots (ring_fetch, above)
            double ring[RB][RE];
            // (a group of RB steps is straight-line code: steps beyond the last row run as no-ops -- `live` false: they load row 0 and
            // store to the dump slots -- rather than leave the group through a branch, and every prefetch is issued, clamped to a row
            // that exists: the compiler's count of outstanding loads is then exact on the one path there is)
#pragma unroll
            for (int b = 0; b < RB; b++) ring_fetch<RE>(ring[b], s, n, (n - 2 - b > 0) ? n - 2 - b : 0, tid, nt);
            for (int jb = n - 2; jb >= 0; jb -= RB) {
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    const bool live = jb - b >= 0;
                    const int j = live ? jb - b : 0;
                    if (live) jj -= (n - j);
                    const double vj = v[j];
                    const bool upd = live && vj != 0;
                    double cs = 0, sn = 0, tau = vj;
                    if (upd) {
                        givens(vn, vj, cs, sn, tau);
                        vn = sn * vj + cs * vn;
                    }
                    if (live && tid == 0) rot[j] = tau;
                    {
                        // (vj == 0: no rotation -- the row stays, w[j] = 0, sd[j] = s(j, j): the same stores with the values selected)
                        double *row = s + jj - j;
#pragma unroll
                        for (int e = 0; e < RE; e++) {
                            const int i = tid + e * nt;
                            const bool mine = live && i >= j && i < n;
                            const double sv = ring[b][e];
                            const double wi = (i == j) ? 0.0 : w[mine ? i : j];
                            const double temp = upd ? cs * sv - sn * wi : sv;
                            const double wn = upd ? sn * sv + cs * wi : wi;
                            double *dst = mine ? row + i : dump + e;
                            *dst = temp;
                            if (mine) { w[i] = wn; if (i == j) sd[j] = temp; }
                        }
                    }
                    ring_fetch<RE>(ring[b], s, n, (j - RB > 0) ? j - RB : 0, tid, nt);
                }
            }
            SOCP_PAR_FOR(i, 0, n) w[i] += vn * u[i];
            if (pf) pf->mark(10, ex.tid);                    // (profile builds: the first sweep)
            bool sing = false;
            // (the rows the first sweep has just stored: a thread reads back its own entries)
            const int jlast = (n - 2 > 0) ? n - 2 : 0;
#pragma unroll
            for (int b = 0; b < RB; b++) ring_fetch<RE>(ring[b], s, n, (b < jlast) ? b : jlast, tid, nt);
            for (int jb = 0; jb < n - 1; jb += RB) {
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    const bool live = jb + b < n - 1;
                    const int j = live ? jb + b : jlast;
                    ex.sync_lds();                           // s(j, j) and w[j] come from the thread that owns element j
                    const double wj = w[j], sjj = sd[j];
                    double cs = 0, sn = 0, tau = 0;
                    const bool upd = live && wj != 0;
                    if (upd) {
                        givens(sjj, wj, cs, sn, tau);
                        if (cs * sjj + sn * wj == 0) sing = true;
                    } else if (live && sjj == 0) {
                        sing = true;
                    }
                    {
                        double *row = s + jj - j;
#pragma unroll
                        for (int e = 0; e < RE; e++) {
                            const int i = tid + e * nt;
                            const bool mine = live && i >= j && i < n;
                            const double sv = ring[b][e], wi = w[mine ? i : j];
                            const double temp = upd ? cs * sv + sn * wi : sv;
                            const double wn = -sn * sv + cs * wi;
                            double *dst = mine ? row + i : dump + e;
                            *dst = temp;
                            if (mine && upd && i != j) w[i] = wn;
                        }
                    }
                    if (live && tid == 0) rot2[j] = (wj != 0) ? tau : wj;
                    if (live) jj += (n - j);
                    ring_fetch<RE>(ring[b], s, n, (j + RB < jlast) ? j + RB : jlast, tid, nt);
                }
            }
            ex.sync();
            const double last = w[n - 1];
            if (ex.tid == 0) s[jj] = last;
            if (last == 0) sing = true;
            return sing;
        }
    }
    for (int nmj = 1; nmj <= n - 1; nmj++) {
        const int j = n - 1 - nmj;
        jj -= (n - j);
        const double vj = v[j];
        double cs = 0, sn = 0, tau = vj;
        if (vj != 0) {
            givens(vn, vj, cs, sn, tau);
            vn = sn * vj + cs * vn;
        }
        if (ex.tid == 0) rot[j] = tau;                       // (read again only after a barrier, in r1mpyq_all)
        if (vj != 0) {
            SOCP_PAR_FOR(i, j, n) {
                const long l = jj + (i - j);
                const double wi = (i == j) ? 0.0 : w[i];     // w[j] = 0 before the rotation reaches it
                const double temp = cs * s[l] - sn * wi;
                w[i] = sn * s[l] + cs * wi;
                s[l] = temp;
                if (i == j) sd[j] = temp;
            }
        } else if (ex.tid == j % ex.nt) {
            w[j] = 0;
            sd[j] = s[jj];
        }
    }
    SOCP_PAR_FOR(i, 0, n) w[i] += vn * u[i];
    if (pf) pf->mark(10, ex.tid);
    bool sing = false;
    for (int j = 0; j < n - 1; j++) {
        ex.sync();                                           // s(j, j) and w[j] come from the thread that owns element j
        const double wj = w[j], sjj = sd[j];
        double cs = 0, sn = 0, tau = 0;
        if (wj != 0) {
            givens(sjj, wj, cs, sn, tau);
            SOCP_PAR_FOR(i, j, n) {
                const long l = jj + (i - j);
                const double temp = cs * s[l] + sn * w[i];
                const double wn = -sn * s[l] + cs * w[i];
                s[l] = temp;
                if (i != j) w[i] = wn;                       // (MINPACK stores the encoding in w[j]: rot2 here)
            }
            if (cs * sjj + sn * wj == 0) sing = true;        // s(j, j) after the rotation, as every thread can compute it
        } else if (sjj == 0) {
            sing = true;
        }
        if (ex.tid == 0) rot2[j] = (wj != 0) ? tau : wj;
        jj += (n - j);
    }
    ex.sync();
    const double last = w[n - 1];
    if (ex.tid == 0) s[jj] = last;
    if (last == 0) sing = true;
    return sing;
}

// The same sweeps for a row of Q in global memory, in pieces of 16 entries = one 128-byte line per thread, fetched with four 16-byte
// loads and written back the same way.  A thread owns a row, so a wavefront's load touches 64 different lines whatever its width:
// with 8-byte loads every line is requested sixteen times over, and with a dozen wavefronts per CU walking 64 rows each the lines
// do not survive in the vector cache between requests.  The ARITHMETIC is rotate_row's -- the same rotations in the same order
// on the same numbers -- so the result is bit-identical.  `a` must be 16-byte aligned (rows of the solver's matrix are: ld is a
// multiple of 8 doubles); the pieces are aligned to 16 entries, the ragged ends go through rotate_row's scalar form.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rotate_row_lines(double *a, const double *c, const double *s, int n, double an, bool first)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int kLine = 16;
    const int last = n - 2;                                  // the sweeps cover j = 0 .. n - 2
    if (last < 2 * kLine) return rotate_row(a, c, s, n, an, first);
    const int top = (last + 1) / kLine * kLine;              // entries [top, last] are the ragged end above the last whole line
    if (first) {
        for (int j = last; j >= top; j--) {
            const double aj = a[j];
            const double temp = c[j] * aj - s[j] * an;
            an = s[j] * aj + c[j] * an;
            a[j] = temp;
        }
        for (int j0 = top - kLine; j0 >= 0; j0 -= kLine) {
            d2 v[kLine / 2];
#pragma unroll
            for (int q = 0; q < kLine / 2; q++) v[q] = *reinterpret_cast<const d2 *>(a + j0 + 2 * q);
#pragma unroll
            for (int u = kLine - 1; u >= 0; u--) {
                const double aj = v[u >> 1][u & 1], cj = c[j0 + u], sj = s[j0 + u];
                const double temp = cj * aj - sj * an;
                an = sj * aj + cj * an;
                v[u >> 1][u & 1] = temp;
            }
#pragma unroll
            for (int q = 0; q < kLine / 2; q++) *reinterpret_cast<d2 *>(a + j0 + 2 * q) = v[q];
        }
    } else {
        for (int j0 = 0; j0 < top; j0 += kLine) {
            d2 v[kLine / 2];
#pragma unroll
            for (int q = 0; q < kLine / 2; q++) v[q] = *reinterpret_cast<const d2 *>(a + j0 + 2 * q);
#pragma unroll
            for (int u = 0; u < kLine; u++) {
                const double aj = v[u >> 1][u & 1], cj = c[j0 + u], sj = s[j0 + u];
                const double temp = cj * aj + sj * an;
                an = -sj * aj + cj * an;
                v[u >> 1][u & 1] = temp;
            }
#pragma unroll
            for (int q = 0; q < kLine / 2; q++) *reinterpret_cast<d2 *>(a + j0 + 2 * q) = v[q];
        }
        for (int j = top; j <= last; j++) {
            const double aj = a[j];
            const double temp = c[j] * aj + s[j] * an;
            an = -s[j] * aj + c[j] * an;
            a[j] = temp;
        }
    }
    return an;
}
#endif

// every row of Q through the two rotation sweeps whose cosines / sines are in f0 .. f3 (a row per thread, no barrier inside)
template <class E>
SOCP_HD void rotate_all_rows(const E &ex, int n, int ld, Work &wk)
{
    const double *c1 = wk.f[0], *s1 = wk.f[1], *c2 = wk.f[2], *s2 = wk.f[3];
    SOCP_PAR_FOR(i, 0, n) {                                  // a row of Q goes through all rotations on its own
        double *a = wk.A + (long)i * ld;
        double an = a[n - 1];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SOCP_SOLVER_NARROW_ROWS)
        an = rotate_row_lines(a, c1, s1, n, an, true);       // (whole cache lines per thread; the same arithmetic)
        an = rotate_row_lines(a, c2, s2, n, an, false);
#else
        an = rotate_row(a, c1, s1, n, an, true);
        an = rotate_row(a, c2, s2, n, an, false);
#endif
        a[n - 1] = an;
    }
}

// minpack.cpp: r1mpyq on Q (n x n, row-major here) and on qtf, with the rotations of the last r1updt (f7, f4) decoded into
// f0 .. f3
template <class E>
SOCP_HD void r1mpyq_all(const E &ex, int n, int ld, Work &wk)
{
    double *c1 = wk.f[0], *s1 = wk.f[1], *c2 = wk.f[2], *s2 = wk.f[3];
    ex.sync();
    SOCP_PAR_FOR(j, 0, n - 1) {
        decode_rotation(wk.f[7][j], c1[j], s1[j]);
        decode_rotation(wk.f[4][j], c2[j], s2[j]);
    }
    ex.sync();
    rotate_all_rows(ex, n, ld, wk);
    if (ex.tid == ex.nt - 1) {                               // qtf is the one-row case (on the thread least likely to own a row)
        double *a = wk.qtf;
        double an = a[n - 1];
        an = rotate_row(a, c1, s1, n, an, true);
        an = rotate_row(a, c2, s2, n, an, false);
        a[n - 1] = an;
    }
    ex.sync();
}

// ---- Q kept as factorised (Config::lazy_q; the throughput flavour only: results move at rounding level).  hybrd needs Q in ONE
// place, Q^T f of the trial residual in Broyden's update, and pays for keeping it current with r1mpyq on the whole matrix after
// every trial step: n^2 doubles read and n^2 written, a quarter of a trial step's memory traffic.  With
// Q_k = Q_0 G_1 ... G_k (G_u = the 2 (n - 1) rotations of update u), Q_k^T f = G_k^T ... G_1^T (Q_0^T f): the rotations of the
// updates since the last refresh are kept -- decoded, 4 n doubles per update, in the workspace area of the packed Householder
// vectors, which only the order-preserving blocked factorisation uses -- and applied to the VECTOR Q_0^T f (what r1mpyq does to
// its one-row argument qtf anyway).  When the list is full the updates are applied to the matrix after all (lazy_flush).
SOCP_HD int lazy_capacity(int n)
{
    const long room = ((long)n * (n + 1) / 2) / (4L * n);
    return (int)(room > 16 ? 16 : room);
}
SOCP_HD bool lazy_applies(const Config &c) { return c.lazy_q != 0 && lazy_capacity(c.n) >= 2; }

#if defined(__HIP_DEVICE_COMPILE__)
// One sweep of r1mpyq's n - 1 rotations through a VECTOR u, by the whole workgroup.  Rotation q touches u[j] and the running last
// entry `an` only (first sweep: j = n - 2 - q, an' = s u[j] + c an; second sweep: j = q, an' = -s u[j] + c an), so `an` goes
// through a chain of affine maps x -> c x + (+-s u[j]) whose composition is associative: every thread composes the maps of its
// chunk of rotations, an exclusive scan over the threads (shuffles inside a wavefront, LDS across them) gives each its incoming
// `an`, and the chunk is then walked again with numbers.  2 chunk + 6 shuffle steps instead of n - 1 dependent ones on one thread.
// red: 2 doubles of LDS per wavefront.  Returns the outgoing `an` (the same in every thread); u is complete when it returns.
struct Affine {
    double a, b;
};
__device__ __forceinline__ Affine affine_after(const Affine &first, const Affine &then) { return {then.a * first.a, then.a * first.b + then.b}; }

template <bool FIRST>
__device__ __forceinline__ double sweep_scan(const BlockExec &ex, int n, double *u, const double *c, const double *s, double an, double *red)
{
    const int m = n - 1;
    const int chunk = (m + ex.nt - 1) / ex.nt;
    const int q0 = ex.tid * chunk, q1 = (q0 + chunk < m) ? q0 + chunk : m;
    Affine mine{1.0, 0.0};
    for (int q = q0; q < q1; q++) {
        const int j = FIRST ? m - 1 - q : q;
        const double sj = s[j], uj = u[j];
        mine = affine_after(mine, Affine{c[j], FIRST ? sj * uj : -(sj * uj)});
    }
    Affine incl = mine;
    const int lane = ex.tid & 63, wave = ex.tid >> 6, nw = (ex.nt + 63) >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Affine o{__shfl_up(incl.a, d), __shfl_up(incl.b, d)};
        if (lane >= d) incl = affine_after(o, incl);
    }
    if (lane == 63) { red[2 * wave] = incl.a; red[2 * wave + 1] = incl.b; }
    ex.sync();
    Affine before{1.0, 0.0};
    for (int w = 0; w < wave; w++) before = affine_after(before, Affine{red[2 * w], red[2 * w + 1]});
    Affine total = before;
    for (int w = wave; w < nw; w++) total = affine_after(total, Affine{red[2 * w], red[2 * w + 1]});
    Affine left{__shfl_up(incl.a, 1), __shfl_up(incl.b, 1)};
    if (lane == 0) left = Affine{1.0, 0.0};
    const Affine pre = affine_after(before, left);
    double a0 = pre.a * an + pre.b;                          // `an` as this thread's chunk finds it
    for (int q = q0; q < q1; q++) {
        const int j = FIRST ? m - 1 - q : q;
        const double cj = c[j], sj = s[j], uj = u[j];
        if (FIRST) { u[j] = cj * uj - sj * a0; a0 = sj * uj + cj * a0; }
        else       { u[j] = cj * uj + sj * a0; a0 = -(sj * uj) + cj * a0; }
    }
    ex.sync();                                               // u complete, red free again
    return total.a * an + total.b;
}
#endif

// u <- G_count^T ... G_1^T u; everyone returns with u complete.  Device: every sweep as a scan (sweep_scan; red = f7, free between
// r1updt calls); host simulation: the serial chain through u[n - 1] as r1mpyq writes it (rounding differs between the two)
template <class E>
SOCP_HD void lazy_apply_to_vector(const E &ex, int n, const Work &wk, int count, double *u)
{
    ex.sync();
#if defined(__HIP_DEVICE_COMPILE__)
    double an = u[n - 1];
    ex.sync();                                               // (everyone has read it before anyone's sweep can finish)
#pragma unroll 1
    for (int k = 0; k < count; k++) {
        const double *c1 = wk.V + (long)k * 4 * n, *s1 = c1 + n, *c2 = s1 + n, *s2 = c2 + n;
        an = sweep_scan<true>(ex, n, u, c1, s1, an, wk.f[7]);
        an = sweep_scan<false>(ex, n, u, c2, s2, an, wk.f[7]);
    }
    if (ex.tid == 0) u[n - 1] = an;
    ex.sync();
#else
    if (ex.tid == 0) {
        double an = u[n - 1];
#pragma unroll 1
        for (int k = 0; k < count; k++) {
            const double *c1 = wk.V + (long)k * 4 * n, *s1 = c1 + n, *c2 = s1 + n, *s2 = c2 + n;
            an = rotate_row(u, c1, s1, n, an, true);
            an = rotate_row(u, c2, s2, n, an, false);
        }
        u[n - 1] = an;
    }
    ex.sync();
#endif
}

// the rotations of the r1updt that has just run (encodings in f7, f4) -> slot `slot` of the list; qtf goes through them now
template <class E>
SOCP_HD void lazy_store(const E &ex, int n, Work &wk, int slot)
{
    double *c1 = wk.V + (long)slot * 4 * n, *s1 = c1 + n, *c2 = s1 + n, *s2 = c2 + n;
    ex.sync();
    SOCP_PAR_FOR(j, 0, n - 1) {
        decode_rotation(wk.f[7][j], c1[j], s1[j]);
        decode_rotation(wk.f[4][j], c2[j], s2[j]);
    }
    ex.sync();
#if defined(__HIP_DEVICE_COMPILE__)
    double an = wk.qtf[n - 1];
    ex.sync();
    an = sweep_scan<true>(ex, n, wk.qtf, c1, s1, an, wk.f[7]);      // (the encodings in f7 are decoded: it is scratch now)
    an = sweep_scan<false>(ex, n, wk.qtf, c2, s2, an, wk.f[7]);
    if (ex.tid == 0) wk.qtf[n - 1] = an;
#else
    if (ex.tid == ex.nt - 1) {
        double *a = wk.qtf;
        double an = a[n - 1];
        an = rotate_row(a, c1, s1, n, an, true);
        an = rotate_row(a, c2, s2, n, an, false);
        a[n - 1] = an;
    }
#endif
    ex.sync();
}

// the list is full: Q <- Q G_1 ... G_count -- r1mpyq `count` times over, each update's tables brought to f0 .. f3 first (the row
// loop is r1mpyq_all's: with the tables read from the list in global memory it took 254 registers instead of 166)
template <class E>
SOCP_HD void lazy_flush(const E &ex, int n, int ld, Work &wk, int count)
{
#pragma unroll 1
    for (int k = 0; k < count; k++) {
        const double *t = wk.V + (long)k * 4 * n;
        ex.sync();                                           // f0 .. f3 are no longer read (the previous update's rows are done)
        SOCP_PAR_FOR(j, 0, n - 1) { wk.f[0][j] = t[j]; wk.f[1][j] = t[n + j]; wk.f[2][j] = t[2 * n + j]; wk.f[3][j] = t[3 * n + j]; }
        ex.sync();
        // (the narrow row sweep: a second copy of the line-wide one in this function took the kernel from 166 to 255 registers, and
        // a full list is rare -- a refresh usually comes first)
        SOCP_PAR_FOR(i, 0, n) {
            double *a = wk.A + (long)i * ld;
            double an = a[n - 1];
            an = rotate_row(a, wk.f[0], wk.f[1], n, an, true);
            an = rotate_row(a, wk.f[2], wk.f[3], n, an, false);
            a[n - 1] = an;
        }
    }
    ex.sync();
}

// ---- the state machine (Core::advance of minpack.cpp).  Every thread of the problem's workgroup calls it with the same
// arguments; scalars are computed by all threads alike and stored by thread 0.
template <class E>
struct Machine {
    const E &ex;
    const Config &c;
    State &st;                 // in global memory; `s` is this thread's copy
    State s;
    Work w;
    Prof prof;
    double *fast_matrix = nullptr;   // device: an LDS buffer of n * ld doubles for the factor work (null: work in place)
    double *blocked_panel = nullptr, *blocked_block = nullptr;   // buffers of factor_blocked (the host simulation's way to run it)
    SOCP_HD Machine(const E &e, const Config &cfg, State &state, double *base, double *fast_vectors = nullptr)
        : ex(e), c(cfg), st(state), s(state), w(base, cfg.n, cfg.ld, fast_vectors)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        // every thread has loaded the same 96 bytes: say so (v_readfirstlane), and the state starts in SCALAR registers -- 24 vector
        // registers less across the whole advance, a third of what the four-wavefront builds spilled
        static_assert(sizeof(State) % sizeof(int) == 0, "State is read word by word");
        int words[sizeof(State) / sizeof(int)];
        __builtin_memcpy(words, &s, sizeof(State));
        for (unsigned k = 0; k < sizeof(State) / sizeof(int); k++) words[k] = __builtin_amdgcn_readfirstlane(words[k]);
        __builtin_memcpy(&s, words, sizeof(State));
#endif
    }

    SOCP_HD void store() { ex.sync(); if (ex.tid == 0) st = s; }
    SOCP_HD void finish(int code) { s.info = code; s.phase = PH_DONE; s.req = RQ_DONE; }
    SOCP_HD void request_jac() { s.jeval = 1; s.phase = PH_JAC; s.req = RQ_JAC; }

    SOCP_HD void request_trial()
    {
        const int n = c.n;
        prof.mark(15, ex.tid);
        const bool gauss_newton = dogleg(ex, n, w, s.delta, &prof, c.fast_sums != 0);
        s.gn = (gauss_newton && !s.sing) ? 1 : 0;           // (sing: a zero on R's diagonal -- the back substitution then divided by a stand-in)
        prof.mark(PF_DOGLEG, ex.tid);
        double *sc = w.f[0];
        SOCP_PAR_FOR(j, 0, n) {
            const double p = -w.wa1[j];
            w.wa1[j] = p;
            w.wa2[j] = w.x[j] + p;
            sc[j] = w.diag[j] * p;
        }
        ex.sync();
        s.pnorm = enorm(n, sc);
        if (s.iter == 1) s.delta = min_of(s.delta, s.pnorm);
        s.phase = PH_TRIAL;
        s.req = RQ_FVEC; s.eval_sel = 1;
        prof.mark(PF_STEP_TAIL, ex.tid);
    }

    SOCP_HD bool after_jacobian()
    {
        const int n = c.n;
        if (c.analytic) s.njev += 1; else s.nfev += n;
        bool sing;
        if (s.pad) {
            // the factor kernel (kernels_solver.hip: factor_blocked on this problem) has been here: A = Q, r, qtf, wa1, wa2 are in place
            sing = s.sing != 0;
            s.pad = 0;
        } else if constexpr (!E::kInlineFactor) {
            // (never reached: launch_advance picks such a build only behind a factor kernel.  Should the engine ever break that rule,
            // the problem ends as "terminated" instead of iterating on an unfactored matrix)
            finish(-3);
            return false;
        } else if (blocked_panel) {
            sing = factor_blocked(ex, n, c.ld, w, blocked_panel, blocked_block);
        } else if (fast_matrix) {
            // the refresh streams its matrix ~2n/3 times: do that on a copy in LDS, bring Q back once
            double *const home = w.A;
            const long len = (long)n * c.ld;
            SOCP_PAR_FOR(e, 0, (int)len) fast_matrix[e] = home[e];
            ex.sync();
            w.A = fast_matrix;
            sing = factor(ex, n, c.ld, w);                   // wa1 = diag(R), wa2 = column norms
            w.A = home;
            SOCP_PAR_FOR(e, 0, (int)len) home[e] = fast_matrix[e];
            ex.sync();
        } else {
            prof.mark(15, ex.tid);
            sing = factor(ex, n, c.ld, w, &prof);
            prof.mark(PF_FACTOR, ex.tid);
        }
        if (s.iter == 1) {
            if (c.mode != 2) SOCP_PAR_FOR(j, 0, n) w.diag[j] = (w.wa2[j] == 0) ? 1.0 : w.wa2[j];
            ex.sync();
            SOCP_PAR_FOR(j, 0, n) w.f[0][j] = w.diag[j] * w.x[j];
            ex.sync();
            s.xnorm = enorm(n, w.f[0]);
            s.delta = c.factor * s.xnorm;
            if (s.delta == 0) s.delta = c.factor;
        }
        s.sing = sing ? 1 : 0;
        s.lazy = 0;                                          // A holds the Q of THIS factorisation
        ex.sync();
        if (c.mode != 2) SOCP_PAR_FOR(j, 0, n) w.diag[j] = max_of(w.diag[j], w.wa2[j]);
        ex.sync();
        prof.mark(PF_JAC_TAIL, ex.tid);
        return true;
    }

    SOCP_HD void after_trial()
    {
        const double p1 = .1, p5 = .5, p001 = .001, p0001 = 1e-4;
        const int n = c.n;
        s.nfev += 1;
        double *f4c = w.f[0], *pc = w.f[1], *pr = w.f[2], *sx = w.f[3];      // fast copies: trial residual, step, qtf + R p, diag x
        SOCP_PAR_FOR(j, 0, n) { f4c[j] = w.wa4[j]; pc[j] = w.wa1[j]; }
        ex.sync();
        const double fnorm1 = enorm(n, f4c);
        double actred = -1;
        if (fnorm1 < s.fnorm) { const double q = fnorm1 / s.fnorm; actred = 1 - q * q; }
        // predicted reduction from |qtf + R p|
        if (c.gn_shortcut && s.gn) {
            // Throughput flavour: p is the Gauss-Newton step -- the back substitution's solution of R x = qtf, negated -- so qtf + R p is
            // what that solve left over: rounding noise, 1e-16 of |qtf|, and the predicted reduction 1 - (noise / |f|)^2 rounds to
            // exactly 1 either way.  MINPACK forms the product regardless (a pass over R per trial step: 15 % of a config-5 trial
            // round, a quarter of a megabyte per problem); here it is taken as the zero it stands for.  Broyden's update below uses
            // the same vector: v moves at rounding level.  Not when R has a zero on its diagonal (the solve divided by a stand-in).
            SOCP_PAR_FOR(i, 0, n) pr[i] = 0.0;
        } else {
            SOCP_PAR_FOR(i, 0, n) {
                pr[i] = w.qtf[i] + dot_run(pc, w.r + row_off(n, i) - i, 1, i, n, 0.0);
            }
        }
        ex.sync();
        const double temp = enorm(n, pr);
        double prered = 0;
        if (temp < s.fnorm) { const double q = temp / s.fnorm; prered = 1 - q * q; }
        const double ratio = prered > 0 ? actred / prered : 0;

        if (ratio < p1) {
            s.ncsuc = 0; s.ncfail += 1; s.delta = p5 * s.delta;
        } else {
            s.ncfail = 0; s.ncsuc += 1;
            if (ratio >= p5 || s.ncsuc > 1) s.delta = max_of(s.delta, s.pnorm / p5);
            if (fabs(ratio - 1) <= p1) s.delta = s.pnorm / p5;
        }
        if (ratio >= p0001) {
            SOCP_PAR_FOR(j, 0, n) { const double xj = w.wa2[j]; w.x[j] = xj; sx[j] = w.diag[j] * xj; w.fvec[j] = f4c[j]; }
            ex.sync();
            s.xnorm = enorm(n, sx);
            s.fnorm = fnorm1;
            s.iter += 1;
        }
        s.nslow1 += 1; if (actred >= p001) s.nslow1 = 0;
        if (s.jeval) s.nslow2 += 1;
        if (actred >= p1) s.nslow2 = 0;

        if (s.delta <= c.xtol * s.xnorm || s.fnorm == 0) { finish(1); return; }
        int code = 0;
        if (s.nfev >= c.maxfev) code = 2;
        if (p1 * max_of(p1 * s.delta, s.pnorm) <= kEpsMch * s.xnorm) code = 3;
        if (s.nslow2 == 5) code = 4;
        if (s.nslow1 == 10) code = 5;
        if (code != 0) { finish(code); return; }
        if (s.ncfail == 2) { request_jac(); return; }

        // Broyden rank-1 update of (Q, R, Q^T f): v -> f4, u -> f5 (r1updt's inputs)
        prof.mark(PF_TRIAL_HEAD, ex.tid);
        const double pnorm = s.pnorm;
        const bool lazy = lazy_applies(c);
        if (lazy && s.lazy > 0) {
            // A is the Q of the last factorisation: Q^T f = (the updates since) applied to A^T f
            double *u = w.f[6];                              // (r1updt sets every entry of its w = f6 before reading it)
            SOCP_PAR_FOR(j, 0, n) u[j] = dot_run(f4c, w.A + j, c.ld, 0, n, 0.0);
            lazy_apply_to_vector(ex, n, w, s.lazy, u);
            SOCP_PAR_FOR(j, 0, n) {
                const double sum = u[j];
                w.f[4][j] = (sum - pr[j]) / pnorm;
                w.f[5][j] = w.diag[j] * ((w.diag[j] * pc[j]) / pnorm);
                if (ratio >= p0001) w.qtf[j] = sum;
            }
        } else {
            SOCP_PAR_FOR(j, 0, n) {
                const double sum = dot_run(f4c, w.A + j, c.ld, 0, n, 0.0);
                w.f[4][j] = (sum - pr[j]) / pnorm;
                w.f[5][j] = w.diag[j] * ((w.diag[j] * pc[j]) / pnorm);
                if (ratio >= p0001) w.qtf[j] = sum;
            }
        }
        ex.sync();
        prof.mark(PF_QTW, ex.tid);
        s.sing = r1updt(ex, n, w, &prof) ? 1 : 0;
        prof.mark(PF_R1UPDT, ex.tid);
        if (lazy) {
            if (s.lazy == lazy_capacity(n)) { lazy_flush(ex, n, c.ld, w, s.lazy); s.lazy = 0; }
            lazy_store(ex, n, w, s.lazy);
            s.lazy += 1;
        } else {
            r1mpyq_all(ex, n, c.ld, w);
        }
        prof.mark(PF_R1MPYQ, ex.tid);
        s.jeval = 0;
        request_trial();
    }

    // user_flag: what the evaluation of the pending request returned (< 0 aborts the solve, shooting.cpp:873)
    SOCP_HD void advance(int flag)
    {
        if (s.phase == PH_DONE) { s.req = RQ_DONE; store(); return; }
        if (s.phase != PH_INIT && flag < 0) { finish(flag); store(); return; }
        switch (s.phase) {
        case PH_INIT: {
            s.info = 0; s.nfev = 0; s.njev = 0;
            bool bad = c.n <= 0 || c.xtol < 0 || c.maxfev <= 0 || c.factor <= 0;
            if (!bad && c.mode == 2)
                for (int j = 0; j < c.n; j++) if (w.diag[j] <= 0) bad = true;
            if (bad) { finish(0); break; }
            s.phase = PH_F0;
            s.req = RQ_FVEC; s.eval_sel = 0;
            break;
        }
        case PH_F0:
            s.nfev = 1;
            SOCP_PAR_FOR(j, 0, c.n) w.f[0][j] = w.fvec[j];
            ex.sync();
            s.fnorm = enorm(c.n, w.f[0]);
            s.iter = 1; s.ncsuc = s.ncfail = s.nslow1 = s.nslow2 = 0;
            request_jac();
            break;
        case PH_JAC:
            if (after_jacobian()) request_trial();
            break;
        case PH_TRIAL:
            after_trial();
            break;
        default:
            s.req = RQ_DONE;
        }
        store();
    }
};

// (re)start a problem from x0 (socp_hybr_start with diag = NULL, mode 1)
template <class E>
SOCP_HD void start(const E &ex, const Config &c, State &st, double *base, const double *x0)
{
    Work w(base, c.n, c.ld);
    SOCP_PAR_FOR(j, 0, c.n) { w.x[j] = x0[j]; w.diag[j] = 1.0; }
    ex.sync();
    if (ex.tid == 0) {
        State s = st;
        s.phase = PH_INIT; s.req = RQ_DONE; s.eval_sel = 0; s.info = 0; s.nfev = 0; s.njev = 0;
        st = s;
    }
    ex.sync();
}

#undef SOCP_PAR_FOR

}  // namespace devsolver
}  // namespace socp
