// minpack.cpp -- Powell hybrid method (MINPACK hybrd / hybrj) as a resumable state machine,
// plus the CMinPack-compatible callback entry points built on it.
//
// Written from the published MINPACK algorithm (More, Garbow, Hillstrom, ANL-80-74; SURVEY.md
// Appendix A): forward-difference or user Jacobian, Householder QR without pivoting, dogleg
// trust-region step, Broyden rank-1 updates of the QR factors through Givens rotations.
// CMinPack is an un-vendored dependency of the reference (src/socp/CMakeLists.txt:11-24); the
// call sites this replaces are shooting.cpp:803-826 (hybrd) and :830-851 (hybrj).
// Iterates are validated against SciPy's MINPACK (scipy.optimize._minpack) in
// tests/test_minpack.py.
//
// Host code by design: the O(n^3) factor work stays on the CPU; what it
// asks for -- residuals and Jacobian columns -- is what the GPU computes.
#include "../../include/socp_solver.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "host_pool.hpp"

namespace {

using socp::Pool;      // host_pool.hpp: a solve's worker threads, handed a batch of tasks per panel / per sweep

constexpr double kEpsMch = DBL_EPSILON;      // dpmpar(1)
constexpr double kGiant = DBL_MAX;           // dpmpar(3)

// Euclidean norm with the three-accumulator scaling of MINPACK's enorm (robust against
// overflow / destructive underflow; keeps iterates comparable with other MINPACK builds).
double enorm(int n, const double *x)
{
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    for (int i = 0; i < n; i++) {
        const double xabs = std::fabs(x[i]);
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
    if (s1 != 0) return x1max * std::sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0) {
        if (s2 >= x3max) return std::sqrt(s2 * (1 + (x3max / s2) * (x3max * s3)));
        return std::sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * std::sqrt(s3);
}

// Householder QR of the n x n matrix a (column-major, ld lda), no column pivoting: on return the
// strict upper triangle holds R's off-diagonal, rdiag its diagonal, the lower trapezoid the
// Householder vectors; acnorm = input column norms.
//
// MINPACK's qrfac is a right-looking column algorithm: reflector j is finished on column j, then applied to
// every later column k, each column on its own (one dot product, one axpy).  For large n (the 832-unknown
// doubleIntegrator problem factors 832 x 832 twice per solve, ~0.8 Gflop each) the work is split by COLUMN: the
// reflectors of a panel of 32 columns are finished serially (that needs only the panel itself), then host threads
// bring the remaining columns up to date, each thread applying the panel's reflectors to its own columns in the
// order 0, 1, ... exactly as the serial code does.  Every number is computed by the same operations in the same
// order, so the factorisation is bit-identical for any thread count.  Threads are started per panel and joined
// (no spinning: the solver may run under a CPU quota).
namespace {
inline void finish_reflector(int n, int j, double *aj, double *rdiag)
{
    double ajnorm = enorm(n - j, aj + j);
    if (ajnorm != 0) {
        if (aj[j] < 0) ajnorm = -ajnorm;
        for (int i = j; i < n; i++) aj[i] /= ajnorm;
        aj[j] += 1;
    }
    rdiag[j] = -ajnorm;            // == 0 marks an identity reflector
}
inline void apply_reflector(int n, int j, const double *aj, double *ak)
{
    double sum = 0;
    for (int i = j; i < n; i++) sum += aj[i] * ak[i];
    const double temp = sum / aj[j];
    for (int i = j; i < n; i++) ak[i] -= temp * aj[i];
}
}  // namespace

void qrfac_nopivot(int n, double *a, int lda, double *rdiag, double *acnorm, int threads)
{
    for (int j = 0; j < n; j++) acnorm[j] = enorm(n, a + (size_t)j * lda);
    if (threads <= 1 || n < 2 * threads) {
        for (int j = 0; j < n; j++) {
            double *aj = a + (size_t)j * lda;
            finish_reflector(n, j, aj, rdiag);
            if (rdiag[j] != 0)
                for (int k = j + 1; k < n; k++) apply_reflector(n, j, aj, a + (size_t)k * lda);
        }
        return;
    }
    const int T = threads, NB = 32;
    for (int j0 = 0; j0 < n; j0 += NB) {
        const int j1 = std::min(j0 + NB, n);
        for (int j = j0; j < j1; j++) {                    // the panel: columns j0 .. j1-1 among themselves
            double *aj = a + (size_t)j * lda;
            finish_reflector(n, j, aj, rdiag);
            if (rdiag[j] != 0)
                for (int k = j + 1; k < j1; k++) apply_reflector(n, j, aj, a + (size_t)k * lda);
        }
        if (j1 >= n) break;
        auto work = [&](int tid) {                          // the rest: each column through reflectors j0 .. j1-1 in order
            for (int k = j1 + tid; k < n; k += T) {
                double *ak = a + (size_t)k * lda;
                for (int j = j0; j < j1; j++)
                    if (rdiag[j] != 0) apply_reflector(n, j, a + (size_t)j * lda, ak);
            }
        };
        const int use = std::min(T, n - j1);
        std::vector<std::thread> pool;
        for (int t = 1; t < use; t++) pool.emplace_back(work, t);
        work(0);
        for (std::thread &th : pool) th.join();
    }
}

// accumulate the orthogonal factor Q (n x n) from the Householder vectors left by qrfac.  Column j of Q is
// e_j pushed through the reflectors j, j-1, ..., 0 -- independent of every other column once the vectors are
// copied out of the way, so the columns are dealt out to threads with the serial operation order per column.
void qform(int n, double *q, int ldq, double *wa, int threads)
{
    if (threads <= 1 || n < 2 * threads) {
        for (int j = 1; j < n; j++)
            for (int i = 0; i < j; i++) q[i + (size_t)j * ldq] = 0;
        for (int l = 0; l < n; l++) {
            const int k = n - 1 - l;
            double *qk = q + (size_t)k * ldq;
            for (int i = k; i < n; i++) { wa[i] = qk[i]; qk[i] = 0; }
            qk[k] = 1;
            if (wa[k] != 0) {
                for (int j = k; j < n; j++) {
                    double *qj = q + (size_t)j * ldq;
                    double sum = 0;
                    for (int i = k; i < n; i++) sum += qj[i] * wa[i];
                    const double temp = sum / wa[k];
                    for (int i = k; i < n; i++) qj[i] -= temp * wa[i];
                }
            }
        }
        return;
    }
    // Householder vectors, packed: v_k = V[off[k] .. off[k] + n - k)
    std::vector<size_t> off(n);
    size_t total = 0;
    for (int k = 0; k < n; k++) { off[k] = total; total += (size_t)(n - k); }
    std::vector<double> V(total);
    for (int k = 0; k < n; k++) std::memcpy(V.data() + off[k], q + (size_t)k * ldq + k, sizeof(double) * (n - k));
    const int T = threads;
    auto work = [&](int tid) {
        for (int j = tid; j < n; j += T) {
            double *qj = q + (size_t)j * ldq;
            for (int i = 0; i < n; i++) qj[i] = 0;
            qj[j] = 1;
            for (int k = j; k >= 0; k--) {
                const double *v = V.data() + off[k] - k;       // v[i] for i = k .. n-1
                if (v[k] == 0) continue;
                double sum = 0;
                for (int i = k; i < n; i++) sum += qj[i] * v[i];
                const double temp = sum / v[k];
                for (int i = k; i < n; i++) qj[i] -= temp * v[i];
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++) pool.emplace_back(work, t);
    work(0);
    for (std::thread &th : pool) th.join();
    (void)wa;
}

// ---------------------------------------------------------------------------------------------------------------------
// One Jacobian refresh's O(n^3) work -- qrfac without pivoting, qtf = Q^T fvec, R packed by rows, qform -- with the
// COLUMNS IN SIMD LANES.  MINPACK's column algorithm gives every column its own serial chain (one dot product, one
// division, one axpy per reflector); chains of different columns never mix.  Walking the matrix row-major, eight columns
// are one vector (and several vectors per task hide the add latency), so the lanes carry eight independent copies of the
// scalar recurrence: each number is produced by the same IEEE operations in the same order as in qrfac_nopivot / qform
// above -- bit-identical (tests/test_minpack.py) -- but the machine is no longer waiting on one 4-cycle add chain.
// Compiled without FMA contraction; the ISA is chosen at load time (AVX-512 / AVX2 / baseline), which changes the vector
// width, never a rounding.  The refresh of the 832-unknown doubleIntegrator problem: qrfac + qform 33 ms (16 threads,
// scalar chains) -> see profiles/ (tests/tools/qr_bench.cpp).
// ---------------------------------------------------------------------------------------------------------------------
namespace colvec {

typedef double v8 __attribute__((vector_size(64)));
typedef long long m8 __attribute__((vector_size(64)));
constexpr int W = 8;                 // columns per vector
#ifndef SOCP_COLVEC_G
#define SOCP_COLVEC_G 2
#endif
constexpr int G = SOCP_COLVEC_G;     // vectors per block: 16 columns advance together.  Measured on the EPYC 9575F at n = 832: one thread 43 ms for
                                     // G = 2, 4 alike (48 for 8); 16 threads 8.7 ms (G = 2), 12-20 (4), 16 (8): smaller blocks balance better
constexpr int CB = W * G;            // columns per block = reflectors per panel: a panel's columns are exactly one block
constexpr int kMinN = 24;            // below this the scalar code is as fast

#ifndef SOCP_ISA_CLONES          // tests build single-ISA variants (-DSOCP_ISA_CLONES= with or without -mavx2) to compare them all
#define SOCP_ISA_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#endif

typedef double v8u __attribute__((vector_size(64), aligned(8)));      // the same eight lanes at any address
// macros, not functions: a helper that returns a 64-byte vector would be compiled for the baseline ISA
#define ld8(p) ((v8)(*reinterpret_cast<const v8u *>(p)))
#define st8(p, x) (*reinterpret_cast<v8u *>(p) = (x))
#define splat(a) (v8{(a), (a), (a), (a), (a), (a), (a), (a)})

// Apply `count` reflectors, in order, to the block of CB columns starting at column first_col of the row-major matrix At
// (row stride ld).  Reflector t acts on rows row0[t] .. n-1 with the vector v[t][0 .. n - row0[t]) (v[t][0] = pivot entry)
// and is skipped when skip[t].  keep_upto[t]: columns <= keep_upto[t] of the block are left alone (the panel's own block:
// reflector j only moves columns > j); pass a value < first_col to move every column.
SOCP_ISA_CLONES
void apply_reflectors(int n, int count, const int *row0, const int *keep_upto, const unsigned char *skip, const double *const *v,
                      double *At, int ld, int first_col)
{
    for (int t = 0; t < count; t++) {
        if (skip[t]) continue;
        const int j = row0[t], len = n - j;
        const double *vt = v[t];
        double *base = At + (size_t)j * ld + first_col;
        v8 sum[G];
        for (int g = 0; g < G; g++) sum[g] = splat(0.0);
        for (int i = 0; i < len; i++) {
            const v8 vi = splat(vt[i]);
            const double *row = base + (size_t)i * ld;
            for (int g = 0; g < G; g++) sum[g] += vi * ld8(row + W * g);
        }
        v8 temp[G];
        const v8 piv = splat(vt[0]);
        for (int g = 0; g < G; g++) temp[g] = sum[g] / piv;
        if (keep_upto[t] < first_col) {
            for (int i = 0; i < len; i++) {
                const v8 vi = splat(vt[i]);
                double *row = base + (size_t)i * ld;
                for (int g = 0; g < G; g++) st8(row + W * g, ld8(row + W * g) - temp[g] * vi);
            }
        } else {
            m8 keep[G] = {};
            for (int g = 0; g < G; g++)
                for (int e = 0; e < W; e++) keep[g][e] = (first_col + W * g + e <= keep_upto[t]) ? -1LL : 0LL;
            for (int i = 0; i < len; i++) {
                const v8 vi = splat(vt[i]);
                double *row = base + (size_t)i * ld;
                for (int g = 0; g < G; g++) {
                    const v8 old = ld8(row + W * g);
                    const v8 upd = old - temp[g] * vi;
                    st8(row + W * g, keep[g] ? old : upd);
                }
            }
        }
    }
}

// qform for the block of CB columns j0 .. j0 + CB - 1 of Q (row-major Qt, row stride ld, initialised to the identity): column j
// is e_j pushed through the reflectors j, j-1, ..., 0 (v_k = V + off[k], rows k .. n-1; skipped when its pivot is zero).
SOCP_ISA_CLONES
void qform_block(int n, int j0, const double *V, const size_t *off, double *Qt, int ld)
{
    const int jtop = std::min(j0 + CB, n) - 1;
    for (int k = jtop; k >= 0; k--) {
        const double *vk = V + off[k];
        if (vk[0] == 0) continue;
        const int len = n - k;
        double *base = Qt + (size_t)k * ld + j0;
        v8 sum[G];
        for (int g = 0; g < G; g++) sum[g] = splat(0.0);
        for (int i = 0; i < len; i++) {
            const v8 vi = splat(vk[i]);
            const double *row = base + (size_t)i * ld;
            for (int g = 0; g < G; g++) sum[g] += ld8(row + W * g) * vi;
        }
        v8 temp[G];
        const v8 piv = splat(vk[0]);
        for (int g = 0; g < G; g++) temp[g] = sum[g] / piv;
        if (k <= j0) {
            for (int i = 0; i < len; i++) {
                const v8 vi = splat(vk[i]);
                double *row = base + (size_t)i * ld;
                for (int g = 0; g < G; g++) st8(row + W * g, ld8(row + W * g) - temp[g] * vi);
            }
        } else {
            m8 keep[G] = {};                           // columns j < k of this block have not reached reflector k yet
            for (int g = 0; g < G; g++)
                for (int e = 0; e < W; e++) keep[g][e] = (j0 + W * g + e < k) ? -1LL : 0LL;
            for (int i = 0; i < len; i++) {
                const v8 vi = splat(vk[i]);
                double *row = base + (size_t)i * ld;
                for (int g = 0; g < G; g++) {
                    const v8 old = ld8(row + W * g);
                    const v8 upd = old - temp[g] * vi;
                    st8(row + W * g, keep[g] ? old : upd);
                }
            }
        }
    }
}

// fjac (column-major, ld ldfjac): in = Jacobian, out = Q.  rdiag / acnorm as qrfac, qtf = Q^T fvec, r = R packed by rows.
void factor(int n, double *fjac, int ldfjac, const double *fvec, double *rdiag, double *acnorm, double *qtf, double *r, Pool &pool)
{
    const int threads = pool.size();
    static const bool trace = std::getenv("SOCP_LINALG_TRACE") != nullptr;      // phase times of every call, to stderr
    using clk = std::chrono::steady_clock;
    auto lap = [&](clk::time_point &t) { const clk::time_point now = clk::now(); const double ms = std::chrono::duration<double, std::milli>(now - t).count(); t = now; return ms; };
    clk::time_point tick = clk::now();
    double t_in = 0, t_panel = 0, t_trail = 0, t_out = 0, t_qform = 0;
    const int ld = ((n + 1 + CB - 1) / CB) * CB;                 // columns 0..n-1, column n = fvec -> qtf, zero padding
    // 64-byte aligned: a block's row segment is exactly four cache lines (no split loads, no line shared by two threads)
    std::vector<double> At_store((size_t)n * ld + 8, 0.0);
    double *const At = At_store.data() + ((64 - (reinterpret_cast<uintptr_t>(At_store.data()) & 63)) & 63) / sizeof(double);
    pool.run((n + 31) / 32, [&](int blk) {                       // column norms and the row-major copy (blocked transpose)
        const int jb = blk * 32, je = std::min(jb + 32, n);
        for (int j = jb; j < je; j++) acnorm[j] = enorm(n, fjac + (size_t)j * ldfjac);
        for (int ib = 0; ib < n; ib += 32)
            for (int j = jb; j < je; j++)
                for (int i = ib; i < std::min(ib + 32, n); i++) At[(size_t)i * ld + j] = fjac[i + (size_t)j * ldfjac];
    });
    // qtf = Q^T fvec rides along as column n: MINPACK's qtf loop (sum = v.q; t = -sum / v_j; q += v t) is the column update
    // (sum = v.a; t = sum / v_j; a -= t v) with both signs flipped, i.e. the same bits.
    for (int i = 0; i < n; i++) At[(size_t)i * ld + n] = fvec[i];

    std::vector<size_t> off(n);                                  // packed Householder vectors: v_k = V[off[k] .. off[k] + n - k)
    size_t total = 0;
    for (int k = 0; k < n; k++) { off[k] = total; total += (size_t)(n - k); }
    std::vector<double> V(total);

    int row0[CB], keep_upto[CB], keep_none[CB];
    unsigned char skip[CB];
    const double *vp[CB];
    for (int t = 0; t < CB; t++) keep_none[t] = -1;
    t_in = lap(tick);
    for (int j0 = 0; j0 < n; j0 += CB) {
        const int j1 = std::min(j0 + CB, n);
        for (int j = j0; j < j1; j++) {
            // finish reflector j on its column, which every earlier reflector has already been through: rows j .. n-1 -> v_j
            double *vj = V.data() + off[j];
            for (int i = j; i < n; i++) vj[i - j] = At[(size_t)i * ld + j];
            double ajnorm = enorm(n - j, vj);
            if (ajnorm != 0) {
                if (vj[0] < 0) ajnorm = -ajnorm;
                for (int i = 0; i < n - j; i++) vj[i] /= ajnorm;
                vj[0] += 1;
            }
            rdiag[j] = -ajnorm;
            const int t = j - j0;
            row0[t] = j; keep_upto[t] = j; skip[t] = (ajnorm == 0); vp[t] = vj;
            // the panel's own block (its later columns; in the last panel also fvec's column): right-looking, at once
            apply_reflectors(n, 1, row0 + t, keep_upto + t, skip + t, vp + t, At, ld, j0);
        }
        t_panel += lap(tick);
        // every block to the right: through the panel's reflectors in order, blocks dealt out to threads
        const int nblk = (ld - (j0 + CB)) / CB;
        pool.run(nblk, [&](int b) { apply_reflectors(n, j1 - j0, row0, keep_none, skip, vp, At, ld, j0 + CB * (b + 1)); });
        t_trail += lap(tick);
    }
    for (int i = 0; i < n; i++) qtf[i] = At[(size_t)i * ld + n];
    for (int i = 0, l = 0; i < n; i++) {                         // R by rows: row i = [rdiag[i], A(i, i+1 .. n-1)]
        r[l++] = rdiag[i];
        for (int k = i + 1; k < n; k++) r[l++] = At[(size_t)i * ld + k];
    }
    std::fill(At, At + (size_t)n * ld, 0.0);                        // Q: columns in lanes again, blocks independent of each other
    for (int j = 0; j < n; j++) At[(size_t)j * ld + j] = 1.0;
    const int qblocks = (n + CB - 1) / CB;
    t_out = lap(tick);
    pool.run(qblocks, [&](int b) { qform_block(n, (qblocks - 1 - b) * CB, V.data(), off.data(), At, ld); });   // heaviest first
    t_qform = lap(tick);
    pool.run((n + 31) / 32, [&](int blk) {
        const int jb = blk * 32;
        for (int ib = 0; ib < n; ib += 32)
            for (int i = ib; i < std::min(ib + 32, n); i++)
                for (int j = jb; j < std::min(jb + 32, n); j++) fjac[i + (size_t)j * ldfjac] = At[(size_t)i * ld + j];
    });
    if (trace)
        std::fprintf(stderr, "[colvec::factor n=%d threads=%d] set-up + transpose %.2f, panels (serial) %.2f, blocks to the right %.2f, qtf + R + identity %.2f, "
                             "qform %.2f, Q back %.2f ms\n", n, threads, t_in, t_panel, t_trail, t_out, t_qform, lap(tick));
}

#undef ld8
#undef st8
#undef splat
}  // namespace colvec

// SOCP_LINALG_VECTOR=0: the scalar column algorithm everywhere (A/B measurements, bit-identity tests)
bool colvec_enabled()
{
    static const bool on = [] { const char *e = std::getenv("SOCP_LINALG_VECTOR"); return !(e && e[0] == '0'); }();
    return on;
}

// dogleg step: minimiser of |R x - qtb| within the ellipsoid |diag x| <= delta, restricted to
// the span of the Gauss-Newton and scaled-gradient directions.  r: upper triangle by rows.
void dogleg(int n, const double *r, const double *diag, const double *qtb, double delta,
            double *x, double *wa1, double *wa2)
{
    // Gauss-Newton direction by back substitution
    int jj = n * (n + 1) / 2;
    for (int k = 1; k <= n; k++) {
        const int j = n - k;                 // 0-based row
        jj -= k;
        int l = jj + 1;
        double sum = 0;
        for (int i = j + 1; i < n; i++) { sum += r[l] * x[i]; l++; }
        double temp = r[jj];
        if (temp == 0) {
            l = j;
            for (int i = 0; i <= j; i++) { temp = std::max(temp, std::fabs(r[l])); l += n - i - 1; }
            temp = kEpsMch * temp;
            if (temp == 0) temp = kEpsMch;
        }
        x[j] = (qtb[j] - sum) / temp;
    }
    for (int j = 0; j < n; j++) { wa1[j] = 0; wa2[j] = diag[j] * x[j]; }
    const double qnorm = enorm(n, wa2);
    if (qnorm <= delta) return;

    // scaled gradient direction
    int l = 0;
    for (int j = 0; j < n; j++) {
        const double temp = qtb[j];
        for (int i = j; i < n; i++) { wa1[i] += r[l] * temp; l++; }
        wa1[j] /= diag[j];
    }
    const double gnorm = enorm(n, wa1);
    double sgnorm = 0;
    double alpha = delta / qnorm;
    if (gnorm != 0) {
        for (int j = 0; j < n; j++) wa1[j] = (wa1[j] / gnorm) / diag[j];
        l = 0;
        for (int j = 0; j < n; j++) {
            double sum = 0;
            for (int i = j; i < n; i++) { sum += r[l] * wa1[i]; l++; }
            wa2[j] = sum;
        }
        double temp = enorm(n, wa2);
        sgnorm = (gnorm / temp) / temp;
        alpha = 0;
        if (sgnorm < delta) {
            const double bnorm = enorm(n, qtb);
            const double dq = delta / qnorm, sd = sgnorm / delta;
            temp = (bnorm / gnorm) * (bnorm / qnorm) * sd;
            const double d1 = temp - dq;
            temp = temp - dq * (sd * sd) + std::sqrt(d1 * d1 + (1 - dq * dq) * (1 - sd * sd));
            alpha = (dq * (1 - sd * sd)) / temp;
        }
    }
    const double temp = (1 - alpha) * std::min(sgnorm, delta);
    for (int j = 0; j < n; j++) x[j] = temp * wa1[j] + alpha * x[j];
}

// Givens pair eliminating b against a, with the one-number encoding MINPACK stores so that
// r1mpyq can replay the rotation: tau = sin if |cos| >= |sin|... (see r1updt in the user guide).
inline void givens(double a, double b, double &cs, double &sn, double &tau)
{
    // eliminate b using pivot a
    if (std::fabs(a) < std::fabs(b)) {
        const double cotan = a / b;
        sn = 0.5 / std::sqrt(0.25 + 0.25 * (cotan * cotan));
        cs = sn * cotan;
        tau = 1;
        if (std::fabs(cs) * kGiant > 1) tau = 1 / cs;
    } else {
        const double tn = b / a;
        cs = 0.5 / std::sqrt(0.25 + 0.25 * (tn * tn));
        sn = cs * tn;
        tau = sn;
    }
}

// rank-1 update of the packed lower-trapezoidal s (here n x n): find orthogonal Q1 with
// (s + u v^T) Q1 lower trapezoidal again; rotations are encoded into v (first sweep) and w.
void r1updt(int n, double *s, const double *u, double *v, double *w, bool &sing)
{
    const int m = n;
    int jj = (n * (2 * m - n + 1)) / 2 - (m - n) - 1;      // 0-based index of the last diagonal
    int l = jj;
    for (int i = n - 1; i < m; i++) { w[i] = s[l]; l++; }
    for (int nmj = 1; nmj <= n - 1; nmj++) {
        const int j = n - 1 - nmj;
        jj -= (m - j);
        w[j] = 0;
        if (v[j] != 0) {
            double cs, sn, tau;
            givens(v[n - 1], v[j], cs, sn, tau);
            v[n - 1] = sn * v[j] + cs * v[n - 1];
            v[j] = tau;
            l = jj;
            for (int i = j; i < m; i++) {
                const double temp = cs * s[l] - sn * w[i];
                w[i] = sn * s[l] + cs * w[i];
                s[l] = temp;
                l++;
            }
        }
    }
    for (int i = 0; i < m; i++) w[i] += v[n - 1] * u[i];
    sing = false;
    for (int j = 0; j < n - 1; j++) {
        if (w[j] != 0) {
            double cs, sn, tau;
            givens(s[jj], w[j], cs, sn, tau);
            l = jj;
            for (int i = j; i < m; i++) {
                const double temp = cs * s[l] + sn * w[i];
                w[i] = -sn * s[l] + cs * w[i];
                s[l] = temp;
                l++;
            }
            w[j] = tau;
        }
        if (s[jj] == 0) sing = true;
        jj += (m - j);
    }
    l = jj;
    for (int i = n - 1; i < m; i++) { s[l] = w[i]; l++; }
    if (s[jj] == 0) sing = true;
}

inline void decode_rotation(double t, double &cs, double &sn)
{
    if (std::fabs(t) > 1) { cs = 1 / t; sn = std::sqrt(1 - cs * cs); }
    else { sn = t; cs = std::sqrt(1 - sn * sn); }
}

// a (m x n, column-major) <- a * Q1 with Q1 replayed from the encodings left in v and w by r1updt
void r1mpyq(int m, int n, double *a, int lda, const double *v, const double *w, Pool *pool = nullptr)
{
    // every row of a goes through the same 2(n-1) rotations and no rotation mixes rows: large matrices are split
    // by ROW RANGE over host threads (same operations per element, bit-identical for any count)
    auto rows = [&](int i0, int i1) {
        double *an = a + (size_t)(n - 1) * lda;
        for (int nmj = 1; nmj <= n - 1; nmj++) {
            const int j = n - 1 - nmj;
            double cs, sn;
            decode_rotation(v[j], cs, sn);
            double *aj = a + (size_t)j * lda;
            for (int i = i0; i < i1; i++) {
                const double temp = cs * aj[i] - sn * an[i];
                an[i] = sn * aj[i] + cs * an[i];
                aj[i] = temp;
            }
        }
        for (int j = 0; j < n - 1; j++) {
            double cs, sn;
            decode_rotation(w[j], cs, sn);
            double *aj = a + (size_t)j * lda;
            for (int i = i0; i < i1; i++) {
                const double temp = cs * aj[i] + sn * an[i];
                an[i] = -sn * aj[i] + cs * an[i];
                aj[i] = temp;
            }
        }
    };
    if (!pool || pool->size() <= 1 || m < 256) { rows(0, m); return; }
    const int T = std::min(pool->size(), m / 64);
    pool->run(T, [&](int t) { rows((int)((long)m * t / T), (int)((long)m * (t + 1) / T)); });
}

enum Phase { PH_INIT, PH_F0, PH_JAC, PH_TRIAL, PH_DONE };

struct Core {
    // configuration
    int n = 0, maxfev = 0, mode = 1, msum = 0, ldfjac = 0, lr = 0;
    double xtol = 0, epsfcn = 0, factor = 0;
    bool analytic = false, bad_input = false;
    int lin_threads = 1;            // host threads for qrfac / qform (bit-identical for any count; see qrfac_nopivot)
    std::shared_ptr<Pool> pool;     // the solve's worker threads: started at its first factorisation, kept until the solver goes
    Pool &workers()
    {
        if (!pool || pool->size() != std::max(1, lin_threads)) pool = std::make_shared<Pool>(lin_threads);
        return *pool;
    }
    // caller-visible arrays
    double *x = nullptr, *fvec = nullptr, *diag = nullptr, *fjac = nullptr, *r = nullptr, *qtf = nullptr;
    double *wa1 = nullptr, *wa2 = nullptr, *wa3 = nullptr, *wa4 = nullptr;
    // iteration state
    Phase phase = PH_INIT;
    int iter = 0, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0, nfev = 0, njev = 0, info = 0;
    bool jeval = false, sing = false;
    double delta = 0, xnorm = 0, fnorm = 0, pnorm = 0;

    int request_jac(const double **xe, double **out)
    {
        jeval = true;
        phase = PH_JAC;
        *xe = x; *out = fjac;
        return SOCP_REQ_JAC;
    }

    int request_trial(const double **xe, double **out)
    {
        // direction p, trial point x + p, scaled step length
        dogleg(n, r, diag, qtf, delta, wa1, wa2, wa3);
        for (int j = 0; j < n; j++) {
            wa1[j] = -wa1[j];
            wa2[j] = x[j] + wa1[j];
            wa3[j] = diag[j] * wa1[j];
        }
        pnorm = enorm(n, wa3);
        if (iter == 1) delta = std::min(delta, pnorm);
        phase = PH_TRIAL;
        *xe = wa2; *out = wa4;
        return SOCP_REQ_FVEC;
    }

    int finish(int code)
    {
        info = code;
        phase = PH_DONE;
        return SOCP_REQ_DONE;
    }

    void after_jacobian()
    {
        if (analytic) njev += 1; else nfev += msum;
        const bool vec = colvec_enabled() && n >= colvec::kMinN;
        if (vec) colvec::factor(n, fjac, ldfjac, fvec, wa1, wa2, qtf, r, workers());     // qrfac + qtf + R + qform, columns in SIMD lanes
        else qrfac_nopivot(n, fjac, ldfjac, wa1, wa2, lin_threads);      // wa1 = diag(R), wa2 = column norms
        if (iter == 1) {
            if (mode != 2)
                for (int j = 0; j < n; j++) diag[j] = wa2[j] == 0 ? 1.0 : wa2[j];
            for (int j = 0; j < n; j++) wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3);
            delta = factor * xnorm;
            if (delta == 0) delta = factor;
        }
        if (vec) {
            sing = false;
            for (int j = 0; j < n; j++) if (wa1[j] == 0) sing = true;
            if (mode != 2)
                for (int j = 0; j < n; j++) diag[j] = std::max(diag[j], wa2[j]);
            return;
        }
        // qtf = Q^T fvec from the Householder vectors
        for (int i = 0; i < n; i++) qtf[i] = fvec[i];
        for (int j = 0; j < n; j++) {
            const double *aj = fjac + (size_t)j * ldfjac;
            if (aj[j] != 0) {
                double sum = 0;
                for (int i = j; i < n; i++) sum += aj[i] * qtf[i];
                const double temp = -sum / aj[j];
                for (int i = j; i < n; i++) qtf[i] += aj[i] * temp;
            }
        }
        // R into packed row storage
        sing = false;
        for (int j = 0; j < n; j++) {
            int l = j;
            for (int i = 0; i < j; i++) { r[l] = fjac[i + (size_t)j * ldfjac]; l += n - i - 1; }
            r[l] = wa1[j];
            if (wa1[j] == 0) sing = true;
        }
        qform(n, fjac, ldfjac, wa1, lin_threads);
        if (mode != 2)
            for (int j = 0; j < n; j++) diag[j] = std::max(diag[j], wa2[j]);
    }

    // returns true when the solve continues with another trial on the current factorisation
    int after_trial(const double **xe, double **out)
    {
        const double p1 = .1, p5 = .5, p001 = .001, p0001 = 1e-4;
        nfev += 1;
        const double fnorm1 = enorm(n, wa4);
        double actred = -1;
        if (fnorm1 < fnorm) { const double q = fnorm1 / fnorm; actred = 1 - q * q; }
        // predicted reduction from |qtf + R p|
        int l = 0;
        for (int i = 0; i < n; i++) {
            double sum = 0;
            for (int j = i; j < n; j++) { sum += r[l] * wa1[j]; l++; }
            wa3[i] = qtf[i] + sum;
        }
        const double temp = enorm(n, wa3);
        double prered = 0;
        if (temp < fnorm) { const double q = temp / fnorm; prered = 1 - q * q; }
        const double ratio = prered > 0 ? actred / prered : 0;

        if (ratio < p1) {
            ncsuc = 0; ncfail += 1; delta = p5 * delta;
        } else {
            ncfail = 0; ncsuc += 1;
            if (ratio >= p5 || ncsuc > 1) delta = std::max(delta, pnorm / p5);
            if (std::fabs(ratio - 1) <= p1) delta = pnorm / p5;
        }
        if (ratio >= p0001) {
            for (int j = 0; j < n; j++) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; fvec[j] = wa4[j]; }
            xnorm = enorm(n, wa2);
            fnorm = fnorm1;
            iter += 1;
        }
        nslow1 += 1; if (actred >= p001) nslow1 = 0;
        if (jeval) nslow2 += 1;
        if (actred >= p1) nslow2 = 0;

        if (delta <= xtol * xnorm || fnorm == 0) return finish(1);
        int code = 0;
        if (nfev >= maxfev) code = 2;
        if (p1 * std::max(p1 * delta, pnorm) <= kEpsMch * xnorm) code = 3;
        if (nslow2 == 5) code = 4;
        if (nslow1 == 10) code = 5;
        if (code != 0) return finish(code);
        if (ncfail == 2) return request_jac(xe, out);

        // Broyden rank-1 update of (Q, R, Q^T f)
        for (int j = 0; j < n; j++) {
            const double *qj = fjac + (size_t)j * ldfjac;
            double sum = 0;
            for (int i = 0; i < n; i++) sum += qj[i] * wa4[i];
            wa2[j] = (sum - wa3[j]) / pnorm;
            wa1[j] = diag[j] * ((diag[j] * wa1[j]) / pnorm);
            if (ratio >= p0001) qtf[j] = sum;
        }
        r1updt(n, r, wa1, wa2, wa3, sing);
        r1mpyq(n, n, fjac, ldfjac, wa2, wa3, (lin_threads > 1 && n >= 256) ? &workers() : nullptr);
        r1mpyq(1, n, qtf, 1, wa2, wa3);
        jeval = false;
        return request_trial(xe, out);
    }

    int advance(int flag, const double **xe, double **out)
    {
        if (phase == PH_DONE) return SOCP_REQ_DONE;
        if (phase != PH_INIT && flag < 0) return finish(flag);
        switch (phase) {
        case PH_INIT:
            info = 0; nfev = 0; njev = 0;
            if (bad_input) return finish(0);
            if (mode == 2)
                for (int j = 0; j < n; j++) if (diag[j] <= 0) return finish(0);
            phase = PH_F0;
            *xe = x; *out = fvec;
            return SOCP_REQ_FVEC;
        case PH_F0:
            nfev = 1;
            fnorm = enorm(n, fvec);
            iter = 1; ncsuc = ncfail = nslow1 = nslow2 = 0;
            return request_jac(xe, out);
        case PH_JAC:
            after_jacobian();
            return request_trial(xe, out);
        case PH_TRIAL:
            return after_trial(xe, out);
        default:
            return SOCP_REQ_DONE;
        }
    }
};

// MINPACK fdjac1 through a one-point callback: dense when ml+mu+1 >= n, banded otherwise.
int fdjac1(cminpack_func_nn fcn, void *p, int n, double *x, const double *fvec, double *fjac, int ldfjac,
           int ml, int mu, double epsfcn, double *wa1, double *wa2)
{
    const double eps = std::sqrt(std::max(epsfcn, kEpsMch));
    const int msum = ml + mu + 1;
    int iflag = 0;
    if (msum >= n) {
        for (int j = 0; j < n; j++) {
            const double temp = x[j];
            double h = eps * std::fabs(temp);
            if (h == 0) h = eps;
            x[j] = temp + h;
            iflag = fcn(p, n, x, wa1, 2);
            if (iflag < 0) return iflag;
            x[j] = temp;
            for (int i = 0; i < n; i++) fjac[i + (size_t)j * ldfjac] = (wa1[i] - fvec[i]) / h;
        }
        return 0;
    }
    for (int k = 0; k < msum; k++) {
        for (int j = k; j < n; j += msum) {
            wa2[j] = x[j];
            double h = eps * std::fabs(wa2[j]);
            if (h == 0) h = eps;
            x[j] = wa2[j] + h;
        }
        iflag = fcn(p, n, x, wa1, 2);
        if (iflag < 0) return iflag;
        for (int j = k; j < n; j += msum) {
            x[j] = wa2[j];
            double h = eps * std::fabs(wa2[j]);
            if (h == 0) h = eps;
            for (int i = 0; i < n; i++) {
                fjac[i + (size_t)j * ldfjac] = 0;
                if (i >= j - mu && i <= j + ml) fjac[i + (size_t)j * ldfjac] = (wa1[i] - fvec[i]) / h;
            }
        }
    }
    return 0;
}

void bind(Core &s, int n, double *x, double *fvec, double xtol, int maxfev, double epsfcn, double *diag, int mode,
          double factor, double *fjac, int ldfjac, double *r, int lr, double *qtf,
          double *wa1, double *wa2, double *wa3, double *wa4)
{
    s.n = n; s.x = x; s.fvec = fvec; s.xtol = xtol; s.maxfev = maxfev; s.epsfcn = epsfcn; s.diag = diag;
    s.mode = mode; s.factor = factor; s.fjac = fjac; s.ldfjac = ldfjac; s.r = r; s.lr = lr; s.qtf = qtf;
    s.wa1 = wa1; s.wa2 = wa2; s.wa3 = wa3; s.wa4 = wa4;
    s.bad_input = n <= 0 || xtol < 0 || maxfev <= 0 || factor <= 0 || ldfjac < n || lr < n * (n + 1) / 2;
    s.phase = PH_INIT;
}

// host threads for the O(n^3) factor work of ONE solve (the blocking entry points): none below n = 192, where a
// factorisation is < 10 Mflop; SOCP_LINALG_THREADS overrides (1 = serial).  Results do not depend on the count.
int auto_lin_threads(int n)
{
    if (const char *e = std::getenv("SOCP_LINALG_THREADS")) { const int t = std::atoi(e); return t < 1 ? 1 : (t > 64 ? 64 : t); }
    if (n < 192) return 1;
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hw ? hw : 1u));
}

}  // namespace

extern "C" {

int socp_hybrd_batched(cminpack_func_nn fcn, socp_fdjac_fn fdjac, void *p, int n, double *x, double *fvec,
                       double xtol, int maxfev, int ml, int mu, double epsfcn, double *diag, int mode,
                       double factor, int nprint, int *nfev, double *fjac, int ldfjac, double *r, int lr,
                       double *qtf, double *wa1, double *wa2, double *wa3, double *wa4)
{
    (void)nprint;
    Core s;
    bind(s, n, x, fvec, xtol, maxfev, epsfcn, diag, mode, factor, fjac, ldfjac, r, lr, qtf, wa1, wa2, wa3, wa4);
    s.analytic = false;
    s.lin_threads = auto_lin_threads(n);
    if (ml < 0 || mu < 0) s.bad_input = true;
    s.msum = std::min(ml + mu + 1, n);
    const double *xe = nullptr;
    double *out = nullptr;
    int flag = 0;
    for (;;) {
        const int req = s.advance(flag, &xe, &out);
        if (req == SOCP_REQ_DONE) break;
        if (req == SOCP_REQ_FVEC) {
            flag = fcn(p, n, xe, out, 1);
        } else if (fdjac) {
            flag = fdjac(p, n, xe, fvec, epsfcn, out, ldfjac);
        } else {
            flag = fdjac1(fcn, p, n, x, fvec, out, ldfjac, ml, mu, epsfcn, wa1, wa2);
        }
    }
    if (nfev) *nfev = s.nfev;
    return s.info;
}

int hybrd(cminpack_func_nn fcn, void *p, int n, double *x, double *fvec, double xtol, int maxfev,
          int ml, int mu, double epsfcn, double *diag, int mode, double factor, int nprint,
          int *nfev, double *fjac, int ldfjac, double *r, int lr, double *qtf,
          double *wa1, double *wa2, double *wa3, double *wa4)
{
    return socp_hybrd_batched(fcn, nullptr, p, n, x, fvec, xtol, maxfev, ml, mu, epsfcn, diag, mode, factor,
                              nprint, nfev, fjac, ldfjac, r, lr, qtf, wa1, wa2, wa3, wa4);
}

int hybrj(cminpack_funcder_nn fcn, void *p, int n, double *x, double *fvec, double *fjac, int ldfjac,
          double xtol, int maxfev, double *diag, int mode, double factor, int nprint,
          int *nfev, int *njev, double *r, int lr, double *qtf,
          double *wa1, double *wa2, double *wa3, double *wa4)
{
    (void)nprint;
    Core s;
    bind(s, n, x, fvec, xtol, maxfev, 0.0, diag, mode, factor, fjac, ldfjac, r, lr, qtf, wa1, wa2, wa3, wa4);
    s.analytic = true;
    s.lin_threads = auto_lin_threads(n);
    const double *xe = nullptr;
    double *out = nullptr;
    int flag = 0;
    for (;;) {
        const int req = s.advance(flag, &xe, &out);
        if (req == SOCP_REQ_DONE) break;
        if (req == SOCP_REQ_FVEC) flag = fcn(p, n, xe, out, fjac, ldfjac, 1);
        else flag = fcn(p, n, xe, fvec, out, ldfjac, 2);
    }
    if (nfev) *nfev = s.nfev;
    if (njev) *njev = s.njev;
    return s.info;
}

/* ---- resumable object ---- */
struct socp_hybr {
    Core core;
    double *x = nullptr, *fvec = nullptr, *diag = nullptr, *fjac = nullptr, *r = nullptr, *qtf = nullptr,
           *wa1 = nullptr, *wa2 = nullptr, *wa3 = nullptr, *wa4 = nullptr;
    std::vector<double> own;             // the workspace of a solver created on its own; empty when it lives in a pool's arena
    double epsfcn = 0;
};

namespace {
// workspace of one solver, in doubles, rounded to whole cache lines
size_t hybr_doubles(int n) { return (((size_t)n * n + (size_t)n * (n + 1) / 2 + 8 * (size_t)n) + 7) / 8 * 8; }

void hybr_init(socp_hybr *s, double *mem, int n, double xtol, int maxfev, double epsfcn, int mode, double factor, int analytic_jac)
{
    s->fjac = mem; mem += (size_t)n * n;
    s->r = mem; mem += (size_t)n * (n + 1) / 2;
    s->x = mem; s->fvec = mem + n; s->diag = mem + 2 * n; s->qtf = mem + 3 * n;
    s->wa1 = mem + 4 * n; s->wa2 = mem + 5 * n; s->wa3 = mem + 6 * n; s->wa4 = mem + 7 * n;
    for (int i = 0; i < n; i++) s->diag[i] = 1.0;
    s->epsfcn = epsfcn;
    bind(s->core, n, s->x, s->fvec, xtol, maxfev, epsfcn, s->diag, mode, factor, s->fjac, n, s->r, n * (n + 1) / 2, s->qtf,
         s->wa1, s->wa2, s->wa3, s->wa4);
    s->core.analytic = analytic_jac != 0;
    s->core.msum = n;
}
}  // namespace

socp_hybr *socp_hybr_create(int n, double xtol, int maxfev, double epsfcn, int mode, double factor, int analytic_jac)
{
    if (n <= 0) return nullptr;
    socp_hybr *s = new socp_hybr;
    s->own.assign(hybr_doubles(n), 0.0);
    hybr_init(s, s->own.data(), n, xtol, maxfev, epsfcn, mode, factor, analytic_jac);
    return s;
}

// A pool of `count` solvers of one size in ONE anonymous mapping, advised to use huge pages: the lock-step engine creates
// thousands of workspaces at once, and 4096 separate allocations of ~100 KB cost 86 ms of first-touch page faults from 16
// threads (a fifth of a 0.38 s sweep).  Fresh anonymous memory is zero; nothing else needs initialising.
struct socp_hybr_pool {
    void *arena = nullptr;
    size_t bytes = 0;
    std::vector<socp_hybr> solvers;
};

socp_hybr_pool *socp_hybr_pool_create(int count, int n, double xtol, int maxfev, double epsfcn, int mode, double factor, int analytic_jac)
{
    if (count < 0 || n <= 0) return nullptr;
    socp_hybr_pool *p = new socp_hybr_pool;
    const size_t per = hybr_doubles(n);
    p->bytes = ((per * sizeof(double) * (size_t)std::max(count, 1)) + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
    p->arena = mmap(nullptr, p->bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p->arena == MAP_FAILED) { delete p; return nullptr; }
#ifdef MADV_HUGEPAGE
    (void)madvise(p->arena, p->bytes, MADV_HUGEPAGE);
#endif
    p->solvers.resize(count);
    for (int i = 0; i < count; i++)
        hybr_init(&p->solvers[i], static_cast<double *>(p->arena) + per * (size_t)i, n, xtol, maxfev, epsfcn, mode, factor, analytic_jac);
    return p;
}

socp_hybr *socp_hybr_pool_get(socp_hybr_pool *p, int i) { return (p && i >= 0 && i < (int)p->solvers.size()) ? &p->solvers[i] : nullptr; }

void socp_hybr_pool_destroy(socp_hybr_pool *p)
{
    if (!p) return;
    if (p->arena && p->arena != MAP_FAILED) (void)munmap(p->arena, p->bytes);
    delete p;
}

void socp_hybr_destroy(socp_hybr *s) { delete s; }

int socp_hybr_start(socp_hybr *s, const double *x0, const double *diag)
{
    if (!s || !x0) return -1;
    std::memcpy(s->x, x0, sizeof(double) * s->core.n);
    if (diag) std::memcpy(s->diag, diag, sizeof(double) * s->core.n);
    else std::fill(s->diag, s->diag + s->core.n, 1.0);
    s->core.phase = PH_INIT;
    return 0;
}

int socp_hybr_advance(socp_hybr *s, int user_flag, const double **x_eval, double **out)
{
    return s->core.advance(user_flag, x_eval, out);
}

// host threads for this solver's factor work (default 1: the lock-step multi-start engine already runs one state
// machine per thread); any count gives the same bits
int socp_hybr_set_threads(socp_hybr *s, int threads)
{
    if (!s || threads < 1) return -1;
    s->core.lin_threads = threads;
    return 0;
}

int socp_hybr_info(const socp_hybr *s) { return s->core.info; }
int socp_hybr_nfev(const socp_hybr *s) { return s->core.nfev; }
int socp_hybr_njev(const socp_hybr *s) { return s->core.njev; }
const double *socp_hybr_x(const socp_hybr *s) { return s->x; }
const double *socp_hybr_fvec(const socp_hybr *s) { return s->fvec; }
double socp_hybr_epsfcn(const socp_hybr *s) { return s->epsfcn; }
void socp_hybr_trust_region(const socp_hybr *s, double *delta, double *xnorm, double *fnorm)
{
    if (delta) *delta = s->core.delta;
    if (xnorm) *xnorm = s->core.xnorm;
    if (fnorm) *fnorm = s->core.fnorm;
}

}  // extern "C"
