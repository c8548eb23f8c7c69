// kernels_factor_fast.hip -- the Jacobian refresh of the device Powell solver in its THROUGHPUT flavour.
//
// What it replaces: solver_dev.hpp `factor` -- qrfac (no pivoting) + Q^T fvec + R + qform of MINPACK's hybrd as the reference
// drives it (shooting.cpp:803-826) -- which keeps MINPACK's per-column operation order bit for bit and therefore
// streams the trailing matrix twice per REFLECTOR: 2048 problems of n = 253 move 354 GB for 2.7 GB of matrices (62 ms).
//
// Here the same factorisation (same Householder vectors v_j = a_j / |a_j| + e_j, same signs: diag(R) = -|a_j| sign(a_jj), same
// Q = H_0 H_1 ... H_{n-1}) is computed as a BLOCKED Householder QR with compact-WY panels of 16 reflectors, the trailing-matrix
// update on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), summation order free:
//
//     the panel   a 16-column strip (<= 256 rows) factorised by ONE wavefront in a ROW layout -- a lane owns rows, all 16 columns of a
//                 row in its registers -- with one batched wave-wide reduction per column (panel_rows, wave_reduce.hpp), V published to
//                 LDS at the end, then T of  H_j0 .. H_j0+15 = I - V T V^T  (larft's recurrence on the Gram matrix V^T V, one MFMA pass)
//     qrfac       TWO panels per pass over the trailing matrix: panel pp; the strip that is panel pp + 1 through panel pp, then factorised;
//                 every later 16-column strip (and the one that holds fvec, column n) through BOTH panels in one load / store:
//                   W = V^T S (MFMA, K = rows)   Y = T^T W (4 MFMA)   S -= V Y (MFMA, K = 16)
//                 with the strip in registers in the MFMA C/D layout (row = 16 chunk + g + 4 reg, column = lane & 15).  Because a sum
//                 over K has no prescribed order here, the K slot of lane group g in step `reg` is simply DEFINED to be that row: the
//                 strip's own registers are the B operand of the first product and the accumulator of the last one -- no layout
//                 change, no LDS round trip; only the A operands (V, T) come from LDS, each read conflict-free.
//     qform       Q = (I - V_0 T_0 V_0^T) ... (I - V_last T_last V_last^T) I: a strip of Q STAYS in a wavefront's registers (it starts
//                 as identity columns) while every panel that reaches it streams through LDS, last panel first; never read, written once.
//
// qrfac has two forms (launch_nch below picks by the number of problems in the launch; profiles/r06_factor_chain_crossover.txt):
//   * from 640 problems up a CHAIN OF LAUNCHES (round 6): per pair of panels qrfac_panel_kernel -- ONE wavefront per problem, every resident
//     wavefront works -- and qrfac_trail_kernel -- four wavefronts per problem for the strips right of the pair --, each instantiated by
//     strip height and launched at the height the pair needs;
//   * below that ONE launch of four-wavefront workgroups (round 5's kernel: factor_fast_kernel<., ., 1>), in which the panels are
//     factorised by one wavefront each while three wait: a launch of few problems is one problem's latency either way.
// qform (factor_fast_kernel<., ., 2>) follows either.  HBM traffic at 2048 x n = 253: 4.5 x the algorithmic bytes (the single launch 4.2 x;
// round 4: 8.3 x; the order-preserving kernel: 130 x).  Results differ from the
// order-preserving kernel at rounding level; the engine uses this kernel only when asked for the throughput flavour
// (SOCP_SOLVER_DEVICE_FAST, or AUTO on a throughput-flavour context).
// Sizes: 39 <= n <= 256 (fast_factor_applies: the strip of a panel must fit the registers of one wavefront); others keep `factor`.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <utility>

#include "solver_launch.hpp"
#include "wave_reduce.hpp"

namespace socp {
namespace devsolver {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int kLdV = 17;       // pitch (doubles) of a panel's V in LDS: rows of 16 entries + 1, so that the lanes of BOTH operand reads of
                               // the trailing update (16 consecutive entries of 4 rows; 4 consecutive entries of 16 rows) spread over the banks

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// The lane coordinates re-defined where they are used.  Everything that depends on the lane alone (which rows and columns a lane
// holds, on which side of a diagonal it sits: several hundred compares in the unrolled code below) is invariant in every loop of
// the kernel; the compiler hoists all of it to the kernel's first lines, keeps the masks in scalar registers for the whole run,
// runs out of them (sgpr_spill_count 670) and then out of vector registers.  A compare costs one instruction where it is needed.
__device__ __forceinline__ int here(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// the value lane `src` holds (src uniform), as a uniform
__device__ __forceinline__ double from_lane(double x, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}
// LDS written by some lanes of this wavefront, read by others: the wave's LDS operations execute in order; keep the compiler
// from moving the reads above the writes
// (not a memory-model fence: that also waits for the wave's outstanding GLOBAL stores -- a round trip to L2 per column of the panel
// when one lane stores diag(R) on the way, which was most of the panel's time)
__device__ __forceinline__ void wave_lds_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// A strip: 16 columns c0 .. c0 + 15, rows row0 .. row0 + 16 NCH - 1, four rows per 16-row chunk and lane:
// S[cc][reg] of the lane with row group g (0 .. 3) and column m (0 .. 15) = A(row0 + 16 cc + g + 4 reg, c0 + m).  With g = lane >> 4,
// m = lane & 15 this is the C/D layout of v_mfma_f64_16x16x4_f64; the panel wave uses g = lane & 3, m = lane >> 2 (a column's four row
// groups in one quad).  Rows >= n and columns >= colmax read as 0.
// One running pointer per lane, advanced by four rows per load (with the row index multiplied out per (chunk, register) the
// compiler hoists 4 NCH row offsets x ld into scalar registers for the whole kernel and spills them), and no per-row lane masks
// except in the one chunk that can be partly below the matrix (4 NCH hoisted exec masks were the other half of the spills):
// chunks before the last one are whole, chunks after it are zero -- uniform branches.
template <int NCH>
__device__ __forceinline__ void strip_load(f64x4 (&S)[NCH], const double *__restrict__ A, int ld, int n, int row0, int nch, int c0, int colmax, int g, int m)
{
    g = here(g); m = here(m);
    const int col = c0 + m;
    const bool cok = col < colmax;
    const double *p = A + (long)(row0 + g) * ld + col;
    const long step = 4L * ld;
#pragma unroll
    for (int cc = 0; cc < NCH; cc++) {
        if (cc + 1 < nch) {
#pragma unroll
            for (int r = 0; r < 4; r++) { S[cc][r] = cok ? *p : 0.0; p += step; }
        } else if (cc + 1 == nch) {
            const int left = n - (row0 + 16 * cc + g);                       // rows of this lane's group still inside the matrix: 4 r < left
#pragma unroll
            for (int r = 0; r < 4; r++) { S[cc][r] = (cok && 4 * r < left) ? *p : 0.0; p += step; }
        } else {
            S[cc] = f64x4{0, 0, 0, 0};
        }
    }
}
// (`skip`: that many chunks at the top are not written -- rows that have left as rows of R; 0, 1 or 2, uniform)
template <int NCH>
__device__ __forceinline__ void strip_store(const f64x4 (&S)[NCH], double *__restrict__ A, int ld, int n, int row0, int nch, int c0, int colmax, int g, int m,
                                            int skip = 0)
{
    g = here(g); m = here(m);
    const int col = c0 + m;
    if (col >= colmax) return;
    double *p = A + (long)(row0 + g) * ld + col;
    const long step = 4L * ld;
#pragma unroll
    for (int cc = 0; cc < NCH; cc++) {
        if (cc < skip) {
            p += 4 * step;
        } else if (cc + 1 < nch) {
#pragma unroll
            for (int r = 0; r < 4; r++) { *p = S[cc][r]; p += step; }
        } else if (cc + 1 == nch) {
            const int left = n - (row0 + 16 * cc + g);
#pragma unroll
            for (int r = 0; r < 4; r++) { if (4 * r < left) *p = S[cc][r]; p += step; }
        }
    }
}

// |column|_2 of the strip's columns from its registers -- the Jacobian's column norms (MINPACK's acnorm) are taken when a strip is
// first loaded, at panel 0, where every strip holds whole columns: no pass of its own over the matrix.  MFMA layout (a column's row
// groups 16 lanes apart).  Plain sum of squares where that is safe, the
// column's largest entry as scale otherwise (tiny / huge entries, zero columns; a NaN is handed on).  Every lane returns its
// column's norm.
template <int NCH>
__device__ __forceinline__ double strip_column_norm(const f64x4 (&S)[NCH])
{
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
    for (int cc = 0; cc < NCH; cc++) {
        q0 = __builtin_fma(S[cc][0], S[cc][0], q0); q1 = __builtin_fma(S[cc][1], S[cc][1], q1);
        q2 = __builtin_fma(S[cc][2], S[cc][2], q2); q3 = __builtin_fma(S[cc][3], S[cc][3], q3);
    }
    double ss = (q0 + q1) + (q2 + q3);
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    if (ss != ss) return ss;
    if (ss > 1e-280 && ss < 1e280) return sqrt(ss);
    double amax = 0.0;
#pragma unroll
    for (int cc = 0; cc < NCH; cc++) {
#pragma unroll
        for (int r = 0; r < 4; r++) amax = fmax(amax, fabs(S[cc][r]));
    }
    amax = fmax(amax, __shfl_xor(amax, 16));
    amax = fmax(amax, __shfl_xor(amax, 32));
    if (!(amax > 0 && amax < INFINITY)) return amax;                         // a zero column, or an infinity handed on
    double s2 = 0.0;
#pragma unroll
    for (int cc = 0; cc < NCH; cc++) {
#pragma unroll
        for (int r = 0; r < 4; r++) { const double x = S[cc][r] / amax; s2 += x * x; }
    }
    s2 += __shfl_xor(s2, 16);
    s2 += __shfl_xor(s2, 32);
    return amax * sqrt(s2);
}

// The first 16 rows of a strip are, once the panel whose rows they are has been applied, rows of R for good: they go to the
// packed R (row i of R: diag, then (i, i + 1 .. n - 1)) and, for the column that carries fvec, to Q^T fvec -- at the moment they are
// in registers, not in a pass of their own over the finished matrix.  `above`: only the entries strictly right of the diagonal
// (the panel's own block; a trailing strip lies wholly right of it).
__device__ __forceinline__ void r_rows_out(const f64x4 &top, double *__restrict__ rpack, double *__restrict__ qtf, int n, int row0, int c0, int g, int m,
                                           bool above)
{
    g = here(g); m = here(m);
    const int col = c0 + m;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = row0 + g + 4 * r;
        if (row < n && (!above || col > row)) {
            if (col < n) rpack[row_off(n, row) + (col - row)] = top[r];
            else if (col == n) qtf[row] = top[r];
        }
    }
}

// The passes over a strip's chunks run in blocks of kBlk chunks behind ONE uniform test per block (is any of it inside the matrix?):
// with a test per chunk every chunk is a basic block of its own, the LDS reads of one chunk cannot be issued while the previous
// chunk still computes, and each pass pays an LDS round trip per chunk -- measured: 6 000 cycles per column of the panel, most of
// the kernel.  Chunks of a block beyond the last row compute on zeros (the strip's registers and V's rows there are zero).  The
// block also bounds what the scheduler may hoist: 4 kBlk operands in flight beside the strip's 4 NCH registers.
// Two chunks per block (round 5; round 4 measured four as best, with the lane-mask branches of that kernel): eight operands in flight
// are enough to cover the LDS round trip, a strip whose chunk count is not a multiple of four computes on fewer zero chunks
// (n = 200: 13 chunks -> 14 instead of 16), and qform spills 37 registers instead of 53.  2048 x n = 253: 4.66 -> 4.46 ms,
// n = 200: 3.33 -> 3.07 ms; one chunk per block 4.58 / 3.10, three 4.58 / 3.19, eight 5.76 / 3.97 (profiles/r05_factor_blk_ab.txt).
// (Reading the NEXT block's operands while this block's products run -- two register sets, written out by hand -- is slower: 4.65 ms.)
#ifndef SOCP_FACTOR_BLK
#define SOCP_FACTOR_BLK 2
#endif
constexpr int kBlk = SOCP_FACTOR_BLK;
#define SOCP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ int blocks_of(int nch) { return (nch + kBlk - 1) / kBlk * kBlk; }      // chunks the passes touch

// S <- (I - V X^T V^T) S for the panel in LDS: Vl[row][kLdV] (rows relative to the panel's first row, zero above the diagonal and
// from the matrix's last row to the end of the block of chunks), Xl[16][16] row-major (X = T applies H_last .. H_first, i.e. the
// panel's Q^T: qrfac; X = T^T applies the panel's Q: qform).  S in the MFMA layout (g = lane >> 4, m = lane & 15).
// `lo`: chunks below it hold zeros of V (uniform): blocks wholly above chunk lo are skipped.
template <int NCH>
__device__ __forceinline__ void strip_apply(f64x4 (&S)[NCH], int nch, const double *Vl, const double *Xl, int lane, int lo = 0)
{
    lane = here(lane);
    const int g = lane >> 4, m = lane & 15;
    const double *vw = Vl + g * kLdV + m;                                    // W = V^T S:  A operand V(row of K slot g, m)
    f64x4 W = {0, 0, 0, 0};
#pragma unroll
    for (int cb = 0; cb < NCH; cb += kBlk) {
        if (cb < nch && cb + kBlk > lo) {
            // the block's operands first, ALL of them in flight, then the products: left to itself the compiler, short of registers,
            // reuses one register pair for every read and waits for each read before the next (an LDS round trip per operand)
            double a[4 * kBlk];
#pragma unroll
            for (int q = 0; q < 4 * kBlk; q++) a[q] = vw[(16 * cb + 4 * q) * kLdV];
            SOCP_SCHED_FENCE();
#pragma unroll
            for (int q = 0; q < 4 * kBlk; q++) if (cb + q / 4 < NCH) W = mfma(a[q], S[cb + q / 4][q % 4], W);   // B: the strip itself, K slot g = row g + 4 r of the chunk
            SOCP_SCHED_FENCE();
        }
    }
    f64x4 Y = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; r++) Y = mfma(Xl[64 * r + lane], W[r], Y);                                   // A: X(g + 4 r, m) = X^T(m, k); B: W(k = g + 4 r, m)
#pragma unroll
    for (int r = 0; r < 4; r++) Y[r] = -Y[r];
    SOCP_SCHED_FENCE();
    const double *vu = Vl + m * kLdV + g;                                    // S -= V Y:  A operand V(16 cc + m, k = g + 4 r)
#pragma unroll
    for (int cb = 0; cb < NCH; cb += kBlk) {
        if (cb < nch && cb + kBlk > lo) {
            double a[4 * kBlk];
#pragma unroll
            for (int q = 0; q < 4 * kBlk; q++) a[q] = vu[16 * (cb + q / 4) * kLdV + 4 * (q % 4)];
            SOCP_SCHED_FENCE();
#pragma unroll
            for (int q = 0; q < 4 * kBlk; q++) if (cb + q / 4 < NCH) S[cb + q / 4] = mfma(a[q], Y[q % 4], S[cb + q / 4]);   // B: -Y(k, m)
            SOCP_SCHED_FENCE();
        }
    }
}

// Development aid (-DSOCP_FACTOR_PROFILE): lane 0 of wave 0 of every workgroup adds the clock ticks between marks to per-phase totals
// (read_factor_profile; scripts/measure_factor.py prints them).  Compiled out otherwise.
#ifdef SOCP_FACTOR_PROFILE
__device__ unsigned long long g_fprof[16];
struct FProf {
    unsigned long long t;
    bool on;
    __device__ explicit FProf(int tid) : t(clock64()), on(tid == 0) {}
    __device__ void mark(int slot)
    {
        const unsigned long long now = clock64();
        if (on) atomicAdd(&g_fprof[slot], now - t);
        t = now;
    }
    // a nested measurement that leaves the outer one running: add the ticks since `from` to `slot`
    __device__ unsigned long long stamp() const { return clock64(); }
    __device__ void add(int slot, unsigned long long from) { if (on) atomicAdd(&g_fprof[slot], clock64() - from); }
};
#else
struct FProf {
    __device__ explicit FProf(int) {}
    __device__ void mark(int) {}
    __device__ unsigned long long stamp() const { return 0; }
    __device__ void add(int, unsigned long long) {}
};
#endif
enum { FP_NORMS = 0, FP_PANEL = 1, FP_PANEL_WAIT = 2, FP_TRAIL = 3, FP_TRAIL_WAIT = 4, FP_RPACK = 5, FP_QLOAD = 6, FP_QSTRIPS = 7, FP_QWAIT = 8,
       FP_LA_APPLY = 9, FP_LA_CONVERT = 10, FP_COLS = 11, FP_LA_STORE = 12, FP_T = 13 };

// 1 / x and 1 / sqrt(x) to double precision from the hardware estimates (relative error ~2^-24) in one cubic step each (error ~1e-22);
// for arguments well inside the exponent range only -- the callers below check that
__device__ __forceinline__ double rcp_in_range(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
__device__ __forceinline__ double rsqrt_in_range(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double d = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * d, __builtin_fma(0.375, d, 0.5), y);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The panel in the ROW layout (round 5).  Round 4's panel kept a column's four row groups in a quad of lanes, so every column
// step costs every lane a dot product and an axpy over its 4 NCH entries PLUS the reflector read back from LDS twice (192 LDS
// operations per column): 110 000 cycles per panel at n = 253, and the chain panel -> apply -> panel was what a problem's time was
// made of (profiles/r05a_factor_phases.txt).  Here a lane owns ROWS: P[q][m] = entry (64 q + lane, m) of the 16-column panel, all
// sixteen columns of a row in one lane's registers.  A column step is then
//     x_k  = (own rows') a_T . a_k  for the columns k >= T                NQ (16 - T) multiply-adds, no data movement
//     w_k  = wave-wide sums of the x_k, ALL OF THEM in one batched reduction (wave_reduce.hpp: 63 instructions), value k -> quad k
//     |a_T| = sqrt(w_T);  v = a_T / |a_T| + e_T;  v . a_k = w_k / |a_T| + a_Tk  (the row-T entry: e_T's share, fetched through LDS)
//     a_k -= tau (v . a_k) v                                               NQ (15 - T) multiply-adds, the coefficient of
//                                                                          column k read from quad k (v_readlane)
// -- one reduction per column instead of a norm AND a dot product, nothing read back from LDS.  The squares are summed
// unscaled; a column whose sum leaves 1e-280 .. 1e280 (tiny / huge entries, a zero column) is rescaled by a power of two --
// exactly -- and the step repeated (uniform branch, rare); a NaN is handed on as MINPACK's enorm does.
template <int NCH> struct Rows { static constexpr int NQ = (NCH + 3) / 4; };

// strip in the MFMA layout -> LDS tile[row][kLdV], rows counted from chunk FIRST of the strip
template <int NCH, int FIRST>
__device__ __forceinline__ void strip_to_tile(const f64x4 (&S)[NCH], double *tile, int lane)
{
    lane = here(lane);
    double *p = tile + (lane >> 4) * kLdV + (lane & 15);
#pragma unroll
    for (int cc = FIRST; cc < NCH; cc++) {
#pragma unroll
        for (int r = 0; r < 4; r++) p[(16 * (cc - FIRST) + 4 * r) * kLdV] = S[cc][r];
    }
}
// ... and back into registers, a lane per row: rows from 16 (NCH - FIRST) on are zero
template <int NCH, int FIRST>
__device__ __forceinline__ void tile_to_rows(double (&P)[Rows<NCH>::NQ][16], const double *tile, int lane)
{
    lane = here(lane);
#pragma unroll
    for (int q = 0; q < Rows<NCH>::NQ; q++) {
        const int row = 64 * q + lane;
        const bool in = row < 16 * (NCH - FIRST);
        const double *p = tile + row * kLdV;
#pragma unroll
        for (int m = 0; m < 16; m++) P[q][m] = in ? p[m] : 0.0;
    }
}
// the factorised panel's V for the matrix cores: tile[row][m] = v_m(row) from the diagonal down, zero above it, zero for a column
// that is no reflector (alive bit clear)
// FIRST = 1: the tile's rows are counted from 16 rows ABOVE the panel's first row (the second panel of a pair: the strips meet both
// panels at one offset); those 16 rows are zero
template <int NCH, int FIRST>
__device__ __forceinline__ void rows_to_tile_V(const double (&P)[Rows<NCH>::NQ][16], unsigned alive, double *tile, int lane)
{
    lane = here(lane);
    if (FIRST && lane < 16) {
#pragma unroll
        for (int m = 0; m < 16; m++) tile[lane * kLdV + m] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < Rows<NCH>::NQ; q++) {
        const int row = 64 * q + lane;
        if (row < 16 * (NCH - FIRST)) {
            double *p = tile + (row + 16 * FIRST) * kLdV;
#pragma unroll
            for (int m = 0; m < 16; m++) p[m] = (((alive >> m) & 1u) && (q > 0 || row >= m)) ? P[q][m] : 0.0;
        }
    }
}
template <int NCH>
__device__ __forceinline__ void tile_to_strip(f64x4 (&S)[NCH], const double *tile, int lane)
{
    lane = here(lane);
    const double *p = tile + (lane >> 4) * kLdV + (lane & 15);
#pragma unroll
    for (int cc = 0; cc < NCH; cc++) {
#pragma unroll
        for (int r = 0; r < 4; r++) S[cc][r] = p[(16 * cc + 4 * r) * kLdV];
    }
}
// rows of R from the panel's own block (strictly right of the diagonal; column n = Q^T fvec): the panel's first 16 rows are lanes 0 .. 15
__device__ __forceinline__ void r_rows_from_rows(const double (&top)[16], double *__restrict__ rpack, double *__restrict__ qtf, int n, int j0, int lane)
{
    lane = here(lane);
    const int row = j0 + lane;
    if (lane < 16 && row < n) {
        double *out = rpack + row_off(n, row) - row;
#pragma unroll
        for (int m = 1; m < 16; m++) {
            const int col = j0 + m;
            if (m > lane) {
                if (col < n) out[col] = top[m];
                else if (col == n) qtf[row] = top[m];
            }
        }
    }
}

template <int NQ, int T>
__device__ __forceinline__ double panel_dots(const double (&a)[NQ], const double (&P)[NQ][16], int lane)
{
    double x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < T) { x[k] = 0.0; continue; }
        // (k == T: the column with itself -- `a`, not P, which differs from it by the power-of-two rescaling of the rare path)
        double s = a[0] * (k == T ? a[0] : P[0][k]);
#pragma unroll
        for (int q = 1; q < NQ; q++) s = __builtin_fma(a[q], k == T ? a[q] : P[q][k], s);
        x[k] = s;
    }
    return reduce16(x, lane);
}

template <int NQ, int T>
__device__ __forceinline__ void panel_step(double (&P)[NQ][16], int np, double *rowbuf, int lane, double &tau_mine, double &rdiag_mine, unsigned &alive)
{
    if (T >= np) return;                                                     // (uniform)
    lane = here(lane);
    // row T of the panel -- final since step T - 1 -- for the quads of its columns: lane T -> LDS -> lane l takes entry l >> 2
    if (lane == T) {
#pragma unroll
        for (int k = T + 1; k < 16; k++) rowbuf[k] = P[0][k];
    }
    double a[NQ];                                                            // column T from its diagonal down
#pragma unroll
    for (int q = 0; q < NQ; q++) a[q] = (q == 0 && lane < T) ? 0.0 : P[q][T];
    double w = panel_dots<NQ, T>(a, P, lane);
    double ss = from_lane(w, 4 * T);
    double scale = 1.0, unscale = 1.0;
    if (!(ss > 1e-280 && ss < 1e280) && ss == ss) {
        // (uniform, rare) a zero column, or squares outside the exponent range: a power of two brings the largest entry into [1/2, 1)
        double amax = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; q++) amax = fmax(amax, fabs(a[q]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off));
        if (amax == 0.0) return;                                             // no reflector: tau = 0, diag(R) = 0, the column stays (zeros)
        if (amax < INFINITY) {
            int e = 0;
            (void)frexp(amax, &e);
            scale = ldexp(1.0, -e);
            unscale = ldexp(1.0, e);
#pragma unroll
            for (int q = 0; q < NQ; q++) a[q] *= scale;
            w = panel_dots<NQ, T>(a, P, lane);
            ss = from_lane(w, 4 * T);
        }
    }
    double ajnorm = ss * rsqrt_in_range(ss);
    const double ajj = from_lane(P[0][T], T) * scale;
    if (ajj < 0) ajnorm = -ajnorm;
    const double inv = rcp_in_range(ajnorm);
    const double tau = rcp_in_range(__builtin_fma(ajj, inv, 1.0));           // 1 / v_T, v_T in [1, 2]: ajnorm carries a(T, T)'s sign
    if (lane == T) { tau_mine = tau; rdiag_mine = -ajnorm * unscale; }
    alive |= 1u << T;
    wave_lds_fence();
    const double coef = __builtin_fma(w, inv, rowbuf[lane >> 2]) * tau;      // (v . a_k) tau in quad k (k > T)
    wave_lds_fence();
    double v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) v[q] = a[q] * inv;
    v[0] = (lane == T) ? v[0] + 1.0 : v[0];
#pragma unroll
    for (int q = 0; q < NQ; q++) P[q][T] = (q == 0 && lane < T) ? P[q][T] : v[q];
#pragma unroll
    for (int k = T + 1; k < 16; k++) {
        const double c = -from_lane(coef, 4 * k);
#pragma unroll
        for (int q = 0; q < NQ; q++) P[q][k] = __builtin_fma(c, v[q], P[q][k]);
    }
}

template <int NQ, int... T>
__device__ __forceinline__ void panel_steps(double (&P)[NQ][16], int np, double *rowbuf, int lane, double &tau_mine, double &rdiag_mine, unsigned &alive,
                                            std::integer_sequence<int, T...>)
{
    (panel_step<NQ, T>(P, np, rowbuf, lane, tau_mine, rdiag_mine, alive), ...);
}

// Factorises the 16-column panel held in the row layout (np of its columns are reflectors): R above the diagonal and the vectors
// from the diagonal down in P, rdiag[0 .. np) to memory, the alive mask (which columns ARE reflectors); returns tau_t in lane t.
// `scratch`: 16 doubles of LDS (the row hand-over).
template <int NCH>
__device__ __forceinline__ double panel_rows(double (&P)[Rows<NCH>::NQ][16], int np, double *scratch, double *__restrict__ rdiag, int lane, unsigned &alive, double &rdiag_mine)
{
    double tau_mine = 0.0;
    rdiag_mine = 0.0;
    alive = 0u;
    panel_steps<Rows<NCH>::NQ>(P, np, scratch, lane, tau_mine, rdiag_mine, alive, std::make_integer_sequence<int, 16>());
    if (here(lane) < np) rdiag[lane] = rdiag_mine;
    return tau_mine;
}

// ... and the panel's T of  H_first .. H_last = I - V T V^T: G = V^T V on the matrix cores, then larft's recurrence
// T(i, t) = -tau_t sum_{k = i}^{t - 1} T(i, k) G(k, t), T(t, t) = tau_t.  Called AFTER the strip has gone home: with the strip's
// 4 NCH registers still live beside it the recurrence spilled a hundred registers per panel, and those round trips to scratch
// memory were most of a panel's time.
template <int NCH>
__device__ __forceinline__ void panel_T(int nch, double tau_mine, const double *Vl, double *Tl, double *Gl, double *__restrict__ Tsave, int lane)
{
    lane = here(lane);
    {
        const int g = lane >> 4, m = lane & 15;
        const double *vw = Vl + g * kLdV + m;
        f64x4 G = {0, 0, 0, 0};
#pragma unroll
        for (int cb = 0; cb < NCH; cb += kBlk) {
            if (cb < nch) {
#pragma unroll
                for (int cc = cb; cc < cb + kBlk && cc < NCH; cc++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const double v = vw[(16 * cc + 4 * r) * kLdV];
                        G = mfma(v, v, G);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) Gl[(g + 4 * r) * 16 + m] = G[r];
    }
    wave_lds_fence();
    {
        const int i = here(lane) & 15;                                       // (every row group computes the same T; group 0 stores it)
        double Trow[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const double tau_t = from_lane(tau_mine, t);
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < t; k++) acc += Trow[k] * Gl[k * 16 + t];      // (Trow[k] = 0 for k < i: T is upper triangular)
            Trow[t] = (i < t) ? -tau_t * acc : (i == t ? tau_t : 0.0);
            SOCP_SCHED_FENCE();
        }
        if (lane < 16) {
#pragma unroll
            for (int t = 0; t < 16; t++) { Tl[i * 16 + t] = Trow[t]; Tsave[i * 16 + t] = Trow[t]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The pair's two strips through the LDS tiles (round 5, STAGED).  With one wavefront per panel doing everything itself, a panel's
// 83 000 cycles were 28 000 of column steps and the rest that wavefront loading its strip (15 000: 64 loads per lane behind the other
// workgroup's traffic), converting, storing the vectors back (10 000) -- while three wavefronts waited.  Here the strips travel
// through the tiles that will hold V anyway: ALL wavefronts bring both strips of the pair in (a quarter each) before the first
// panel starts; the panel wavefronts read their strip from LDS; the vectors go back to A from the tiles by the wavefronts that
// have nothing else to do ([A]'s during [C], [C]'s at the head of the trailing pass).  No register array lives across a barrier.
template <int NCH>
__device__ __forceinline__ void pair_tiles_load(double *V0, double *V1, const double *__restrict__ A, int ld, int n, int j0, int nch, bool two, int wave, int lane)
{
    lane = here(lane);
    const int g = lane >> 4, m = lane & 15;
    const bool ok0 = j0 + m <= n, ok1 = two && j0 + 16 + m <= n;             // (column n = fvec rides along)
#pragma unroll
    for (int q = 0; q < (NCH + 3) / 4; q++) {
        const int cc = wave + 4 * q;                                         // (uniform)
        if (cc < NCH) {
            double v0[4] = {0.0, 0.0, 0.0, 0.0}, v1[4] = {0.0, 0.0, 0.0, 0.0};
            if (cc < nch) {                                                  // (chunks below the matrix: zeros, no loads issued)
                const double *p = A + (long)(j0 + 16 * cc + g) * ld + j0 + m;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool in = j0 + 16 * cc + g + 4 * r < n;
                    v0[r] = (in && ok0) ? p[0] : 0.0;
                    v1[r] = (in && ok1) ? p[16] : 0.0;
                    p += 4L * ld;
                }
            }
            double *t0 = V0 + (16 * cc + g) * kLdV + m, *t1 = V1 + (16 * cc + g) * kLdV + m;
#pragma unroll
            for (int r = 0; r < 4; r++) { t0[4 * r * kLdV] = v0[r]; t1[4 * r * kLdV] = v1[r]; }
        }
    }
}
// a tile's vectors back to A: columns c0 .. c0 + 15 (<= n), rows j0 + 16 first .., chunk cc by share `idx` of `shares` (uniform)
template <int NCH>
__device__ __forceinline__ void tile_store(const double *tile, double *__restrict__ A, int ld, int n, int j0, int nch, int c0, int first, int idx, int shares, int lane)
{
    lane = here(lane);
    const int g = lane >> 4, m = lane & 15;
    if (c0 + m > n) return;
#pragma unroll
    for (int q = 0; q < NCH; q++) {
        const int cc = first + idx + shares * q;                             // (uniform)
        if (cc < NCH && cc < nch) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * cc + g + 4 * r;
                if (j0 + row < n) A[(long)(j0 + row) * ld + c0 + m] = tile[row * kLdV + m];
            }
        }
    }
}
// |column|_2 of a tile's 16 columns (pair 0: whole columns), one wavefront: lane = (column, quarter of the rows); strip_column_norm's rule
template <int NCH>
__device__ __forceinline__ void tile_column_norms(const double *tile, double *__restrict__ out, int ncols, int lane)
{
    lane = here(lane);
    const int col = lane & 15, part = lane >> 4;
    const double *p = tile + (part * 4 * NCH) * kLdV + col;
    double ss = 0.0;
#pragma unroll 8
    for (int i = 0; i < 4 * NCH; i++) { const double v = p[i * kLdV]; ss = __builtin_fma(v, v, ss); }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    double nrm = ss;
    if (ss == ss) {
        if (ss > 1e-280 && ss < 1e280) {
            nrm = sqrt(ss);
        } else {
            double amax = 0.0;
            for (int i = 0; i < 4 * NCH; i++) amax = fmax(amax, fabs(p[i * kLdV]));
            amax = fmax(amax, __shfl_xor(amax, 16));
            amax = fmax(amax, __shfl_xor(amax, 32));
            nrm = amax;
            if (amax > 0 && amax < INFINITY) {
                double s2 = 0.0;
                for (int i = 0; i < 4 * NCH; i++) { const double x = p[i * kLdV] / amax; s2 += x * x; }
                s2 += __shfl_xor(s2, 16);
                s2 += __shfl_xor(s2, 32);
                nrm = amax * sqrt(s2);
            }
        }
    }
    if (part == 0 && col < ncols) out[col] = nrm;
}

// ---------------------------------------------------------------------------------------------------------------------------
// qrfac as a CHAIN OF LAUNCHES (round 6).  Round 5's qrfac was one launch of four-wavefront workgroups in which a problem's panels were
// factorised by ONE wavefront while the other three waited -- holding their registers and the workgroup's 75 KB of LDS, so that a CU had two
// problems resident and, in a panel phase, one working wavefront per problem.  Timed on their own (profiles/r06_factor_probe.txt) the
// two halves of that launch simply ADD: 1.65 ms of panel phases + 1.65 ms of trailing passes at 2048 x n = 253, 0.8 + 0.2 ms at 4096 x n = 85.
// Now, per pair of panels (pp, pp + 1):
//   qrfac_panel_kernel   ONE wavefront per problem (64-thread workgroups, one LDS tile): strip pp -> row layout -> the 16 column steps -> V, T;
//                        strip pp + 1 through panel pp on the matrix cores, then factorised the same way.  Every resident wavefront works; a CU
//                        holds as many problems as tiles fit (4 at 16 chunks ... 12 at 4).  V goes back to A's lower trapezoid (zero above
//                        the diagonal), T to the workspace (Tsave), R's rows and Q^T f's entries to their places.
//   qrfac_trail_kernel   four wavefronts per problem: both panels' V, T from A / Tsave into the two LDS tiles, then every strip right of the pair
//                        through both panels in one load / store (round 5's [E], unchanged).
// The kernels are instantiated by strip height and each pair's launches take the height that pair NEEDS (n = 253: 16, 16, 12, 12, 8, 6, 4, 4
// chunks): late pairs compute on fewer zero rows and fit more problems per CU.  2 ceil(npanels / 2) launches instead of one; they queue behind
// each other on the stream (no host synchronisation in between).
template <int NCH, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void qrfac_panel_kernel(Config c, State *states, double *ws, long ws_stride, const int *__restrict__ list, int count, int pp, int final_launch)
{
    extern __shared__ double lds[];
    double *V0 = lds, *T0 = V0 + 16 * NCH * kLdV, *Gl = T0 + 256;
    const int lane = threadIdx.x, g = lane >> 4, m = lane & 15;
    const int n = c.n, ld = c.ld;
    const int npanels = (n + 15) >> 4;
    const int j0 = 16 * pp, j1 = j0 + 16, nch = (n - j0 + 15) >> 4;
    const bool two = pp + 1 < npanels;
    for (int b = blockIdx.x; b < count; b += gridDim.x) {
        const int p = list[b];
        Work w(ws + (long)p * ws_stride, n, ld, lds);
        double *A = w.A, *rdiag = w.wa1, *acnorm = w.wa2, *Tsave = w.V;
        FProf prof(lane);
        if (pp == 0) {
            // fvec rides along as column n (the Jacobian's column norms are taken from the strips as they are first loaded)
            for (int i = lane; i < n; i += 64) A[(long)i * ld + n] = w.fvec[i];
            __syncthreads();
        }
        prof.mark(FP_NORMS);
        {   // ---- panel pp: strip -> row layout -> the 16 columns -> V (LDS tile, A), R's rows, T
            f64x4 S[NCH];
            const int np0 = (n - j0 < 16) ? n - j0 : 16;
            const unsigned long long t_la = prof.stamp();
            strip_load<NCH>(S, A, ld, n, j0, nch, j0, n + 1, g, m);          // (column n = fvec rides along when it falls into this strip)
            if (pp == 0) {
                const double nrm = strip_column_norm<NCH>(S);
                if (g == 0 && m < n) acnorm[m] = nrm;
            }
            double P[Rows<NCH>::NQ][16];
            strip_to_tile<NCH, 0>(S, V0, lane);
            prof.add(FP_LA_APPLY, t_la);
            const unsigned long long t_cv = prof.stamp();
            wave_lds_fence();
            tile_to_rows<NCH, 0>(P, V0, lane);
            wave_lds_fence();
            prof.add(FP_LA_CONVERT, t_cv);
            const unsigned long long t_cols = prof.stamp();
            unsigned alive;
            double rdiag_mine;
            const double tau = panel_rows<NCH>(P, np0, V0, rdiag + j0, lane, alive, rdiag_mine);
            prof.add(FP_COLS, t_cols);
            const unsigned long long t_st = prof.stamp();
            r_rows_from_rows(P[0], w.r, w.qtf, n, j0, lane);
            rows_to_tile_V<NCH, 0>(P, alive, V0, lane);
            wave_lds_fence();
            tile_to_strip<NCH>(S, V0, lane);
            strip_store<NCH>(S, A, ld, n, j0, nch, j0, n + 1, g, m, 0);      // (A keeps the vectors: the trailing launch and qform read them back)
            if (lane < np0) w.r[row_off(n, j0 + lane)] = rdiag_mine;
            prof.add(FP_LA_STORE, t_st);
            const unsigned long long t_T = prof.stamp();
            panel_T<NCH>(nch, tau, V0, T0, Gl, Tsave + 256 * pp, lane);
            prof.add(FP_T, t_T);
        }
        if (two) {
            // ---- the strip that is panel pp + 1 through panel pp (matrix cores, V and T from the tile), then factorised; the tile is free
            // once the apply has read it and takes the strip's rows from the second chunk on (16 zero rows on top of its V: the trailing
            // strips meet both panels at one offset)
            f64x4 S[NCH];
            const unsigned long long t_la = prof.stamp();
            strip_load<NCH>(S, A, ld, n, j0, nch, j1, n + 1, g, m);
            if (pp == 0) {
                const double nrm = strip_column_norm<NCH>(S);
                if (g == 0 && 16 + m < n) acnorm[16 + m] = nrm;
            }
            wave_lds_fence();                                                // (T0: written by lanes 0 .. 15, read by all)
            strip_apply<NCH>(S, nch, V0, T0, lane);
            prof.add(FP_LA_APPLY, t_la);
            const unsigned long long t_cv = prof.stamp();
            // its first 16 rows are rows of R now (packed R / Q^T fvec); the rest is panel pp + 1.  (A keeps nothing of those rows: qform zeroes them.)
            r_rows_out(S[0], w.r, w.qtf, n, j0, j1, g, m, false);
            SOCP_SCHED_FENCE();
            double P[Rows<NCH>::NQ][16];
            strip_to_tile<NCH, 1>(S, V0, lane);
            wave_lds_fence();
            tile_to_rows<NCH, 1>(P, V0, lane);
            wave_lds_fence();
            const int np1 = (n - j1 < 16) ? n - j1 : 16;
            prof.add(FP_LA_CONVERT, t_cv);
            const unsigned long long t_cols = prof.stamp();
            unsigned alive;
            double rdiag_mine;
            const double tau = panel_rows<NCH>(P, np1, V0, rdiag + j1, lane, alive, rdiag_mine);
            prof.add(FP_COLS, t_cols);
            const unsigned long long t_st = prof.stamp();
            r_rows_from_rows(P[0], w.r, w.qtf, n, j1, lane);
            rows_to_tile_V<NCH, 1>(P, alive, V0, lane);
            wave_lds_fence();
            tile_to_strip<NCH>(S, V0, lane);
            strip_store<NCH>(S, A, ld, n, j0, nch, j1, n + 1, g, m, 1);
            if (lane < np1) w.r[row_off(n, j1 + lane)] = rdiag_mine;
            prof.add(FP_LA_STORE, t_st);
            const unsigned long long t_T = prof.stamp();
            panel_T<NCH>(nch, tau, V0, T0, Gl, Tsave + 256 * (pp + 1), lane);
            prof.add(FP_T, t_T);
        }
        prof.mark(FP_PANEL);
        if (final_launch) {
            // (the last launch of the chain: "singular" = a zero on R's diagonal; the flags the advance kernel reads)
            __syncthreads();
            int zero = 0;
            for (int j = lane; j < n; j += 64) zero |= (rdiag[j] == 0) ? 1 : 0;
            const int sing = __syncthreads_or(zero);
            if (lane == 0) { states[p].sing = sing ? 1 : 0; states[p].pad = 1; }
        }
        __syncthreads();                                                     // (the tile is free for the next problem of this workgroup)
    }
}

template <int NCH, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void qrfac_trail_kernel(Config c, State *states, double *ws, long ws_stride, const int *__restrict__ list, int count, int pp, int final_launch)
{
    extern __shared__ double lds[];
    constexpr int kPanelDoubles = 16 * NCH * kLdV + 256;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, m = lane & 15;
    const int n = c.n, ld = c.ld;
    const int npanels = (n + 15) >> 4;
    const int j0 = 16 * pp, j1 = j0 + 16, nch = (n - j0 + 15) >> 4;
    const bool two = pp + 1 < npanels;
    double *V0 = lds, *T0 = V0 + 16 * NCH * kLdV, *V1 = lds + kPanelDoubles, *T1 = V1 + 16 * NCH * kLdV;
    for (int b = blockIdx.x; b < count; b += gridDim.x) {
        const int p = list[b];
        Work w(ws + (long)p * ws_stride, n, ld, lds);
        double *A = w.A, *rdiag = w.wa1, *acnorm = w.wa2, *Tsave = w.V;
        FProf prof(tid);
        {
            // both panels' vectors from A's lower trapezoid (where the panel launch left them: zero above the diagonal and in the columns that
            // are no reflectors) into the tiles, rows counted from panel pp's first row -- panel pp + 1's tile with 16 zero rows on top --,
            // zero from the matrix's last row to the tile's end; T as panel_T left it
            const int t = tid & 15, r0 = tid >> 4;
            const bool c0ok = j0 + t <= n, c1ok = two && j1 + t <= n;
            double x0[NCH], x1[NCH];
#pragma unroll
            for (int it = 0; it < NCH; it++) {
                const int row = r0 + 16 * it, arow = j0 + row;
                const bool in = it < nch && arow < n;
                const double *q = A + (long)arow * ld + j0 + t;
                x0[it] = (in && c0ok) ? q[0] : 0.0;
                x1[it] = (in && c1ok && it > 0) ? q[16] : 0.0;
            }
            const double xt0 = Tsave[256 * pp + tid], xt1 = two ? Tsave[256 * (pp + 1) + tid] : 0.0;
#pragma unroll
            for (int it = 0; it < NCH; it++) {
                V0[(r0 + 16 * it) * kLdV + t] = x0[it];
                V1[(r0 + 16 * it) * kLdV + t] = x1[it];
            }
            T0[tid] = xt0;
            T1[tid] = xt1;
        }
        __syncthreads();
        prof.mark(FP_QLOAD);
        // the strips right of the panel(s); their first 16 (32) rows leave as rows of R and are not written back
        for (int c0 = j0 + (two ? 32 : 16) + 16 * wave; c0 <= n; c0 += 64) {
            f64x4 S[NCH];
            strip_load<NCH>(S, A, ld, n, j0, nch, c0, n + 1, g, m);
            if (pp == 0) {
                const double nrm = strip_column_norm<NCH>(S);
                if (g == 0 && c0 + m < n) acnorm[c0 + m] = nrm;
            }
            strip_apply<NCH>(S, nch, V0, T0, lane);
            r_rows_out(S[0], w.r, w.qtf, n, j0, c0, g, m, false);
            if (two) {
                strip_apply<NCH>(S, nch, V1, T1, lane);
                if (NCH > 1) r_rows_out(S[NCH > 1 ? 1 : 0], w.r, w.qtf, n, j1, c0, g, m, false);
            }
            strip_store<NCH>(S, A, ld, n, j0, nch, c0, n + 1, g, m, two ? 2 : 1);
        }
        prof.mark(FP_TRAIL);
        if (final_launch) {
            __syncthreads();
            int zero = 0;
            for (int j = tid; j < n; j += 256) zero |= (rdiag[j] == 0) ? 1 : 0;
            const int sing = __syncthreads_or(zero);
            if (tid == 0) { states[p].sing = sing ? 1 : 0; states[p].pad = 1; }
        }
        __syncthreads();
        prof.mark(FP_TRAIL_WAIT);
    }
}

// Round 5's forms: qrfac as ONE launch (what a launch of few problems takes: launch_nch below) of four-wavefront workgroups (PHASE 1: the pair's
// strips through the LDS tiles, the panels by one wavefront each while three wait), and qform (PHASE 2), which is what runs behind either qrfac.
// One workgroup of 256 threads per problem.  LDS (doubles): two panels' V[16 NCH][kLdV] and T[256] (qrfac: the pair being applied; qform:
// the panel being applied and the next one on its way), Gl[256]
// WPE: wavefronts per SIMD the registers are budgeted for -- 2 for the 16-chunk strip (128 of 256 VGPRs are the strip), more for the
// smaller strips: the kernel is latency-bound, other problems' wavefronts are what fills its waits
template <int NCH, int WPE, int PHASE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void factor_fast_kernel(Config c, State *states, double *ws, long ws_stride, const int *__restrict__ list, int count)
{
    static_assert(PHASE == 1 || PHASE == 2, "qrfac or qform");
    extern __shared__ double lds[];
    constexpr int kPanelDoubles = 16 * NCH * kLdV + 256;
    double *Gl = lds + 2 * kPanelDoubles;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, m = lane & 15;
    // (the wavefront's number as the SCALAR it is: which wavefront takes a panel, a chunk, a strip are scalar branches then)
    const int n = c.n, ld = c.ld;
    for (int b = blockIdx.x; b < count; b += gridDim.x) {
        const int p = list[b];
        Work w(ws + (long)p * ws_stride, n, ld, lds);
        double *A = w.A, *rdiag = w.wa1, *acnorm = w.wa2, *Tsave = w.V;
        FProf prof(tid);
        const int npanels = (n + 15) >> 4;
        if constexpr (PHASE == 1) {
        // ---- fvec rides along as column n (the column norms of the Jacobian are taken from the strips as they are first loaded)
        for (int i = tid; i < n; i += 256) A[(long)i * ld + n] = w.fvec[i];
        __syncthreads();
        prof.mark(FP_NORMS);
        // ---- qrfac, TWO panels per pass over the trailing matrix.  For the panels pp, pp + 1:
        //   [A] one wavefront factorises panel pp (row layout, panel_rows) -> V, T into buffer 0;
        //   [C] one wavefront takes the strip that is panel pp + 1 through panel pp and factorises it -> buffer 1 (its rows counted
        //       from panel pp's first row: 16 zero rows on top, so one strip of registers meets both panels at the same offsets);
        //   [E] all four wavefronts take the strips right of both panels through pp and pp + 1 in ONE load / store.
        // The wavefronts take turns at [A] and [C] (two workgroups share a CU: their serial parts should not share a SIMD).
        // The pair's strips travel through the LDS tiles (pair_tiles_load / tile_store above).
        double *V0 = lds, *T0 = V0 + 16 * NCH * kLdV, *V1 = lds + kPanelDoubles, *T1 = V1 + 16 * NCH * kLdV;
        for (int pp = 0; pp < npanels; pp += 2) {
            const int j0 = 16 * pp, j1 = j0 + 16, nch = (n - j0 + 15) >> 4;
            const bool two = pp + 1 < npanels;
            const int wa = pp & 3, wb = (pp + 1) & 3;                        // (pp is even: wa in {0, 2}, wb in {1, 3})
            {
                const unsigned long long t_la = prof.stamp();
                pair_tiles_load<NCH>(V0, V1, A, ld, n, j0, nch, two, wave, lane);
                __syncthreads();
                prof.add(FP_LA_APPLY, t_la);
                // (pair 0 holds whole columns: the Jacobian's column norms.  A tile is read by the wavefront that will overwrite it, or
                // before the barrier that lets that wavefront start)
                if (pp == 0 && two && wave == 3) tile_column_norms<NCH>(V1, acnorm + 16, n - 16, lane);
                if (wave == wa) {
                    const int np0 = (n - j0 < 16) ? n - j0 : 16;
                    if (pp == 0) tile_column_norms<NCH>(V0, acnorm, n, lane);
                    double P[Rows<NCH>::NQ][16];
                    const unsigned long long t_cv = prof.stamp();
                    tile_to_rows<NCH, 0>(P, V0, lane);
                    wave_lds_fence();
                    prof.add(FP_LA_CONVERT, t_cv);
                    const unsigned long long t_cols = prof.stamp();
                    unsigned alive;
                    double rdiag_mine;
                    const double tau = panel_rows<NCH>(P, np0, Gl, rdiag + j0, lane, alive, rdiag_mine);
                    prof.add(FP_COLS, t_cols);
                    const unsigned long long t_st = prof.stamp();
                    r_rows_from_rows(P[0], w.r, w.qtf, n, j0, lane);
                    rows_to_tile_V<NCH, 0>(P, alive, V0, lane);
                    wave_lds_fence();
                    if (lane < np0) w.r[row_off(n, j0 + lane)] = rdiag_mine;
                    prof.add(FP_LA_STORE, t_st);
                    const unsigned long long t_T = prof.stamp();
                    panel_T<NCH>(nch, tau, V0, T0, Gl, Tsave + 256 * pp, lane);
                    prof.add(FP_T, t_T);
                }
                prof.mark(FP_PANEL);
                __syncthreads();
                prof.mark(FP_PANEL_WAIT);
                if (two) {
                    if (wave == wb) {
                        f64x4 S[NCH];
                        const unsigned long long t_la2 = prof.stamp();
                        tile_to_strip<NCH>(S, V1, lane);
                        wave_lds_fence();
                        strip_apply<NCH>(S, nch, V0, T0, lane);
                        prof.add(FP_LA_APPLY, t_la2);
                        const unsigned long long t_cv = prof.stamp();
                        r_rows_out(S[0], w.r, w.qtf, n, j0, j1, g, m, false);
                        SOCP_SCHED_FENCE();
                        double P[Rows<NCH>::NQ][16];
                        strip_to_tile<NCH, 1>(S, V1, lane);
                        wave_lds_fence();
                        tile_to_rows<NCH, 1>(P, V1, lane);
                        wave_lds_fence();
                        const int np1 = (n - j1 < 16) ? n - j1 : 16;
                        prof.add(FP_LA_CONVERT, t_cv);
                        const unsigned long long t_cols = prof.stamp();
                        unsigned alive;
                        double rdiag_mine;
                        const double tau = panel_rows<NCH>(P, np1, Gl, rdiag + j1, lane, alive, rdiag_mine);
                        prof.add(FP_COLS, t_cols);
                        const unsigned long long t_st = prof.stamp();
                        r_rows_from_rows(P[0], w.r, w.qtf, n, j1, lane);
                        rows_to_tile_V<NCH, 1>(P, alive, V1, lane);
                        wave_lds_fence();
                        if (lane < np1) w.r[row_off(n, j1 + lane)] = rdiag_mine;
                        prof.add(FP_LA_STORE, t_st);
                        const unsigned long long t_T = prof.stamp();
                        panel_T<NCH>(nch, tau, V1, T1, Gl, Tsave + 256 * (pp + 1), lane);
                        prof.add(FP_T, t_T);
                    } else {
                        // panel pp's vectors back to A (qform reads them from there): the three wavefronts without a panel
                        tile_store<NCH>(V0, A, ld, n, j0, nch, j0, 0, (wave - wb - 1) & 3, 3, lane);
                    }
                    prof.mark(FP_PANEL);
                    __syncthreads();
                    prof.mark(FP_PANEL_WAIT);
                    tile_store<NCH>(V1, A, ld, n, j0, nch, j1, 1, wave, 4, lane);     // (its first chunk: rows of R, zeros in the tile)
                } else {
                    tile_store<NCH>(V0, A, ld, n, j0, nch, j0, 0, wave, 4, lane);
                }
            }
            // [E] the strips right of the panel(s); their first 16 (32) rows leave as rows of R and are not written back
            for (int c0 = j0 + (two ? 32 : 16) + 16 * wave; c0 <= n; c0 += 64) {
                f64x4 S[NCH];
                strip_load<NCH>(S, A, ld, n, j0, nch, c0, n + 1, g, m);
                if (pp == 0) {
                    const double nrm = strip_column_norm<NCH>(S);
                    if (g == 0 && c0 + m < n) acnorm[c0 + m] = nrm;
                }
                strip_apply<NCH>(S, nch, V0, T0, lane);
                r_rows_out(S[0], w.r, w.qtf, n, j0, c0, g, m, false);
                if (two) {
                    strip_apply<NCH>(S, nch, V1, T1, lane);
                    if (NCH > 1) r_rows_out(S[NCH > 1 ? 1 : 0], w.r, w.qtf, n, j1, c0, g, m, false);
                }
                strip_store<NCH>(S, A, ld, n, j0, nch, c0, n + 1, g, m, two ? 2 : 1);
            }
            prof.mark(FP_TRAIL);
            __syncthreads();
            prof.mark(FP_TRAIL_WAIT);
        }
        // ---- (Q^T fvec and the packed R have been written row block by row block as the strips passed) "singular":
        int zero = 0;
        for (int j = tid; j < n; j += 256) zero |= (rdiag[j] == 0) ? 1 : 0;
        const int sing = __syncthreads_or(zero);
        prof.mark(FP_RPACK);
        if (tid == 0) { states[p].sing = sing ? 1 : 0; states[p].pad = 1; }
        }   // PHASE == 1
        if constexpr (PHASE == 2) {
        // ---- qform (round 5): a strip of Q STAYS in a wavefront's registers while every panel that reaches it streams through LDS.
        // Q = H_0 ... H_last applied to the identity: the 16 columns c0 .. c0 + 15 start as identity columns and are touched by the
        // panels p <= c0 / 16 only (a later panel acts on rows below the columns' ones), last panel first.  So a strip is never READ
        // and is written ONCE (round 4: read and written once per pair of panels, 5 GB of the launch's 22); what is re-read instead are
        // the panels' vectors -- from A's lower triangle, where qrfac left them -- once per ROUND of four strips (one per wavefront),
        // double-buffered: the next panel's loads are in flight while the current one is applied.  Rounds go from the last strips to
        // the first: a strip's store overwrites the vectors of its own panel, which only the strips from it on need.
        {
            const int nstrips = npanels, nch_all = (n + 15) >> 4;
            const int t = tid & 15, r0 = tid >> 4;
            // (rounds are counted from the LAST strip: the short round, if any, is the one of the first strips, which few panels reach.
            // A wavefront holds TWO strips where its registers allow -- strips of <= 8 chunks, n <= 128 --: a panel then serves eight
            // strips per trip through LDS, and two applies stand behind the next panel's fetch instead of one)
            constexpr int SPW = NCH <= 8 ? 2 : 1, RS = 4 * SPW;
            for (int hi = nstrips; hi > 0; hi -= RS) {
                const int s0 = hi >= RS ? hi - RS : 0;
                int sw[SPW];
                bool have[SPW];
                f64x4 S[SPW][NCH];
                {
                    const int gi = here(g), mi = here(m);
#pragma unroll
                    for (int u = 0; u < SPW; u++) {
                        sw[u] = s0 + 4 * u + wave;                           // this wavefront's strip(s)
                        have[u] = sw[u] < hi;
#pragma unroll
                        for (int cc = 0; cc < NCH; cc++) {
#pragma unroll
                            for (int r = 0; r < 4; r++) S[u][cc][r] = (cc == sw[u] && gi + 4 * r == mi && 16 * sw[u] + mi < n) ? 1.0 : 0.0;
                        }
                    }
                }
                const int ptop = hi - 1;
                // panel q's vectors at ABSOLUTE rows (zero above its diagonal: the strips meet every panel at one offset) and T^T
                double xv[NCH], xt;
                auto fetch = [&](int q) {
                    const int jq = 16 * q, npq = (n - jq < 16) ? n - jq : 16;
#pragma unroll
                    for (int it = 0; it < NCH; it++) {
                        const int row = r0 + 16 * it;
                        xv[it] = (it < nch_all && row < n && row - jq >= t && t < npq) ? A[(long)row * ld + jq + t] : 0.0;
                    }
                    xt = Tsave[256 * q + tid];
                };
                auto deposit = [&](double *Vb, double *Tb) {
#pragma unroll
                    for (int it = 0; it < NCH; it++) Vb[(r0 + 16 * it) * kLdV + t] = xv[it];
                    Tb[(tid & 15) * 16 + (tid >> 4)] = xt;                   // X = T^T
                };
                fetch(ptop);
                deposit(lds, lds + 16 * NCH * kLdV);
                __syncthreads();
                prof.mark(FP_QLOAD);
                for (int q = ptop; q >= 0; q--) {
                    double *Vc = lds + ((ptop - q) & 1) * kPanelDoubles, *Tc = Vc + 16 * NCH * kLdV;
                    double *Vn = lds + ((ptop - q + 1) & 1) * kPanelDoubles, *Tn = Vn + 16 * NCH * kLdV;
                    if (q > 0) fetch(q - 1);
#pragma unroll
                    for (int u = 0; u < SPW; u++)
                        if (have[u] && q <= sw[u]) strip_apply<NCH>(S[u], nch_all, Vc, Tc, lane, q);
                    if (q > 0) deposit(Vn, Tn);
                    prof.mark(FP_QSTRIPS);
                    __syncthreads();
                    prof.mark(FP_QWAIT);
                }
#pragma unroll
                for (int u = 0; u < SPW; u++)
                    if (have[u]) strip_store<NCH>(S[u], A, ld, n, 0, nch_all, 16 * sw[u], n, g, m);
            }
        }
        }   // PHASE == 2
        __syncthreads();
    }
}

template <auto Kernel>
hipError_t raise_lds_limit_fast()
{
    static std::atomic<unsigned long long> raised{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && ((raised.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    // (the kernel also has a few hundred bytes of static LDS -- the workgroup-wide "or" -- so the dynamic part may not claim all 160 KB)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e == hipSuccess && dev < 64) raised.fetch_or(1ull << dev, std::memory_order_release);
    return e;
}

template <int NCH, int WPE, int PHASE>
hipError_t launch_phase(hipStream_t st, const PoolDev &pool, const int *d_list, int count)
{
    const size_t lds_bytes = sizeof(double) * (size_t)(2 * (16 * NCH * kLdV + 256) + 256);
    if (lds_bytes > 65536) {
        const hipError_t raised = raise_lds_limit_fast<factor_fast_kernel<NCH, WPE, PHASE>>();
        if (raised != hipSuccess) return raised;
    }
    hipLaunchKernelGGL((factor_fast_kernel<NCH, WPE, PHASE>), dim3((unsigned)count), dim3(256), lds_bytes, st, pool.cfg, pool.states, pool.ws, pool.ws_stride, d_list, count);
    return hipGetLastError();
}

// ---- the chain's launches.  Wavefronts per SIMD the registers are budgeted for: the panel launch holds as many problems per CU as tiles
// fit (38.9 KB at 16 chunks: 4 = one wavefront per SIMD, which may then use the whole register file; 30 KB at 12: 5; 21.5 KB at 8: 7;
// 17 KB at 6: 9; 13 KB at 4: 12); the trailing launch is round 5's [E]: two wavefronts per SIMD for the tall strips (181 / 224 registers), three for the
// 6- and 8-chunk ones (121 / 144 registers fit 168: 4096 x n = 127 1.61 -> 1.55 ms, n = 96 0.965 -> 0.93; profiles/r06_factor_trail_wpe_ab.txt), four for 4 chunks.
#ifndef SOCP_FACTOR_PANEL_WPE16
#define SOCP_FACTOR_PANEL_WPE16 1
#endif
#ifndef SOCP_FACTOR_PANEL_WPE12
#define SOCP_FACTOR_PANEL_WPE12 2
#endif
#ifndef SOCP_FACTOR_PANEL_WPE8
#define SOCP_FACTOR_PANEL_WPE8 2
#endif
#ifndef SOCP_FACTOR_PANEL_WPE6
#define SOCP_FACTOR_PANEL_WPE6 2
#endif
#ifndef SOCP_FACTOR_PANEL_WPE4
#define SOCP_FACTOR_PANEL_WPE4 2
#endif
#ifndef SOCP_FACTOR_CHAIN_MIN_DEFAULT
#define SOCP_FACTOR_CHAIN_MIN_DEFAULT 640
#endif
template <int NCH> struct ChainBudget;
template <> struct ChainBudget<16> { static constexpr int panel = SOCP_FACTOR_PANEL_WPE16, trail = 2; };
#ifndef SOCP_FACTOR_TRAIL_WPE12
#define SOCP_FACTOR_TRAIL_WPE12 2
#endif
#ifndef SOCP_FACTOR_TRAIL_WPE8
#define SOCP_FACTOR_TRAIL_WPE8 3
#endif
#ifndef SOCP_FACTOR_TRAIL_WPE6
#define SOCP_FACTOR_TRAIL_WPE6 3
#endif
template <> struct ChainBudget<12> { static constexpr int panel = SOCP_FACTOR_PANEL_WPE12, trail = SOCP_FACTOR_TRAIL_WPE12; };
template <> struct ChainBudget<8> { static constexpr int panel = SOCP_FACTOR_PANEL_WPE8, trail = SOCP_FACTOR_TRAIL_WPE8; };
template <> struct ChainBudget<6> { static constexpr int panel = SOCP_FACTOR_PANEL_WPE6, trail = SOCP_FACTOR_TRAIL_WPE6; };
template <> struct ChainBudget<4> { static constexpr int panel = SOCP_FACTOR_PANEL_WPE4, trail = 4; };

template <int NCH>
hipError_t launch_panel(hipStream_t st, const PoolDev &pool, const int *d_list, int count, int pp, bool final_launch)
{
    const size_t lds_bytes = sizeof(double) * (size_t)(16 * NCH * kLdV + 512);
    hipLaunchKernelGGL((qrfac_panel_kernel<NCH, ChainBudget<NCH>::panel>), dim3((unsigned)count), dim3(64), lds_bytes, st, pool.cfg, pool.states, pool.ws,
                       pool.ws_stride, d_list, count, pp, final_launch ? 1 : 0);
    return hipGetLastError();
}
template <int NCH>
hipError_t launch_trail(hipStream_t st, const PoolDev &pool, const int *d_list, int count, int pp, bool final_launch)
{
    const size_t lds_bytes = sizeof(double) * (size_t)(2 * (16 * NCH * kLdV + 256));
    if (lds_bytes > 65536) {
        const hipError_t raised = raise_lds_limit_fast<qrfac_trail_kernel<NCH, ChainBudget<NCH>::trail>>();
        if (raised != hipSuccess) return raised;
    }
    hipLaunchKernelGGL((qrfac_trail_kernel<NCH, ChainBudget<NCH>::trail>), dim3((unsigned)count), dim3(256), lds_bytes, st, pool.cfg, pool.states, pool.ws,
                       pool.ws_stride, d_list, count, pp, final_launch ? 1 : 0);
    return hipGetLastError();
}
// one launch of the chain at the strip height pair pp needs (chunks from panel pp's first row down to the matrix's last row)
hipError_t launch_chain_step(bool trail, hipStream_t st, const PoolDev &pool, const int *d_list, int count, int pp, bool final_launch)
{
    const int nch = (pool.cfg.n - 16 * pp + 15) >> 4;
#define SOCP_CHAIN_STEP(NCH) (trail ? launch_trail<NCH>(st, pool, d_list, count, pp, final_launch) : launch_panel<NCH>(st, pool, d_list, count, pp, final_launch))
    if (nch <= 4) return SOCP_CHAIN_STEP(4);
    if (nch <= 6) return SOCP_CHAIN_STEP(6);
    if (nch <= 8) return SOCP_CHAIN_STEP(8);
    if (nch <= 12) return SOCP_CHAIN_STEP(12);
    return SOCP_CHAIN_STEP(16);
#undef SOCP_CHAIN_STEP
}

struct PairPlan {
    bool has_trail, last_pair;
};
PairPlan plan_of(int n, int pp)
{
    const int npanels = (n + 15) >> 4, j0 = 16 * pp;
    const bool two = pp + 1 < npanels;
    return {j0 + (two ? 32 : 16) <= n, pp + 2 >= npanels};
}

hipError_t launch_qrfac_chain(hipStream_t st, const PoolDev &pool, const int *d_list, int count)
{
    const int n = pool.cfg.n, npanels = (n + 15) >> 4;
    for (int pp = 0; pp < npanels; pp += 2) {
        const PairPlan pl = plan_of(n, pp);
        hipError_t e = launch_chain_step(false, st, pool, d_list, count, pp, pl.last_pair && !pl.has_trail);
        if (e == hipSuccess && pl.has_trail) e = launch_chain_step(true, st, pool, d_list, count, pp, pl.last_pair);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Which form of qrfac a launch takes.  The chain pays where the launch holds more problems than the single kernel keeps resident (two
// four-wavefront workgroups per CU = 512 problems): there the single kernel runs its problems in rounds, each round's panel phases with one
// wavefront of four working, and the chain's panel launch -- every resident wavefront working -- is what shortens the refresh (4096 x n = 85:
// 1.24 -> 0.80 ms).  A launch of FEW problems is one problem's latency either way, the same serial panels, and the chain only adds its launches'
// start-up (the sweeps' later refreshes are such launches: M = 9, n = 127, 283 rounds, +2.7 % with the chain everywhere).  Measured crossover
// (profiles/r06_factor_chain_crossover.txt): 512 problems and fewer -- one residency -- the single launch is 10-20 % faster, from 768 up the
// chain is (n = 85: 0.24 against 0.30 ms at 768, 0.80 against 1.23 at 4096; n = 253: even from 768 on, 1-3 % from 2048 on).
// SOCP_FACTOR_CHAIN_MIN=<problems>: the chain from that many problems up (0: always, a huge number: never).
int chain_min_problems()
{
    static const int v = [] {
        const char *e = std::getenv("SOCP_FACTOR_CHAIN_MIN");
        return e ? std::atoi(e) : SOCP_FACTOR_CHAIN_MIN_DEFAULT;
    }();
    return v;
}

template <int NCH, int WPE>
hipError_t launch_nch(hipStream_t st, const PoolDev &pool, const int *d_list, int count)
{
    // (Tried on top and not kept: the chain for two halves of the launch's problems on two streams, a half's panel launch -- vector ALUs,
    // latency -- beside the other half's trailing launch -- memory, matrix cores --, with the trailing launches alternating through events or
    // the two chains running free: 4.62-4.68 resp. 4.34-4.36 ms against 4.33-4.36 at 2048 x n = 253, +3 ... +7 % at n = 85 / 127
    // (profiles/r06_factor_halves_ab.txt): what the two kinds of launch do not share in execution units they share in LDS and registers.)
    const hipError_t e = count >= chain_min_problems() ? launch_qrfac_chain(st, pool, d_list, count) : launch_phase<NCH, WPE, 1>(st, pool, d_list, count);
    return e != hipSuccess ? e : launch_phase<NCH, WPE, 2>(st, pool, d_list, count);
}

}  // namespace

hipError_t read_factor_profile(unsigned long long out[16], bool reset)
{
    for (int k = 0; k < 16; k++) out[k] = 0;
#ifdef SOCP_FACTOR_PROFILE
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fprof), sizeof(unsigned long long) * 16);
    if (e != hipSuccess) return e;
    if (reset) {
        const unsigned long long zero[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_fprof), zero, sizeof(zero));
    }
    return e;
#else
    (void)reset;
    return hipSuccess;
#endif
}

// n <= 256: a panel's strip (n rows x 16 columns) lives in the registers of one wavefront; the T factors of the panels
// (256 doubles each) are kept in the workspace's spare n (n + 1) / 2 doubles until qform, which they fit from n = 39 on
// (n = 33 .. 38 excepted) -- below that the order-preserving kernel is fast anyway.
// (n = 32 would pass the storage test as well; it is refused so that the rule is the documented range, every size of which is
// under test -- tests/test_gpu_factor_fast.py)
bool fast_factor_applies(int n) { return n >= 39 && n <= 256 && (long)n * (n + 1) / 2 >= 256L * ((n + 15) / 16); }

hipError_t launch_factor_fast(hipStream_t st, const PoolDev &pool, const int *d_list, int count)
{
    if (count <= 0) return hipSuccess;
    const int n = pool.cfg.n;
    if (!fast_factor_applies(n) || pool.cfg.ld < n + 1) return hipErrorInvalidValue;
    // the strip's register count follows the problem size: 16 rows per chunk
    // (wavefronts per SIMD the registers are budgeted for: profiles/r05_factor_fast_wpe_ab.txt -- the 4-chunk strip is 13 % faster with
    // four whatever it spills; 6 and 8 chunks run no slower with two, and spill 46 / 52 registers instead of 314 / 231)
#ifndef SOCP_FACTOR_WPE_SMALL
#define SOCP_FACTOR_WPE_SMALL 4
#endif
#ifndef SOCP_FACTOR_WPE_MID
#define SOCP_FACTOR_WPE_MID 2
#endif
    if (n <= 64) return launch_nch<4, SOCP_FACTOR_WPE_SMALL>(st, pool, d_list, count);
    if (n <= 96) return launch_nch<6, SOCP_FACTOR_WPE_MID>(st, pool, d_list, count);
    if (n <= 128) return launch_nch<8, SOCP_FACTOR_WPE_MID>(st, pool, d_list, count);
    if (n <= 192) return launch_nch<12, 2>(st, pool, d_list, count);
    return launch_nch<16, 2>(st, pool, d_list, count);
}

}  // namespace devsolver
}  // namespace socp
