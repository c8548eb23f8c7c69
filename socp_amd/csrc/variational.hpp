// variational.hpp -- the hybrj path on the device: integration of the augmented
// state [X ; dX/dX0] with ONE WAVEFRONT PER TRAJECTORY and assembly of the analytic shooting Jacobian.
//
// Augmented state layout (shooting.cpp:1003-1005): L = (s+1)*s doubles, X[0..s) the
// state, X[s*(k+1)+i] = dX_k/dX0_i.  For the double integrator s = 12, L = 156: lane l of the wave
// owns elements l, l+64, l+128.  Every RK4 stage publishes the stage vector to LDS, and each lane
// then reads what its elements need from there: the sensitivity rows of OTHER state components
// (d/dt R = (df/dX) R couples rows), and the three costates the control law reads.  Operation
// order per element is the reference's (odeTools.cpp:89-98; doubleIntegrator.cpp:113-213), the
// translation unit is compiled -ffp-contract=off; the path contains no exp, so results are
// bit-identical to the x86 path.  Under the adaptive integrator (SOCP_INT_DOPRI5) the same wavefront-per-trajectory form
// runs Dormand-Prince 5(4) with per-WAVE step control: traj_var_wave_dopri5_kernel.
#pragma once
#include "dev_common.hpp"

namespace socp {

// K_var: B augmented trajectories, one wave each.  X0, Xf: [B][L].  pp_params (may be null): per-problem parameter blocks
// [B / M][pp_stride] (dev_common.hpp), trajectory b belongs to problem b / M.
template <class Mdl>
__global__ __launch_bounds__(64) void traj_var_wave_kernel(ModelParams P, const double *__restrict__ t0,
                                                           const double *__restrict__ tf,
                                                           const double *__restrict__ X0,
                                                           double *__restrict__ Xf,
                                                           const double *__restrict__ pp_params, int pp_stride, int M)
{
    constexpr int L = (Mdl::S + 1) * Mdl::S;
    constexpr int K = (L + 63) / 64;
    __shared__ double Y[L];
    const int lane = threadIdx.x;
    const long b = blockIdx.x;
    if (pp_params) {
        const double *src = pp_params + (b / M) * pp_stride;
#pragma unroll
        for (int k = 0; k < kMaxParams; k++)
            if (k < pp_stride - 2) P.p[k] = src[k];
    }
    double X[K], F1[K], Fs[K], F[K], V[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int e = lane + 64 * k;
        X[k] = e < L ? X0[b * L + e] : 0.0;
    }
    // publish a stage vector, then evaluate this lane's elements of the variational RHS
    auto stage = [&](double ts, const double (&v)[K], double (&out)[K]) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) { const int e = lane + 64 * k; if (e < L) Y[e] = v[k]; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) { const int e = lane + 64 * k; out[k] = e < L ? Mdl::aug_rhs(P, ts, e, Y) : 0.0; }
    };
    const double ta = t0[b], tb = tf[b];
    const double dt = (tb - ta) / P.step_nbr;
    double t = ta;
    int guard = P.step_nbr + 8;                         // see Lane::integrate: never reached unless t += dt stalls
    while (t < (tb - dt / 2) && guard-- > 0) {          // wave-uniform: same t in every lane
        const double step = (t + dt > tb) ? (tb - t) : dt;
        const double h2 = step / 2.0;
        stage(t, X, F1);
#pragma unroll
        for (int k = 0; k < K; k++) V[k] = X[k] + h2 * F1[k];
        stage(t + step / 2.0, V, Fs);
#pragma unroll
        for (int k = 0; k < K; k++) V[k] = X[k] + h2 * Fs[k];
        stage(t + step / 2.0, V, F);
#pragma unroll
        for (int k = 0; k < K; k++) { V[k] = X[k] + step * F[k]; Fs[k] = Fs[k] + F[k]; }
        stage(t + step, V, F);
        const double h6 = step / 6.0;
#pragma unroll
        for (int k = 0; k < K; k++) X[k] = X[k] + h6 * (F1[k] + (F[k] + 2.0 * Fs[k]));
        t += dt;
    }
#pragma unroll
    for (int k = 0; k < K; k++) { const int e = lane + 64 * k; if (e < L) Xf[b * L + e] = X[k]; }
}

// The same trajectories under the ADAPTIVE integrator -- what the reference runs for every integrate() call when built with
// -D_USE_BOOST, the isJac = 1 ones of the hybrj path included (odeTools.cpp:129-134; default ModelInt model.hpp:395-414; caller
// shooting.cpp:996-1130): Dormand-Prince 5(4) on the whole augmented state, the error norm max_i |e_i| / (tol + tol (|x_i| + h |k1_i|))
// over all (s + 1) s entries.  One wavefront per trajectory as above, step control PER WAVE: every lane computes its elements'
// share of the norm, a wave-wide maximum makes it uniform, and the whole wave accepts, rejects and resizes together.  The
// controller, the stage sums and the step budget are those of Lane::dopri5_try / integrate_dopri5 (integrator.hpp) element for
// element -- a restatement of Boost.Odeint's published algorithm, [ext] PARITY UNPINNED.
template <class Mdl>
__global__ __launch_bounds__(64) void traj_var_wave_dopri5_kernel(ModelParams P, const double *__restrict__ t0,
                                                                  const double *__restrict__ tf,
                                                                  const double *__restrict__ X0,
                                                                  double *__restrict__ Xf,
                                                                  const double *__restrict__ pp_params, int pp_stride, int M)
{
    constexpr int L = (Mdl::S + 1) * Mdl::S;
    constexpr int K = (L + 63) / 64;
    constexpr int kBudget = 50000;                     // trial steps per segment (integrator.hpp: kAdaptiveStepBudget)
    __shared__ double Y[L];
    const int lane = threadIdx.x;
    const long b = blockIdx.x;
    if (pp_params) {
        const double *src = pp_params + (b / M) * pp_stride;
#pragma unroll
        for (int k = 0; k < kMaxParams; k++)
            if (k < pp_stride - 2) P.p[k] = src[k];
    }
    double X[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int e = lane + 64 * k;
        X[k] = e < L ? X0[b * L + e] : 0.0;
    }
    auto stage = [&](double ts, const double (&v)[K], double (&out)[K]) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) { const int e = lane + 64 * k; if (e < L) Y[e] = v[k]; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) { const int e = lane + 64 * k; out[k] = e < L ? Mdl::aug_rhs(P, ts, e, Y) : 0.0; }
    };
    // maximum over the wave, a NaN winning (the reference's loop: if (e > err || e != e) err = e)
    auto wave_max = [](double v) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(v, off);
            if (o > v || o != o) v = o;
        }
        return v;
    };
    constexpr double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
    constexpr double b21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40, b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9,
                     b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729,
                     b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176, b65 = -5103.0 / 18656,
                     c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
    constexpr double dc1 = 35.0 / 384 - 5179.0 / 57600, dc3 = 500.0 / 1113 - 7571.0 / 16695, dc4 = 125.0 / 192 - 393.0 / 640,
                     dc5 = -2187.0 / 6784 - -92097.0 / 339200, dc6 = 11.0 / 84 - 187.0 / 2100, dc7 = -1.0 / 40;
    const double eps = 2.220446049250313e-16;
    const double ta = t0[b], tb = tf[b];
    double t = ta, h = (tb - ta) / P.step_nbr;
    bool poison = false;
    if (h > 0) {                                        // zero-length / backward segment: no step
        double k1[K], k2[K], k3[K], k4[K], k5[K], k6[K], kn[K], xn[K], y[K];
        bool have_k1 = false;
        int budget = kBudget;                           // every wave reaches an exit (running out: NaN, like odeint's step_adjustment_error)
        while (tb - t > eps && budget > 0 && !poison) {
            while (t + h - tb <= eps && budget > 0) {
                if (!have_k1) { stage(t, X, k1); have_k1 = true; }
                int tries = 0;
                bool ok = false;
                do {
                    const double hh = h, tt = t;
#pragma unroll
                    for (int k = 0; k < K; k++) y[k] = 1.0 * X[k] + hh * b21 * k1[k];
                    stage(tt + hh * a2, y, k2);
#pragma unroll
                    for (int k = 0; k < K; k++) y[k] = 1.0 * X[k] + hh * b31 * k1[k] + hh * b32 * k2[k];
                    stage(tt + hh * a3, y, k3);
#pragma unroll
                    for (int k = 0; k < K; k++) y[k] = 1.0 * X[k] + hh * b41 * k1[k] + hh * b42 * k2[k] + hh * b43 * k3[k];
                    stage(tt + hh * a4, y, k4);
#pragma unroll
                    for (int k = 0; k < K; k++) y[k] = 1.0 * X[k] + hh * b51 * k1[k] + hh * b52 * k2[k] + hh * b53 * k3[k] + hh * b54 * k4[k];
                    stage(tt + hh * a5, y, k5);
#pragma unroll
                    for (int k = 0; k < K; k++)
                        y[k] = 1.0 * X[k] + hh * b61 * k1[k] + hh * b62 * k2[k] + hh * b63 * k3[k] + hh * b64 * k4[k] + hh * b65 * k5[k];
                    stage(tt + hh, y, k6);
#pragma unroll
                    for (int k = 0; k < K; k++)
                        xn[k] = 1.0 * X[k] + hh * c1 * k1[k] + hh * c3 * k3[k] + hh * c4 * k4[k] + hh * c5 * k5[k] + hh * c6 * k6[k];
                    stage(tt + hh, xn, kn);
                    double err = 0;
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        if (lane + 64 * k < L) {
                            double e = hh * dc1 * k1[k] + hh * dc3 * k3[k] + hh * dc4 * k4[k] + hh * dc5 * k5[k] + hh * dc6 * k6[k] + hh * dc7 * kn[k];
                            e = fabs(e) / (P.tol + P.tol * (1.0 * fabs(X[k]) + 1.0 * hh * fabs(k1[k])));
                            if (e > err || e != e) err = e;
                        }
                    }
                    err = wave_max(err);                // uniform from here on: the wave steps as one
                    budget--;
                    if (!(err <= 1.0)) {
                        double f = 0.9 * pow(err, -1.0 / 3.0);
                        if (!(f > 0.2)) f = 0.2;
                        h = hh * f;
                        ok = false;
                    } else {
                        t = tt + hh;
                        if (err < 0.5) {
                            const double floor5 = 1.0 / 3125.0;
                            const double e = err > floor5 ? err : floor5;
                            h = hh * (0.9 * pow(e, -1.0 / 5.0));
                        }
                        ok = true;
                    }
                } while (!ok && ++tries < 500);
                if (!ok) { poison = true; break; }
#pragma unroll
                for (int k = 0; k < K; k++) { X[k] = xn[k]; k1[k] = kn[k]; }
            }
            h = tb - t;
            have_k1 = false;
        }
        if (budget <= 0 && tb - t > eps) poison = true;
    }
#pragma unroll
    for (int k = 0; k < K; k++) { const int e = lane + 64 * k; if (e < L) Xf[b * L + e] = poison ? __builtin_nan("") : X[k]; }
}

// K_vprep: augmented initial states and segment bounds of one unknown vector z:
// Xaug[i] = [z_i ; I] (shooting.cpp:1003-1005,1060-1064), t0[i], tf[i] from the timeline.
// One block per (problem, segment): Zb[np][n] -> Xaug[np * M][L], t0 / tf[np * M]; per-problem boundary tables when set.
template <class Mdl>
__global__ void var_prepare_kernel(ProblemDev pb, const double *__restrict__ Zb, double *__restrict__ Xaug,
                                   double *__restrict__ t0, double *__restrict__ tf)
{
    constexpr int S = Mdl::S, L = (Mdl::S + 1) * Mdl::S;
    const long prob = blockIdx.x / pb.M;
    const int i = blockIdx.x - (int)prob * pb.M;
    const double *z = Zb + prob * pb.n;
    if (pb.pp_time) pb.time = pb.pp_time + prob * (pb.M + 1);
    for (int e = threadIdx.x; e < L; e += blockDim.x) {
        double v;
        if (e < S) v = z[S * i + e];
        else { const int k = (e - S) / S, c = (e - S) - k * S; v = (k == c) ? 1.0 : 0.0; }
        Xaug[(long)blockIdx.x * L + e] = v;
    }
    if (threadIdx.x == 0) { t0[blockIdx.x] = node_time(pb, z, i); tf[blockIdx.x] = node_time(pb, z, i + 1); }
}

// K_vasm: analytic shooting Jacobian from the integrated augmented states (shooting.cpp:996-1130 with
// the blocks of model.hpp:104-120,149-183,305-326 and shooting.cpp:1524-1555), one thread per segment,
// written column-major (the hand-over layout, shooting.cpp:889-893) into a zeroed n x n matrix.
// The reference's quirks are kept: only d/dt_end terms exist, and at a FREE interior time the copy loop
// of shooting.cpp:1070 also drops the time term one block to the right.
// One thread per (problem, segment); Zb[np][n], Xtf_all[np * M][L] -> Fjac[np][n * n].
template <class Mdl>
__global__ void var_assemble_kernel(ModelParams P, ProblemDev pb, int np, const double *__restrict__ Zb,
                                    const double *__restrict__ Xtf_all, double *__restrict__ Fjac)
{
    constexpr int S = Mdl::S, D = Mdl::D, L = (Mdl::S + 1) * Mdl::S;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int M = pb.M, n = pb.n;
    if (gid >= (long)np * M) return;
    const long prob = gid / M;
    const int i = (int)(gid - prob * M);
    load_problem_block(pb, prob, P, pb);                 // per-problem parameters / boundary tables when set (P, pb are by-value copies)
    const double *z = Zb + prob * n;
    double *fjac = Fjac + prob * (long)n * n;
    auto J = [&](int row, int col) -> double & { return fjac[row + (long)n * col]; };
    const double t1 = node_time(pb, z, i), t2 = node_time(pb, z, i + 1);
    const double *Xtf = Xtf_all + gid * L;
    const int index = S * (i + 1);
    auto ident = [](int k, int c) -> double { return k == c ? 1.0 : 0.0; };   // sensitivity block of [z ; I]

    if (i == 0) {
        const int *mx = pb.mode_x;
        const int col_t = pb.ft_row[0];
        double X1[S], fx[S], dH[S + 1];
#pragma unroll
        for (int k = 0; k < S; k++) X1[k] = z[k];
        for (int k = 0; k < D; k++) {
            const int src = (mx[k] == 1) ? (k + D) : k;                       // row of dX/dX0 copied (model.hpp:104-120)
            for (int j = 0; j < S; j++) J(k, j) = ident(src, j);
        }
        if (col_t >= 0) {                                                     // InitialHFunction, model.hpp:256-289
            Mdl::rhs(P, 0, 0, t1, X1, fx);
            Mdl::dhamiltonian(P, t1, X1, dH);
            for (int k = 0; k < D; k++) J(k, col_t) = (mx[k] == 1) ? fx[k + D] : fx[k];
            for (int c = 0; c < S; c++) {
                double acc = 0;
                for (int k = 0; k < S; k++) acc += dH[k] * ident(k, c);
                J(col_t, c) = acc;
            }
            double acc = 0;
            for (int k = 0; k < S; k++) acc += dH[k] * fx[k];
            J(col_t, col_t) = acc + dH[S];
        }
    }
    if (i < M - 1) {
        const int *mx = pb.mode_x + (i + 1) * D;
        const int col_t = pb.ft_row[i + 1];
        double Xs[S], Xp[S], fxt[S], fxp[S];
#pragma unroll
        for (int k = 0; k < S; k++) { Xs[k] = Xtf[k]; Xp[k] = z[index + k]; }
        Mdl::rhs(P, 0, 0, t2, Xs, fxt);
        Mdl::rhs(P, 0, 0, t2, Xp, fxp);
        // Rows of shooting::MultipleShootingFunction with isJac = 1 (shooting.cpp:1524-1555).  Each entry is ONE store through ONE
        // pointer with the value selected by the mode -- not a store per arm of a three-way divergent branch, the shape hipcc 7.2
        // mis-compiled in segment_residual (integrator.hpp; profiles/r05_fault_08f2e7a_isa.txt).
        //   FIXED      row j: dX(t2-)_j/dz_prev                 row j + D: e_j on the node's own block
        //   CONTINUOUS row j: dX_j/dz_prev, -e_j                row j + D: dX_{j+D}/dz_prev, -e_{j+D}
        //   FREE       the model's SwitchingStateFunction(..., isJac = 1): optional trait switching_state_jac, else zero rows
        //              (the default hook is a no-op on a block the caller has just zeroed, shooting.cpp:1067,1535-1538)
        for (int j = 0; j < D; j++) {
            const int mode = mx[j];
            const bool fixed = mode == 0, free_ = mode == 1;
            double aX[S], aP[S], bX[S], bP[S];                                // FREE only: partials of the two rows
            bool hook = false;
            if constexpr (has_switching_state_jac<Mdl>::value) {
                if (free_) {
#pragma unroll
                    for (int k = 0; k < S; k++) aX[k] = aP[k] = bX[k] = bP[k] = 0.0;
                    Mdl::switching_state_jac(P, t2, j, Xs, Xp, pb.xnode + (i + 1) * S, aX, aP, bX, bP);
                    hook = true;
                }
            }
            for (int c = 0; c < S; c++) {
                double ha = 0, hb = 0, hap = 0, hbp = 0;                      // the hook's rows chained through dX(t2-)/dz_prev
                if constexpr (has_switching_state_jac<Mdl>::value) {
                    if (hook) {
                        for (int k = 0; k < S; k++) { ha += aX[k] * Xtf[S * (k + 1) + c]; hb += bX[k] * Xtf[S * (k + 1) + c]; }
                        hap = aP[c]; hbp = bP[c];
                    }
                }
                const double sj = Xtf[S * (j + 1) + c], sjd = Xtf[S * (j + D + 1) + c];
                J(index + j, index - S + c) = free_ ? ha : sj;
                J(index + j, index + c) = fixed ? 0.0 : (free_ ? hap : -ident(j, c));
                J(index + j + D, index - S + c) = fixed ? 0.0 : (free_ ? hb : sjd);
                J(index + j + D, index + c) = fixed ? ident(j, c) : (free_ ? hbp : -ident(j + D, c));
            }
            if (col_t >= 0) {
                double ta = 0, tb = 0;                                        // the free-time column of the hook's rows
                if constexpr (has_switching_state_jac<Mdl>::value) {
                    if (hook)
                        for (int k = 0; k < S; k++) { ta += aX[k] * fxt[k] + aP[k] * fxp[k]; tb += bX[k] * fxt[k] + bP[k] * fxp[k]; }
                }
                J(index + j, col_t) = fixed ? fxt[j] : (free_ ? ta : fxt[j] - fxp[j]);
                J(index + j + D, col_t) = fixed ? fxp[j] : (free_ ? tb : fxt[j + D] - fxp[j + D]);
            }
        }
        if (col_t >= 0) {
            // shooting.cpp:1070: the copy loop runs one entry past the 4d block, so the time term also
            // lands at column index + 2d of the same rows (== col_t when that node is the last interior one)
            const int spill = index + S;
            if (spill < n && spill != col_t)
                for (int k = 0; k < S; k++) J(index + k, spill) = J(index + k, col_t);
            // model.hpp:305-326 SwitchingTimesFunction, isJac = 1
            double dH[S + 1], dHp[S + 1];
            Mdl::dhamiltonian(P, t2, Xs, dH);
            Mdl::dhamiltonian(P, t2, Xp, dHp);
            for (int c = 0; c < S; c++) {
                double a = 0, b = 0;
                for (int k = 0; k < S; k++) {
                    a += dH[k] * Xtf[S * (k + 1) + c];
                    b -= dHp[k] * ident(k, c);
                }
                J(col_t, index - S + c) = a;
                J(col_t, index + c) = b;
            }
            double acc = 0;
            for (int k = 0; k < S; k++) acc += dH[k] * fxt[k] - dHp[k] * fxp[k];
            J(col_t, col_t) = acc + (dH[S] - dHp[S]);
        }
    }
    if (i == M - 1) {
        const int *mx = pb.mode_x + M * D;
        const int col_t = pb.ft_row[M];
        for (int k = 0; k < D; k++) {
            const int src = (mx[k] == 1) ? (k + D + 1) : (k + 1);
            for (int j = 0; j < S; j++) J(D + k, S * i + j) = Xtf[S * src + j];
        }
        if (col_t >= 0) {                                                     // FinalHFunction, model.hpp:149-183
            double Xs[S], fx[S], dH[S + 1];
#pragma unroll
            for (int k = 0; k < S; k++) Xs[k] = Xtf[k];
            Mdl::rhs(P, 0, 0, t2, Xs, fx);
            Mdl::dhamiltonian(P, t2, Xs, dH);
            for (int k = 0; k < D; k++) J(D + k, col_t) = (mx[k] == 1) ? fx[k + D] : fx[k];
            for (int c = 0; c < S; c++) {
                double acc = 0;
                for (int k = 0; k < S; k++) acc += dH[k] * Xtf[S * (k + 1) + c];
                J(col_t, S * i + c) = acc;
            }
            double acc = 0;
            for (int k = 0; k < S; k++) acc += dH[k] * fx[k];
            J(col_t, col_t) = acc + dH[S];
        }
    }
}

// K_veval: Model(t, X, isJac = 1) on an augmented state and Hamiltonian(t, X, isJac = 1), one point
// per thread (host mirror's virtuals; not hot)
template <class Mdl>
__global__ void var_eval_kernel(ModelParams P, int what, int B, const double *__restrict__ tin, const double *__restrict__ Xin, int len,
                                double *__restrict__ out)
{
    constexpr int S = Mdl::S, L = (Mdl::S + 1) * Mdl::S;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *X = Xin + (long)b * len;
    const double t = tin ? tin[b] : 0.0;
    if (what == 0) {
        for (int e = 0; e < L; e++) out[(long)b * L + e] = Mdl::aug_rhs(P, t, e, X);
    } else {
        double dH[S + 1];
        Mdl::dhamiltonian(P, t, X, dH);
        for (int k = 0; k <= S; k++) out[(long)b * (S + 1) + k] = dH[k];
    }
}

// ---- launchers, one set per model with variational equations (the in-tree double integrator; plugins with the optional
// aug_rhs / dhamiltonian trait, plugin_impl.hpp) ----------------------------------------------------------------------------
namespace varimpl {

template <class Mdl>
hipError_t traj(hipStream_t st, const ModelParams &P, int B, const double *t0, const double *tf, const double *X0, double *Xf)
{
    if (B <= 0) return hipSuccess;
    if (P.integrator == 1) hipLaunchKernelGGL(traj_var_wave_dopri5_kernel<Mdl>, dim3(B), dim3(64), 0, st, P, t0, tf, X0, Xf, (const double *)nullptr, 0, 1);
    else hipLaunchKernelGGL(traj_var_wave_kernel<Mdl>, dim3(B), dim3(64), 0, st, P, t0, tf, X0, Xf, (const double *)nullptr, 0, 1);
    return hipGetLastError();
}

template <class Mdl>
hipError_t jacobian(hipStream_t st, const ModelParams &P, const ProblemDev &pb, int np, const double *z, double *Xaug, double *Xtf,
                    double *t0, double *tf, double *fjac)
{
    if (np <= 0) return hipSuccess;
    const unsigned B = (unsigned)((long)np * pb.M);                 // one wavefront per (problem, segment)
    hipLaunchKernelGGL(var_prepare_kernel<Mdl>, dim3(B), dim3(64), 0, st, pb, z, Xaug, t0, tf);
    if (P.integrator == 1) hipLaunchKernelGGL(traj_var_wave_dopri5_kernel<Mdl>, dim3(B), dim3(64), 0, st, P, t0, tf, Xaug, Xtf, pb.pp_params, pb.pp_stride, pb.M);
    else hipLaunchKernelGGL(traj_var_wave_kernel<Mdl>, dim3(B), dim3(64), 0, st, P, t0, tf, Xaug, Xtf, pb.pp_params, pb.pp_stride, pb.M);
    hipError_t e = hipMemsetAsync(fjac, 0, sizeof(double) * (size_t)np * pb.n * pb.n, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(var_assemble_kernel<Mdl>, dim3((B + 63) / 64), dim3(64), 0, st, P, pb, np, z, Xtf, fjac);
    return hipGetLastError();
}

template <class Mdl>
hipError_t eval(hipStream_t st, const ModelParams &P, int what, int B, const double *t, const double *X, int len, double *out)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(var_eval_kernel<Mdl>, dim3((B + 63) / 64), dim3(64), 0, st, P, what, B, t, X, len, out);
    return hipGetLastError();
}

}  // namespace varimpl

}  // namespace socp
