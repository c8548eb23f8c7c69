// integrator.hpp -- per-lane fixed-step RK4 driver and the three lane-per-trajectory kernels
// (trajectory batch, fused shooting residual, fused FD-Jacobian column batch).
//
// Included by kernels_exact.hip (compiled -ffp-contract=off with the reference-order models)
// and by kernels_fast.hip (contraction on, restructured models).  One lane owns one
// trajectory: its s-vector, the RK4 stage vectors and all RHS temporaries live in VGPRs, so
// the inner loop touches neither LDS nor HBM; a 64-lane wave is 64 independent trajectories.
#pragma once
#include <type_traits>

#include "dev_common.hpp"

namespace socp {

// Optional model traits (absent = false).  kCustomTraj: the model overrides model::ComputeTraj and brings its
// own driver  compute_traj<INTEG>(P, a0, a1, t0, tf, X, observer)  that may rewrite its two per-lane auxiliary
// scalars (models_interceptor.hpp).  kCustomFinal: it overrides Final[H]Function through final_row() /
// final_h_offset().
template <class M, class = void> struct has_custom_traj : std::false_type {};
template <class M> struct has_custom_traj<M, std::void_t<decltype(M::kCustomTraj)>> : std::bool_constant<M::kCustomTraj> {};
// optional trait switching_state(P, t, j, X, Xp, Xd, f_state, f_costate): the model's SwitchingStateFunction (model.hpp:339-341,
// called by MultipleShootingFunction for a FREE state component of an INTERIOR node, shooting.cpp:1535-1538) -- the two residual
// rows of component j.  Without it the rows are zero: the reference's default hook is a no-op and leaves what its (zero-
// initialised, then reused) scratch vector holds.
template <class M, class = void> struct has_switching_state : std::false_type {};
template <class M> struct has_switching_state<M, std::void_t<decltype(&M::switching_state)>> : std::true_type {};
// optional trait switching_state_jac(P, t, j, X, Xp, Xd, dfs_dX, dfs_dXp, dfc_dX, dfc_dXp): the same hook with isJac = 1 (hybrj path,
// shooting.cpp:1524-1538) -- the partial derivatives of the two rows of component j with respect to the state before (X) and after
// (Xp) the node, S entries each.  var_assemble_kernel chains them through the sensitivity block and forms the free-time column the
// way MultipleShootingFunction does for its own rows (d/dX . f(X) + d/dXp . f(Xp)).  Without it the rows of the analytic Jacobian
// are zero: the reference's default hook is a no-op on a block its caller has just zeroed.
template <class M, class = void> struct has_switching_state_jac : std::false_type {};
template <class M> struct has_switching_state_jac<M, std::void_t<decltype(&M::switching_state_jac)>> : std::true_type {};
template <class M, class = void> struct has_custom_final : std::false_type {};
template <class M> struct has_custom_final<M, std::void_t<decltype(M::kCustomFinal)>> : std::bool_constant<M::kCustomFinal> {};

constexpr int kAdaptiveStepBudget = 50000;

struct NoObserver {
    template <class... A> __device__ __forceinline__ void operator()(A &&...) const {}
};
struct NoStepHook {
    template <class X> __device__ __forceinline__ bool operator()(double, X &) const { return false; }
};

template <class Mdl>
struct Lane {
    static constexpr int S = Mdl::S;
    static constexpr int D = Mdl::D;

    // odeTools.cpp:89-98.  Stage states X + (step/2.0) F, stage times t + step/2.0 (twice) and
    // t + step, update X + (step/6.0)*(F1 + (F4 + 2.0*(F2 + F3))): this association order is
    // the parity contract.
    __device__ static __forceinline__ void rk4(const ModelParams &P, double sw0, double sw1,
                                              double t, double (&X)[S], double step)
    {
        if constexpr (Mdl::kRefOrder) {
            double F1[S], Fs[S], F[S], Y[S];
            const double h2 = step / 2.0;
            const double th = t + step / 2.0;
            Mdl::rhs(P, sw0, sw1, t, X, F1);
#pragma unroll
            for (int i = 0; i < S; i++) Y[i] = X[i] + h2 * F1[i];
            Mdl::rhs(P, sw0, sw1, th, Y, Fs);                 // F2
#pragma unroll
            for (int i = 0; i < S; i++) Y[i] = X[i] + h2 * Fs[i];
            Mdl::rhs(P, sw0, sw1, th, Y, F);                  // F3
#pragma unroll
            for (int i = 0; i < S; i++) { Y[i] = X[i] + step * F[i]; Fs[i] = Fs[i] + F[i]; }   // F2 + F3
            Mdl::rhs(P, sw0, sw1, t + step, Y, F);            // F4
            const double h6 = step / 6.0;
#pragma unroll
            for (int i = 0; i < S; i++) X[i] = X[i] + h6 * (F1[i] + (F[i] + 2.0 * Fs[i]));
        } else {
            // throughput flavour: running sum acc = F1 + 2 F2 + 2 F3 (+ F4) instead of keeping F1 and
            // F2+F3 alive -- 14 fewer live doubles, which is what lets three waves share a SIMD
            double A[S], F[S], Y[S];
            const double h2 = 0.5 * step;
            const double th = t + h2;
            Mdl::rhs(P, sw0, sw1, t, X, A);
#pragma unroll
            for (int i = 0; i < S; i++) Y[i] = X[i] + h2 * A[i];
            Mdl::rhs(P, sw0, sw1, th, Y, F);
#pragma unroll
            for (int i = 0; i < S; i++) { Y[i] = X[i] + h2 * F[i]; A[i] = A[i] + 2.0 * F[i]; }
            Mdl::rhs(P, sw0, sw1, th, Y, F);
#pragma unroll
            for (int i = 0; i < S; i++) { Y[i] = X[i] + step * F[i]; A[i] = A[i] + 2.0 * F[i]; }
            Mdl::rhs(P, sw0, sw1, t + step, Y, F);
            const double h6 = step * (1.0 / 6.0);
#pragma unroll
            for (int i = 0; i < S; i++) X[i] = X[i] + h6 * (A[i] + F[i]);
        }
    }

    // ---- adaptive Dormand-Prince 5(4), what the reference runs when built with -D_USE_BOOST:
    // integrate_adaptive(make_dense_output<runge_kutta_dopri5>(tol, tol), ode, X, t0, tf, dt) (odeTools.cpp:129-134).
    // [ext] Boost.Odeint is not vendored; restated from its published algorithm: FSAL stages
    // summed left to right, error max_i |e_i| / (tol + tol (|x_i| + dt |k1_i|)) on the OLD state, reject ->
    // dt *= max(0.9 err^-1/3, 0.2), accept with err < 0.5 -> dt *= 0.9 max(5^-5, err)^-1/5, stepping while
    // t + dt <= tf and finishing with dt = tf - t.  Step control is PER LANE: lanes of a wave take different
    // numbers of steps and wait for the slowest (the loop runs under the exec mask).
    __device__ static __forceinline__ bool dopri5_try(const ModelParams &P, const double &sw0, const double &sw1, double &t, double &dt,
                                                     const double (&x)[S], const double (&k1)[S], double (&xn)[S], double (&kn)[S])
    {
        constexpr double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
        constexpr double b21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40, b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9,
                         b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729,
                         b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176, b65 = -5103.0 / 18656,
                         c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
        constexpr double dc1 = 35.0 / 384 - 5179.0 / 57600, dc3 = 500.0 / 1113 - 7571.0 / 16695, dc4 = 125.0 / 192 - 393.0 / 640,
                         dc5 = -2187.0 / 6784 - -92097.0 / 339200, dc6 = 11.0 / 84 - 187.0 / 2100, dc7 = -1.0 / 40;
        double k2[S], k3[S], k4[S], k5[S], k6[S], y[S];
        const double h = dt, tt = t;
#pragma unroll
        for (int i = 0; i < S; i++) y[i] = 1.0 * x[i] + h * b21 * k1[i];
        Mdl::rhs(P, sw0, sw1, tt + h * a2, y, k2);
#pragma unroll
        for (int i = 0; i < S; i++) y[i] = 1.0 * x[i] + h * b31 * k1[i] + h * b32 * k2[i];
        Mdl::rhs(P, sw0, sw1, tt + h * a3, y, k3);
#pragma unroll
        for (int i = 0; i < S; i++) y[i] = 1.0 * x[i] + h * b41 * k1[i] + h * b42 * k2[i] + h * b43 * k3[i];
        Mdl::rhs(P, sw0, sw1, tt + h * a4, y, k4);
#pragma unroll
        for (int i = 0; i < S; i++) y[i] = 1.0 * x[i] + h * b51 * k1[i] + h * b52 * k2[i] + h * b53 * k3[i] + h * b54 * k4[i];
        Mdl::rhs(P, sw0, sw1, tt + h * a5, y, k5);
#pragma unroll
        for (int i = 0; i < S; i++) y[i] = 1.0 * x[i] + h * b61 * k1[i] + h * b62 * k2[i] + h * b63 * k3[i] + h * b64 * k4[i] + h * b65 * k5[i];
        Mdl::rhs(P, sw0, sw1, tt + h, y, k6);
#pragma unroll
        for (int i = 0; i < S; i++) xn[i] = 1.0 * x[i] + h * c1 * k1[i] + h * c3 * k3[i] + h * c4 * k4[i] + h * c5 * k5[i] + h * c6 * k6[i];
        Mdl::rhs(P, sw0, sw1, tt + h, xn, kn);
        double err = 0;
#pragma unroll
        for (int i = 0; i < S; i++) {
            double e = h * dc1 * k1[i] + h * dc3 * k3[i] + h * dc4 * k4[i] + h * dc5 * k5[i] + h * dc6 * k6[i] + h * dc7 * kn[i];
            e = fabs(e) / (P.tol + P.tol * (1.0 * fabs(x[i]) + 1.0 * h * fabs(k1[i])));
            if (e > err || e != e) err = e;
        }
        if (!(err <= 1.0)) {
            double f = 0.9 * pow(err, -1.0 / 3.0);
            if (!(f > 0.2)) f = 0.2;
            dt = h * f;
            return false;
        }
        t = tt + h;
        if (err < 0.5) {
            const double floor5 = 1.0 / 3125.0;                 // 5^-5
            const double e = err > floor5 ? err : floor5;
            dt = h * (0.9 * pow(e, -1.0 / 5.0));
        }
        return true;
    }

    // `hook(t, X)` runs before every step and reports whether it rewrote X (the interceptor's chart change): the
    // FSAL derivative is then recomputed.  sw0/sw1 are references because such a hook may also change them.
    // `after(t, X)` sees the state at the end of every ACCEPTED step: the observer of the reference's adaptive integrate()
    // (odeTools.cpp:103-123 with the Boost branch at :108 -- integrate_adaptive calls it at t0 and after each step).
    template <class Hook = NoStepHook, class After = NoObserver>
    __device__ static __forceinline__ void integrate_dopri5(const ModelParams &P, const double &sw0, const double &sw1,
                                                           double t0, double tf, double (&X)[S], Hook &&hook = Hook(), After &&after = After())
    {
        const double eps = 2.220446049250313e-16;
        double t = t0, h = (tf - t0) / P.step_nbr;
        if (!(h > 0)) return;                                   // zero-length / backward segment: no step
        double k1[S], xn[S], kn[S];
        bool have_k1 = false;
        // Every lane reaches an exit: at most kAdaptiveStepBudget trial steps per segment (odeint itself has no limit;
        // a trajectory that runs into a singularity of the dynamics would otherwise hold its whole wave for minutes).
        // Running out is reported like odeint's step_adjustment_error: the result is NaN.
        int budget = kAdaptiveStepBudget;
        while (tf - t > eps && budget > 0) {
            while (t + h - tf <= eps && budget > 0) {
                if (hook(t, X)) have_k1 = false;
                if (!have_k1) { Mdl::rhs(P, sw0, sw1, t, X, k1); have_k1 = true; }
                int tries = 0;
                bool ok;
                do {
                    ok = dopri5_try(P, sw0, sw1, t, h, X, k1, xn, kn);
                    budget--;
                } while (!ok && ++tries < 500);
                if (!ok) {                                       // odeint throws step_adjustment_error: poison the result
#pragma unroll
                    for (int i = 0; i < S; i++) X[i] = __builtin_nan("");
                    return;
                }
#pragma unroll
                for (int i = 0; i < S; i++) { X[i] = xn[i]; k1[i] = kn[i]; }
                after(t, X);
            }
            h = tf - t;
            have_k1 = false;
        }
        if (budget <= 0 && tf - t > eps) {
#pragma unroll
            for (int i = 0; i < S; i++) X[i] = __builtin_nan("");
        }
    }

    // model::ComputeTraj.  a0/a1 are the lane's auxiliary scalars; only a kCustomTraj model writes them.
    template <int INTEG>
    __device__ static __forceinline__ void integrate_with(const ModelParams &P, double &a0, double &a1,
                                                         double t0, double tf, double (&X)[S])
    {
        if constexpr (has_custom_traj<Mdl>::value) Mdl::template compute_traj<INTEG>(P, a0, a1, t0, tf, X, NoObserver());
        else if constexpr (INTEG == 1) integrate_dopri5(P, a0, a1, t0, tf, X);
        else integrate(P, a0, a1, t0, tf, X);
    }

    // model.hpp:395-414 / goddard.cpp:298-317 (dt) + odeTools.cpp:128-146 (loop): t is
    // accumulated by t += dt, the last step is clamped to tf - t, and a segment with
    // tf <= t0 + dt/2 (zero length or backward) takes no step at all.
    __device__ static __forceinline__ void integrate(const ModelParams &P, double sw0, double sw1,
                                                    double t0, double tf, double (&X)[S])
    {
        const double dt = (tf - t0) / P.step_nbr;
        double t = t0;
        // The loop takes stepNbr steps (one more or less when rounding moves the last comparison).  The
        // counter only stops the degenerate case the reference loops on forever -- dt below the spacing of
        // t, where t += dt no longer advances -- because a wave that never finishes hangs the device.
        int guard = P.step_nbr + 8;
        while (t < (tf - dt / 2) && guard-- > 0) {
            const double step = (t + dt > tf) ? (tf - t) : dt;
            rk4(P, sw0, sw1, t, X, step);
            t += dt;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// K_traj: B independent trajectories, X0[B][S] -> Xf[B][S].
// HBM side: the wave's 64 rows are one contiguous 64*S*8-byte span; it is read and written with
// consecutive lanes on consecutive doubles (coalesced) and transposed through LDS so that each
// lane ends up with its own row in registers.
// ---------------------------------------------------------------------------------------------
// WPE = cap on waves per SIMD (amdgpu_waves_per_eu max): the launcher picks ceil(waves / 1024) so that a
// grid smaller than the chip is spread one (or two) waves per SIMD instead of being packed three deep on
// a fraction of the SIMDs (launch_impl.hpp).
template <class Mdl, int WPE, int INTEG = 0, bool PERPROB = false>      // PERPROB unused here (one launch-macro shape for all hot kernels)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, WPE))) void traj_lane_kernel(ModelParams P, int B,
                                                       const double *__restrict__ t0,
                                                       const double *__restrict__ tf,
                                                       const double *__restrict__ sw,
                                                       const double *__restrict__ X0,
                                                       double *__restrict__ Xf)
{
    constexpr int S = Mdl::S;
    constexpr int LD = S + 1;                       // padded row: conflict-free lane-strided access
    __shared__ double tile[64 * LD];
    const int lane = threadIdx.x;
    const long row0 = (long)blockIdx.x * 64;
    const int rows = (B - row0) < 64 ? (int)(B - row0) : 64;
    const double *src = X0 + row0 * S;
    for (int idx = lane; idx < rows * S; idx += 64) tile[(idx / S) * LD + (idx % S)] = src[idx];
    __syncthreads();

    double X[S];
    const bool live = lane < rows;
    if (live) {
#pragma unroll
        for (int k = 0; k < S; k++) X[k] = tile[lane * LD + k];
        const long b = row0 + lane;
        double s0 = sw ? sw[2 * b] : P.sw0;
        double s1 = sw ? sw[2 * b + 1] : P.sw1;
        Lane<Mdl>::template integrate_with<INTEG>(P, s0, s1, t0[b], tf[b], X);
#pragma unroll
        for (int k = 0; k < S; k++) tile[lane * LD + k] = X[k];
    }
    __syncthreads();
    double *dst = Xf + row0 * S;
    for (int idx = lane; idx < rows * S; idx += 64) dst[idx] = tile[(idx / S) * LD + (idx % S)];
}

// K_dense: ONE trajectory with the state after every step kept (trace replay, shooting.cpp:496-544 ->
// the observer form of integrate, odeTools.cpp:103-123).  Row 0 is (t0, X0); row k the accumulated
// time t and state after k steps -- exactly what the reference's observer is shown.  Not hot.
// A kCustomTraj model reports the rows its own ComputeTraj traces, with the two auxiliary scalars of each row in
// aux[row][2] (may be null).
template <class Mdl, int INTEG = 0>
__global__ __launch_bounds__(64) void traj_dense_kernel(ModelParams P, double t0, double tf, double sw0, double sw1,
                                  const double *__restrict__ X0, double *__restrict__ dense,
                                  double *__restrict__ times, int cap, int *__restrict__ rows, double *__restrict__ aux)
{
    constexpr int S = Mdl::S;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double X[S];
    if constexpr (has_custom_traj<Mdl>::value) {
#pragma unroll
        for (int k = 0; k < S; k++) X[k] = X0[k];
        int r = 0;
        double a0 = sw0, a1 = sw1;
        Mdl::template compute_traj<INTEG>(P, a0, a1, t0, tf, X, [&](double t, const double (&Xr)[S], double b0, double b1) {
            if (r < cap) {
#pragma unroll
                for (int k = 0; k < S; k++) dense[(long)r * S + k] = Xr[k];
                times[r] = t;
                if (aux) { aux[2 * r] = b0; aux[2 * r + 1] = b1; }
            }
            r++;
        });
        // one more row: the state as ComputeTraj returns it (back in the model's default chart) and the flags it leaves
        if (r < cap) {
#pragma unroll
            for (int k = 0; k < S; k++) dense[(long)r * S + k] = X[k];
            times[r] = tf;
            if (aux) { aux[2 * r] = a0; aux[2 * r + 1] = a1; }
        }
        *rows = r + 1;
        return;
    }
#pragma unroll
    for (int k = 0; k < S; k++) { X[k] = X0[k]; dense[k] = X[k]; }
    times[0] = t0;
    if (aux) { aux[0] = sw0; aux[1] = sw1; }
    int r = 1;
    auto keep = [&](double t, const double (&Xr)[S]) {
        if (r < cap) {
#pragma unroll
            for (int k = 0; k < S; k++) dense[(long)r * S + k] = Xr[k];
            times[r] = t;
            if (aux) { aux[2 * r] = sw0; aux[2 * r + 1] = sw1; }
        }
        r++;
    };
    if constexpr (INTEG == 1) {
        // the adaptive integrator's own steps: row k = time and state at the end of the k-th accepted step
        Lane<Mdl>::integrate_dopri5(P, sw0, sw1, t0, tf, X, NoStepHook(), keep);
    } else {
        const double dt = (tf - t0) / P.step_nbr;
        double t = t0;
        int guard = P.step_nbr + 8;
        while (t < (tf - dt / 2) && guard-- > 0) {
            const double step = (t + dt > tf) ? (tf - t) : dt;
            Lane<Mdl>::rk4(P, sw0, sw1, t, X, step);
            t += dt;
            keep(t, X);
        }
    }
    *rows = r;
}

// ---------------------------------------------------------------------------------------------
// Shooting residual pieces shared by K_res and K_fdj.  One trajectory = (row, segment i).
// It owns these residual rows (shooting.cpp:945-990):
//   i == 0     : F[0..d)            initial rows (+ H row at s*M when t0 is FREE)
//   i <  M-1   : F[s(i+1) .. +s)    continuity at node i+1 (+ free-time row of node i+1)
//   i == M-1   : F[d..2d)           final rows (+ H row when tf is FREE)
// `Emit` receives (row index, value).
// ---------------------------------------------------------------------------------------------
template <class Mdl, int INTEG, class ZRead, class Emit>
__device__ __forceinline__ void segment_residual(const ModelParams &P, const ProblemDev &pb,
                                                 const ZRead &z, int i, Emit &&emit)
{
    constexpr int S = Mdl::S;
    constexpr int D = Mdl::D;
    const int M = pb.M;

    // timeline entries this segment needs (shooting.cpp:1579-1617)
    auto jt = [&](int j) -> double { const int kind = pb.node_kind[j]; return kind >= 0 ? z(kind) : pb.time[j]; };
    auto nt = [&](int k) -> double {
        const int kind = pb.node_kind[k];
        if (kind >= -1) return jt(k);
        const int a = pb.lo[k], b = pb.hi[k];
        const double ta = jt(a), tb = jt(b);
        return ta + (k - a) * (tb - ta) / (b - a);
    };
    const double t1 = nt(i), t2 = nt(i + 1);
    // model switching times = FREE node times with index < M, in node order (:1604,1615)
    // (a model with its own ComputeTraj keeps its two auxiliary scalars for itself: they are not switching times)
    double sw0 = P.sw0, sw1 = P.sw1;
    if constexpr (!has_custom_traj<Mdl>::value) {
        if (pb.sw_node0 >= 0) sw0 = nt(pb.sw_node0);
        if (pb.sw_node1 >= 0) sw1 = nt(pb.sw_node1);
    }

    double X[S];
    if (i == 0) {
        // model.hpp:196-228 InitialFunction / :239-290 InitialHFunction, isJac == 0.
        // Done BEFORE the integration so the node state is not live across it (register budget).
#pragma unroll
        for (int k = 0; k < S; k++) X[k] = z(k);
#pragma unroll
        for (int j = 0; j < D; j++) {
            const bool fr = pb.mode_x[j] == 1;
            emit(j, fr ? X[j + D] : X[j] - pb.xnode[j]);
        }
        if (pb.ft_row[0] >= 0) emit(pb.ft_row[0], Mdl::hamiltonian(P, sw0, sw1, t1, X));
    } else {
#pragma unroll
        for (int k = 0; k < S; k++) X[k] = z(S * i + k);
    }
    Lane<Mdl>::template integrate_with<INTEG>(P, sw0, sw1, t1, t2, X);     // shooting.cpp:943 Move(t1, X1, t2)

    if (i < M - 1) {
        double Xp[S];
#pragma unroll
        for (int k = 0; k < S; k++) Xp[k] = z(S * (i + 1) + k);
        // free interior time: model::SwitchingTimesFunction (shooting.cpp:964-968)
        if (pb.ft_row[i + 1] >= 0) emit(pb.ft_row[i + 1], Mdl::switching_fn(P, sw0, sw1, t2, X, Xp));
        // shooting::MultipleShootingFunction, isJac == 0 (shooting.cpp:1511-1576)
        const int *mx = pb.mode_x + (i + 1) * D;
        const double *xd = pb.xnode + (i + 1) * S;
#pragma unroll
        for (int j = 0; j < D; j++) {
            const int row = S * (i + 1) + j;
            // FIXED pins both sides; CONTINUOUS: state and costate jumps; FREE: the model's SwitchingStateFunction
            // (shooting.cpp:1535-1538).  Written as selects -- ONE store per row through ONE pointer, a branch only around the
            // optional hook.  As a three-way branch with an emit() pair in every arm (commit 08f2e7a) hipcc 7.2 mis-compiled the
            // example plugin's residual kernel: the backend sank the `emit(row + D, .)` store of component 1 to the join and left
            // its ADDRESS register undefined on the all-CONTINUOUS path (LLVM IR correct, ISA not: profiles/r05_fault_08f2e7a_isa.txt).
            // An all-CONTINUOUS wave then stored through whatever the register held last -- component 0's X[0] - Xp[0], i.e.
            // address 0 for a continuous iterate (the fault seen), an arbitrary address for a discontinuous one (no fault, a
            // stray write).  scripts/isa_store_audit.py looks for that shape in every kernel's ISA; tests/test_gpu_plugin.py
            // compares whole guarded output buffers on a discontinuous iterate.
            const int mode = mx[j];
            const double xdj = xd[j];
            double fs = 0.0, fc = 0.0;
            if constexpr (has_switching_state<Mdl>::value) {
                if (mode == 1) Mdl::switching_state(P, t2, j, X, Xp, xd, fs, fc);
            }
            const double a = mode == 0 ? X[j] - xdj : (mode == 1 ? fs : X[j] - Xp[j]);
            const double b = mode == 0 ? Xp[j] - xdj : (mode == 1 ? fc : X[j + D] - Xp[j + D]);
            emit(row, a);
            emit(row + D, b);
        }
    }
    if (i == M - 1) {
        // model.hpp:90-122 FinalFunction / :133-185 FinalHFunction, isJac == 0
        const int *mx = pb.mode_x + M * D;
        const double *xd = pb.xnode + M * S;
        if constexpr (has_custom_final<Mdl>::value) {
#pragma unroll
            for (int j = 0; j < D; j++) emit(D + j, Mdl::final_row(P, j, mx[j], X, xd));
            if (pb.ft_row[M] >= 0) emit(pb.ft_row[M], Mdl::hamiltonian(P, sw0, sw1, t2, X) + Mdl::final_h_offset(P));
        } else {
#pragma unroll
            for (int j = 0; j < D; j++) {
                const bool fr = mx[j] == 1;
                emit(D + j, fr ? X[j + D] : X[j] - xd[j]);
            }
            if (pb.ft_row[M] >= 0) emit(pb.ft_row[M], Mdl::hamiltonian(P, sw0, sw1, t2, X));
        }
    }
}

// Row-owned tiles.  A workgroup of the residual kernels owns R = min(64 / M, ...) WHOLE residual rows: lane
// l < R*M integrates segment l % M of local row l / M and drops the residual entries it owns into an LDS
// tile [R][n]; after the barrier the tile -- one contiguous R*n*8-byte span of the output -- is stored with
// consecutive lanes on consecutive doubles, i.e. as full cache lines (direct per-lane stores are 8 bytes at
// a stride of n*8 and cost ~1.7x the algorithmic write traffic, profiles/r01).  rows_per_block = 0 selects
// the direct form (problems whose row tile would not fit the LDS budget, or M > 64).
__device__ __forceinline__ void store_tile(const double *tile, double *dst, long count)
{
    for (long idx = threadIdx.x; idx < count; idx += 64) dst[idx] = tile[idx];
}

// K_res: Z[B][n] -> F[B][n]; one lane = (row, segment).
// PERPROB: row b carries its own parameter / boundary blocks (dev_common.hpp load_problem_block) -- a separate
// instantiation, so the shared-parameter kernels keep their parameters in scalar registers.
template <class Mdl, int WPE, int INTEG = 0, bool PERPROB = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, WPE))) void residual_lane_kernel(ModelParams P, ProblemDev pb, int B,
                                                              const double *__restrict__ Z,
                                                              double *__restrict__ F, int rows_per_block)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];
    const int M = pb.M, n = pb.n;
    // one code path for both output forms: `out` is the lane's row either in the LDS tile or in F itself
    long b;
    int i;
    double *out;
    bool live;
    const long row0 = (long)blockIdx.x * rows_per_block;
    const int rows = rows_per_block ? ((B - row0) < rows_per_block ? (int)(B - row0) : rows_per_block) : 0;
    if (rows_per_block == 0) {
        const long T = (long)blockIdx.x * 64 + threadIdx.x;
        live = T < (long)B * M;
        b = T / M;
        i = (int)(T - b * M);
        out = F + b * n;
    } else {
        const int lane = threadIdx.x, lr = lane / M;
        live = lane < rows * M;
        b = row0 + lr;
        i = lane - lr * M;
        out = tile + (long)lr * n;
    }
    if (live) {
        const double *zr = Z + b * n;
        auto z = [=](int k) -> double { return zr[k]; };
        if constexpr (PERPROB) {
            ModelParams Pq = P;
            ProblemDev pq = pb;
            load_problem_block(pb, b, Pq, pq);
            segment_residual<Mdl, INTEG>(Pq, pq, z, i, [=](int row, double v) { out[row] = v; });
        } else {
            segment_residual<Mdl, INTEG>(P, pb, z, i, [=](int row, double v) { out[row] = v; });
        }
    }
    if (rows_per_block == 0) return;                 // uniform over the workgroup
    __syncthreads();
    store_tile(tile, F + row0 * n, (long)rows * n);
}

// MINPACK fdjac1 step: h = eps*|z_j|, or eps when that is zero
__device__ __forceinline__ double fd_step(double zj, double eps)
{
    const double h = eps * fabs(zj);
    return h == 0 ? eps : h;
}

// K_fdj: fdjac1 columns of `np` problems as one batch.  pairs[t] = (column j, segment i) to
// integrate; the unknown vector of column j is z with z_j + h_j generated on the fly, so the
// perturbation matrix never exists in HBM: reads are z[n] and fvec[n] per problem (cache
// resident), writes are the Jacobian entries fjac[row + n*j] = (F_j[row] - fvec[row]) / h_j.
template <class Mdl, int WPE, int INTEG = 0, bool PERPROB = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, WPE))) void fdjac_lane_kernel(ModelParams P, ProblemDev pb, int np, int T,
                                                           const int2 *__restrict__ pairs,
                                                           const double *__restrict__ Zb,
                                                           const double *__restrict__ Fvec,
                                                           double eps, double *__restrict__ Fjac)
{
    const long tid = (long)blockIdx.x * 64 + threadIdx.x;
    if (tid >= (long)np * T) return;
    const long prob = tid / T;
    const int2 pr = pairs[tid - prob * T];
    const int j = pr.x, i = pr.y;
    const double *zb = Zb + prob * pb.n;
    const double *fvec = Fvec + prob * pb.n;
    const double h = fd_step(zb[j], eps);
    // z_k (+ h when k is the perturbed column).  Written as an ADD of h or -0.0 (x + -0.0 == x bit for bit, -0.0 included):
    // a select between the loaded value and a precomputed z_j + h makes the compiler select between two ADDRESSES
    // (the global one and a stack copy of z_j + h) and load through a flat pointer -- 16 bytes of scratch per lane.
    auto z = [=](int k) -> double { return zb[k] + (k == j ? h : -0.0); };
    double *col = Fjac + prob * (long)pb.n * pb.n + (long)pb.n * j;
    if constexpr (PERPROB) {
        ModelParams Pq = P;
        ProblemDev pq = pb;
        load_problem_block(pb, prob, Pq, pq);
        segment_residual<Mdl, INTEG>(Pq, pq, z, i, [=](int row, double v) { col[row] = (v - fvec[row]) / h; });
    } else {
        segment_residual<Mdl, INTEG>(P, pb, z, i, [=](int row, double v) { col[row] = (v - fvec[row]) / h; });
    }
}

// K_fdr: the (n+1) residual rows of a forward-difference Jacobian -- row 0 at z, row j+1 at
// z + h_j e_j -- for `np` problems in ONE launch (no dependency between base and perturbed
// trajectories).  Rows[np][n+1][n]; the perturbation matrix is generated on the fly; output through
// row-owned LDS tiles (see above) so that the residual rows are written as full lines.
template <class Mdl, int WPE, int INTEG = 0, bool PERPROB = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, WPE))) void fdrows_lane_kernel(ModelParams P, ProblemDev pb, int np,
                                                            const double *__restrict__ Zb, double eps,
                                                            double *__restrict__ Rows, int rows_per_block)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];
    const int M = pb.M, n = pb.n;
    const long total_rows = (long)np * (n + 1);          // virtual residual rows: (problem, FD row)
    // one code path for both output forms: `out` is the lane's row either in the LDS tile or in Rows itself
    long vrow;
    int i;
    double *out;
    bool live;
    const long row0 = (long)blockIdx.x * rows_per_block;
    const int rows = rows_per_block ? ((total_rows - row0) < rows_per_block ? (int)(total_rows - row0) : rows_per_block) : 0;
    if (rows_per_block == 0) {
        const long tid = (long)blockIdx.x * 64 + threadIdx.x;
        live = tid < total_rows * M;
        vrow = tid / M;
        i = (int)(tid - vrow * M);
        out = Rows + vrow * n;
    } else {
        const int lane = threadIdx.x, lr = lane / M;
        live = lane < rows * M;
        vrow = row0 + lr;
        i = lane - lr * M;
        out = tile + (long)lr * n;
    }
    if (live) {
        const long prob = vrow / (n + 1);
        const int row = (int)(vrow - prob * (n + 1));    // 0 = base, j+1 = column j
        const int j = row - 1;
        const double *zb = Zb + prob * n;
        const double hj = j >= 0 ? fd_step(zb[j], eps) : -0.0;
        auto z = [=](int k) -> double { return zb[k] + (k == j ? hj : -0.0); };      // see fdjac_lane_kernel
        if constexpr (PERPROB) {
            ModelParams Pq = P;
            ProblemDev pq = pb;
            load_problem_block(pb, prob, Pq, pq);
            segment_residual<Mdl, INTEG>(Pq, pq, z, i, [=](int r, double v) { out[r] = v; });
        } else {
            segment_residual<Mdl, INTEG>(P, pb, z, i, [=](int r, double v) { out[r] = v; });
        }
    }
    if (rows_per_block == 0) return;                 // uniform over the workgroup
    __syncthreads();
    store_tile(tile, Rows + row0 * n, (long)rows * n);
}

#ifdef SOCP_DEFINE_COMMON
// K_fdd: Jacobian from the rows of K_fdr: fjac[p][r + n*j] = (Rows[p][j+1][r] - Rows[p][0][r]) / h_j
__global__ void fd_diff_kernel(int n, int np, const double *__restrict__ Zb, double eps,
                               const double *__restrict__ Rows, double *__restrict__ Fjac)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)n * n;
    if (tid >= np * per) return;
    const long prob = tid / per;
    const int e = (int)(tid - prob * per);
    const int j = e / n, r = e - j * n;
    const double h = fd_step(Zb[prob * n + j], eps);
    const double *rows = Rows + prob * (long)(n + 1) * n;
    Fjac[tid] = (rows[(long)(j + 1) * n + r] - rows[r]) / h;
}
#endif

// K_eval: model::Model / Control / Hamiltonian for a batch of (t, X) points (set-up and trace
// paths of the host mirror; not on the hot path).
template <class Mdl>
__global__ __launch_bounds__(64) void eval_lane_kernel(ModelParams P, int what, int B,
                                                       const double *__restrict__ t,
                                                       const double *__restrict__ sw,
                                                       const double *__restrict__ Xin,
                                                       double *__restrict__ out)
{
    constexpr int S = Mdl::S;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double X[S];
#pragma unroll
    for (int k = 0; k < S; k++) X[k] = Xin[(long)b * S + k];
    const double s0 = sw ? sw[2 * b] : P.sw0;
    const double s1 = sw ? sw[2 * b + 1] : P.sw1;
    if (what == 0) {
        double dX[S];
        Mdl::rhs(P, s0, s1, t[b], X, dX);
#pragma unroll
        for (int k = 0; k < S; k++) out[(long)b * S + k] = dX[k];
    } else if (what == 1) {
        double u[3];
        Mdl::control_only(P, s0, s1, t[b], X, u);
        for (int k = 0; k < Mdl::NU; k++) out[(long)Mdl::NU * b + k] = u[k];
    } else {
        out[b] = Mdl::hamiltonian(P, s0, s1, t[b], X);
    }
}

}  // namespace socp
