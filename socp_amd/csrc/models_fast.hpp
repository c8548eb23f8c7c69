// models_fast.hpp -- restructured device dynamics for the throughput flavour.
//
// Same mathematics as goddard.cpp:48-185 / doubleIntegrator.cpp:49-259, different rounding:
// the 62 divisions and 7 square roots of one reference Goddard RHS call become three inverse
// square roots (1/r, 1/v, 1/|p_v|), one reciprocal (1/m) and one exp; the gravity-gradient block
// is factored as  p_i' = (A - 3 (p_v.r) / r^5) x_i + p_v,i / r^3 ; |u| is |alpha| (saturated)
// instead of a recomputed norm; compiled with FMA contraction.  Every result differs from the
// reference-order flavour at rounding level only; tests/test_gpu_parity.py states and checks
// the tolerance (<= 1e-8 relative after 1e4 RK4 steps) and the converged-solution
// parity.  The bang / singular / off law (mu2 <= 0) keeps the reference-order singular-control
// expression (rare branch, one arc).
#pragma once
#include "models_exact.hpp"

namespace socp {

// 1/x to double precision from the hardware estimate r0 (relative error e ~ 2^-24) in ONE cubic step:
// 1/x = r0 / (1 - e) with e = 1 - x r0, so r0 (1 + e + e^2) is off by e^3 ~ 1e-22 (no div_scale/fixup:
// arguments here are O(1) masses, never subnormal or huge).  4 instructions.
__device__ __forceinline__ double fast_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    const double p = __builtin_fma(e, e, e);
    return __builtin_fma(r, p, r);
}

// 1/sqrt(x) to double precision, one cubic step from the hardware estimate y0: with d = 1 - x y0^2,
// 1/sqrt(x) = y0 (1 - d)^(-1/2) = y0 (1 + d/2 + 3/8 d^2 + 5/16 d^3 ...); y0 + y0 d (1/2 + 3/8 d) is off by
// 5/16 d^3 ~ 1e-22.  6 instructions.
__device__ __forceinline__ double fast_rsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double d = __builtin_fma(-(x * y), y, 1.0);
    const double p = __builtin_fma(0.375, d, 0.5);
    return __builtin_fma(y * d, p, y);
}

// exp(x) without the library's overflow / underflow selects: n = round(x log2 e), t = x - n ln2 (two-part
// ln2), exp(t) = 1 + t + t^2 g(t) with g a degree-9 interpolant of (e^t - 1 - t)/t^2 at the Chebyshev nodes of
// |t| <= ln2/2 (coefficients solved in 80-digit arithmetic; evaluated error <= 1 ulp), scaled by ldexp --
// which itself saturates to 0 / inf; NaN propagates through the polynomial.  The air-density term of the
// Goddard model has |x| = kr |r - 1| of order 10.
__device__ __forceinline__ double fast_exp(double x)
{
    const double n = __builtin_rint(x * 1.4426950408889634);
    double t = __builtin_fma(-n, 0x1.62e42fefa39efp-1, x);
    t = __builtin_fma(-n, 0x1.abc9e3b39803fp-56, t);
    double p = 0x1.af38d53857513p-26;
    p = __builtin_fma(p, t, 0x1.2891a8c1d838dp-22);
    p = __builtin_fma(p, t, 0x1.71de0d9c145d0p-19);
    p = __builtin_fma(p, t, 0x1.a019b8ef67c6cp-16);
    p = __builtin_fma(p, t, 0x1.a01a01a7c8d47p-13);
    p = __builtin_fma(p, t, 0x1.6c16c17893833p-10);
    p = __builtin_fma(p, t, 0x1.11111111109adp-7);
    p = __builtin_fma(p, t, 0x1.5555555553d4fp-5);
    p = __builtin_fma(p, t, 0x1.5555555555556p-3);
    p = __builtin_fma(p, t, 0x1.0000000000001p-1);
    p = __builtin_fma(p, t, 1.0);
    p = __builtin_fma(p, t, 1.0);
    return ldexp(p, (int)n);
}

template <bool SMOOTH>
struct GoddardFastT {
    static constexpr int D = 7;
    static constexpr int S = 14;
    static constexpr int NU = 3;
    static constexpr bool kRefOrder = false;

    __device__ static __forceinline__ void rhs(const ModelParams &P, double sw0, double sw1, double t,
                                              const double (&X)[S], double (&dX)[S])
    {
        const double x = X[0], y = X[1], z = X[2], vx = X[3], vy = X[4], vz = X[5], mass = X[6];
        const double p_x = X[7], p_y = X[8], p_z = X[9], p_vx = X[10], p_vy = X[11], p_vz = X[12], p_mass = X[13];
        const double b = P.p[GP_B], C = P.p[GP_C], KD = P.p[GP_KD], kr = P.p[GP_KR];

        const double r2 = x*x + y*y + z*z;
        const double v2 = vx*vx + vy*vy + vz*vz;
        const double q2 = p_vx*p_vx + p_vy*p_vy + p_vz*p_vz;
        const double ir = fast_rsqrt(r2), iv = fast_rsqrt(v2), iq = fast_rsqrt(q2);
        const double r = r2 * ir, v = v2 * iv, norm_pv = q2 * iq;
        const double im = fast_rcp(mass);
        const double pvdotv = p_vx*vx + p_vy*vy + p_vz*vz;
        const double pvdotr = p_vx*x + p_vy*y + p_vz*z;
        const double E = fast_exp(-kr*(r - 1));
        const double ir2 = ir * ir;              // g = 1/r^2
        const double ir3 = ir2 * ir;             // g/r

        // control (goddard.cpp:104-185)
        const double Cm = C * im;
        const double Switch = P.p[GP_MU1] - b*p_mass - Cm*norm_pv;
        double alpha = 0;
        if constexpr (SMOOTH) {
            alpha = __builtin_fmax(-Switch * (0.5 / P.p[GP_MU2]), 0.0);      // Switch < 0 ? -Switch/(2 mu2) : 0
        } else if (P.p[GP_MU2] > 0) {
            if (Switch < 0) alpha = -Switch * (0.5 / P.p[GP_MU2]);
        }
        if constexpr (!SMOOTH) if (!(P.p[GP_MU2] > 0)) {
            if (t <= sw0) {
                alpha = 1.0;
            } else if (t > sw0 && t <= sw1) {
                if (P.p[GP_SING] < 0) {
                    GoddardExact::Common c;
                    c.r = r; c.v = v; c.pvdotv = pvdotv; c.g = ir2; c.norm_pv = norm_pv; c.E = E;
                    alpha = GoddardExact::singular_control(P, c, X);
                } else {
                    alpha = P.p[GP_SING];
                }
            }
        }
        const double u_max = P.p[GP_UMAX];
        // u = -p_v/|p_v| * alpha, rescaled to |u| = u_max when |alpha| > u_max (:167-176)
        // (alpha * u_max/|alpha| = sign(alpha) * u_max: no division needed)
        double norm_u, a_eff;
        if constexpr (SMOOTH) {
            norm_u = __builtin_fmin(alpha, u_max);           // the smooth law gives alpha >= 0
            a_eff = norm_u;
        } else {
            const double a_abs = fabs(alpha);
            norm_u = a_abs > u_max ? u_max : a_abs;
            a_eff = copysign(norm_u, alpha);
        }
        const double ua = -a_eff * iq;                       // u = ua * p_v
        const double pvdotu = -a_eff * norm_pv;

        // state equations (:81-87), thrust term C/m u_i written as (C/m ua) p_v,i
        const double Dm = KD * E * im;           // KD exp(-kr(r-1)) / m
        const double Dv = Dm * v;
        const double Tm = Cm * ua;
        dX[0] = vx;
        dX[1] = vy;
        dX[2] = vz;
        dX[3] = Tm*p_vx - Dv*vx - ir3*x;
        dX[4] = Tm*p_vy - Dv*vy - ir3*y;
        dX[5] = Tm*p_vz - Dv*vz - ir3*z;
        dX[6] = -b*norm_u;
        // costate equations (:91-97)
        const double W = -(kr * Dv * pvdotv * ir) - 3.0 * ir3 * ir2 * pvdotr;
        dX[7] = W*x + ir3*p_vx;
        dX[8] = W*y + ir3*p_vy;
        dX[9] = W*z + ir3*p_vz;
        const double DG = Dm * (pvdotv * iv);
        dX[10] = DG*vx + (Dv*p_vx - p_x);
        dX[11] = DG*vy + (Dv*p_vy - p_y);
        dX[12] = DG*vz + (Dv*p_vz - p_z);
        dX[13] = im * (Cm*pvdotu - Dv*pvdotv);
    }

    __device__ static void control_only(const ModelParams &P, double sw0, double sw1, double t,
                                        const double (&X)[S], double (&u)[3])
    {
        GoddardExactT<SMOOTH>::control_only(P, sw0, sw1, t, X, u);
    }
    __device__ static double hamiltonian(const ModelParams &P, double sw0, double sw1, double t, const double (&X)[S])
    {
        return GoddardExactT<SMOOTH>::hamiltonian(P, sw0, sw1, t, X);
    }
    __device__ static double switching_fn(const ModelParams &P, double sw0, double sw1, double t,
                                          const double (&X)[S], const double (&Xp)[S])
    {
        return GoddardExactT<SMOOTH>::switching_fn(P, sw0, sw1, t, X, Xp);
    }
};

using GoddardFast = GoddardFastT<false>;
using GoddardFastSmooth = GoddardFastT<true>;

// The double integrator has one division per control component and one rare square root: nothing
// to restructure beyond what contraction gives.
using DIntFast = DIntExact;
// SEIR model, throughput flavour: the reference's ten divisions per right-hand side are all by the model constants
// Tinf, Tinc, N -- here three reciprocals that do not depend on the state (the compiler hoists them out of the
// time loop) and one shared infection flux Rt S I / (Tinf N); contraction on.  Rounding-level differences only.
struct CovidFast : CovidExact {
    static constexpr bool kRefOrder = false;

    __device__ static __forceinline__ void rhs(const ModelParams &P, double, double, double,
                                              const double (&X)[S], double (&dX)[S])
    {
        const double Sx = X[0], E = X[1], I = X[2], R = X[3], pS = X[4], pE = X[5], pI = X[6], pR = X[7];
        const double iTinf = fast_rcp(P.p[CP_TINF]), iTinc = fast_rcp(P.p[CP_TINC]), iN = fast_rcp(P.p[CP_N]);
        const double k = iTinf * iN;
        double u = (pE - pS) * Sx * I * k * P.p[CP_R0];
        u = __builtin_fmin(__builtin_fmax(u, P.p[CP_UMIN]), P.p[CP_UMAX]);
        const double Rt = P.p[CP_R0] * (1 - u);
        const double dI = I - P.p[CP_IMAX];
        const double Ipen = dI >= 0 ? -P.p[CP_MUI] * dI : 0.0;
        const double flux = Rt * k * Sx * I;
        const double EoT = E * iTinc, IoT = I * iTinf;
        const double dp = (pS - pE) * R * k;
        dX[0] = -flux;
        dX[1] = flux - EoT;
        dX[2] = EoT - IoT;
        dX[3] = IoT;
        dX[4] = dp * I;
        dX[5] = (pE - pI) * iTinc;
        dX[6] = dp * Sx + (pI - pR) * iTinf + Ipen;
        dX[7] = 0;
    }
};

}  // namespace socp
