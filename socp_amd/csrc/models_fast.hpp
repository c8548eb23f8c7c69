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

// 1/x to double precision from the hardware estimate + two Newton steps (no div_scale/fixup:
// arguments here are O(1) masses, never subnormal or huge)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// 1/sqrt(x) to double precision: hardware estimate + two coupled Newton steps
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    double e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-h * y, y, 0.5);
    return __builtin_fma(y, e, y);
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
        const double E = exp(-kr*(r - 1));
        const double ir2 = ir * ir;              // g = 1/r^2
        const double ir3 = ir2 * ir;             // g/r

        // control (goddard.cpp:104-185)
        const double Cm = C * im;
        const double Switch = P.p[GP_MU1] - b*p_mass - Cm*norm_pv;
        double alpha = 0;
        if (SMOOTH || P.p[GP_MU2] > 0) {
            if (Switch < 0) alpha = -Switch * (0.5 / P.p[GP_MU2]);
        } else if constexpr (!SMOOTH) {
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
        const double a_abs = fabs(alpha);
        // u = -p_v/|p_v| * alpha, rescaled to |u| = u_max when |alpha| > u_max (:167-176)
        // (alpha * u_max/|alpha| = sign(alpha) * u_max: no division needed)
        const double norm_u = a_abs > u_max ? u_max : a_abs;
        const double a_eff = copysign(norm_u, alpha);
        const double ua = -a_eff * iq;
        const double u0 = p_vx * ua, u1 = p_vy * ua, u2 = p_vz * ua;
        const double pvdotu = -a_eff * norm_pv;

        // state equations (:81-87)
        const double Dm = KD * E * im;           // KD exp(-kr(r-1)) / m
        const double Dv = Dm * v;
        dX[0] = vx;
        dX[1] = vy;
        dX[2] = vz;
        dX[3] = Cm*u0 - Dv*vx - ir3*x;
        dX[4] = Cm*u1 - Dv*vy - ir3*y;
        dX[5] = Cm*u2 - Dv*vz - ir3*z;
        dX[6] = -b*norm_u;
        // costate equations (:91-97)
        const double W = -(kr * Dv * pvdotv * ir) - 3.0 * ir3 * ir2 * pvdotr;
        dX[7] = W*x + ir3*p_vx;
        dX[8] = W*y + ir3*p_vy;
        dX[9] = W*z + ir3*p_vz;
        const double G = pvdotv * iv;
        dX[10] = Dm*(G*vx + p_vx*v) - p_x;
        dX[11] = Dm*(G*vy + p_vy*v) - p_y;
        dX[12] = Dm*(G*vz + p_vz*v) - p_z;
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
using CovidFast = CovidExact;     // IEEE +,-,*,/ only: nothing to restructure

}  // namespace socp
