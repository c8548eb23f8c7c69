// models_exact.hpp -- device dynamics in the REFERENCE OPERATION ORDER.
//
// These are the parity anchors on the GPU: every product, quotient and sum is formed in the
// association order of the reference expressions, the translation unit is compiled with
// -ffp-contract=off, and divisions / square roots are the IEEE-correct expansions.  Repeated
// sub-expressions (the reference evaluates exp(-kr(r-1)) ten times per call, and Control()
// recomputes what Model() already has) are bitwise-equal values, so naming them once does not
// change any result.  The only rounding that can differ from the x86 reference is inside
// ocml's exp.  Citations: file:line into bherisse/socp.
#pragma once
#include "dev_common.hpp"
#include "exp_glibc.hpp"

namespace socp {

// ---- IEEE quotients that share a denominator ---------------------------------------------------------------
// The compiler expands every double division n / d into the hardware's correctly rounded sequence
//     d' = div_scale(d), n' = div_scale(n);  r0 = rcp(d');  r1 = r0 + r0 (1 - d' r0);  r2 = r1 + r1 (1 - d' r1);
//     q0 = n' r2;  e = n' - d' q0;  q1 = q0 + e r2;  result = div_fixup(q1)
// (11 instructions plus hazard nops).  The scaling and the fix-up only act when an operand is zero, infinite,
// NaN, subnormal, or when the exponents are extreme; otherwise they pass the operands through and the first five
// steps depend on d alone.  `Den` does those five steps once and `n / Den` the last three, with the SAME
// instructions in the same order -- so for a denominator within 2^-400 .. 2^400 and numerators within
// 2^-568 .. 2^368 every quotient has exactly the bits of the compiler's n / d -- with one exception that no
// arithmetic can see: a NEGATIVE-ZERO numerator over a positive denominator gives +0 where the fix-up step would
// restore -0 (the shared denominators r, v, m, |p_v| are positive; a zero quotient only ever meets an addition).  The models below ask den_ok()
// for each shared denominator and fall back to plain division (DenT = double) otherwise; the reference's
// right-hand sides divide ~56 times by four distinct quantities (r, v, m, |p_v|).
struct Den {
    double d, r;
    __device__ __forceinline__ explicit Den(double den) : d(den)
    {
        const double r0 = __builtin_amdgcn_rcp(den);
        const double f0 = __builtin_fma(-den, r0, 1.0);
        const double r1 = __builtin_fma(r0, f0, r0);
        const double f1 = __builtin_fma(-den, r1, 1.0);
        r = __builtin_fma(r1, f1, r1);
    }
};
__device__ __forceinline__ double operator/(double n, const Den &D)
{
    const double q0 = n * D.r;
    const double e = __builtin_fma(-D.d, q0, n);
    return __builtin_fma(e, D.r, q0);
}
__device__ __forceinline__ bool den_ok(double d)
{
    const double a = fabs(d);
    return a > 0x1p-400 && a < 0x1p400;              // false for NaN
}

// The same idea for sqrt: the compiler's correctly rounded double sqrt is  scale-if-tiny, v_rsq, eight fused steps,
// unscale, and a class test that passes 0 / inf / NaN through (22 instructions).  For an argument within
// 2^-800 .. 2^800 the scaling and the class test are pass-through; sqrt_in_range() is the remaining ten
// instructions, in the compiler's order.
__device__ __forceinline__ bool sqrt_arg_ok(double x) { return x > 0x1p-800 && x < 0x1p800; }
__device__ __forceinline__ double sqrt_in_range(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

enum { GP_C = 0, GP_B, GP_KD, GP_KR, GP_UMAX, GP_MU1, GP_MU2, GP_SING };
enum { DP_UMAX = 0, DP_AMAX, DP_MUT };
enum { CP_R0 = 0, CP_TINF, CP_TINC, CP_N, CP_IMAX, CP_MUI, CP_UMIN, CP_UMAX };

// SMOOTH = true: specialised for the quadratic-cost law mu2 > 0 (goddard.cpp:137-145); the host
// selects it from the parameter block, so the kernel carries no bang/singular/off code.  Both
// instantiations evaluate identical expressions on the path they share.
template <bool SMOOTH>
struct GoddardExactT {
    static constexpr int D = 7;
    static constexpr int S = 14;
    static constexpr int NU = 3;
    static constexpr bool kRefOrder = true;

    // quantities both Model() and Control() derive from the state (goddard.cpp:66-76,121-130)
    struct Common {
        double r, v, pvdotv, g, norm_pv, E;
        bool shared;               // r, v, |p_v| and m are all in range: shared-denominator quotients, short sqrt
    };

    __device__ static __forceinline__ Common common(const ModelParams &P, const double (&X)[S])
    {
        Common c;
        const double r2 = X[0]*X[0] + X[1]*X[1] + X[2]*X[2];
        const double v2 = X[3]*X[3] + X[4]*X[4] + X[5]*X[5];
        const double q2 = X[10]*X[10] + X[11]*X[11] + X[12]*X[12];
        c.pvdotv = X[10]*X[3] + X[11]*X[4] + X[12]*X[5];
        // (squares within 2^-800 .. 2^800 <=> norms within 2^-400 .. 2^400, the den_ok range)
        c.shared = sqrt_arg_ok(r2) && sqrt_arg_ok(v2) && sqrt_arg_ok(q2) && den_ok(X[6]);
        if (c.shared) {
            c.r = sqrt_in_range(r2);
            c.v = sqrt_in_range(v2);
            c.norm_pv = sqrt_in_range(q2);
            const Den R(c.r);
            c.g = 1 / R / R;                                  // same bits as 1 / r / r (see Den)
        } else {
            c.r = sqrt(r2);
            c.v = sqrt(v2);
            c.norm_pv = sqrt(q2);
            c.g = 1 / c.r / c.r;
        }
        c.E = exp_glibc(-P.p[GP_KR]*(c.r - 1));
        return c;
    }

    // costate derivatives of position and velocity (goddard.cpp:91-96, reused at :214-219)
    // DenT = double: plain IEEE division; DenT = Den: the same quotients from shared reciprocals (see Den)
    template <class DenT = double>
    __device__ static __forceinline__ void costate_dots(const ModelParams &P, const Common &c,
                                                       const double (&X)[S], double (&pd)[6])
    {
        const double x = X[0], y = X[1], z = X[2], vx = X[3], vy = X[4], vz = X[5];
        const double p_x = X[7], p_y = X[8], p_z = X[9], p_vx = X[10], p_vy = X[11], p_vz = X[12];
        const double KD = P.p[GP_KD], kr = P.p[GP_KR];
        const double g = c.g, pvdotv = c.pvdotv, E = c.E;
        const DenT r(c.r), mass(X[6]);
        const double v = c.v;
        const double A = -kr*KD / mass*v*E;          // common left prefix of :91-93
        const double Q = KD / mass*E;                // common left prefix of :94-96
        pd[0] = A*x / r*pvdotv + g*(p_vx*(1 - 3 * x*x / r / r) / r - p_vy * 3 * x*y / r / r / r - p_vz * 3 * x*z / r / r / r);
        pd[1] = A*y / r*pvdotv + g*(-p_vx * 3 * y*x / r / r / r + p_vy*(1 - 3 * y*y / r / r) / r - p_vz * 3 * y*z / r / r / r);
        pd[2] = A*z / r*pvdotv + g*(-p_vx * 3 * z*x / r / r / r - p_vy * 3 * z*y / r / r / r + p_vz*(1 - 3 * z*z / r / r) / r);
        const DenT dv(c.v);
        pd[3] = -p_x + Q*(pvdotv*vx / dv + p_vx*v);
        pd[4] = -p_y + Q*(pvdotv*vy / dv + p_vy*v);
        pd[5] = -p_z + Q*(pvdotv*vz / dv + p_vz*v);
    }

    // goddard.cpp:188-253
    __device__ static double singular_control(const ModelParams &P, const Common &c, const double (&X)[S])
    {
        const double x = X[0], y = X[1], z = X[2], vx = X[3], vy = X[4], vz = X[5], mass = X[6];
        const double p_x = X[7], p_y = X[8], p_z = X[9], p_vx = X[10], p_vy = X[11], p_vz = X[12];
        const double b = P.p[GP_B], C = P.p[GP_C], KD = P.p[GP_KD], kr = P.p[GP_KR];
        const double r = c.r, v = c.v, g = c.g, pvdotv = c.pvdotv, norm_pv = c.norm_pv;
        const double rdotv = x*vx + y*vy + z*vz;
        const double D = KD*c.E;
        double pd[6];
        costate_dots(P, c, X, pd);
        const double prdotdotpv = pd[0]*p_vx + pd[1]*p_vy + pd[2]*p_vz;
        const double prdotpvdot = p_x*pd[3] + p_y*pd[4] + p_z*pd[5];
        const double prdotpv = p_x*p_vx + p_y*p_vy + p_z*p_vz;
        const double pvdotdotv = pd[3]*vx + pd[4]*vy + pd[5]*vz;
        const double pvdotdotpv = pd[3]*p_vx + pd[4]*p_vy + pd[5]*p_vz;
        const double vdotg = vx*g*x / r + vy*g*y / r + vz*g*z / r;
        const double pvdotg = p_vx*g*x / r + p_vy*g*y / r + p_vz*g*z / r;

        const double au = 2 * norm_pv*C / mass*pvdotv
            + 2 * pvdotv*C / mass*norm_pv
            - b / mass*(2 * pvdotv*pvdotv + norm_pv*norm_pv*v*v)
            - b / D*v*prdotpv - C / D*prdotpv / v*pvdotv / norm_pv;

        const double bu = -2 * norm_pv*norm_pv*(vdotg + D / mass*v*v*v) + 2 * v*v*pvdotdotpv
            - 2 * pvdotv*(pvdotg + D / mass*v*pvdotv - pvdotdotv)
            + b / C*(2 * norm_pv*pvdotv*(vdotg + D / mass*v*v*v) + norm_pv*v*v*(pvdotg + D / mass*v*pvdotv - pvdotdotv) - v*v*pvdotv / norm_pv*pvdotdotpv)
            - mass / D*kr*rdotv / r*v*prdotpv + mass / D*prdotpv / v*(vdotg + D / mass*v*v*v) - mass / D*v*(prdotdotpv + prdotpvdot);

        return bu / au;
    }

    // goddard.cpp:104-185
    template <class DenT = double>
    __device__ static __forceinline__ void control(const ModelParams &P, const Common &c, double sw0, double sw1,
                                                  double t, const double (&X)[S], double (&u)[3])
    {
        const double p_vx = X[10], p_vy = X[11], p_vz = X[12], p_mass = X[13];
        const DenT mass(X[6]), npv(c.norm_pv);
        const double Switch = P.p[GP_MU1] - P.p[GP_B]*p_mass - P.p[GP_C] / mass*c.norm_pv;
        double alpha_u = 0;
        if (SMOOTH || P.p[GP_MU2] > 0) {
            if (Switch < 0) alpha_u = -Switch / 2 / P.p[GP_MU2];
        } else if constexpr (!SMOOTH) {
            if (t <= sw0) {
                alpha_u = 1.0;
            } else if (t > sw0 && t <= sw1) {
                alpha_u = (P.p[GP_SING] < 0) ? singular_control(P, c, X) : P.p[GP_SING];
            }
        }
        u[0] = -p_vx*alpha_u / npv;
        u[1] = -p_vy*alpha_u / npv;
        u[2] = -p_vz*alpha_u / npv;
        const double norm_u = fabs(alpha_u);
        const double u_max = P.p[GP_UMAX];
        if (norm_u > u_max) {
            u[0] = u[0] / norm_u*u_max;
            u[1] = u[1] / norm_u*u_max;
            u[2] = u[2] / norm_u*u_max;
        }
    }

    // goddard.cpp:48-101
    __device__ static __forceinline__ void rhs(const ModelParams &P, double sw0, double sw1, double t,
                                              const double (&X)[S], double (&dX)[S])
    {
        const Common c = common(P, X);
        if (c.shared) rhs_with<Den>(P, c, sw0, sw1, t, X, dX);
        else rhs_with<double>(P, c, sw0, sw1, t, X, dX);
    }

    template <class DenT>
    __device__ static __forceinline__ void rhs_with(const ModelParams &P, const Common &c, double sw0, double sw1, double t,
                                                   const double (&X)[S], double (&dX)[S])
    {
        double u[3];
        control<DenT>(P, c, sw0, sw1, t, X, u);
        const double x = X[0], y = X[1], z = X[2], vx = X[3], vy = X[4], vz = X[5];
        const DenT mass(X[6]), cr(c.r);
        const double p_vx = X[10], p_vy = X[11], p_vz = X[12];
        const double b = P.p[GP_B], C = P.p[GP_C], KD = P.p[GP_KD];
        const double norm_u = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        const double pvdotu = p_vx*u[0] + p_vy*u[1] + p_vz*u[2];
        const double W = -KD*c.v;                    // common left prefix of :84-86
        double pd[6];
        costate_dots<DenT>(P, c, X, pd);
        dX[0] = vx;
        dX[1] = vy;
        dX[2] = vz;
        dX[3] = W*vx*c.E / mass - c.g*x / cr + C*u[0] / mass;
        dX[4] = W*vy*c.E / mass - c.g*y / cr + C*u[1] / mass;
        dX[5] = W*vz*c.E / mass - c.g*z / cr + C*u[2] / mass;
        dX[6] = -b*norm_u;
        dX[7] = pd[0];
        dX[8] = pd[1];
        dX[9] = pd[2];
        dX[10] = pd[3];
        dX[11] = pd[4];
        dX[12] = pd[5];
        dX[13] = -KD*c.E / mass / mass*c.v*c.pvdotv + C / mass / mass*pvdotu;
    }

    // goddard.cpp:104 as a standalone entry (trace / model::Control)
    __device__ static void control_only(const ModelParams &P, double sw0, double sw1, double t,
                                        const double (&X)[S], double (&u)[3])
    {
        const Common c = common(P, X);
        control(P, c, sw0, sw1, t, X, u);
    }

    // goddard.cpp:256-295
    __device__ static double hamiltonian(const ModelParams &P, double sw0, double sw1, double t, const double (&X)[S])
    {
        const Common c = common(P, X);
        double u[3];
        control(P, c, sw0, sw1, t, X, u);
        const double x = X[0], y = X[1], z = X[2], vx = X[3], vy = X[4], vz = X[5], mass = X[6];
        const double p_x = X[7], p_y = X[8], p_z = X[9], p_vx = X[10], p_vy = X[11], p_vz = X[12], p_mass = X[13];
        const double b = P.p[GP_B], C = P.p[GP_C], KD = P.p[GP_KD];
        const double norm_u = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        const double W = -KD*c.v;
        const double H = P.p[GP_MU1]*norm_u + P.p[GP_MU2]*norm_u*norm_u
            + p_x*vx + p_y*vy + p_z*vz
            + p_vx*(W*vx*c.E / mass - c.g*x / c.r + C*u[0] / mass)
            + p_vy*(W*vy*c.E / mass - c.g*y / c.r + C*u[1] / mass)
            + p_vz*(W*vz*c.E / mass - c.g*z / c.r + C*u[2] / mass)
            - p_mass*b*norm_u;
        return H;
    }

    // goddard.cpp:343-370: the free interior-time row is H(t, X-) (Xp unused)
    __device__ static double switching_fn(const ModelParams &P, double sw0, double sw1, double t,
                                          const double (&X)[S], const double (&)[S])
    {
        return hamiltonian(P, sw0, sw1, t, X);
    }
};

using GoddardExact = GoddardExactT<false>;        // general law (any mu2)
using GoddardExactSmooth = GoddardExactT<true>;   // mu2 > 0 only

struct DIntExact {
    static constexpr int D = 6;
    static constexpr int S = 12;
    static constexpr int NU = 3;
    static constexpr bool kRefOrder = true;

    // doubleIntegrator.cpp:218-259
    __device__ static __forceinline__ void control_only(const ModelParams &P, double, double, double,
                                                       const double (&X)[S], double (&u)[3])
    {
        const double a_max = P.p[DP_AMAX], u_max = P.p[DP_UMAX];
        u[0] = -X[9] / a_max;
        u[1] = -X[10] / a_max;
        u[2] = -X[11] / a_max;
        const double norm_u = sqrt(u[0]*u[0] + u[1]*u[1] + u[2]*u[2]);
        if (norm_u > u_max) {
            u[0] = u[0] / norm_u*u_max;
            u[1] = u[1] / norm_u*u_max;
            u[2] = u[2] / norm_u*u_max;
        }
    }

    // doubleIntegrator.cpp:67-108
    __device__ static __forceinline__ void rhs(const ModelParams &P, double sw0, double sw1, double t,
                                              const double (&X)[S], double (&dX)[S])
    {
        double u[3];
        control_only(P, sw0, sw1, t, X, u);
        const double a_max = P.p[DP_AMAX];
        dX[0] = X[3];  dX[1] = X[4];  dX[2] = X[5];
        dX[3] = a_max * u[0];  dX[4] = a_max * u[1];  dX[5] = a_max * u[2];
        dX[6] = 0;  dX[7] = 0;  dX[8] = 0;
        dX[9] = -X[6];  dX[10] = -X[7];  dX[11] = -X[8];
    }

    // doubleIntegrator.cpp:264-300, isJac == 0
    __device__ static double hamiltonian(const ModelParams &P, double sw0, double sw1, double t, const double (&X)[S])
    {
        double u[3];
        control_only(P, sw0, sw1, t, X, u);
        const double a_max = P.p[DP_AMAX];
        const double norm_u = sqrt(u[0]*u[0] + u[1]*u[1] + u[2]*u[2]);
        return P.p[DP_MUT] + a_max * a_max*norm_u*norm_u / 2 + X[6] * X[3] + X[7] * X[4] + X[8] * X[5]
             + a_max * (X[9]*u[0] + X[10] * u[1] + X[11] * u[2]);
    }

    // model.hpp:299-304 default: H(t, X-) - H(t, X+)
    __device__ static double switching_fn(const ModelParams &P, double sw0, double sw1, double t,
                                          const double (&X)[S], const double (&Xp)[S])
    {
        return hamiltonian(P, sw0, sw1, t, X) - hamiltonian(P, sw0, sw1, t, Xp);
    }
};

// SEIR epidemic model with a social-distancing control (covid19.cpp:53-165), dim 4.  IEEE operations
// only (no exp / sqrt): bit-identical to the x86 path.
struct CovidExact {
    static constexpr int D = 4;
    static constexpr int S = 8;
    static constexpr int NU = 1;
    static constexpr bool kRefOrder = true;

    // covid19.cpp:97-126
    __device__ static __forceinline__ double control_scalar(const ModelParams &P, const double (&X)[S])
    {
        double u = (X[5] - X[4])*X[0]*X[2] / P.p[CP_TINF] / P.p[CP_N] * P.p[CP_R0];
        if (u <= P.p[CP_UMIN]) u = P.p[CP_UMIN];
        if (u >= P.p[CP_UMAX]) u = P.p[CP_UMAX];
        return u;
    }
    __device__ static void control_only(const ModelParams &P, double, double, double, const double (&X)[S], double (&u)[3])
    {
        u[0] = control_scalar(P, X); u[1] = 0; u[2] = 0;
    }

    // covid19.cpp:53-95.  All ten divisions are by the model constants Tinf, Tinc, N: shared denominators (see Den).
    __device__ static __forceinline__ void rhs(const ModelParams &P, double, double, double,
                                              const double (&X)[S], double (&dX)[S])
    {
        if (den_ok(P.p[CP_TINF]) && den_ok(P.p[CP_TINC]) && den_ok(P.p[CP_N])) rhs_with<Den>(P, X, dX);
        else rhs_with<double>(P, X, dX);
    }

    template <class DenT>
    __device__ static __forceinline__ void rhs_with(const ModelParams &P, const double (&X)[S], double (&dX)[S])
    {
        const double Sx = X[0], E = X[1], I = X[2], R = X[3], pS = X[4], pE = X[5], pI = X[6], pR = X[7];
        const double R0 = P.p[CP_R0];
        const DenT Tinf(P.p[CP_TINF]), Tinc(P.p[CP_TINC]), N(P.p[CP_N]);
        double u = (X[5] - X[4])*X[0]*X[2] / Tinf / N * R0;             // control_scalar (:97-126) with the shared denominators
        if (u <= P.p[CP_UMIN]) u = P.p[CP_UMIN];
        if (u >= P.p[CP_UMAX]) u = P.p[CP_UMAX];
        const double Rt = R0 * (1 - u);
        double Ipen = 0;
        if (I >= P.p[CP_IMAX]) Ipen = -P.p[CP_MUI]*(I - P.p[CP_IMAX]);
        dX[0] = -Rt / Tinf / N*Sx*I;
        dX[1] = Rt / Tinf / N*Sx*I - E / Tinc;
        dX[2] = E / Tinc - I / Tinf;
        dX[3] = I / Tinf;
        dX[4] = (pS - pE)*R*I / Tinf / N;
        dX[5] = (pE - pI) / Tinc;
        dX[6] = (pS - pE)*R*Sx / Tinf / N + (pI - pR) / Tinf + Ipen;
        dX[7] = 0;
    }

    // covid19.cpp:128-165
    __device__ static double hamiltonian(const ModelParams &P, double, double, double, const double (&X)[S])
    {
        const double Sx = X[0], E = X[1], I = X[2], pS = X[4], pE = X[5], pI = X[6], pR = X[7];
        const double R0 = P.p[CP_R0], Tinf = P.p[CP_TINF], Tinc = P.p[CP_TINC], N = P.p[CP_N];
        const double u = control_scalar(P, X);
        const double Rt = R0 * (1 - u);
        double Ipen = 0;
        if (I >= P.p[CP_IMAX]) Ipen = P.p[CP_MUI]*(I - P.p[CP_IMAX])*(I - P.p[CP_IMAX]) / 2;
        return u*u / 2 + Ipen
            + pS * (-Rt / Tinf / N*Sx*I)
            + pE * (Rt / Tinf / N*Sx*I - E / Tinc)
            + pI * (E / Tinc - I / Tinf)
            + pR * (I / Tinf);
    }

    // model.hpp:299-304 default
    __device__ static double switching_fn(const ModelParams &P, double sw0, double sw1, double t,
                                          const double (&X)[S], const double (&Xp)[S])
    {
        return hamiltonian(P, sw0, sw1, t, X) - hamiltonian(P, sw0, sw1, t, Xp);
    }
};

}  // namespace socp
