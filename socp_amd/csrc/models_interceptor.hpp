// models_interceptor.hpp -- device twin of the reference's `interceptor` model (endo-atmospheric
// launch vehicle, two coordinate charts, powered + coasting stage; interceptor.cpp:34-998).
//
// State [h, v, a1, a2, L, l ; p_h, p_v, p_a1, p_a2, p_L, p_l] with (a1, a2) = (gamma, chi) in chart 1 and
// (theta, phi) in chart 2.  Unlike the other models the dynamics depend on two per-trajectory flags the
// reference keeps in its data_struct: stageMode (1 = powered) and currentChart.  They travel in the two
// per-lane auxiliary scalars every model entry point receives (the Goddard model's switching times):
//     a0 = stage mode (0.0 / 1.0),   a1 = chart (1.0 / 2.0).
// The model integrates with its own driver (kCustomTraj): interceptor::ComputeTraj + ModelInt
// (:101-128, :162-218) -- stepNbr RK4 steps per stage, the chart re-chosen before every step, the
// state handed back in chart 1 -- and replaces the final-boundary rows (kCustomFinal; :221-272).
//
// Operation order follows the cited expressions term by term; sin/cos/tan/atan2/acos/exp are the device
// library's (<= 1 ulp from libm, not bit-identical).  The reference's two Eigen calls in the chart change
// (6x6 PartialPivLU solve and matrix-vector product) are a textbook partial-pivot LU and row sums here,
// written with static indices only so the 6x6 stays in registers.
#pragma once
#include "exp_glibc.hpp"
#include "integrator.hpp"

namespace socp {

// ---- helpers of the throughput flavour (FAST = true; compiled with FMA contraction) -----------------------
namespace ifast {
// 1/x and 1/sqrt(x) in one cubic step from the hardware estimate (error e^3 ~ 1e-22; arguments are O(1..1e7)
// physical quantities, never subnormal)
__device__ __forceinline__ double rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
__device__ __forceinline__ double rsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double d = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * d, __builtin_fma(0.375, d, 0.5), y);
}
// sin and cos together: k = round(2x/pi), r = x - k pi/2 with a two-part pi/2 (Cody-Waite; exact enough for
// |x| up to ~1e4, the flight-path / heading / latitude angles are O(1)), degree-15 / degree-16 Taylor
// polynomials on |r| <= pi/4 (remainders 5e-17, 2e-18), quadrant by select.  ~1 ulp; 33 instructions instead
// of the library's ~120 with its large-argument path.
__device__ __forceinline__ void sincos(double x, double &s, double &c)
{
    const double k = __builtin_rint(x * 0.6366197723675814);
    double r = __builtin_fma(-k, 0x1.921fb54442d18p+0, x);
    r = __builtin_fma(-k, 0x1.1a62633145c07p-54, r);
    const double r2 = r * r;
    double p = -1.0 / 1307674368000.0;
    p = __builtin_fma(p, r2, 1.0 / 6227020800.0);
    p = __builtin_fma(p, r2, -1.0 / 39916800.0);
    p = __builtin_fma(p, r2, 1.0 / 362880.0);
    p = __builtin_fma(p, r2, -1.0 / 5040.0);
    p = __builtin_fma(p, r2, 1.0 / 120.0);
    p = __builtin_fma(p, r2, -1.0 / 6.0);
    const double sr = __builtin_fma(r * r2, p, r);
    double q = 1.0 / 20922789888000.0;
    q = __builtin_fma(q, r2, -1.0 / 87178291200.0);
    q = __builtin_fma(q, r2, 1.0 / 479001600.0);
    q = __builtin_fma(q, r2, -1.0 / 3628800.0);
    q = __builtin_fma(q, r2, 1.0 / 40320.0);
    q = __builtin_fma(q, r2, -1.0 / 720.0);
    q = __builtin_fma(q, r2, 1.0 / 24.0);
    const double cr = __builtin_fma(r2 * r2, q, __builtin_fma(-0.5, r2, 1.0));
    const int n = (int)k;
    const bool swap = n & 1;
    const double s0 = swap ? cr : sr, c0 = swap ? sr : cr;
    s = (n & 2) ? -s0 : s0;
    c = ((n + 1) & 2) ? -c0 : c0;
}
}  // namespace ifast

template <bool FAST>
struct InterceptorT {
    static constexpr int D = 6;
    static constexpr int S = 12;
    static constexpr int NU = 2;
    static constexpr bool kRefOrder = true;
    static constexpr bool kCustomTraj = true;
    static constexpr bool kCustomFinal = true;
    static constexpr bool kOneWavePerSimd = true;      // launch-table hint: instantiate the WPE = 1 kernels only
    // FAST: the right-hand side is restructured (reciprocals instead of the ~25 divisions, cos/sin(beta) from the
    // atan2 arguments instead of atan2 + sincos, tan = sin/cos, short sincos, contraction); same mathematics,
    // rounding-level differences.  Control(), Hamiltonian(), the chart change and the drivers are shared.

    // parameter slots = include/socp_hip.h SOCP_INTERCEPTOR_* (interceptor.hpp:28-46 order, then R_Earth, mu0, chartLimit)
    enum { C0 = 0, HR, D0, ETA, PROP, EMPTY, Q, VE, ALPHA_MAX, UMAX, AMAX, MU_GFT, MUT, MUV, MUC, REARTH, MU0, CHART_LIMIT };

    struct Com { double mass, c_max, d, r, g, ft; };

    // the temporaries Model_k / Control_k / Hamiltonian_k all start with (:292-308), ComputeMass (:981-997)
    __device__ static __forceinline__ Com common(const ModelParams &P, double stage, double t, double h)
    {
        Com c;
        const double qm_mass = P.p[Q] * P.p[MU_GFT];
        const double t1 = P.p[PROP] / P.p[Q];
        const double qm = stage * P.p[Q] * P.p[MU_GFT];
        c.mass = P.p[EMPTY] + P.p[PROP] - qm_mass * (stage == 1.0 ? t : t1);
        const double e = exp_glibc(-h / P.p[HR]);
        c.c_max = P.p[C0] * e * (P.p[PROP] + P.p[EMPTY]) / c.mass;
        c.d = P.p[D0] * e * (P.p[PROP] + P.p[EMPTY]) / c.mass;
        c.r = h + P.p[REARTH];
        c.g = P.p[MU0] / c.r / c.r * P.p[MU_GFT];
        c.ft = P.p[VE] * qm;
        return c;
    }

    // Control_1 (:338-385) / Control_2 (:506-552).  c1 = cos(gamma) or cos(theta); the costate pair enters as
    // (p_a1, +p_a2) in chart 1 and (p_a1, -p_a2) in chart 2.
    __device__ static __forceinline__ void control_k(const ModelParams &P, const Com &c, bool chart1, double v, double c1,
                                                    double p_v, double p_a1, double p_a2, double &u, double &beta,
                                                    double &sb, double &cb)
    {
        const double mass = c.mass, c_max = c.c_max, ft = c.ft, alpha_max = P.p[ALPHA_MAX], eta = P.p[ETA];
        beta = chart1 ? atan2(p_a2, p_a1 * c1) : atan2(-p_a2, p_a1 * c1);
        sincos(beta, &sb, &cb);
        const double A = p_a1 * (v * c_max * cb + ft * cb * alpha_max / mass / v);
        const double B = p_a2 * (v * c_max * sb / c1 + ft * sb / c1 * alpha_max / mass / v);
        const double den = p_v * (2 * eta * c_max * v * v + ft * alpha_max * alpha_max / mass) - P.p[MUC];
        u = (chart1 ? A + B : A - B) / den;
        if (fabs(u) > P.p[UMAX]) u = P.p[UMAX] * u / fabs(u);
    }

    __device__ static __forceinline__ void control_only(const ModelParams &P, double stage, double chart, double t,
                                                       const double (&X)[S], double (&uo)[3])
    {
        const Com c = common(P, stage, t, X[0]);
        double u, beta, sb, cb;
        control_k(P, c, chart == 1.0, X[1], cos(X[2]), X[7], X[8], X[9], u, beta, sb, cb);
        uo[0] = u; uo[1] = beta; uo[2] = 0;
    }

    // Model_1 (:275-335) / Model_2 (:440-503)
    __device__ static __forceinline__ void rhs_ref(const ModelParams &P, double stage, double chart, double t,
                                              const double (&X)[S], double (&Xdot)[S])
    {
        const Com c = common(P, stage, t, X[0]);
        const double v = X[1], L = X[4];
        const double p_h = X[6], p_v = X[7], p_a1 = X[8], p_a2 = X[9], p_L = X[10], p_l = X[11];
        const double mass = c.mass, c_max = c.c_max, d = c.d, r = c.r, g = c.g, ft = c.ft, eta = P.p[ETA], hr = P.p[HR];
        double s1, c1, s2, c2, sL, cL;
        sincos(X[2], &s1, &c1);
        sincos(X[3], &s2, &c2);
        sincos(L, &sL, &cL);
        const double tL = tan(L);
        double u, beta, sb, cb;
        control_k(P, c, chart == 1.0, v, c1, p_v, p_a1, p_a2, u, beta, sb, cb);
        const double alpha = P.p[ALPHA_MAX] * u;
        double sa, ca;
        sincos(alpha, &sa, &ca);
        if (chart == 1.0) {
            const double sg = s1, cg = c1, sc = s2, cc = c2, p_gamma = p_a1, p_chi = p_a2;
            Xdot[0] = v * sg;
            Xdot[1] = -(d + eta * c_max * u * u) * v * v - g * sg + ft * ca / mass;
            Xdot[2] = v * c_max * u * cb - g / v * cg + ft * sa * cb / mass / v + v * cg / r;
            Xdot[3] = v * c_max * u * sb / cg + ft * sa * sb / cg / mass / v + v * cg * tL * sc / r;
            Xdot[4] = v * cg * cc / r;
            Xdot[5] = v * cg * sc / cL / r;
            Xdot[6] = -p_v / hr * (d + eta * c_max * u * u) * v * v - 2 * g / r * (p_gamma / v * cg + p_v * sg)
                      + p_L * v * cg * cc / r / r + p_gamma * v * cg / r / r + p_gamma * v * c_max * u * cb / hr
                      + p_l * v * cg * sc / cL / r / r + p_chi * v * cg * tL * sc / r / r + p_chi * v * c_max * u * sb / cg / hr;
            Xdot[7] = -(p_L * cg * cc / r + p_l * cg * sc / cL / r + p_h * sg
                        + p_gamma * (c_max * u * cb + g / v / v * cg - ft * sa * cb / mass / v / v + cg / r)
                        + p_chi * (c_max * u * sb / cg - ft * sa * sb / cg / mass / v / v + cg * tL * sc / r)
                        - p_v * 2 * (d + eta * c_max * u * u) * v);
            Xdot[8] = v * (p_L * sg * cc / r + p_l * sg * sc / cL / r - p_h * cg)
                      - g * (p_gamma / v * sg - p_v * cg)
                      + p_gamma * v * sg / r + p_chi * v * sg * tL * sc / r
                      - p_chi * (v * c_max * u * sb + ft * sa * sb / mass / v) * sg / cg / cg;
            Xdot[9] = v * (p_L * cg * sc / r - p_l * cg * cc / cL / r - p_chi * cg * tL * cc / r);
            Xdot[10] = -p_l * v * cg * sc * sL / cL / cL / r - p_chi * v * cg * (1 + tL * tL) * sc / r;
            Xdot[11] = 0.0;
        } else {
            const double st = s1, ct = c1, sp = s2, cp = c2, p_theta = p_a1, p_phi = p_a2;
            const double tt = tan(X[2]);
            Xdot[0] = -v * ct * cp;
            Xdot[1] = -(d + eta * c_max * u * u) * v * v + g * ct * cp + ft * ca / mass;
            Xdot[2] = v * c_max * u * cb + v * st * (cp + sp * tL) / r
                      + (ft * sa * cb / (mass * v) - g * st * cp / v);
            Xdot[3] = -v * c_max * u * sb / ct
                      + v * ct * (sp + tt * tt * (sp - tL * cp)) / r
                      - (ft * sa * sb / (mass * v * ct) + g * sp / (v * ct));
            Xdot[4] = v * ct * sp / r;
            Xdot[5] = v * st / (r * cL);
            Xdot[6] = -p_v / hr * (d + eta * c_max * u * u) * v * v - 2 * g / r * (p_theta * st * cp / v + p_phi * sp / ct / v - p_v * ct * cp)
                      + p_L * v * ct * sp / r / r + v * p_theta * st * (cp + sp * tL) / r / r + p_theta * v * c_max * u * cb / hr
                      + p_l * v * st / cL / r / r + v * p_phi * ct * (sp + tt * tt * (sp - tL * cp)) / r / r - p_phi * v * c_max * u * sb / ct / hr;
            Xdot[7] = -(p_L * ct * sp / r + p_l * st / (r * cL) - p_h * ct * cp
                        + p_theta * (c_max * u * cb + g / v / v * st * cp - ft * sa * cb / mass / v / v + st * (cp + sp * tL) / r)
                        + p_phi * (-c_max * u * sb / ct + g / v / v * sp / ct + ft * sa * sb / ct / mass / v / v + ct * (sp + tt * tt * (sp - tL * cp)) / r)
                        - p_v * 2 * (d + eta * c_max * u * u) * v);
            Xdot[8] = -v * (-p_L * st * sp / r + p_l * ct / (r * cL) + p_h * st * cp)
                      - g * (-p_theta * ct * cp / v - p_phi * sp * tt / (v * ct) - p_v * st * cp)
                      - p_theta * v * ct * (cp + sp * tL) / r + p_phi * v * st * (sp + tt * tt * (sp - tL * cp)) / r
                      - p_phi * v * ct * (2 * tt * (1 + tt * tt) * (sp - tL * cp)) / r
                      - p_phi * (-v * c_max * u * sb - ft * sa * sb / mass / v) * tt / ct;
            Xdot[9] = -v * (p_h * ct * sp + p_L * ct * cp / r)
                      - g * (p_theta * st * sp / v - p_phi * cp / (v * ct) - p_v * ct * sp)
                      - p_theta * (v * st * (-sp + cp * tL) / r)
                      - p_phi * v * ct * (cp + tt * tt * (cp + tL * sp)) / r;
            Xdot[10] = -p_l * v * st * tL / cL / r - v * (1 + tL * tL) * (p_theta * st * sp - p_phi * ct * cp * tt * tt) / r;
            Xdot[11] = 0.0;
        }
    }

    __device__ static __forceinline__ void rhs(const ModelParams &P, double stage, double chart, double t,
                                              const double (&X)[S], double (&Xdot)[S])
    {
        if constexpr (FAST) rhs_fast(P, stage, chart, t, X, Xdot);
        else rhs_ref(P, stage, chart, t, X, Xdot);
    }

    // The same right-hand side, restructured for throughput.  Notation: i* = reciprocal of *, vr = v/r,
    // K = d + eta c_max u^2 (drag + induced drag), An = ft sin(alpha)/(m v) (thrust-normal turn rate),
    // Q = v c_max u + An (total turn rate magnitude).
    __device__ static __forceinline__ void rhs_fast(const ModelParams &P, double stage, double chart, double t,
                                                   const double (&X)[S], double (&Xdot)[S])
    {
        const bool chart1 = chart == 1.0;
        const double h = X[0], v = X[1], L = X[4];
        const double p_h = X[6], p_v = X[7], p_a1 = X[8], p_a2 = X[9], p_L = X[10], p_l = X[11];
        const double eta = P.p[ETA], alpha_max = P.p[ALPHA_MAX];
        // ComputeMass + the common temporaries (:292-308, :981-997)
        const double qm1 = P.p[Q] * P.p[MU_GFT];
        const double mass = P.p[EMPTY] + P.p[PROP] - qm1 * (stage == 1.0 ? t : P.p[PROP] * ifast::rcp(P.p[Q]));
        const double im = ifast::rcp(mass), ihr = ifast::rcp(P.p[HR]);
        const double dens = exp(-h * ihr) * (P.p[PROP] + P.p[EMPTY]) * im;
        const double c_max = P.p[C0] * dens, d = P.p[D0] * dens;
        const double r = h + P.p[REARTH];
        const double ir = ifast::rcp(r), iv = ifast::rcp(v);
        const double g = P.p[MU0] * ir * ir * P.p[MU_GFT];
        const double ft = P.p[VE] * (stage * qm1);
        double s1, c1, s2, c2, sL, cL;
        ifast::sincos(X[2], s1, c1);
        ifast::sincos(X[3], s2, c2);
        ifast::sincos(L, sL, cL);
        const double ic1 = ifast::rcp(c1), icL = ifast::rcp(cL);
        const double tL = sL * icL;
        // control (:338-385 / :506-552): cos/sin(beta) straight from the atan2 arguments
        const double by = chart1 ? p_a2 : -p_a2, bx = p_a1 * c1;
        const double b2 = bx * bx + by * by;
        double cb = 1.0, sb = 0.0;                          // atan2(0, 0) = 0
        if (b2 > 0) { const double ib = ifast::rsqrt(b2); cb = bx * ib; sb = by * ib; }
        const double W = v * c_max + ft * alpha_max * im * iv;
        const double num = p_a1 * cb * W + by * sb * ic1 * W;      // chart 2: - p_phi (...) = by (...)
        const double den = p_v * (2 * eta * c_max * v * v + ft * alpha_max * alpha_max * im) - P.p[MUC];
        double u = num / den;
        if (fabs(u) > P.p[UMAX]) u = copysign(P.p[UMAX], u);
        double sa, ca;
        ifast::sincos(alpha_max * u, sa, ca);

        const double K = d + eta * c_max * u * u;
        const double vr = v * ir;
        const double An = ft * sa * im * iv;
        const double Q = v * c_max * u + An;
        const double Kv2 = K * v * v;
        if (chart1) {
            const double sg = s1, cg = c1, sc = s2, cc = c2, p_gamma = p_a1, p_chi = p_a2, icg = ic1;
            const double vrcg = vr * cg;
            const double lat = p_L * cc + p_l * sc * icL;             // (p_L cos chi + p_l sin chi / cos L)
            Xdot[0] = v * sg;
            Xdot[1] = -Kv2 - g * sg + ft * ca * im;
            Xdot[2] = Q * cb - g * iv * cg + vrcg;
            Xdot[3] = Q * sb * icg + vrcg * tL * sc;
            Xdot[4] = vrcg * cc;
            Xdot[5] = vrcg * sc * icL;
            Xdot[6] = -p_v * ihr * Kv2 - 2 * g * ir * (p_gamma * iv * cg + p_v * sg)
                      + ir * (vrcg * (lat + p_gamma + p_chi * tL * sc))
                      + ihr * v * c_max * u * (p_gamma * cb + p_chi * sb * icg);
            Xdot[7] = -(ir * cg * lat + p_h * sg
                        + p_gamma * (c_max * u * cb + g * iv * iv * cg - An * iv * cb + cg * ir)
                        + p_chi * ((c_max * u - An * iv) * sb * icg + cg * tL * sc * ir)
                        - 2 * p_v * K * v);
            Xdot[8] = vr * sg * (lat + p_gamma + p_chi * tL * sc) - v * p_h * cg
                      - g * (p_gamma * iv * sg - p_v * cg)
                      - p_chi * Q * sb * sg * icg * icg;
            Xdot[9] = vrcg * (p_L * sc - p_l * cc * icL - p_chi * tL * cc);
            Xdot[10] = -vrcg * sc * (p_l * sL * icL * icL + p_chi * (1 + tL * tL));
            Xdot[11] = 0.0;
        } else {
            const double st = s1, ct = c1, sp = s2, cp = c2, p_theta = p_a1, p_phi = p_a2, ict = ic1;
            const double tt = st * ict, tt2 = tt * tt;
            const double A = cp + sp * tL;                            // cos phi + sin phi tan L
            const double B = sp + tt2 * (sp - tL * cp);               // sin phi + tan^2 theta (sin phi - tan L cos phi)
            const double givt = g * iv;
            Xdot[0] = -v * ct * cp;
            Xdot[1] = -Kv2 + g * ct * cp + ft * ca * im;
            Xdot[2] = Q * cb + vr * st * A - givt * st * cp;
            Xdot[3] = -Q * sb * ict + vr * ct * B - givt * sp * ict;
            Xdot[4] = vr * ct * sp;
            Xdot[5] = vr * st * icL;
            Xdot[6] = -p_v * ihr * Kv2 - 2 * g * ir * (iv * (p_theta * st * cp + p_phi * sp * ict) - p_v * ct * cp)
                      + ir * vr * (p_L * ct * sp + p_theta * st * A + p_l * st * icL + p_phi * ct * B)
                      + ihr * v * c_max * u * (p_theta * cb - p_phi * sb * ict);
            Xdot[7] = -(ir * (p_L * ct * sp + p_l * st * icL) - p_h * ct * cp
                        + p_theta * (c_max * u * cb + givt * iv * st * cp - An * iv * cb + st * A * ir)
                        + p_phi * ((-c_max * u + An * iv) * sb * ict + givt * iv * sp * ict + ct * B * ir)
                        - 2 * p_v * K * v);
            Xdot[8] = -v * (-p_L * st * sp * ir + p_l * ct * ir * icL + p_h * st * cp)
                      + givt * (p_theta * ct * cp + p_phi * sp * tt * ict) + g * p_v * st * cp
                      - p_theta * vr * ct * A + p_phi * vr * st * B
                      - p_phi * vr * ct * (2 * tt * (1 + tt2) * (sp - tL * cp))
                      + p_phi * Q * sb * tt * ict;
            Xdot[9] = -v * (p_h * ct * sp + p_L * ct * cp * ir)
                      - givt * (p_theta * st * sp - p_phi * cp * ict) + g * p_v * ct * sp
                      - p_theta * vr * st * (-sp + cp * tL)
                      - p_phi * vr * ct * (cp + tt2 * (cp + tL * sp));
            Xdot[10] = -p_l * vr * st * tL * icL - vr * (1 + tL * tL) * (p_theta * st * sp - p_phi * ct * cp * tt2);
            Xdot[11] = 0.0;
        }
    }

    // Hamiltonian_1 (:388-437) / Hamiltonian_2 (:555-604)
    __device__ static __forceinline__ double hamiltonian(const ModelParams &P, double stage, double chart, double t,
                                                        const double (&X)[S])
    {
        const Com c = common(P, stage, t, X[0]);
        const double v = X[1], L = X[4];
        const double p_h = X[6], p_v = X[7], p_a1 = X[8], p_a2 = X[9], p_L = X[10], p_l = X[11];
        const double mass = c.mass, c_max = c.c_max, d = c.d, r = c.r, g = c.g, ft = c.ft, eta = P.p[ETA];
        double s1, c1, s2, c2;
        sincos(X[2], &s1, &c1);
        sincos(X[3], &s2, &c2);
        const double cL = cos(L), tL = tan(L);
        double u, beta, sb, cb;
        control_k(P, c, chart == 1.0, v, c1, p_v, p_a1, p_a2, u, beta, sb, cb);
        const double alpha = P.p[ALPHA_MAX] * u;
        double sa, ca;
        sincos(alpha, &sa, &ca);
        if (chart == 1.0) {
            const double sg = s1, cg = c1, sc = s2, cc = c2, p_gamma = p_a1, p_chi = p_a2;
            return p_L * v * cg * cc / r
                   + p_l * v * cg * sc / cL / r
                   + p_h * v * sg
                   + p_gamma * (v * c_max * u * cb - g / v * cg + ft * sa * cb / mass / v + v * cg / r)
                   + p_chi * (v * c_max * u * sb / cg + ft * sa * sb / cg / mass / v + v * cg * tL * sc / r)
                   - p_v * ((d + eta * c_max * u * u) * v * v + g * sg - ft * ca / mass)
                   + P.p[MUC] * u * u / 2;
        }
        const double st = s1, ct = c1, sp = s2, cp = c2, p_theta = p_a1, p_phi = p_a2;
        const double tt = tan(X[2]);
        return p_L * v * ct * sp / r
               + p_l * v * st / (r * cL)
               - p_h * v * ct * cp
               + p_theta * (v * c_max * u * cb + v * st * (cp + sp * tL) / r + (ft * sa * cb / (mass * v) - g * st * cp / v))
               + p_phi * (-v * c_max * u * sb / ct + v * ct * (sp + tt * tt * (sp - tL * cp)) / r - (ft * sa * sb / (mass * v * ct) + g * sp / (v * ct)))
               - p_v * ((d + eta * c_max * u * u) * v * v - g * ct * cp - ft * ca / mass)
               + P.p[MUC] * u * u / 2;
    }

    // model::SwitchingTimesFunction default (model.hpp:299-328): H(t, X-) - H(t, X+)
    __device__ static __forceinline__ double switching_fn(const ModelParams &P, double stage, double chart, double t,
                                                         const double (&X)[S], const double (&Xp)[S])
    {
        return hamiltonian(P, stage, chart, t, X) - hamiltonian(P, stage, chart, t, Xp);
    }

    // ---- chart change (ConversionState12 :607-730, ConversionState21 :733-843) -------------------
    // Jacobians of (Earth-frame position, velocity) w.r.t. (h, L, l, angle1, angle2, v), stored as the reference
    // fills them: J[row][col] = Jac(row, col).
    __device__ static __forceinline__ void jac_chart1(double r, double v, double L, double l, double gamma, double chi, double (&J)[6][6])
    {
        double sL, cL, sl, cl, sg, cg, sc, cc;
        sincos(L, &sL, &cL); sincos(l, &sl, &cl); sincos(gamma, &sg, &cg); sincos(chi, &sc, &cc);
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) J[i][j] = 0.0;
        J[0][0] = cL * cl; J[1][0] = -r * sL * cl; J[2][0] = -r * cL * sl;
        J[0][1] = cL * sl; J[1][1] = -r * sL * sl; J[2][1] = r * cL * cl;
        J[0][2] = sL;      J[1][2] = r * cL;
        J[1][3] = (-cL * cl * cg * cc - sL * cl * sg) * v;
        J[2][3] = (sL * sl * cg * cc - cl * cg * sc - cL * sl * sg) * v;
        J[3][3] = (sL * cl * sg * cc + sl * sg * sc + cL * cl * cg) * v;
        J[4][3] = (sL * cl * cg * sc - sl * cg * cc) * v;
        J[5][3] = (-sL * cl * cg * cc - sl * cg * sc + cL * cl * sg) * v;
        J[1][4] = (-cL * sl * cg * cc - sL * sl * sg) * v;
        J[2][4] = (-sL * cl * cg * cc - sl * cg * sc + cL * cl * sg) * v;
        J[3][4] = (sL * sl * sg * cc - cl * sg * sc + cL * sl * cg) * v;
        J[4][4] = (sL * sl * cg * sc + cl * cg * cc) * v;
        J[5][4] = (-sL * sl * cg * cc + cl * cg * sc + cL * sl * sg) * v;
        J[1][5] = (-sL * cg * cc + cL * sg) * v;
        J[3][5] = (-cL * sg * cc + sL * cg) * v;
        J[4][5] = -cL * cg * sc * v;
        J[5][5] = (cL * cg * cc + sL * sg) * v;
    }

    __device__ static __forceinline__ void jac_chart2(double r, double v, double L, double l, double theta, double phi, double (&J)[6][6])
    {
        double sL, cL, sl, cl, st, ct, sp, cp;
        sincos(L, &sL, &cL); sincos(l, &sl, &cl); sincos(theta, &st, &ct); sincos(phi, &sp, &cp);
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) J[i][j] = 0.0;
        J[0][0] = cL * cl; J[1][0] = -r * sL * cl; J[2][0] = -r * cL * sl;
        J[0][1] = cL * sl; J[1][1] = -r * sL * sl; J[2][1] = r * cL * cl;
        J[0][2] = sL;      J[1][2] = r * cL;
        J[1][3] = (-cL * cl * ct * sp + sL * cl * ct * cp) * v;
        J[2][3] = (sL * sl * ct * sp - cl * st + cL * sl * ct * cp) * v;
        J[3][3] = (sL * cl * st * sp - sl * ct + cL * cl * st * cp) * v;
        J[4][3] = (-sL * cl * ct * cp + cL * cl * ct * sp) * v;
        J[5][3] = (-sL * cl * ct * sp - sl * st - cL * cl * ct * cp) * v;
        J[1][4] = (-cL * sl * ct * sp + sL * sl * ct * cp) * v;
        J[2][4] = (-sL * cl * ct * sp - sl * st - cL * cl * ct * cp) * v;
        J[3][4] = (sL * sl * st * sp + cl * ct + cL * sl * st * cp) * v;
        J[4][4] = (-sL * sl * ct * cp + cL * sl * ct * sp) * v;
        J[5][4] = (-sL * sl * ct * sp + cl * st - cL * sl * ct * cp) * v;
        J[1][5] = (-sL * ct * sp - cL * ct * cp) * v;
        J[3][5] = (-cL * st * sp + sL * st * cp) * v;
        J[4][5] = (cL * ct * cp + sL * ct * sp) * v;
        J[5][5] = (cL * ct * sp - sL * ct * cp) * v;
    }

    // y <- A^-1 y, LU with partial pivoting (first largest |entry| of the column).  Row exchanges are selects over
    // statically indexed registers: no dynamically indexed array, hence no scratch memory.
    __device__ static __forceinline__ void lu6_solve(double (&A)[6][6], double (&y)[6])
    {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            int piv = k;
            double big = fabs(A[k][k]);
#pragma unroll
            for (int i = k + 1; i < 6; i++)
                if (fabs(A[i][k]) > big) { big = fabs(A[i][k]); piv = i; }
#pragma unroll
            for (int i = k + 1; i < 6; i++) {
                const bool ex = piv == i;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const double a = A[k][j], b = A[i][j];
                    A[k][j] = ex ? b : a;
                    A[i][j] = ex ? a : b;
                }
                const double a = y[k], b = y[i];
                y[k] = ex ? b : a;
                y[i] = ex ? a : b;
            }
            if (big != 0) {
#pragma unroll
                for (int i = k + 1; i < 6; i++) A[i][k] /= A[k][k];
            }
#pragma unroll
            for (int i = k + 1; i < 6; i++)
#pragma unroll
                for (int j = k + 1; j < 6; j++) A[i][j] -= A[i][k] * A[k][j];
        }
#pragma unroll
        for (int i = 1; i < 6; i++)
#pragma unroll
            for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
#pragma unroll
        for (int i = 5; i >= 0; i--) {
#pragma unroll
            for (int j = i + 1; j < 6; j++) y[i] -= A[i][j] * y[j];
            y[i] /= A[i][i];
        }
    }

    // X (chart `from`) -> X in the other chart; both directions share the solve and the product
    __device__ static __forceinline__ void change_chart(const ModelParams &P, bool from1, double (&X)[S])
    {
        constexpr double kPi = 3.14159265358979323846;
        const double eps = 1e-18;
        const double v = X[1], a1 = X[2], a2 = X[3], L = X[4], l = X[5];
        const double r = X[0] + P.p[REARTH];
        double n1, n2;
        double s1, c1, s2, c2;
        sincos(a1, &s1, &c1);
        sincos(a2, &s2, &c2);
        if (from1) {
            // gamma, chi -> theta, phi (:618-639)
            if (a1 == kPi / 2.0) { n1 = 0; n2 = -kPi; }
            else if (a1 == -kPi / 2.0) { n1 = 0; n2 = 0; }
            else {
                n1 = acos(sqrt(s1 * s1 + c1 * c1 * c2 * c2));
                if (c1 * s2 < 0) n1 = -n1;
                const double cn = cos(n1);
                const double sinPhi = c1 * c2 / cn;
                if (fabs(sinPhi) < eps && s1 / cn < 0) n2 = 0;
                else if (fabs(sinPhi) < eps && s1 / cn > 0) n2 = -kPi;
                else if (sinPhi > 0) n2 = acos(-s1 / cn);
                else n2 = -acos(-s1 / cn);
            }
        } else {
            // theta, phi -> gamma, chi (:746-771)
            if (a1 == kPi / 2.0) { n1 = 0; n2 = kPi / 2.0; }
            else if (a1 == -kPi / 2.0) { n1 = 0; n2 = -kPi / 2.0; }
            else {
                n1 = acos(sqrt(s1 * s1 + c1 * c1 * s2 * s2));
                if (c1 * c2 > 0) n1 = -n1;
                const double cn = cos(n1);
                const double sinChi = s1 / cn;
                if (fabs(sinChi) < eps && s2 * c1 / cn > 0) n2 = 0;
                else if (fabs(sinChi) < eps && s2 * c1 / cn < 0) n2 = -kPi;
                else if (sinChi > 0) n2 = acos(s2 * c1 / cn);
                else n2 = -acos(s2 * c1 / cn);
            }
        }
        double Jf[6][6], Jt[6][6];
        if (from1) { jac_chart1(r, v, L, l, a1, a2, Jf); jac_chart2(r, v, L, l, n1, n2, Jt); }
        else { jac_chart2(r, v, L, l, a1, a2, Jf); jac_chart1(r, v, L, l, n1, n2, Jt); }
        // costates in the reference's component order (h, L, l, angle1, angle2, v) (:710-727, :824-841)
        double p[6] = {X[6], X[10], X[11], X[8], X[9], X[7]};
        lu6_solve(Jf, p);
        double q[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double acc = Jt[i][0] * p[0];
#pragma unroll
            for (int j = 1; j < 6; j++) acc += Jt[i][j] * p[j];
            q[i] = acc;
        }
        X[2] = n1; X[3] = n2;
        X[6] = q[0]; X[7] = q[5]; X[8] = q[3]; X[9] = q[4]; X[10] = q[1]; X[11] = q[2];
    }

    // SetChart (:953-978)
    __device__ static __forceinline__ bool set_chart(const ModelParams &P, double &chart, double (&X)[S])
    {
        if (fabs(cos(X[2])) >= P.p[CHART_LIMIT]) return false;
        change_chart(P, chart == 1.0, X);
        chart = chart == 1.0 ? 2.0 : 1.0;
        return true;
    }

    // One RK4 step in the reference's association order (odeTools.cpp:78-87, the function-pointer overload
    // interceptor::ModelInt calls), with the four stages as ONE loop body: the right-hand side is ~10 transcendental
    // calls, and four inlined copies of it would only inflate the kernel.
    __device__ static __forceinline__ void rk4_step(const ModelParams &P, double stage, double chart, double t, double (&X)[S], double step)
    {
        double F1[S], Fs[S], F[S], Y[S];
        const double h2 = step / 2.0;
#pragma unroll
        for (int i = 0; i < S; i++) { Y[i] = X[i]; F1[i] = 0; Fs[i] = 0; }
#pragma unroll 1
        for (int st = 0; st < 4; st++) {
            const double tt = st == 0 ? t : (st == 3 ? t + step : t + step / 2.0);
            rhs(P, stage, chart, tt, Y, F);
            const double cnext = st == 2 ? step : h2;
#pragma unroll
            for (int i = 0; i < S; i++) {
                if (st == 0) F1[i] = F[i];
                else if (st == 1) Fs[i] = F[i];
                else if (st == 2) Fs[i] = Fs[i] + F[i];
                Y[i] = X[i] + cnext * F[i];
            }
        }
        const double h6 = step / 6.0;
#pragma unroll
        for (int i = 0; i < S; i++) X[i] = X[i] + h6 * (F1[i] + (F[i] + 2.0 * Fs[i]));
    }

    // interceptor::ComputeTraj (:162-218) over interceptor::ModelInt (:101-128).  INTEG == 1 replaces the
    // stepNbr fixed steps of a stage by adaptive Dormand-Prince steps (the chart is still re-chosen before
    // every step) -- an extension: the reference's interceptor never reaches odeTools::integrate.
    // obs(t, X, stage, chart) sees the rows the reference traces: stage start and after every step.
    template <int INTEG, class Obs>
    __device__ static __forceinline__ void compute_traj(const ModelParams &P, double &stage, double &chart,
                                                       double t0, double tf, double (&X)[S], Obs &&obs)
    {
        chart = 1.0;
        const double t1 = P.p[PROP] / P.p[Q];
        const bool two = t0 < t1 && tf > t1;
        stage = t0 < t1 ? 1.0 : 0.0;
        const int phases = two ? 2 : 1;
#pragma unroll 1
        for (int ph = 0; ph < phases; ph++) {
            const double ta = ph == 0 ? t0 : t1;
            const double tb = (two && ph == 0) ? t1 : tf;
            if (ph == 1) stage = 0.0;
            obs(ta, X, stage, chart);
            if constexpr (INTEG == 1) {
                // the observer sees the state after every accepted step (the last one ends at tb)
                Lane<InterceptorT>::integrate_dopri5(P, stage, chart, ta, tb, X,
                                                         [&](double, double (&Xc)[S]) { return set_chart(P, chart, Xc); },
                                                         [&](double ts, const double (&Xs)[S]) { obs(ts, Xs, stage, chart); });
            } else {
                const double dt = (tb - ta) / P.step_nbr;
                double t = ta;
#pragma unroll 1
                for (int i = 0; i < P.step_nbr; i++) {
                    set_chart(P, chart, X);
                    rk4_step(P, stage, chart, t, X, dt);
                    t += dt;
                    obs(t, X, stage, chart);
                }
            }
        }
        if (chart == 2.0) change_chart(P, false, X);      // handed back in chart 1; the flag keeps its value (:213-215)
    }

    // FinalFunction / FinalHFunction overrides (:221-272): altitude row scaled by hr, free final velocity
    // -> p_v + muV, heading row replaced by p_chi when the target flight-path angle is vertical
    __device__ static __forceinline__ double final_row(const ModelParams &P, int j, int mode, const double (&X)[S], const double *xd)
    {
        if (mode == 1) return j == 1 ? X[j + D] + P.p[MUV] : X[j + D];
        double f = X[j] - xd[j];
        if (j == 0) f = f / P.p[HR];
        if (j == 3 && fabs(cos(xd[2])) < 1e-5) f = X[j + D];
        return f;
    }
    // the free-final-time row is H + muT (:270)
    __device__ static __forceinline__ double final_h_offset(const ModelParams &P) { return P.p[MUT]; }
};

using InterceptorModel = InterceptorT<false>;      // reference operation order (kernels_interceptor.hip)
using InterceptorFast = InterceptorT<true>;        // throughput flavour (kernels_interceptor_fast.hip)

}  // namespace socp
