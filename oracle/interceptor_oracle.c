/*
 * interceptor_oracle.c -- CPU ORACLE for the interceptor model (test infrastructure, NOT product code).
 *
 * PARITY UNPINNED: the reference's interceptor.cpp includes "Eigen/Dense" (interceptor.cpp:15), which
 * this image lacks, so that translation unit cannot be compiled here and the reference ships no
 * output of it (no golden trace, no asserting test).  This file is therefore a
 * restatement from the source text alone.  What stands in for a pin (tests/test_oracle_interceptor.py):
 *   - the costate equations are checked against -dH/dx of the restated Hamiltonian by central
 *     differences in both charts (a transcription slip in either breaks it);
 *   - chart 1 -> 2 -> 1 is the identity and H is invariant under the chart change;
 *   - the 6x6 solve is checked against numpy.
 * The two Eigen calls (Jac.lu().solve(p), Jac*p; interceptor.cpp:715-716, :829-830) are restated as
 * textbook partial-pivot LU and row-times-vector sums; Eigen's internal summation order is not
 * reproducible without Eigen, so chart switches agree with the reference only to rounding.
 *
 * Citations are file:line into /root/reference/src/models/interceptor/interceptor.cpp.
 * Arithmetic order follows the cited expressions term by term (trigonometric values are named once
 * and reused -- the same values the reference recomputes).
 */
#include "socp_oracle.h"

#include <math.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* interceptor.cpp:34-66: constructor defaults; ModelInt steps with its own data->stepNbr = 50 */
void orc_interceptor_init(orc_model *m)
{
    memset(m, 0, sizeof(*m));
    m->model_id = ORC_MODEL_INTERCEPTOR;
    m->dim = 6;
    m->step_nbr = 50;
    m->p[IP_C0] = 0.00075;  m->p[IP_HR] = 7500;     m->p[IP_D0] = 0.00005; m->p[IP_ETA] = 0.442;
    m->p[IP_PROP] = 200;    m->p[IP_EMPTY] = 200;   m->p[IP_Q] = 10;       m->p[IP_VE] = 1500;
    m->p[IP_ALPHA_MAX] = M_PI / 6; m->p[IP_UMAX] = 1; m->p[IP_AMAX] = 1500; m->p[IP_MU_GFT] = 1;
    m->p[IP_MUT] = 0;       m->p[IP_MUV] = 1;       m->p[IP_MUC] = 0;
    m->p[IP_REARTH] = 6378145; m->p[IP_MU0] = 3.986e14; m->p[IP_CHART_LIMIT] = 0.1;
    m->chart = 1;
    m->stage = 0;
}

/* the temporaries every Model_k / Control_k / Hamiltonian_k starts with (:292-308 and its repeats) */
typedef struct {
    double mass, c_max, d, r, g, ft, eta, hr, alpha_max, u_max, muC;
} icom;

/* interceptor.cpp:981-997 */
static double compute_mass(const orc_model *m, double t)
{
    const double *p = m->p;
    double qm = p[IP_Q] * p[IP_MU_GFT];
    double t1 = p[IP_PROP] / p[IP_Q];
    if (m->stage == 1) return p[IP_EMPTY] + p[IP_PROP] - qm * t;
    return p[IP_EMPTY] + p[IP_PROP] - qm * t1;
}

static void common(const orc_model *m, double t, const double *X, icom *c)
{
    const double *p = m->p;
    double h = X[0];
    double qm = m->stage * p[IP_Q] * p[IP_MU_GFT];
    c->mass = compute_mass(m, t);
    c->c_max = p[IP_C0] * exp(-h / p[IP_HR]) * (p[IP_PROP] + p[IP_EMPTY]) / c->mass;
    c->d = p[IP_D0] * exp(-h / p[IP_HR]) * (p[IP_PROP] + p[IP_EMPTY]) / c->mass;
    c->r = h + p[IP_REARTH];
    c->g = p[IP_MU0] / c->r / c->r * p[IP_MU_GFT];
    c->ft = p[IP_VE] * qm;
    c->eta = p[IP_ETA];
    c->hr = p[IP_HR];
    c->alpha_max = p[IP_ALPHA_MAX];
    c->u_max = p[IP_UMAX];
    c->muC = p[IP_MUC];
}

/* interceptor.cpp:338-385 (chart 1) and :506-552 (chart 2): beta, then the unsaturated u, then clipping */
static void control_1(const orc_model *m, double t, const double *X, double *uc)
{
    icom c; common(m, t, X, &c);
    double v = X[1], gamma = X[2], p_v = X[7], p_gamma = X[8], p_chi = X[9];
    double mass = c.mass, c_max = c.c_max, ft = c.ft, alpha_max = c.alpha_max, eta = c.eta;
    double cg = cos(gamma);
    double beta = atan2(p_chi, p_gamma * cg);
    double cb = cos(beta), sb = sin(beta);
    double u = (p_gamma * (v * c_max * cb + ft * cb * alpha_max / mass / v)
                + p_chi * (v * c_max * sb / cg + ft * sb / cg * alpha_max / mass / v))
               / (p_v * (2 * eta * c_max * v * v + ft * alpha_max * alpha_max / mass) - c.muC);
    if (fabs(u) > c.u_max) u = c.u_max * u / fabs(u);
    uc[0] = u;
    uc[1] = beta;
}

static void control_2(const orc_model *m, double t, const double *X, double *uc)
{
    icom c; common(m, t, X, &c);
    double v = X[1], theta = X[2], p_v = X[7], p_theta = X[8], p_phi = X[9];
    double mass = c.mass, c_max = c.c_max, ft = c.ft, alpha_max = c.alpha_max, eta = c.eta;
    double ct = cos(theta);
    double beta = atan2(-p_phi, p_theta * ct);
    double cb = cos(beta), sb = sin(beta);
    double u = (p_theta * (v * c_max * cb + ft * cb * alpha_max / mass / v)
                - p_phi * (v * c_max * sb / ct + ft * sb / ct * alpha_max / mass / v))
               / (p_v * (2 * eta * c_max * v * v + ft * alpha_max * alpha_max / mass) - c.muC);
    if (fabs(u) > c.u_max) u = c.u_max * u / fabs(u);
    uc[0] = u;
    uc[1] = beta;
}

/* interceptor.cpp:275-335: NED-frame chart (h, v, gamma, chi, L, l) */
static void model_1(const orc_model *m, double t, const double *X, double *Xdot)
{
    icom c; common(m, t, X, &c);
    double v = X[1], gamma = X[2], chi = X[3], L = X[4];
    double p_h = X[6], p_v = X[7], p_gamma = X[8], p_chi = X[9], p_L = X[10], p_l = X[11];
    double mass = c.mass, c_max = c.c_max, d = c.d, r = c.r, g = c.g, ft = c.ft, eta = c.eta, hr = c.hr;
    double uc[2];
    control_1(m, t, X, uc);
    double u = uc[0], beta = uc[1];
    double alpha = c.alpha_max * u;
    double sg = sin(gamma), cg = cos(gamma), sc = sin(chi), cc = cos(chi), sL = sin(L), cL = cos(L), tL = tan(L);
    double sb = sin(beta), cb = cos(beta), sa = sin(alpha), ca = cos(alpha);

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
}

/* interceptor.cpp:388-437; uc = (u, beta) as Control_1 returns them (the `_at` form lets a test hold the control
 * fixed: the thrust terms use sin/cos(alpha) while the control law is their small-angle optimum, so
 * dH/du != 0 in the powered stage and only the PARTIAL derivatives of H give the costate equations) */
static double hamiltonian_1_at(const orc_model *m, double t, const double *X, const double *uc)
{
    icom c; common(m, t, X, &c);
    double v = X[1], gamma = X[2], chi = X[3], L = X[4];
    double p_h = X[6], p_v = X[7], p_gamma = X[8], p_chi = X[9], p_L = X[10], p_l = X[11];
    double mass = c.mass, c_max = c.c_max, d = c.d, r = c.r, g = c.g, ft = c.ft, eta = c.eta;
    double u = uc[0], beta = uc[1];
    double alpha = c.alpha_max * u;
    double sg = sin(gamma), cg = cos(gamma), sc = sin(chi), cc = cos(chi), cL = cos(L), tL = tan(L);
    double sb = sin(beta), cb = cos(beta), sa = sin(alpha), ca = cos(alpha);

    return p_L * v * cg * cc / r
           + p_l * v * cg * sc / cL / r
           + p_h * v * sg
           + p_gamma * (v * c_max * u * cb - g / v * cg + ft * sa * cb / mass / v + v * cg / r)
           + p_chi * (v * c_max * u * sb / cg + ft * sa * sb / cg / mass / v + v * cg * tL * sc / r)
           - p_v * ((d + eta * c_max * u * u) * v * v + g * sg - ft * ca / mass)
           + c.muC * u * u / 2;
}

/* interceptor.cpp:440-503: second chart (h, v, theta, phi, L, l), regular where cos(gamma) -> 0 */
static void model_2(const orc_model *m, double t, const double *X, double *Xdot)
{
    icom c; common(m, t, X, &c);
    double v = X[1], theta = X[2], phi = X[3], L = X[4];
    double p_h = X[6], p_v = X[7], p_theta = X[8], p_phi = X[9], p_L = X[10], p_l = X[11];
    double mass = c.mass, c_max = c.c_max, d = c.d, r = c.r, g = c.g, ft = c.ft, eta = c.eta, hr = c.hr;
    double uc[2];
    control_2(m, t, X, uc);
    double u = uc[0], beta = uc[1];
    double alpha = c.alpha_max * u;
    double st = sin(theta), ct = cos(theta), tt = tan(theta), sp = sin(phi), cp = cos(phi);
    double cL = cos(L), tL = tan(L);
    double sb = sin(beta), cb = cos(beta), sa = sin(alpha), ca = cos(alpha);

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

/* interceptor.cpp:555-604 */
static double hamiltonian_2_at(const orc_model *m, double t, const double *X, const double *uc)
{
    icom c; common(m, t, X, &c);
    double v = X[1], theta = X[2], phi = X[3], L = X[4];
    double p_h = X[6], p_v = X[7], p_theta = X[8], p_phi = X[9], p_L = X[10], p_l = X[11];
    double mass = c.mass, c_max = c.c_max, d = c.d, r = c.r, g = c.g, ft = c.ft, eta = c.eta;
    double u = uc[0], beta = uc[1];
    double alpha = c.alpha_max * u;
    double st = sin(theta), ct = cos(theta), tt = tan(theta), sp = sin(phi), cp = cos(phi);
    double cL = cos(L), tL = tan(L);
    double sb = sin(beta), cb = cos(beta), sa = sin(alpha), ca = cos(alpha);

    return p_L * v * ct * sp / r
           + p_l * v * st / (r * cL)
           - p_h * v * ct * cp
           + p_theta * (v * c_max * u * cb + v * st * (cp + sp * tL) / r + (ft * sa * cb / (mass * v) - g * st * cp / v))
           + p_phi * (-v * c_max * u * sb / ct + v * ct * (sp + tt * tt * (sp - tL * cp)) / r - (ft * sa * sb / (mass * v * ct) + g * sp / (v * ct)))
           - p_v * ((d + eta * c_max * u * u) * v * v - g * ct * cp - ft * ca / mass)
           + c.muC * u * u / 2;
}

/* interceptor.cpp:69-98: dispatch on the chart the state is currently expressed in */
void orc_interceptor_rhs(const orc_model *m, double t, const double *X, double *Xdot)
{
    if (m->chart == 1) model_1(m, t, X, Xdot); else model_2(m, t, X, Xdot);
}
void orc_interceptor_control(const orc_model *m, double t, const double *X, double *uc)
{
    if (m->chart == 1) control_1(m, t, X, uc); else control_2(m, t, X, uc);
}
double orc_interceptor_hamiltonian_at(const orc_model *m, double t, const double *X, const double *u_beta)
{
    return m->chart == 1 ? hamiltonian_1_at(m, t, X, u_beta) : hamiltonian_2_at(m, t, X, u_beta);
}
double orc_interceptor_hamiltonian(const orc_model *m, double t, const double *X)
{
    double uc[2];
    orc_interceptor_control(m, t, X, uc);
    return orc_interceptor_hamiltonian_at(m, t, X, uc);
}

/* ---- chart change ------------------------------------------------------------------------- */
/* d(position, velocity in the Earth frame)/d(h, L, l, angle1, angle2, v) transposed, as the
 * reference fills it: entry (row, col) below is Jac(row, col) of interceptor.cpp:659-682 (chart 1
 * angles gamma, chi) and :684-708 (chart 2 angles theta, phi). */
static void jac_chart1(double r, double v, double L, double l, double gamma, double chi, double J[6][6])
{
    double cL = cos(L), sL = sin(L), cl = cos(l), sl = sin(l), cg = cos(gamma), sg = sin(gamma), cc = cos(chi), sc = sin(chi);
    memset(J, 0, sizeof(double) * 36);
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

static void jac_chart2(double r, double v, double L, double l, double theta, double phi, double J[6][6])
{
    double cL = cos(L), sL = sin(L), cl = cos(l), sl = sin(l), ct = cos(theta), st = sin(theta), cp = cos(phi), sp = sin(phi);
    memset(J, 0, sizeof(double) * 36);
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

/* x = A^-1 b by LU with partial pivoting (first largest |entry| of the column), unit-lower forward and
 * upper backward substitution; y = B x row by row, sums left to right.  [ext] stands for Eigen's
 * PartialPivLU::solve and operator* (see header). */
void orc_lu6_solve(const double A_in[6][6], const double *b, double *x)
{
    double A[6][6], y[6];
    memcpy(A, A_in, sizeof(A));
    memcpy(y, b, sizeof(y));
    for (int k = 0; k < 6; k++) {
        int piv = k;
        double big = fabs(A[k][k]);
        for (int i = k + 1; i < 6; i++)
            if (fabs(A[i][k]) > big) { big = fabs(A[i][k]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < 6; j++) { double tmp = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = tmp; }
            double tmp = y[k]; y[k] = y[piv]; y[piv] = tmp;
        }
        if (big != 0)
            for (int i = k + 1; i < 6; i++) A[i][k] /= A[k][k];
        for (int i = k + 1; i < 6; i++)
            for (int j = k + 1; j < 6; j++) A[i][j] -= A[i][k] * A[k][j];
    }
    for (int i = 1; i < 6; i++)
        for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
    for (int i = 5; i >= 0; i--) {
        for (int j = i + 1; j < 6; j++) y[i] -= A[i][j] * y[j];
        y[i] /= A[i][i];
    }
    memcpy(x, y, sizeof(y));
}

static void matvec6(const double B[6][6], const double *x, double *y)
{
    for (int i = 0; i < 6; i++) {
        double acc = B[i][0] * x[0];
        for (int j = 1; j < 6; j++) acc += B[i][j] * x[j];
        y[i] = acc;
    }
}

/* costate transport p_to = J_to * J_from^-1 * p_from in the reference's component order
 * (h, L, l, angle1, angle2, v) <-> X[6], X[10], X[11], X[8], X[9], X[7] (:710-727, :824-841) */
static void transport_costate(const double Jfrom[6][6], const double Jto[6][6], const double *Xin, double *Xout)
{
    double p[6] = { Xin[6], Xin[10], Xin[11], Xin[8], Xin[9], Xin[7] }, tmp[6], q[6];
    orc_lu6_solve(Jfrom, p, tmp);
    matvec6(Jto, tmp, q);
    Xout[6] = q[0]; Xout[7] = q[5]; Xout[8] = q[3]; Xout[9] = q[4]; Xout[10] = q[1]; Xout[11] = q[2];
}

/* interceptor.cpp:607-730 */
void orc_interceptor_chart12(const orc_model *m, const double *X1, double *X2)
{
    const double eps = 1e-18;
    double v = X1[1], gamma = X1[2], chi = X1[3], L = X1[4], l = X1[5];
    double r = X1[0] + m->p[IP_REARTH];
    double out[12];
    memcpy(out, X1, sizeof(out));
    if (gamma == M_PI / 2.0) { out[2] = 0; out[3] = -M_PI; }
    else if (gamma == -M_PI / 2.0) { out[2] = 0; out[3] = 0; }
    else {
        double sg = sin(gamma), cg = cos(gamma), sc = sin(chi), cc = cos(chi);
        out[2] = acos(sqrt(sg * sg + cg * cg * cc * cc));
        if (cg * sc < 0) out[2] = -acos(sqrt(sg * sg + cg * cg * cc * cc));
        double c2 = cos(out[2]);
        double sinPhi = cg * cc / c2;
        if (fabs(sinPhi) < eps && sg / c2 < 0) out[3] = 0;
        else if (fabs(sinPhi) < eps && sg / c2 > 0) out[3] = -M_PI;
        else if (sinPhi > 0) out[3] = acos(-sg / c2);
        else out[3] = -acos(-sg / c2);
    }
    double J1[6][6], J2[6][6];
    jac_chart1(r, v, L, l, gamma, chi, J1);
    jac_chart2(r, v, L, l, out[2], out[3], J2);
    transport_costate(J1, J2, X1, out);
    memcpy(X2, out, sizeof(out));
}

/* interceptor.cpp:733-843 */
void orc_interceptor_chart21(const orc_model *m, const double *X2, double *X1)
{
    const double eps = 1e-18;
    double v = X2[1], theta = X2[2], phi = X2[3], L = X2[4], l = X2[5];
    double r = X2[0] + m->p[IP_REARTH];
    double out[12];
    memcpy(out, X2, sizeof(out));
    if (theta == M_PI / 2.0) { out[2] = 0; out[3] = M_PI / 2.0; }
    else if (theta == -M_PI / 2.0) { out[2] = 0; out[3] = -M_PI / 2.0; }
    else {
        double st = sin(theta), ct = cos(theta), sp = sin(phi), cp = cos(phi);
        out[2] = acos(sqrt(st * st + ct * ct * sp * sp));
        if (ct * cp > 0) out[2] = -acos(sqrt(st * st + ct * ct * sp * sp));
        double c1 = cos(out[2]);
        double sinChi = st / c1;
        if (fabs(sinChi) < eps && sp * ct / c1 > 0) out[3] = 0;
        else if (fabs(sinChi) < eps && sp * ct / c1 < 0) out[3] = -M_PI;
        else if (sinChi > 0) out[3] = acos(sp * ct / c1);
        else out[3] = -acos(sp * ct / c1);
    }
    double J1[6][6], J2[6][6];
    jac_chart1(r, v, L, l, out[2], out[3], J1);
    jac_chart2(r, v, L, l, theta, phi, J2);
    transport_costate(J2, J1, X2, out);
    memcpy(X1, out, sizeof(out));
}

/* interceptor.cpp:953-978: leave a chart when |cos(X[2])| drops below chartLimit */
static void set_chart(orc_model *m, double *X)
{
    if (fabs(cos(X[2])) >= m->p[IP_CHART_LIMIT]) return;
    if (m->chart == 1) { orc_interceptor_chart12(m, X, X); m->chart = 2; }
    else { orc_interceptor_chart21(m, X, X); m->chart = 1; }
}

/* interceptor.cpp:101-128: stepNbr steps of size dt, chart chosen before every step; t by t += dt.
 * obs (may be NULL) sees (t, X, chart, stage) at the start and after each step -- the trace rows. */
static void model_int(orc_model *m, double t0, double *X, double tf, orc_interceptor_observer obs, void *ctx)
{
    double t = t0;
    double dt = (tf - t0) / m->step_nbr;
    if (obs) obs(ctx, t, X, m->chart, m->stage);
    for (int i = 0; i < m->step_nbr; i++) {
        set_chart(m, X);
        orc_rk4_step(m, t, X, dt, 0);
        t += dt;
        if (obs) obs(ctx, t, X, m->chart, m->stage);
    }
}

/* interceptor.cpp:162-218: powered stage until t1 = propellant_mass/q, coasting after; the state is
 * handed back in chart 1 but data->currentChart and data->stageMode keep their last values */
void orc_interceptor_compute_traj_obs(orc_model *m, double t0, const double *X0, double tf, double *Xf,
                                      orc_interceptor_observer obs, void *ctx)
{
    double X[12];
    memcpy(X, X0, sizeof(X));
    m->chart = 1;
    double t1 = m->p[IP_PROP] / m->p[IP_Q];
    if (t0 < t1) {
        m->stage = 1;
        if (tf > t1) {
            model_int(m, t0, X, t1, obs, ctx);
            m->stage = 0;
            model_int(m, t1, X, tf, obs, ctx);
        } else {
            model_int(m, t0, X, tf, obs, ctx);
        }
    } else {
        m->stage = 0;
        model_int(m, t0, X, tf, obs, ctx);
    }
    if (m->chart == 2) orc_interceptor_chart21(m, X, X);
    memcpy(Xf, X, sizeof(X));
}

static int set_chart_hook(orc_model *m, double t, double *X)
{
    (void)t;
    const int before = m->chart;
    set_chart(m, X);
    return m->chart != before;
}

void orc_interceptor_compute_traj_adaptive(orc_model *m, double t0, const double *X0, double tf, double tol, double *Xf)
{
    double X[12];
    memcpy(X, X0, sizeof(X));
    m->chart = 1;
    const double t1 = m->p[IP_PROP] / m->p[IP_Q];
    const int two = t0 < t1 && tf > t1;
    m->stage = t0 < t1 ? 1 : 0;
    for (int ph = 0; ph < (two ? 2 : 1); ph++) {
        const double ta = ph == 0 ? t0 : t1;
        const double tb = (two && ph == 0) ? t1 : tf;
        if (ph == 1) m->stage = 0;
        orc_integrate_dopri5_hook(m, X, ta, tb, (tb - ta) / m->step_nbr, tol, 0, set_chart_hook);
    }
    if (m->chart == 2) orc_interceptor_chart21(m, X, X);
    memcpy(Xf, X, sizeof(X));
}

void orc_interceptor_compute_traj(orc_model *m, double t0, const double *X0, double tf, double *Xf)
{
    orc_interceptor_compute_traj_obs(m, t0, X0, tf, Xf, 0, 0);
}

/* interceptor.cpp:221-245 (FinalFunction) and :248-272 (FinalHFunction without its H row) */
void orc_interceptor_final_rows(const orc_model *m, const double *Xtf, const double *Xf, const int *mode_x, double *fvec)
{
    const int n = 6;
    for (int j = 0; j < n; j++) {
        if (mode_x[j] == 1) {
            fvec[j] = Xtf[j + n];
            if (j == 1) fvec[j] = Xtf[j + n] + m->p[IP_MUV];
        } else {
            fvec[j] = Xtf[j] - Xf[j];
            if (j == 0) fvec[j] = fvec[j] / m->p[IP_HR];
            if (j == 3 && fabs(cos(Xf[2])) < 1e-5) fvec[j] = Xtf[j + n];
        }
    }
}

/* interceptor.cpp:846-950: closed-form costate guess (IFAC 2017 paper); fills Xi[6..12) */
void orc_interceptor_init_analytical(const orc_model *m, double ti, double *Xi, double tf, const double *Xf)
{
    (void)tf;
    const double *P = m->p;
    double h = Xi[0], v = Xi[1], gamma = Xi[2], chi = Xi[3], L = Xi[4], l = Xi[5];
    double hf = Xf[0], gammaf = Xf[2], chif = Xf[3], Lf = Xf[4], lf = Xf[5];
    double mass = compute_mass(m, ti);
    double c_max = P[IP_C0] * exp(-h / P[IP_HR]) * (P[IP_PROP] + P[IP_EMPTY]) / mass;
    double d = P[IP_D0] * exp(-h / P[IP_HR]) * (P[IP_PROP] + P[IP_EMPTY]) / mass;
    double r = h + P[IP_REARTH], rf = hf + P[IP_REARTH];
    double eta = P[IP_ETA], hr = P[IP_HR];
    double b = sqrt(c_max * d / (2 * eta));
    double cL = cos(L), sL = sin(L), cl = cos(l), sl = sin(l), cLf = cos(Lf), sLf = sin(Lf), clf = cos(lf), slf = sin(lf);
    double sg = sin(gamma), cg = cos(gamma), sc = sin(chi), cc = cos(chi), tg = tan(gamma);
    double ex = rf * cLf * clf - r * cL * cl, ey = rf * cLf * slf - r * cL * sl, ez = rf * sLf - r * sL;
    double R = sqrt(ex * ex + ey * ey + ez * ez);
    double Rdot = -(ex * (sg * cL * cl - cg * cc * sL * cl - cg * sc * sl)
                    + ey * (sg * cL * sl - cg * cc * sL * sl + cg * sc * cl)
                    + ez * (sg * sL + cL * cg * cc)) / R;
    double bdot = -c_max * d * sg * sqrt(2 * eta / (c_max * d)) / (2 * eta * hr);
    double bR = b * R, ep = exp(bR), em = exp(-bR), w = bdot * R + b * Rdot;
    double N1 = ep - em - 2 * b * R, D1 = 4 + ep * (bR - 2) - em * (bR + 2);
    double dN1 = w * (ep + em - 2), dD1 = w * (ep * (bR - 2) + (em * (bR + 2)) + ep - em);
    double k1 = b * R * (ep - em - 2 * b * R) / (4 + ep * (bR - 2) - em * (bR + 2));
    double k1dot = w * N1 / D1 + (b * R * (dN1 * D1 - dD1 * N1) / (D1 * D1));
    double N2 = ep * (bR - 1) + em * (bR + 1), D2 = 4 + ep * (bR - 2) - em * (bR + 2);
    double dN2 = b * R * w * (ep - em), dD2 = w * (ep * (bR - 2) + (em * (bR + 2)) + ep - em);
    double k2 = b * R * (ep * (bR - 1) + em * (bR + 1)) / (4 + ep * (bR - 2) - em * (bR + 2));
    double k2dot = w * N2 / D2 + (b * R * (dN2 * D2 - dD2 * N2) / (D2 * D2));
    double k3 = 2 + k1 - k2, k3dot = k1dot - k2dot;
    /* lambda_1: elevation of the line of sight */
    double l1;
    double dist = fabs(rf * (cL * cLf * cl * clf + cL * cLf * sl * slf + sL * sLf) - r);
    double x_E_R = -cL * cl * ex - cL * sl * ey - sL * ez;
    if (R == 0) l1 = gammaf;
    else if (dist / R >= 1 && x_E_R > 0) l1 = -M_PI / 2.0;
    else if (dist / R >= 1) l1 = M_PI / 2.0;
    else if (x_E_R > 0) l1 = -asin(dist / R);
    else l1 = asin(dist / R);
    /* lambda_2: azimuth of the line of sight */
    double l2;
    double tlam2 = r - rf * (cLf * clf * cL * cl + cLf * slf * cL * sl + sLf * sL);
    double px = rf * cLf * clf + (tlam2 - r) * cL * cl, py = rf * cLf * slf + (tlam2 - r) * cL * sl, pz = rf * sLf + (tlam2 - r) * sL;
    double normProj = sqrt(px * px + py * py + pz * pz);
    double prodScal = -px * sL * cl - py * sL * sl + pz * cL;
    double coordProj_el = -sl * px + cl * py;
    if (normProj == 0) l2 = 0;
    else if (prodScal / normProj <= -1) l2 = M_PI;
    else if (prodScal / normProj >= 1) l2 = 0;
    else if (coordProj_el >= 0) l2 = acos(prodScal / normProj);
    else l2 = -acos(prodScal / normProj);
    double u1 = -(k1 * (gammaf - l1) / R + k2 * sin(gamma - l1) / R + k3 * cg / (2 * hr)) / c_max;
    double u2 = -(k1 * (chif - l2) * cg / R + k2 * sin(chi - l2) * cg / R) / c_max;
    double d1DivC = sg / (c_max * hr);
    double s1 = sin(gamma - l1), c1 = cos(gamma - l1), s2 = sin(chi - l2), c2 = cos(chi - l2);
    double du1 = d1DivC * c_max * u1 -
                 (k1dot * (gammaf - l1) / R + k1 * s1 / (R * R) - k1 * (gammaf - l1) * Rdot / (R * R) + k2dot * s1 / R +
                  k2 * c1 * (c_max * u1 + s1 / R) / R - k2 * s1 * Rdot / (R * R) + k3dot * cg / (2 * hr) -
                  c_max * u1 * k3 * sg / (2 * hr)) / c_max;
    double du2 = d1DivC * c_max * u2 -
                 (k1dot * cg * (chif - l2) / R - k1 * c_max * u1 * sg * (chif - l2) / R + k1 * cg * s2 / (R * R) -
                  k1 * cg * (chif - l2) * Rdot / (R * R) + k2dot * cg * s2 / R - k2 * sg * s2 * c_max * u1 / R +
                  k2 * cg * c2 * (c_max * u2 / cg + s2 / R) / R - k2 * cg * s2 * Rdot / (R * R)) / c_max;
    double pg, pc;
    Xi[7] = -1;
    Xi[8] = pg = 2 * eta * u1;
    Xi[9] = pc = 2 * eta * u2 * cg;
    Xi[6] = (-sg * c_max * u1 * pg * cg - sg * eta * c_max * cg * u1 * u1 - sg * d * cg -
             sg * eta * c_max * cg * u2 * u2 - sg * c_max * u2 * pc - c_max * u2 * tg * pc * cg +
             2 * du1 * eta * cg * cg) / cg;
    Xi[10] = r * (sg * c_max * u2 * tg * pc * cc - 2 * sg * sg * sc * eta * cg * du2 +
                  2 * sg * sg * sc * eta * cg * c_max * u1 * u2 * tg - 2 * sg * du1 * eta * cg * cc -
                  c_max * u1 * pg * cg * cg * cc - eta * c_max * cg * cg * u1 * u1 * cc -
                  2 * cg * cg * cg * sc * eta * du2 + 2 * cg * cg * cg * sc * eta * c_max * u1 * u2 * tg -
                  d * cg * cg * cc - eta * c_max * cg * cg * u2 * u2 * cc - c_max * u2 * pc * cc * cg) / cg;
    Xi[11] = -r * cL * (-sc * sg * c_max * u2 * tg * pc + 2 * sc * sg * du1 * eta * cg +
                        sc * c_max * u1 * pg * cg * cg + sc * eta * c_max * cg * cg * u1 * u1 +
                        sc * d * cg * cg + sc * eta * c_max * cg * cg * u2 * u2 + sc * c_max * u2 * pc * cg -
                        2 * eta * cg * cg * cg * du2 * cc - 2 * eta * cg * du2 * cc * sg * sg +
                        2 * eta * cg * cg * cg * c_max * u1 * u2 * tg * cc +
                        2 * eta * cg * c_max * u1 * u2 * tg * cc * sg * sg) / cg;
    Xi[8] = v * Xi[8];
    Xi[9] = v * Xi[9];
    Xi[6] = v * Xi[6];
    Xi[10] = v * Xi[10];
    Xi[11] = v * Xi[11];
}
