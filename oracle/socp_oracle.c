/*
 * socp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See socp_oracle.h for the rules of use and how this file is pinned.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * Citations are file:line into /root/reference (bherisse/socp @ 2025-09-05).
 */
#include "socp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* model layer                                                         */
/* ------------------------------------------------------------------ */

/* defaults: goddard.cpp:23-40 (dim 7, switching times 0.0227/0.08, parameters_struct
 * goddard.hpp:28-37); doubleIntegrator.cpp:26-34 (dim 6, stepNbr 30). */
void orc_model_init(orc_model *m, int model_id)
{
    if (model_id == ORC_MODEL_INTERCEPTOR) { orc_interceptor_init(m); return; }
    memset(m, 0, sizeof(*m));
    m->model_id = model_id;
    if (model_id == ORC_MODEL_GODDARD) {
        m->dim = 7;
        m->step_nbr = 10;
        m->p[GP_C] = 3.5;  m->p[GP_B] = 7.0;   m->p[GP_KD] = 310.0; m->p[GP_KR] = 500.0;
        m->p[GP_UMAX] = 1.0; m->p[GP_MU1] = 1.0; m->p[GP_MU2] = 0.0; m->p[GP_SING] = -1.0;
        m->nsw = 2; m->sw[0] = 0.0227; m->sw[1] = 0.08;
    } else if (model_id == ORC_MODEL_COVID19) {
        /* covid19.cpp:25-38; ModelInt uses its own data->stepNbr = 1000 (:36, :171-173) */
        m->dim = 4;
        m->step_nbr = 1000;
        m->p[CP_R0] = 4; m->p[CP_TINF] = 10; m->p[CP_TINC] = 5; m->p[CP_N] = 1;
        m->p[CP_IMAX] = 0.1; m->p[CP_MUI] = 1; m->p[CP_UMIN] = -10; m->p[CP_UMAX] = 20;
    } else {
        m->dim = 6;
        m->step_nbr = 30;
        m->p[DP_UMAX] = 1.0; m->p[DP_AMAX] = 1.0; m->p[DP_MUT] = 0.01;
    }
}

int orc_control_dim(const orc_model *m) { return m->model_id == ORC_MODEL_COVID19 ? 1 : (m->model_id == ORC_MODEL_INTERCEPTOR ? 2 : 3); }

int orc_state_len(const orc_model *m, int is_jac)
{
    int s = 2 * m->dim;
    return is_jac ? (s + 1) * s : s;
}

/* goddard.cpp:188-253 -- closed-form singular-arc thrust magnitude bu/au */
double orc_goddard_singular_control(const orc_model *m, double t, const double *X)
{
    (void)t;
    double x = X[0], y = X[1], z = X[2], vx = X[3], vy = X[4], vz = X[5], mass = X[6];
    double p_x = X[7], p_y = X[8], p_z = X[9], p_vx = X[10], p_vy = X[11], p_vz = X[12];
    double r = sqrt(x*x + y*y + z*z);
    double v = sqrt(vx*vx + vy*vy + vz*vz);
    double rdotv = x*vx + y*vy + z*vz;
    double pvdotv = p_vx*vx + p_vy*vy + p_vz*vz;
    double b = m->p[GP_B], C = m->p[GP_C], KD = m->p[GP_KD], kr = m->p[GP_KR];
    double g = 1 / r / r;
    double norm_pv = sqrt(p_vx*p_vx + p_vy*p_vy + p_vz*p_vz);
    double D = KD*exp(-kr*(r - 1));

    /* :214-219, same expressions as the costate equations of Model */
    double p_xdot = -kr*KD / mass*v*exp(-kr*(r - 1))*x / r*pvdotv + g*(p_vx*(1 - 3 * x*x / r / r) / r - p_vy * 3 * x*y / r / r / r - p_vz * 3 * x*z / r / r / r);
    double p_ydot = -kr*KD / mass*v*exp(-kr*(r - 1))*y / r*pvdotv + g*(-p_vx * 3 * y*x / r / r / r + p_vy*(1 - 3 * y*y / r / r) / r - p_vz * 3 * y*z / r / r / r);
    double p_zdot = -kr*KD / mass*v*exp(-kr*(r - 1))*z / r*pvdotv + g*(-p_vx * 3 * z*x / r / r / r - p_vy * 3 * z*y / r / r / r + p_vz*(1 - 3 * z*z / r / r) / r);
    double p_vxdot = -p_x + KD / mass*exp(-kr*(r - 1))*(pvdotv*vx / v + p_vx*v);
    double p_vydot = -p_y + KD / mass*exp(-kr*(r - 1))*(pvdotv*vy / v + p_vy*v);
    double p_vzdot = -p_z + KD / mass*exp(-kr*(r - 1))*(pvdotv*vz / v + p_vz*v);

    /* :221-231 */
    double prdotdotpv = p_xdot*p_vx + p_ydot*p_vy + p_zdot*p_vz;
    double prdotpvdot = p_x*p_vxdot + p_y*p_vydot + p_z*p_vzdot;
    double prdotpv = p_x*p_vx + p_y*p_vy + p_z*p_vz;
    double pvdotdotv = p_vxdot*vx + p_vydot*vy + p_vzdot*vz;
    double pvdotdotpv = p_vxdot*p_vx + p_vydot*p_vy + p_vzdot*p_vz;
    double vdotg = vx*g*x / r + vy*g*y / r + vz*g*z / r;
    double pvdotg = p_vx*g*x / r + p_vy*g*y / r + p_vz*g*z / r;

    /* :236-244 */
    double au = 2 * norm_pv*C / mass*pvdotv
        + 2 * pvdotv*C / mass*norm_pv
        - b / mass*(2 * pvdotv*pvdotv + norm_pv*norm_pv*v*v)
        - b / D*v*prdotpv - C / D*prdotpv / v*pvdotv / norm_pv;

    double bu = -2 * norm_pv*norm_pv*(vdotg + D / mass*v*v*v) + 2 * v*v*pvdotdotpv
        - 2 * pvdotv*(pvdotg + D / mass*v*pvdotv - pvdotdotv)
        + b / C*(2 * norm_pv*pvdotv*(vdotg + D / mass*v*v*v) + norm_pv*v*v*(pvdotg + D / mass*v*pvdotv - pvdotdotv) - v*v*pvdotv / norm_pv*pvdotdotpv)
        - mass / D*kr*rdotv / r*v*prdotpv + mass / D*prdotpv / v*(vdotg + D / mass*v*v*v) - mass / D*v*(prdotdotpv + prdotpvdot);

    return bu / au;
}

/* goddard.cpp:104-185 */
static void goddard_control(const orc_model *m, double t, const double *X, double *u)
{
    double mass = X[6], p_vx = X[10], p_vy = X[11], p_vz = X[12], p_mass = X[13];
    double b = m->p[GP_B], C = m->p[GP_C];
    double norm_pv = sqrt(p_vx*p_vx + p_vy*p_vy + p_vz*p_vz);
    double alpha_u = 0;
    double Switch = m->p[GP_MU1] - b*p_mass - C / mass*norm_pv;       /* :135 */

    if (m->p[GP_MU2] > 0) {                                           /* :137-145 */
        if (Switch < 0) alpha_u = -Switch / 2 / m->p[GP_MU2];
        else alpha_u = 0;
    } else {                                                          /* :146-162 */
        if (t <= m->sw[0]) {
            alpha_u = 1.0;
        } else if (t > m->sw[0] && t <= m->sw[1]) {
            if (m->p[GP_SING] < 0) alpha_u = orc_goddard_singular_control(m, t, X);
            else alpha_u = m->p[GP_SING];
        } else {
            alpha_u = 0;
        }
    }
    u[0] = -p_vx*alpha_u / norm_pv;                                   /* :163-165 */
    u[1] = -p_vy*alpha_u / norm_pv;
    u[2] = -p_vz*alpha_u / norm_pv;

    double norm_u = fabs(alpha_u);
    double u_max = m->p[GP_UMAX];
    if (norm_u > u_max) {                                             /* :171-176 */
        u[0] = u[0] / norm_u*u_max;
        u[1] = u[1] / norm_u*u_max;
        u[2] = u[2] / norm_u*u_max;
    }
}

/* goddard.cpp:48-101 */
static void goddard_model(const orc_model *m, double t, const double *X, double *Xdot)
{
    double x = X[0], y = X[1], z = X[2], vx = X[3], vy = X[4], vz = X[5], mass = X[6];
    double p_x = X[7], p_y = X[8], p_z = X[9], p_vx = X[10], p_vy = X[11], p_vz = X[12];
    double r = sqrt(x*x + y*y + z*z);
    double v = sqrt(vx*vx + vy*vy + vz*vz);
    double pvdotv = p_vx*vx + p_vy*vy + p_vz*vz;
    double b = m->p[GP_B], C = m->p[GP_C], KD = m->p[GP_KD], kr = m->p[GP_KR];
    double g = 1 / r / r;
    double u[3];
    goddard_control(m, t, X, u);
    double norm_u = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double pvdotu = p_vx*u[0] + p_vy*u[1] + p_vz*u[2];

    Xdot[0] = vx;
    Xdot[1] = vy;
    Xdot[2] = vz;
    Xdot[3] = -KD*v*vx*exp(-kr*(r - 1)) / mass - g*x / r + C*u[0] / mass;
    Xdot[4] = -KD*v*vy*exp(-kr*(r - 1)) / mass - g*y / r + C*u[1] / mass;
    Xdot[5] = -KD*v*vz*exp(-kr*(r - 1)) / mass - g*z / r + C*u[2] / mass;
    Xdot[6] = -b*norm_u;
    Xdot[7] = -kr*KD / mass*v*exp(-kr*(r - 1))*x / r*pvdotv + g*(p_vx*(1 - 3 * x*x / r / r) / r - p_vy * 3 * x*y / r / r / r - p_vz * 3 * x*z / r / r / r);
    Xdot[8] = -kr*KD / mass*v*exp(-kr*(r - 1))*y / r*pvdotv + g*(-p_vx * 3 * y*x / r / r / r + p_vy*(1 - 3 * y*y / r / r) / r - p_vz * 3 * y*z / r / r / r);
    Xdot[9] = -kr*KD / mass*v*exp(-kr*(r - 1))*z / r*pvdotv + g*(-p_vx * 3 * z*x / r / r / r - p_vy * 3 * z*y / r / r / r + p_vz*(1 - 3 * z*z / r / r) / r);
    Xdot[10] = -p_x + KD / mass*exp(-kr*(r - 1))*(pvdotv*vx / v + p_vx*v);
    Xdot[11] = -p_y + KD / mass*exp(-kr*(r - 1))*(pvdotv*vy / v + p_vy*v);
    Xdot[12] = -p_z + KD / mass*exp(-kr*(r - 1))*(pvdotv*vz / v + p_vz*v);
    Xdot[13] = -KD*exp(-kr*(r - 1)) / mass / mass*v*pvdotv + C / mass / mass*pvdotu;
}

/* goddard.cpp:256-295 */
static double goddard_hamiltonian(const orc_model *m, double t, const double *X)
{
    double x = X[0], y = X[1], z = X[2], vx = X[3], vy = X[4], vz = X[5], mass = X[6];
    double p_x = X[7], p_y = X[8], p_z = X[9], p_vx = X[10], p_vy = X[11], p_vz = X[12], p_mass = X[13];
    double r = sqrt(x*x + y*y + z*z);
    double v = sqrt(vx*vx + vy*vy + vz*vz);
    double b = m->p[GP_B], C = m->p[GP_C], KD = m->p[GP_KD], kr = m->p[GP_KR];
    double g = 1 / r / r;
    double u[3];
    goddard_control(m, t, X, u);
    double norm_u = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);

    double H = m->p[GP_MU1]*norm_u + m->p[GP_MU2]*norm_u*norm_u
        + p_x*vx + p_y*vy + p_z*vz
        + p_vx*(-KD*v*vx*exp(-kr*(r - 1)) / mass - g*x / r + C*u[0] / mass)
        + p_vy*(-KD*v*vy*exp(-kr*(r - 1)) / mass - g*y / r + C*u[1] / mass)
        + p_vz*(-KD*v*vz*exp(-kr*(r - 1)) / mass - g*z / r + C*u[2] / mass)
        - p_mass*b*norm_u;
    return H;
}

/* doubleIntegrator.cpp:218-259 */
static void dint_control(const orc_model *m, const double *X, double *u)
{
    double a_max = m->p[DP_AMAX], u_max = m->p[DP_UMAX];
    u[0] = -X[9] / a_max;
    u[1] = -X[10] / a_max;
    u[2] = -X[11] / a_max;
    double norm_u = sqrt(u[0]*u[0] + u[1]*u[1] + u[2]*u[2]);
    if (norm_u > u_max) {
        u[0] = u[0] / norm_u*u_max;
        u[1] = u[1] / norm_u*u_max;
        u[2] = u[2] / norm_u*u_max;
    }
}

/* doubleIntegrator.cpp:67-108 (state part f, also :142-153) */
static void dint_f(const orc_model *m, const double *X, double *f)
{
    double a_max = m->p[DP_AMAX];
    double u[3];
    dint_control(m, X, u);
    f[0] = X[3];  f[1] = X[4];  f[2] = X[5];
    f[3] = a_max * u[0];  f[4] = a_max * u[1];  f[5] = a_max * u[2];
    f[6] = 0;  f[7] = 0;  f[8] = 0;
    f[9] = -X[6];  f[10] = -X[7];  f[11] = -X[8];
}

/* doubleIntegrator.cpp:113-213 -- [f ; (df/dX) R] with the constant df/dX of :155-166
 * and the dense triple loop of :193-200 (zero terms are summed as the reference does). */
static void dint_model_jac(const orc_model *m, const double *X, double *XJdot)
{
    enum { S = 12 };
    double dfdX[S * S];
    memset(dfdX, 0, sizeof(dfdX));
    dfdX[S*0 + 3] = 1;  dfdX[S*1 + 4] = 1;  dfdX[S*2 + 5] = 1;
    dfdX[S*3 + 9] = -1; dfdX[S*4 + 10] = -1; dfdX[S*5 + 11] = -1;
    dfdX[S*9 + 6] = -1; dfdX[S*10 + 7] = -1; dfdX[S*11 + 8] = -1;
    dint_f(m, X, XJdot);
    for (int i = 0; i < S; i++) {
        for (int j = 0; j < S; j++) {
            double acc = 0;
            for (int k = 0; k < S; k++) acc += dfdX[S*i + k] * X[S*(k + 1) + j];
            XJdot[S + S*i + j] = acc;
        }
    }
}

/* doubleIntegrator.cpp:264-300 */
static void dint_hamiltonian(const orc_model *m, const double *X, int is_jac, double *H)
{
    double vx = X[3], vy = X[4], vz = X[5], p_x = X[6], p_y = X[7], p_z = X[8];
    double p_vx = X[9], p_vy = X[10], p_vz = X[11];
    double a_max = m->p[DP_AMAX];
    double u[3];
    dint_control(m, X, u);
    double norm_u = sqrt(u[0]*u[0] + u[1]*u[1] + u[2]*u[2]);
    if (!is_jac) {
        H[0] = m->p[DP_MUT] + a_max * a_max*norm_u*norm_u / 2 + p_x * vx + p_y * vy + p_z * vz
             + a_max * (p_vx*u[0] + p_vy * u[1] + p_vz * u[2]);
    } else {
        double dH[13] = { 0., 0., 0., p_x, p_y, p_z, vx, vy, vz, -p_vx, -p_vy, -p_vz, 0. };
        memcpy(H, dH, sizeof(dH));
    }
}

/* covid19.cpp:97-126 */
static double covid_control(const orc_model *m, const double *X)
{
    double S = X[0], I = X[2], pS = X[4], pE = X[5];
    double u = (pE - pS)*S*I / m->p[CP_TINF] / m->p[CP_N] * m->p[CP_R0];
    if (u <= m->p[CP_UMIN]) u = m->p[CP_UMIN];
    if (u >= m->p[CP_UMAX]) u = m->p[CP_UMAX];
    return u;
}

/* covid19.cpp:53-95 */
static void covid_model(const orc_model *m, const double *X, double *Xdot)
{
    double S = X[0], E = X[1], I = X[2], R = X[3], pS = X[4], pE = X[5], pI = X[6], pR = X[7];
    double R0 = m->p[CP_R0], Tinf = m->p[CP_TINF], Tinc = m->p[CP_TINC], N = m->p[CP_N];
    double u = covid_control(m, X);
    double Rt = R0 * (1 - u);
    double Ipen = 0;
    if (I >= m->p[CP_IMAX]) Ipen = -m->p[CP_MUI]*(I - m->p[CP_IMAX]);
    Xdot[0] = -Rt / Tinf / N*S*I;
    Xdot[1] = Rt / Tinf / N*S*I - E / Tinc;
    Xdot[2] = E / Tinc - I / Tinf;
    Xdot[3] = I / Tinf;
    Xdot[4] = (pS - pE)*R*I / Tinf / N;
    Xdot[5] = (pE - pI) / Tinc;
    Xdot[6] = (pS - pE)*R*S / Tinf / N + (pI - pR) / Tinf + Ipen;
    Xdot[7] = 0;
}

/* covid19.cpp:128-165 */
static double covid_hamiltonian(const orc_model *m, const double *X)
{
    double S = X[0], E = X[1], I = X[2], pS = X[4], pE = X[5], pI = X[6], pR = X[7];
    double R0 = m->p[CP_R0], Tinf = m->p[CP_TINF], Tinc = m->p[CP_TINC], N = m->p[CP_N];
    double u = covid_control(m, X);
    double Rt = R0 * (1 - u);
    double Ipen = 0;
    if (I >= m->p[CP_IMAX]) Ipen = m->p[CP_MUI]*(I - m->p[CP_IMAX])*(I - m->p[CP_IMAX]) / 2;
    double H = u*u / 2 + Ipen
        + pS * (-Rt / Tinf / N*S*I)
        + pE * (Rt / Tinf / N*S*I - E / Tinc)
        + pI * (E / Tinc - I / Tinf)
        + pR * (I / Tinf);
    return H;
}

void orc_control(const orc_model *m, double t, const double *X, double *u3)
{
    if (m->model_id == ORC_MODEL_GODDARD) goddard_control(m, t, X, u3);
    else if (m->model_id == ORC_MODEL_COVID19) u3[0] = covid_control(m, X);
    else if (m->model_id == ORC_MODEL_INTERCEPTOR) orc_interceptor_control(m, t, X, u3);
    else dint_control(m, X, u3);
}

/* odeTools.hpp:37-40 -> model::Model; goddard ignores isJac (goddard.cpp:48),
 * doubleIntegrator dispatches on it (doubleIntegrator.cpp:49-62). */
void orc_rhs(const orc_model *m, double t, const double *X, int is_jac, double *Xdot)
{
    if (m->model_id == ORC_MODEL_GODDARD) goddard_model(m, t, X, Xdot);
    else if (m->model_id == ORC_MODEL_COVID19) covid_model(m, X, Xdot);
    else if (m->model_id == ORC_MODEL_INTERCEPTOR) orc_interceptor_rhs(m, t, X, Xdot);
    else if (is_jac) dint_model_jac(m, X, Xdot);
    else dint_f(m, X, Xdot);
}

void orc_hamiltonian(const orc_model *m, double t, const double *X, int is_jac, double *H)
{
    if (m->model_id == ORC_MODEL_GODDARD) H[0] = goddard_hamiltonian(m, t, X);
    else if (m->model_id == ORC_MODEL_COVID19) H[0] = covid_hamiltonian(m, X);
    else if (m->model_id == ORC_MODEL_INTERCEPTOR) H[0] = orc_interceptor_hamiltonian(m, t, X);
    else dint_hamiltonian(m, X, is_jac, H);
}

/* ------------------------------------------------------------------ */
/* ODE layer                                                           */
/* ------------------------------------------------------------------ */

#define ORC_MAX_LEN 256   /* (2*7+1)*14 = 210 is the largest state here */

/* odeTools.cpp:89-98: X <- X + (step/6.0)*(F1 + (F4 + 2.0*(F2+F3))), stage states
 * X + (step/2.0)*F, stage times t + step/2.0 (twice) and t + step. */
void orc_rk4_step(const orc_model *m, double t, double *X, double step, int is_jac)
{
    int n = orc_state_len(m, is_jac);
    /* goddard's Model returns a 2d vector whatever isJac is (goddard.cpp:49) */
    if (m->model_id != ORC_MODEL_DOUBLE_INTEGRATOR) n = 2 * m->dim;
    double F1[ORC_MAX_LEN], F2[ORC_MAX_LEN], F3[ORC_MAX_LEN], F4[ORC_MAX_LEN], Y[ORC_MAX_LEN];
    double h2 = step / 2.0;

    orc_rhs(m, t, X, is_jac, F1);
    for (int i = 0; i < n; i++) Y[i] = X[i] + h2 * F1[i];
    orc_rhs(m, t + step / 2.0, Y, is_jac, F2);
    for (int i = 0; i < n; i++) Y[i] = X[i] + h2 * F2[i];
    orc_rhs(m, t + step / 2.0, Y, is_jac, F3);
    for (int i = 0; i < n; i++) Y[i] = X[i] + step * F3[i];
    orc_rhs(m, t + step, Y, is_jac, F4);

    double h6 = step / 6.0;
    for (int i = 0; i < n; i++)
        X[i] = X[i] + h6 * (F1[i] + (F4[i] + 2.0 * (F2[i] + F3[i])));
}

/* odeTools.cpp:128-146 (non-Boost branch): t accumulated by t += dt; last step clamped
 * to tf - t when t + dt > tf; zero steps when tf <= t0 + dt/2 (incl. backward segments). */
long orc_integrate(const orc_model *m, double *X, double t0, double tf, double dt, int is_jac)
{
    double t = t0;
    long steps = 0;
    while (t < (tf - dt / 2)) {
        if (t + dt > tf) orc_rk4_step(m, t, X, tf - t, is_jac);
        else orc_rk4_step(m, t, X, dt, is_jac);
        t += dt;
        steps++;
    }
    return steps;
}

/* ---- Dormand-Prince 5(4), controlled, as Boost.Odeint drives it [ext] ------------------------------
 * runge_kutta_dopri5::do_step_impl: stage sums a1*x1 + a2*x2 + ... left to right (scale_sumN);
 * default_error_checker: max_i |err_i| / (eps_abs + eps_rel*(|x_i| + dt*|dxdt_i|)) with the OLD state;
 * default_step_adjuster: reject -> dt *= max(0.9 err^(-1/3), 1/5); accept and err < 0.5 ->
 * dt *= 0.9 max(5^-5, err)^(-1/5); integrate_adaptive with a dense-output stepper: step while
 * t + dt <= tf, then re-initialise with dt = tf - t (which also drops the FSAL derivative). */
static int dopri5_try_step(const orc_model *m, int n, int is_jac, const double *x, const double *k1, double *t, double *dt,
                           double tol, double *xnew, double *knew)
{
    static const double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
    static const double b21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40, b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9,
        b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729,
        b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176, b65 = -5103.0 / 18656,
        c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
    static const double dc1 = 35.0 / 384 - 5179.0 / 57600, dc3 = 500.0 / 1113 - 7571.0 / 16695, dc4 = 125.0 / 192 - 393.0 / 640,
        dc5 = -2187.0 / 6784 - -92097.0 / 339200, dc6 = 11.0 / 84 - 187.0 / 2100, dc7 = -1.0 / 40;
    double k2[ORC_MAX_LEN], k3[ORC_MAX_LEN], k4[ORC_MAX_LEN], k5[ORC_MAX_LEN], k6[ORC_MAX_LEN], y[ORC_MAX_LEN];
    const double h = *dt, tt = *t;
    for (int i = 0; i < n; i++) y[i] = 1.0 * x[i] + h * b21 * k1[i];
    orc_rhs(m, tt + h * a2, y, is_jac, k2);
    for (int i = 0; i < n; i++) y[i] = 1.0 * x[i] + h * b31 * k1[i] + h * b32 * k2[i];
    orc_rhs(m, tt + h * a3, y, is_jac, k3);
    for (int i = 0; i < n; i++) y[i] = 1.0 * x[i] + h * b41 * k1[i] + h * b42 * k2[i] + h * b43 * k3[i];
    orc_rhs(m, tt + h * a4, y, is_jac, k4);
    for (int i = 0; i < n; i++) y[i] = 1.0 * x[i] + h * b51 * k1[i] + h * b52 * k2[i] + h * b53 * k3[i] + h * b54 * k4[i];
    orc_rhs(m, tt + h * a5, y, is_jac, k5);
    for (int i = 0; i < n; i++) y[i] = 1.0 * x[i] + h * b61 * k1[i] + h * b62 * k2[i] + h * b63 * k3[i] + h * b64 * k4[i] + h * b65 * k5[i];
    orc_rhs(m, tt + h, y, is_jac, k6);
    for (int i = 0; i < n; i++) xnew[i] = 1.0 * x[i] + h * c1 * k1[i] + h * c3 * k3[i] + h * c4 * k4[i] + h * c5 * k5[i] + h * c6 * k6[i];
    orc_rhs(m, tt + h, xnew, is_jac, knew);
    double err = 0;
    for (int i = 0; i < n; i++) {
        double e = h * dc1 * k1[i] + h * dc3 * k3[i] + h * dc4 * k4[i] + h * dc5 * k5[i] + h * dc6 * k6[i] + h * dc7 * knew[i];
        e = fabs(e) / (tol + tol * (1.0 * fabs(x[i]) + 1.0 * h * fabs(k1[i])));
        if (e > err || e != e) err = e;
    }
    if (!(err <= 1.0)) {                                   /* reject (also on NaN) */
        double f = 0.9 * pow(err, -1.0 / 3.0);
        if (!(f > 0.2)) f = 0.2;
        *dt = h * f;
        return 0;
    }
    *t = tt + h;
    if (err < 0.5) {
        double e = err > pow(5.0, -5.0) ? err : pow(5.0, -5.0);
        *dt = h * (0.9 * pow(e, -1.0 / 5.0));
    }
    return 1;
}

/* the same on the AUGMENTED state [X ; dX/dX0] (is_jac = 1): with -D_USE_BOOST every integrate() call of the reference goes through
 * the adaptive stepper, the variational trajectories of the hybrj path included (odeTools.cpp:129-134, model.hpp:395-414,
 * shooting.cpp:996-1130), the error norm taken over all (2d + 1) 2d entries.  [ext] parity unpinned like the state-only form. */
static long integrate_dopri5_any(orc_model *m, int is_jac, double *X, double t0, double tf, double dt, double tol, long *rejected,
                                 orc_step_hook hook);

long orc_integrate_dopri5_hook(orc_model *m, double *X, double t0, double tf, double dt, double tol, long *rejected,
                               orc_step_hook hook)
{
    return integrate_dopri5_any(m, 0, X, t0, tf, dt, tol, rejected, hook);
}

long orc_integrate_dopri5_jac(const orc_model *m, double *X, double t0, double tf, double dt, double tol, long *rejected)
{
    return integrate_dopri5_any((orc_model *)m, 1, X, t0, tf, dt, tol, rejected, 0);
}

static long integrate_dopri5_any(orc_model *m, int is_jac, double *X, double t0, double tf, double dt, double tol, long *rejected,
                                 orc_step_hook hook)
{
    const int n = orc_state_len(m, is_jac);
    const double eps = DBL_EPSILON;
    double k1[ORC_MAX_LEN], xn[ORC_MAX_LEN], kn[ORC_MAX_LEN];
    double t = t0, h = dt;
    long steps = 0, rej = 0, budget = ORC_ADAPTIVE_BUDGET;
    int have_k1 = 0;
    if (!(h > 0)) { if (rejected) *rejected = 0; return 0; }          /* zero-length / backward: nothing to do */
    while (tf - t > eps && budget > 0) {                              /* less_with_sign(t, tf, dt) */
        while (t + h - tf <= eps && budget > 0) {                     /* less_eq_with_sign(t + dt, tf, dt) */
            if (hook && hook(m, t, X)) have_k1 = 0;
            if (!have_k1) { orc_rhs(m, t, X, is_jac, k1); have_k1 = 1; }
            int tries = 0, ok;
            do {
                ok = dopri5_try_step(m, n, is_jac, X, k1, &t, &h, tol, xn, kn);
                budget--;
                if (!ok) rej++;
            } while (!ok && ++tries < 500);
            if (!ok) {                                                /* odeint's step_adjustment_error */
                for (int i = 0; i < n; i++) X[i] = NAN;
                if (rejected) *rejected = rej;
                return -1;
            }
            memcpy(X, xn, sizeof(double) * n);
            memcpy(k1, kn, sizeof(double) * n);
            steps++;
        }
        h = tf - t;                                                   /* initialize(x, t, tf - t) */
        have_k1 = 0;
    }
    if (budget <= 0 && tf - t > eps)
        for (int i = 0; i < n; i++) X[i] = NAN;
    if (rejected) *rejected = rej;
    return steps;
}

long orc_integrate_dopri5(const orc_model *m, double *X, double t0, double tf, double dt, double tol, long *rejected)
{
    return orc_integrate_dopri5_hook((orc_model *)m, X, t0, tf, dt, tol, rejected, 0);
}

/* model.hpp:395-414 / goddard.cpp:298-317: dt = (tf - t0)/stepNbr, then integrate. */
long orc_model_int(const orc_model *m, double t0, const double *X0, double tf, int is_jac, double *Xf)
{
    int n = orc_state_len(m, is_jac);
    double dt = (tf - t0) / m->step_nbr;
    if (Xf != X0) memcpy(Xf, X0, sizeof(double) * n);
    return orc_integrate(m, Xf, t0, tf, dt, is_jac);
}

void orc_compute_traj(orc_model *m, double t0, const double *X0, double tf, int is_jac, double *Xf)
{
    if (m->model_id == ORC_MODEL_INTERCEPTOR) {
        double X[12];
        memcpy(X, X0, sizeof(X));
        if (m->integrator == 1) orc_interceptor_compute_traj_adaptive(m, t0, X, tf, m->tol, Xf);
        else orc_interceptor_compute_traj(m, t0, X, tf, Xf);
    } else if (m->integrator == 1) {
        int n = orc_state_len(m, is_jac);
        if (Xf != X0) memcpy(Xf, X0, sizeof(double) * n);
        if (is_jac) orc_integrate_dopri5_jac(m, Xf, t0, tf, (tf - t0) / m->step_nbr, m->tol, 0);
        else orc_integrate_dopri5(m, Xf, t0, tf, (tf - t0) / m->step_nbr, m->tol, 0);
    } else {
        orc_model_int(m, t0, X0, tf, is_jac, Xf);
    }
}

void orc_integrate_batch(const orc_model *m, int B, const double *t0, const double *tf,
                         const double *aux_sw, const double *X0, double *Xf, int is_jac)
{
    int n = orc_state_len(m, is_jac);
    for (int b = 0; b < B; b++) {
        orc_model mm = *m;
        if (aux_sw) { mm.nsw = 2; mm.sw[0] = aux_sw[2*b]; mm.sw[1] = aux_sw[2*b + 1]; }
        orc_compute_traj(&mm, t0[b], X0 + (size_t)n*b, tf[b], is_jac, Xf + (size_t)n*b);
    }
}

/* ------------------------------------------------------------------ */
/* shooting layer                                                      */
/* ------------------------------------------------------------------ */

/* shooting.cpp:179,196 */
int orc_num_param(const orc_problem *p)
{
    int n = 2 * p->dim * p->num_multi;
    for (int j = 0; j <= p->num_multi; j++) if (p->mode_t[j] == ORC_FREE) n++;
    return n;
}

/* shooting.cpp:1579-1617.  Side effect (:1615 -> goddard.cpp:373-377): the model's
 * switching times become the FREE node times with index < M, in node order. */
void orc_compute_timeline(orc_model *m, const orc_problem *p, const double *z, double *tl)
{
    int M = p->num_multi;
    int nbr = 2 * p->dim * M;
    int cur = 0, nsw = 0;
    for (int j = 0; j <= M; j++) {
        if (p->mode_t[j] == ORC_FIXED) {
            tl[j] = p->time[j];
            for (int k = cur + 1; k < j; k++)
                tl[k] = tl[cur] + (k - cur) * (tl[j] - tl[cur]) / (j - cur);
            cur = j;
        }
        if (p->mode_t[j] == ORC_FREE) {
            nbr += 1;
            tl[j] = z[nbr - 1];
            if (j < M && nsw < ORC_MAX_SWITCH) m->sw[nsw++] = tl[j];
            for (int k = cur + 1; k < j; k++)
                tl[k] = tl[cur] + (k - cur) * (tl[j] - tl[cur]) / (j - cur);
            cur = j;
        }
    }
    m->nsw = nsw;
}

/* model.hpp:196-228 (Initial) and :90-122 (Final), isJac == 0: identical row rule */
static void boundary_rows(int d, const double *Xt, const double *Xd, const int *mode, double *f)
{
    for (int j = 0; j < d; j++) {
        if (mode[j] == ORC_FREE) f[j] = Xt[j + d];
        else f[j] = Xt[j] - Xd[j];
    }
}

/* model.hpp:196-228 / :90-122, isJac == 1: d x 2d block taken from the sensitivity rows */
static void boundary_rows_jac(int d, const double *Xt, const int *mode, double *f)
{
    int s = 2 * d;
    for (int j = 0; j < d; j++) {
        int src = (mode[j] == ORC_FREE) ? (j + d + 1) : (j + 1);
        for (int i = 0; i < s; i++) f[s * j + i] = Xt[s * src + i];
    }
}

/* model.hpp:133-185 / :239-290, isJac == 1: (d+1) x (2d+1) block incl. time column and H row */
static void boundary_h_rows_jac(const orc_model *m, double t, const double *Xt, const int *mode, double *f)
{
    int d = m->dim, s = 2 * d, w = s + 1;
    double fx[ORC_MAX_LEN], dH[ORC_MAX_LEN];
    orc_rhs(m, t, Xt, 0, fx);
    for (int j = 0; j < d; j++) {
        int src = (mode[j] == ORC_FREE) ? (j + d + 1) : (j + 1);
        for (int i = 0; i < s; i++) f[w * j + i] = Xt[s * src + i];
        f[w * j + s] = (mode[j] == ORC_FREE) ? fx[j + d] : fx[j];
    }
    orc_hamiltonian(m, t, Xt, 1, dH);
    for (int i = 0; i < s; i++) {
        f[w * d + i] = 0;
        for (int k = 0; k < s; k++) f[w * d + i] += dH[k] * Xt[s * (k + 1) + i];
    }
    f[w * d + s] = 0;
    for (int k = 0; k < s; k++) f[w * d + s] += dH[k] * fx[k];
    f[w * d + s] += dH[s];
}

/* model.hpp:299-328 default; goddard.cpp:343-370 overrides with H(t, X) alone. isJac == 0. */
static double switching_times_function(const orc_model *m, double t, const double *X, const double *Xp)
{
    double h, hp;
    orc_hamiltonian(m, t, X, 0, &h);
    if (m->model_id == ORC_MODEL_GODDARD) return h;
    orc_hamiltonian(m, t, Xp, 0, &hp);
    return h - hp;
}

/* model.hpp:305-326, isJac == 1: [dH/dX R | -dHp/dX Rp | time term], length 4d+1 */
static void switching_times_function_jac(const orc_model *m, double t, const double *X, const double *Xp, double *f)
{
    int d = m->dim, s = 2 * d;
    double fxt[ORC_MAX_LEN], fxp[ORC_MAX_LEN], dH[ORC_MAX_LEN], dHp[ORC_MAX_LEN];
    for (int i = 0; i < 4 * d + 1; i++) f[i] = 0;
    orc_rhs(m, t, X, 0, fxt);
    orc_hamiltonian(m, t, X, 1, dH);
    orc_rhs(m, t, Xp, 0, fxp);
    orc_hamiltonian(m, t, Xp, 1, dHp);
    for (int i = 0; i < s; i++) {
        for (int k = 0; k < s; k++) {
            f[i] += dH[k] * X[s * (k + 1) + i];
            f[s + i] -= dHp[k] * Xp[s * (k + 1) + i];
        }
    }
    for (int k = 0; k < s; k++) f[4 * d] += dH[k] * fxt[k] - dHp[k] * fxp[k];
    f[4 * d] += dH[s] - dHp[s];
}

/* one residual block by name (model.hpp:90-328); the assembly below uses the same static pieces */
int orc_residual_block(const orc_model *m, int which, double t, const double *X, const double *other,
                       const int *mode, int is_jac, double *out)
{
    const int d = m->dim, s = 2 * d;
    if (which == 4) {
        if (!is_jac) { out[0] = switching_times_function(m, t, X, other); return 1; }
        if (m->model_id == ORC_MODEL_GODDARD) { orc_hamiltonian(m, t, X, 1, out); return 1; }   /* goddard.cpp:369: H(t, X, isJac), and its Hamiltonian ignores isJac (:256-295) */
        switching_times_function_jac(m, t, X, other, out);
        return 4 * d + 1;
    }
    const int with_h = (which == 1 || which == 3);
    if (!is_jac) {
        boundary_rows(d, X, other, mode, out);
        if (with_h) orc_hamiltonian(m, t, X, 0, &out[d]);
        return with_h ? d + 1 : d;
    }
    if (with_h) { boundary_h_rows_jac(m, t, X, mode, out); return (d + 1) * (s + 1); }
    boundary_rows_jac(d, X, mode, out);
    return d * s;
}

/* shooting.cpp:1511-1576, isJac == 0.  FREE state mode at an interior node defers to
 * model::SwitchingStateFunction, a no-op by default (model.hpp:339-341): rows untouched. */
static void multiple_shooting_rows(int d, const double *X, const double *Xp, const double *Xd,
                                   const int *mode, double *f)
{
    for (int j = 0; j < d; j++) {
        switch (mode[j]) {
        case ORC_FIXED:
            f[j] = X[j] - Xd[j];
            f[j + d] = Xp[j] - Xd[j];
            break;
        case ORC_FREE:
            break;
        default: /* CONTINUOUS and the reference's default: branch */
            f[j] = X[j] - Xp[j];
            f[j + d] = X[j + d] - Xp[j + d];
            break;
        }
    }
}

/* shooting.cpp:1524-1555, isJac == 1: 2d rows of stride 4d+1 */
static void multiple_shooting_rows_jac(const orc_model *m, double t, const double *X, const double *Xp,
                                       const int *mode, int mode_t, double *f)
{
    int d = m->dim, s = 2 * d, w = 4 * d + 1;
    double fxt[ORC_MAX_LEN], fxp[ORC_MAX_LEN];
    orc_rhs(m, t, X, 0, fxt);
    orc_rhs(m, t, Xp, 0, fxp);
    for (int j = 0; j < d; j++) {
        if (mode[j] == ORC_FIXED) {
            for (int i = 0; i < s; i++) {
                f[w * j + i] = X[s * (j + 1) + i];
                f[w * (j + d) + s + i] = Xp[s * (j + 1) + i];
            }
            if (mode_t == ORC_FREE) {
                f[w * j + 4 * d] = fxt[j];
                f[w * (j + d) + 4 * d] = fxp[j];
            }
        } else if (mode[j] == ORC_CONTINUOUS) {
            for (int i = 0; i < s; i++) {
                f[w * j + i] = X[s * (j + 1) + i];
                f[w * j + s + i] = -Xp[s * (j + 1) + i];
                f[w * (j + d) + i] = X[s * (j + d + 1) + i];
                f[w * (j + d) + s + i] = -Xp[s * (j + d + 1) + i];
            }
            if (mode_t == ORC_FREE) {
                f[w * j + 4 * d] = fxt[j] - fxp[j];
                f[w * (j + d) + 4 * d] = fxt[j + d] - fxp[j + d];
            }
        }
    }
}

/* shooting.cpp:918-993 */
void orc_shooting_function(orc_model *m, const orc_problem *p, const double *z, double *fvec)
{
    int d = p->dim, s = 2 * d, M = p->num_multi;
    double *tl = (double *)malloc(sizeof(double) * (M + 1));
    double X1[ORC_MAX_LEN], Xtf[ORC_MAX_LEN], Xp[ORC_MAX_LEN], rows[ORC_MAX_LEN];
    memcpy(X1, z, sizeof(double) * s);
    orc_compute_timeline(m, p, z, tl);
    memset(rows, 0, sizeof(rows));

    int nbr = s * M;
    for (int i = 0; i < M; i++) {
        double t1 = tl[i], t2 = tl[i + 1];
        orc_compute_traj(m, t1, X1, t2, 0, Xtf);                          /* Move -> model::ComputeTraj */
        int index = s * (i + 1);
        if (i == 0) {
            double f0[ORC_MAX_LEN];
            boundary_rows(d, X1, p->xnode, p->mode_x, f0);
            for (int k = 0; k < d; k++) fvec[k] = f0[k];
            if (p->mode_t[0] != ORC_FIXED) {
                double h;
                orc_hamiltonian(m, tl[0], X1, 0, &h);                     /* model.hpp:253 */
                fvec[s * M] = h;
                nbr += 1;
            }
        }
        if (i < M - 1) {
            memcpy(Xp, z + index, sizeof(double) * s);
            if (p->mode_t[i + 1] == ORC_FREE) {
                fvec[nbr] = switching_times_function(m, t2, Xtf, Xp);
                nbr += 1;
            }
            multiple_shooting_rows(d, Xtf, Xp, p->xnode + (size_t)s * (i + 1), p->mode_x + (size_t)d * (i + 1), rows);
            for (int k = 0; k < s; k++) fvec[index + k] = rows[k];
            memcpy(X1, Xp, sizeof(double) * s);
        }
        if (i == M - 1) {
            double fN[ORC_MAX_LEN];
            if (m->model_id == ORC_MODEL_INTERCEPTOR)                    /* interceptor.cpp:221-272 overrides */
                orc_interceptor_final_rows(m, Xtf, p->xnode + (size_t)s * M, p->mode_x + (size_t)d * M, fN);
            else
                boundary_rows(d, Xtf, p->xnode + (size_t)s * M, p->mode_x + (size_t)d * M, fN);
            for (int k = 0; k < d; k++) fvec[k + d] = fN[k];
            if (p->mode_t[M] != ORC_FIXED) {
                double h;
                orc_hamiltonian(m, t2, Xtf, 0, &h);                       /* model.hpp:147 */
                if (m->model_id == ORC_MODEL_INTERCEPTOR) h = h + m->p[IP_MUT];   /* interceptor.cpp:270 */
                fvec[nbr] = h;
                nbr += 1;
            }
        }
    }
    free(tl);
}

static void augmented_identity(int s, const double *state, double *X)
{
    memset(X, 0, sizeof(double) * (s + 1) * s);
    for (int i = 0; i < s; i++) X[i] = state[i];
    for (int i = 0; i < s; i++) X[s * (i + 1) + i] = 1;
}

/* shooting.cpp:996-1130 (serial variational Jacobian, row-major fjac[n*row+col]).
 * Quirk kept: at a FREE interior time the copy loop :1070 runs j <= 4d, so the time
 * term is also written at column index+2d (harmless for M == 2, where that IS the
 * free-time column). */
void orc_shooting_jacobian(orc_model *m, const orc_problem *p, const double *z, double *fjac)
{
    int d = p->dim, s = 2 * d, M = p->num_multi, n = orc_num_param(p);
    int L = (s + 1) * s;
    double *tl = (double *)malloc(sizeof(double) * (M + 1));
    double *X1 = (double *)malloc(sizeof(double) * L);
    double *Xtf = (double *)malloc(sizeof(double) * L);
    double *Xp = (double *)malloc(sizeof(double) * L);
    double *blk = (double *)malloc(sizeof(double) * (size_t)s * (4 * d + 1) + sizeof(double) * (d + 1) * (s + 1));
    memset(fjac, 0, sizeof(double) * (size_t)n * n);
    augmented_identity(s, z, X1);
    orc_compute_timeline(m, p, z, tl);

    int nbr = s * M;
    for (int i = 0; i < M; i++) {
        double t1 = tl[i], t2 = tl[i + 1];
        orc_compute_traj(m, t1, X1, t2, 1, Xtf);                          /* Move(..., isJac = 1) -> model::ComputeTraj: the selected integrator */
        int index = s * (i + 1);
        if (i == 0) {
            if (p->mode_t[0] == ORC_FIXED) {
                boundary_rows_jac(d, X1, p->mode_x, blk);
                for (int k = 0; k < d; k++)
                    for (int j = 0; j < s; j++) fjac[(size_t)n * k + j] = blk[s * k + j];
            } else {
                int w = s + 1;
                boundary_h_rows_jac(m, tl[0], X1, p->mode_x, blk);
                for (int k = 0; k < d; k++) {
                    for (int j = 0; j < s; j++) fjac[(size_t)n * k + j] = blk[w * k + j];
                    fjac[(size_t)n * k + nbr] = blk[w * k + s];
                }
                for (int j = 0; j < s; j++) fjac[(size_t)n * nbr + j] = blk[w * d + j];
                fjac[(size_t)n * nbr + nbr] = blk[w * d + s];
                nbr += 1;
            }
        }
        if (i < M - 1) {
            int w = 4 * d + 1;
            augmented_identity(s, z + index, Xp);
            memset(blk, 0, sizeof(double) * (size_t)s * w);
            multiple_shooting_rows_jac(m, t2, Xtf, Xp, p->mode_x + (size_t)d * (i + 1), p->mode_t[i + 1], blk);
            if (p->mode_t[i + 1] == ORC_FREE) {
                double sf[ORC_MAX_LEN];
                for (int k = 0; k < s; k++) {
                    for (int j = 0; j < 4 * d + 1; j++) {
                        int col = index - s + j;
                        if (col < n) fjac[(size_t)n * (index + k) + col] = blk[w * k + j];
                    }
                    fjac[(size_t)n * (index + k) + nbr] = blk[w * k + 4 * d];
                }
                switching_times_function_jac(m, t2, Xtf, Xp, sf);
                for (int j = 0; j < 4 * d; j++) fjac[(size_t)n * nbr + (index - s + j)] = sf[j];
                fjac[(size_t)n * nbr + nbr] = sf[4 * d];
                nbr += 1;
            } else {
                for (int k = 0; k < s; k++)
                    for (int j = 0; j < 4 * d; j++)
                        fjac[(size_t)n * (index + k) + (index - s + j)] = blk[w * k + j];
            }
            memcpy(X1, Xp, sizeof(double) * L);
        }
        if (i == M - 1) {
            if (p->mode_t[M] == ORC_FIXED) {
                boundary_rows_jac(d, Xtf, p->mode_x + (size_t)d * M, blk);
                for (int k = 0; k < d; k++)
                    for (int j = 0; j < s; j++) fjac[(size_t)n * (d + k) + (s * i + j)] = blk[s * k + j];
            } else {
                int w = s + 1;
                boundary_h_rows_jac(m, t2, Xtf, p->mode_x + (size_t)d * M, blk);
                for (int k = 0; k < d; k++) {
                    for (int j = 0; j < s; j++) fjac[(size_t)n * (d + k) + (s * i + j)] = blk[w * k + j];
                    fjac[(size_t)n * (d + k) + nbr] = blk[w * k + s];
                }
                for (int j = 0; j < s; j++) fjac[(size_t)n * nbr + (s * i + j)] = blk[w * d + j];
                fjac[(size_t)n * nbr + nbr] = blk[w * d + s];
                nbr += 1;
            }
        }
    }
    free(tl); free(X1); free(Xtf); free(Xp); free(blk);
}

/* MINPACK fdjac1, dense branch (ml+mu+1 >= n as the reference passes ml = mu = n-1,
 * shooting.cpp:789-790); [ext] algorithm, SURVEY Appendix A. */
void orc_fdjac1(orc_model *m, const orc_problem *p, const double *z, const double *fvec,
                double epsfcn, double *fjac)
{
    int n = orc_num_param(p);
    double eps = sqrt(epsfcn > DBL_EPSILON ? epsfcn : DBL_EPSILON);
    double *x = (double *)malloc(sizeof(double) * n);
    double *wa = (double *)malloc(sizeof(double) * n);
    memcpy(x, z, sizeof(double) * n);
    for (int j = 0; j < n; j++) {
        double temp = x[j];
        double h = eps * fabs(temp);
        if (h == 0) h = eps;
        x[j] = temp + h;
        orc_shooting_function(m, p, x, wa);
        x[j] = temp;
        for (int i = 0; i < n; i++) fjac[i + (size_t)j * n] = (wa[i] - fvec[i]) / h;
    }
    free(x); free(wa);
}

void orc_residual_batch(orc_model *m, const orc_problem *p, int B, const double *Z, double *F)
{
    int n = orc_num_param(p);
    for (int b = 0; b < B; b++) orc_shooting_function(m, p, Z + (size_t)n * b, F + (size_t)n * b);
}
