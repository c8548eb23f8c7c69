// interceptor.cpp -- host side of the interceptor model mirror (reference: interceptor.cpp:19-998).
#include "interceptor.hpp"

#include <cmath>
#include <stdexcept>

#include "socp_hip.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct interceptor::data_struct {
    int n;
    parameters_struct parameters;
    real R_Earth, mu0;
    int stepNbr;                 // ModelInt steps with this, not model::stepNbr (interceptor.cpp:59,105)
    std::string strFileTrace;
    int currentChart;            // as the last ComputeTraj left it (interceptor.cpp:213-215 keeps it)
    real chartLimit;
    int stageMode;
};

interceptor::interceptor(std::string the_fileTrace) : model(6), data(new data_struct)
{
    // interceptor.cpp:37-64
    data->n = dim;
    parameters_struct &p = data->parameters;
    p.c0 = 0.00075; p.hr = 7500; p.d0 = 0.00005; p.eta = 0.442;
    p.propellant_mass = 200; p.empty_mass = 200; p.q = 10; p.ve = 1500;
    p.alpha_max = M_PI / 6; p.u_max = 1; p.a_max = 1500;
    p.r_2p = 0; p.t_2p = 0;
    p.mu_gft = 1; p.muT = 0; p.muV = 1; p.muC = 0;
    data->R_Earth = 6378145;
    data->mu0 = 3.986e14;
    data->stepNbr = 50;
    data->strFileTrace = the_fileTrace;
    strFileTrace = the_fileTrace;
    data->currentChart = 1;
    data->chartLimit = 0.1;
    data->stageMode = 0;
    std::ofstream wipe(data->strFileTrace.c_str(), std::ios::trunc);
}

interceptor::~interceptor() { delete data; }
interceptor::parameters_struct &interceptor::GetParameterData() { return data->parameters; }

int interceptor::DeviceModelId() const { return SOCP_MODEL_INTERCEPTOR; }
int interceptor::DeviceStepNumber() const { return data->stepNbr; }
std::vector<real> interceptor::DeviceSwitchingTimes() const
{
    return std::vector<real>{(real)data->stageMode, (real)data->currentChart};
}

int interceptor::DeviceParams(double *out, int cap) const
{
    if (cap < SOCP_INTERCEPTOR_NPARAMS) return 0;
    const parameters_struct &p = data->parameters;
    const double v[SOCP_INTERCEPTOR_NPARAMS] = {p.c0, p.hr, p.d0, p.eta, p.propellant_mass, p.empty_mass, p.q, p.ve, p.alpha_max,
                                                p.u_max, p.a_max, p.mu_gft, p.muT, p.muV, p.muC, data->R_Earth, data->mu0,
                                                data->chartLimit};
    for (int i = 0; i < SOCP_INTERCEPTOR_NPARAMS; i++) out[i] = v[i];
    return SOCP_INTERCEPTOR_NPARAMS;
}

// evaluated in the chart / stage the object is currently in, like the reference (interceptor.cpp:69-98)
interceptor::mstate interceptor::Model(real const &t, mstate const &X, int) const { return DeviceEval(SOCP_EVAL_RHS, t, X, 0); }
interceptor::mcontrol interceptor::Control(real const &t, mstate const &X) const { return DeviceEval(SOCP_EVAL_CONTROL, t, X, 0); }
interceptor::mstate interceptor::Hamiltonian(real const &t, mstate const &X, int) const { return DeviceEval(SOCP_EVAL_HAMILTONIAN, t, X, 0); }
int interceptor::GetMode(real const &, mstate const &) const { return data->stageMode; }

// interceptor.cpp:981-997
real interceptor::ComputeMass(real const &t, mstate const &) const
{
    const parameters_struct &p = data->parameters;
    const real qm = p.q * p.mu_gft;
    const real t1 = p.propellant_mass / p.q;
    return p.empty_mass + p.propellant_mass - qm * (data->stageMode == 1 ? t : t1);
}

interceptor::mstate interceptor::ComputeTraj(real const &t0, mstate const &X0, real const &tf, int isTrace, int isJac)
{
    if (isJac) throw std::runtime_error("interceptor::ComputeTraj: the model has no variational equations (modelOrder 0)");
    if (AdaptiveIntegrator() && isTrace)
        throw std::runtime_error("interceptor::ComputeTraj: trace replay is available with the fixed-step integrator only");
    socp_ctx *ctx = DeviceContext();
    const int S = 2 * dim;
    if ((int)X0.size() != S) throw std::runtime_error("interceptor::ComputeTraj: state must have 12 entries");
    if (AdaptiveIntegrator()) {
        // adaptive steps (an extension, see models_interceptor.hpp): end state only; the chart flag is not reported
        mstate Xf(S);
        if (socp_integrate_batch(ctx, 1, &t0, &tf, nullptr, X0.data(), Xf.data(), 0) != SOCP_OK)
            throw std::runtime_error(std::string("interceptor::ComputeTraj: ") + socp_last_error(ctx));
        const real t1 = data->parameters.propellant_mass / data->parameters.q;
        data->stageMode = (t0 < t1 && !(tf > t1)) ? 1 : 0;
        data->currentChart = 1;
        return Xf;
    }
    // rows the reference traces: per stage 1 + stepNbr; plus the returned state
    const int cap = 2 * (data->stepNbr + 1) + 1;
    std::vector<double> dense((size_t)cap * S), times(cap), aux((size_t)cap * 2);
    int rows = 0;
    if (socp_integrate_dense_aux(ctx, t0, tf, nullptr, X0.data(), dense.data(), times.data(), aux.data(), cap, &rows) != SOCP_OK)
        throw std::runtime_error(std::string("interceptor::ComputeTraj: ") + socp_last_error(ctx));
    if (rows < 1 || rows > cap) throw std::runtime_error("interceptor::ComputeTraj: unexpected row count from the device");
    const int last = rows - 1;
    data->stageMode = (int)aux[2 * last];
    data->currentChart = (int)aux[2 * last + 1];
    if (isTrace) {
        // interceptor::Trace (interceptor.cpp:131-151): t, X, control (u, beta), H, chart -- control and H in each row's
        // own chart and stage, one batched evaluation for all rows
        const int nr = last;
        std::vector<double> u((size_t)nr * 2), H(nr);
        if (socp_eval_batch(ctx, SOCP_EVAL_CONTROL, nr, times.data(), aux.data(), dense.data(), S, u.data(), 0) != SOCP_OK ||
            socp_eval_batch(ctx, SOCP_EVAL_HAMILTONIAN, nr, times.data(), aux.data(), dense.data(), S, H.data(), 0) != SOCP_OK)
            throw std::runtime_error(std::string("interceptor::ComputeTraj: ") + socp_last_error(ctx));
        std::stringstream ss;
        for (int k = 0; k < nr; k++) {
            ss << times[k] << "\t";
            for (int j = 0; j < S; j++) ss << dense[(size_t)k * S + j] << "\t";
            ss << u[2 * k] << "\t" << u[2 * k + 1] << "\t";
            ss << H[k] << "\t";
            ss << (int)aux[2 * k + 1] << std::endl;
        }
        std::ofstream fileTrace(data->strFileTrace.c_str(), std::ios::app);
        fileTrace << ss.str();
    }
    return mstate(dense.begin() + (size_t)last * S, dense.begin() + (size_t)(last + 1) * S);
}

// interceptor.cpp:221-245
void interceptor::FinalFunction(real const &, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int) const
{
    const int n = data->n;
    for (int j = 0; j < n; j++) {
        if (mode_X[j] == 1) {
            fvec[j] = X_tf[j + n];
            if (j == 1) fvec[j] = X_tf[j + n] + data->parameters.muV;
        } else {
            fvec[j] = X_tf[j] - Xf[j];
            if (j == 0) fvec[j] = fvec[j] / data->parameters.hr;
            if (j == 3 && std::fabs(std::cos(Xf[2])) < 1e-5) fvec[j] = X_tf[j + n];
        }
    }
}

// interceptor.cpp:248-272
void interceptor::FinalHFunction(real const &tf, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const
{
    FinalFunction(tf, X_tf, Xf, mode_X, fvec, isJac);
    fvec[data->n] = Hamiltonian(tf, X_tf, isJac)[0] + data->parameters.muT;
}

// interceptor.cpp:846-950.  Closed-form guess of the initial costate from the line-of-sight geometry to the
// rendez-vous point; set-up arithmetic, once per problem.
void interceptor::InitAnalytical(real const &ti, mstate &Xi, real const &, mstate &Xf) const
{
    using std::sin; using std::cos; using std::tan; using std::exp; using std::sqrt; using std::acos; using std::asin; using std::fabs;
    const parameters_struct &P = data->parameters;
    const real h = Xi[0], v = Xi[1], gamma = Xi[2], chi = Xi[3], L = Xi[4], l = Xi[5];
    const real hf = Xf[0], gammaf = Xf[2], chif = Xf[3], Lf = Xf[4], lf = Xf[5];
    const real mass = ComputeMass(ti, Xi);
    const real c_max = P.c0 * exp(-h / P.hr) * (P.propellant_mass + P.empty_mass) / mass;
    const real d = P.d0 * exp(-h / P.hr) * (P.propellant_mass + P.empty_mass) / mass;
    const real r = h + data->R_Earth, rf = hf + data->R_Earth;
    const real eta = P.eta, hr = P.hr;
    const real b = sqrt(c_max * d / (2 * eta));
    const real cL = cos(L), sL = sin(L), cl = cos(l), sl = sin(l), cLf = cos(Lf), sLf = sin(Lf), clf = cos(lf), slf = sin(lf);
    const real sg = sin(gamma), cg = cos(gamma), sc = sin(chi), cc = cos(chi), tg = tan(gamma);
    // line of sight to the rendez-vous point in the Earth frame
    const real ex = rf * cLf * clf - r * cL * cl, ey = rf * cLf * slf - r * cL * sl, ez = rf * sLf - r * sL;
    const real R = sqrt(ex * ex + ey * ey + ez * ez);
    const real Rdot = -(ex * (sg * cL * cl - cg * cc * sL * cl - cg * sc * sl)
                        + ey * (sg * cL * sl - cg * cc * sL * sl + cg * sc * cl)
                        + ez * (sg * sL + cL * cg * cc)) / R;
    const real bdot = -c_max * d * sg * sqrt(2 * eta / (c_max * d)) / (2 * eta * hr);
    const real bR = b * R, ep = exp(bR), em = exp(-bR), w = bdot * R + b * Rdot;
    const real N1 = ep - em - 2 * b * R, D1 = 4 + ep * (bR - 2) - em * (bR + 2);
    const real dN1 = w * (ep + em - 2), dD1 = w * (ep * (bR - 2) + (em * (bR + 2)) + ep - em);
    const real k1 = b * R * (ep - em - 2 * b * R) / (4 + ep * (bR - 2) - em * (bR + 2));
    const real k1dot = w * N1 / D1 + (b * R * (dN1 * D1 - dD1 * N1) / (D1 * D1));
    const real N2 = ep * (bR - 1) + em * (bR + 1), D2 = 4 + ep * (bR - 2) - em * (bR + 2);
    const real dN2 = b * R * w * (ep - em), dD2 = w * (ep * (bR - 2) + (em * (bR + 2)) + ep - em);
    const real k2 = b * R * (ep * (bR - 1) + em * (bR + 1)) / (4 + ep * (bR - 2) - em * (bR + 2));
    const real k2dot = w * N2 / D2 + (b * R * (dN2 * D2 - dD2 * N2) / (D2 * D2));
    const real k3 = 2 + k1 - k2, k3dot = k1dot - k2dot;
    // elevation of the line of sight
    real l1;
    const real dist = fabs(rf * (cL * cLf * cl * clf + cL * cLf * sl * slf + sL * sLf) - r);
    const real x_E_R = -cL * cl * ex - cL * sl * ey - sL * ez;
    if (R == 0) l1 = gammaf;
    else if (dist / R >= 1 && x_E_R > 0) l1 = -M_PI / 2.0;
    else if (dist / R >= 1) l1 = M_PI / 2.0;
    else if (x_E_R > 0) l1 = -asin(dist / R);
    else l1 = asin(dist / R);
    // azimuth of the line of sight
    real l2;
    const real tlam2 = r - rf * (cLf * clf * cL * cl + cLf * slf * cL * sl + sLf * sL);
    const real px = rf * cLf * clf + (tlam2 - r) * cL * cl, py = rf * cLf * slf + (tlam2 - r) * cL * sl, pz = rf * sLf + (tlam2 - r) * sL;
    const real normProj = sqrt(px * px + py * py + pz * pz);
    const real prodScal = -px * sL * cl - py * sL * sl + pz * cL;
    const real coordProj_el = -sl * px + cl * py;
    if (normProj == 0) l2 = 0;
    else if (prodScal / normProj <= -1) l2 = M_PI;
    else if (prodScal / normProj >= 1) l2 = 0;
    else if (coordProj_el >= 0) l2 = acos(prodScal / normProj);
    else l2 = -acos(prodScal / normProj);
    const real u1 = -(k1 * (gammaf - l1) / R + k2 * sin(gamma - l1) / R + k3 * cg / (2 * hr)) / c_max;
    const real u2 = -(k1 * (chif - l2) * cg / R + k2 * sin(chi - l2) * cg / R) / c_max;
    const real d1DivC = sg / (c_max * hr);
    const real s1 = sin(gamma - l1), c1 = cos(gamma - l1), s2 = sin(chi - l2), c2 = cos(chi - l2);
    const real du1 = d1DivC * c_max * u1 -
                     (k1dot * (gammaf - l1) / R + k1 * s1 / (R * R) - k1 * (gammaf - l1) * Rdot / (R * R) + k2dot * s1 / R +
                      k2 * c1 * (c_max * u1 + s1 / R) / R - k2 * s1 * Rdot / (R * R) + k3dot * cg / (2 * hr) -
                      c_max * u1 * k3 * sg / (2 * hr)) / c_max;
    const real du2 = d1DivC * c_max * u2 -
                     (k1dot * cg * (chif - l2) / R - k1 * c_max * u1 * sg * (chif - l2) / R + k1 * cg * s2 / (R * R) -
                      k1 * cg * (chif - l2) * Rdot / (R * R) + k2dot * cg * s2 / R - k2 * sg * s2 * c_max * u1 / R +
                      k2 * cg * c2 * (c_max * u2 / cg + s2 / R) / R - k2 * cg * s2 * Rdot / (R * R)) / c_max;
    const real pg = 2 * eta * u1, pc = 2 * eta * u2 * cg;
    Xi[7] = -1;
    Xi[8] = pg;
    Xi[9] = pc;
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
