// model.cpp -- host side of the `model` mirror: default residual blocks and the device hook.
#include "model.hpp"

#include <stdexcept>

#include "socp_hip.h"

model::model(int const &_stateDim, int _modelOrder, int _stepNbr, std::string _fileTrace)
    : dim(_stateDim), modelOrder(_modelOrder), strFileTrace(_fileTrace), stepNbr(_stepNbr)
{
    parameters.clear();
    std::ofstream wipe(strFileTrace.c_str(), std::ios::trunc);   // model.hpp:51-54: start from an empty trace file
}

model::~model()
{
    if (deviceCtx_) socp_ctx_destroy(deviceCtx_);
}

socp_ctx *model::DeviceContext() const
{
    const int id = DeviceModelId();
    if (id == 0)
        throw std::runtime_error("model: no device dynamics registered for this class (DeviceModelId() == 0); "
                                 "socp_amd integrates on the GPU only");
    if (!deviceCtx_) {
        if (socp_ctx_create(&deviceCtx_, id, -1) != SOCP_OK)
            throw std::runtime_error(std::string("model: cannot create device context: ") + socp_last_error(nullptr));
    }
    // the continuation loop mutates parameters through a raw real& (shooting.cpp:695-707) and users
    // poke stepNbr / switching times directly: re-pack on every use, never cache
    double p[SOCP_MAX_NPARAMS];
    const int np = DeviceParams(p, SOCP_MAX_NPARAMS);
    if (socp_ctx_set_params(deviceCtx_, p, np) != SOCP_OK || socp_ctx_set_step_number(deviceCtx_, DeviceStepNumber()) != SOCP_OK)
        throw std::runtime_error(std::string("model: ") + socp_last_error(deviceCtx_));
    if (socp_ctx_set_integrator(deviceCtx_, AdaptiveIntegrator() ? SOCP_INT_DOPRI5 : SOCP_INT_RK4, odeIntTol) != SOCP_OK)
        throw std::runtime_error(std::string("model: ") + socp_last_error(deviceCtx_));
    if (deviceVariant_ >= 0 && socp_ctx_set_variant(deviceCtx_, deviceVariant_) != SOCP_OK)
        throw std::runtime_error(std::string("model: ") + socp_last_error(deviceCtx_));
    const std::vector<real> sw = DeviceSwitchingTimes();
    socp_ctx_set_switching_times(deviceCtx_, sw.data(), (int)sw.size());
    return deviceCtx_;
}

model::mstate model::DeviceEval(int what, real const &t, mstate const &X, int isJac) const
{
    socp_ctx *ctx = DeviceContext();
    const int s = 2 * dim;
    const int out_len = what == SOCP_EVAL_RHS ? (int)X.size() : (what == SOCP_EVAL_CONTROL ? socp_ctx_control_dim(ctx) : (isJac ? s + 1 : 1));
    mstate out(out_len);
    if (socp_eval_batch(ctx, what, 1, &t, nullptr, X.data(), (int)X.size(), out.data(), isJac) != SOCP_OK)
        throw std::runtime_error(std::string("model: ") + socp_last_error(ctx));
    return out;
}

// model.hpp:395-414
model::mstate model::ModelInt(real const &t0, mstate const &X, real const &tf, int isTrace, int isJac)
{
    const real dt = (tf - t0) / stepNbr;
    mstate Xs = X;
    if (isTrace) {
        std::stringstream ss;
        integrate(modelStruct(this, isJac), Xs, t0, tf, dt, observerStruct(this, ss));
        std::ofstream fileTrace(strFileTrace.c_str(), std::ios::app);
        fileTrace << ss.str();
    } else {
        integrate(modelStruct(this, isJac), Xs, t0, tf, dt);
    }
    return Xs;
}

namespace {
template <class Stream>
void trace_row(const model &m, real const &t, model::mstate const &X, Stream &file)
{
    // model.hpp:422-462: t, X[0..2d), control, H -- tab separated, one row per call
    const model::mstate u = m.Control(t, X);
    const real H = m.Hamiltonian(t, X, 0)[0];
    file << t << "\t";
    for (int k = 0; k < 2 * m.dim; k++) file << X[k] << "\t";
    for (size_t k = 0; k < u.size(); k++) file << u[k] << "\t";
    file << H << std::endl;
}
}  // namespace

void model::Trace(real const &t, mstate const &X, std::ofstream &file) const { trace_row(*this, t, X, file); }
void model::Trace(real const &t, mstate const &X, std::stringstream &file) const { trace_row(*this, t, X, file); }

// ---- default residual blocks (model.hpp:90-328) ------------------------------------------------
// isJac == 0: per component j, FREE -> transversality X[j+d] = 0, otherwise X[j] = target[j].
// isJac == 1: the same rows differentiated, read from the sensitivity block of the augmented state
// (row k of dX/dX0 sits at X[2d*(k+1) ...]).
namespace {
void rows_value(int d, model::mstate const &Xt, model::mstate const &target, std::vector<int> const &mode, std::vector<real> &f)
{
    for (int j = 0; j < d; j++) f[j] = (mode[j] == model::FREE) ? Xt[j + d] : Xt[j] - target[j];
}
void rows_sens(int d, int stride, model::mstate const &Xt, std::vector<int> const &mode, std::vector<real> &f)
{
    const int s = 2 * d;
    for (int j = 0; j < d; j++) {
        const int src = (mode[j] == model::FREE) ? (j + d + 1) : (j + 1);
        for (int i = 0; i < s; i++) f[stride * j + i] = Xt[s * src + i];
    }
}
}  // namespace

void model::FinalFunction(real const &, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const
{
    if (isJac == 0) rows_value(dim, X_tf, Xf, mode_X, fvec);
    else rows_sens(dim, 2 * dim, X_tf, mode_X, fvec);
}

void model::InitialFunction(real const &, mstate const &X_t0, mstate const &X0, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const
{
    if (isJac == 0) rows_value(dim, X_t0, X0, mode_X, fvec);
    else rows_sens(dim, 2 * dim, X_t0, mode_X, fvec);
}

namespace {
// model.hpp:133-185 / :239-290: boundary rows plus the H = 0 row of a free boundary time
void h_block(const model &m, real const &t, model::mstate const &Xt, model::mstate const &target,
             std::vector<int> const &mode, std::vector<real> &f, int isJac)
{
    const int d = m.dim, s = 2 * d, w = s + 1;
    if (isJac == 0) {
        rows_value(d, Xt, target, mode, f);
        f[d] = m.Hamiltonian(t, Xt, 0)[0];
        return;
    }
    const model::mstate state(Xt.begin(), Xt.begin() + s);
    const model::mstate fx = m.Model(t, state, 0);
    rows_sens(d, w, Xt, mode, f);
    for (int j = 0; j < d; j++) f[w * j + s] = (mode[j] == model::FREE) ? fx[j + d] : fx[j];
    const model::mstate dH = m.Hamiltonian(t, state, 1);
    for (int i = 0; i < s; i++) {
        real acc = 0;
        for (int k = 0; k < s; k++) acc += dH[k] * Xt[s * (k + 1) + i];
        f[w * d + i] = acc;
    }
    real acc = 0;
    for (int k = 0; k < s; k++) acc += dH[k] * fx[k];
    f[w * d + s] = acc + dH[s];
}
}  // namespace

void model::FinalHFunction(real const &tf, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const
{
    h_block(*this, tf, X_tf, Xf, mode_X, fvec, isJac);
}

void model::InitialHFunction(real const &t0, mstate const &X_t0, mstate const &X0, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const
{
    h_block(*this, t0, X_t0, X0, mode_X, fvec, isJac);
}

// model.hpp:299-328: H(t, X-) - H(t, X+) at a free interior time, or its derivative row
model::mstate model::SwitchingTimesFunction(real const &t, mstate const &X, mstate const &Xp, int isJac) const
{
    if (isJac == 0) return mstate(1, Hamiltonian(t, X, 0)[0] - Hamiltonian(t, Xp, 0)[0]);
    const int s = 2 * dim;
    mstate f(2 * s + 1, 0);
    const mstate a(X.begin(), X.begin() + s), b(Xp.begin(), Xp.begin() + s);
    const mstate fa = Model(t, a, 0), fb = Model(t, b, 0);
    const mstate dHa = Hamiltonian(t, a, 1), dHb = Hamiltonian(t, b, 1);
    for (int i = 0; i < s; i++)
        for (int k = 0; k < s; k++) {
            f[i] += dHa[k] * X[s * (k + 1) + i];
            f[s + i] -= dHb[k] * Xp[s * (k + 1) + i];
        }
    for (int k = 0; k < s; k++) f[2 * s] += dHa[k] * fa[k] - dHb[k] * fb[k];
    f[2 * s] += dHa[s] - dHb[s];
    return f;
}
