// odeTools.cpp -- host side of the ODE toolbox mirror: segments of models with device dynamics go to the GPU; the one-step
// helpers and the integration of user classes WITHOUT a device twin (only the reference's host virtuals) run here.
#include "odeTools.hpp"

#include <iostream>
#include <stdexcept>

#include "model.hpp"
#include "socp_hip.h"

odeTools::odeVector odeTools::MultState(real a, odeVector const &X)
{
    odeVector Y(X);
    for (real &y : Y) y *= a;
    return Y;
}

odeTools::odeVector odeTools::AddState(odeVector const &X, odeVector const &Y)
{
    odeVector Z(X);
    for (size_t i = 0; i < Z.size(); i++) Z[i] += Y[i];
    return Z;
}

namespace {
#ifdef _USE_BOOST
bool g_adaptive = true;
#else
bool g_adaptive = false;
#endif
}  // namespace

void odeTools::UseAdaptiveIntegrator(bool on) { g_adaptive = on; }
bool odeTools::AdaptiveIntegrator() { return g_adaptive; }

namespace {
// nullptr when the object has no device dynamics (a user class that only overrides the reference's host virtuals)
model *device_model(odeTools *ode)
{
    model *m = dynamic_cast<model *>(ode);
    return (m && m->DeviceModelId() != 0) ? m : nullptr;
}

void warn_host_path_once()
{
    static bool said = false;
    if (said) return;
    said = true;
    std::cerr << "socp_amd: this model has no device dynamics (model::DeviceModelId() == 0): its host virtual Model() is "
                 "integrated on the CPU, one call at a time -- give it a device twin (include/socp_plugin.h) to run on the GPU"
              << std::endl;
}

// One explicit Runge-Kutta step of the reference's three one-step helpers (odeTools.cpp:46-98), for any right-hand side
// f(t, X) -> dX/dt.  Operation order as there: stage states X + (step/2.0) F, stage times t + step/2.0 (twice) and t + step,
// RK4 update X + (step/6.0) (F1 + (F4 + 2.0 (F2 + F3))).
template <class Rhs>
odeTools::odeVector rk_step(int order, real t, odeTools::odeVector const &X, real step, Rhs &&f)
{
    const size_t n = X.size();
    odeTools::odeVector Y(n), out(n);
    const odeTools::odeVector F1 = f(t, X);
    if (order == 1) {
        for (size_t i = 0; i < n; i++) out[i] = X[i] + step * F1[i];
        return out;
    }
    const real half = step / 2.0;
    for (size_t i = 0; i < n; i++) Y[i] = X[i] + half * F1[i];
    const odeTools::odeVector F2 = f(t + step / 2.0, Y);
    if (order == 2) {
        for (size_t i = 0; i < n; i++) out[i] = X[i] + step * F2[i];
        return out;
    }
    for (size_t i = 0; i < n; i++) Y[i] = X[i] + half * F2[i];
    const odeTools::odeVector F3 = f(t + step / 2.0, Y);
    for (size_t i = 0; i < n; i++) Y[i] = X[i] + step * F3[i];
    const odeTools::odeVector F4 = f(t + step, Y);
    const real sixth = step / 6.0;
    for (size_t i = 0; i < n; i++) out[i] = X[i] + sixth * (F1[i] + (F4[i] + 2.0 * (F2[i] + F3[i])));
    return out;
}

typedef odeTools::odeVector (*rhs_fn)(real const &, odeTools::odeVector const &, void *);
odeTools::odeVector step_fn(int order, real t, odeTools::odeVector const &X, real step, rhs_fn fn, void *context)
{
    return rk_step(order, t, X, step, [&](real tt, odeTools::odeVector const &Y) { return fn(tt, Y, context); });
}
void step_ode(int order, real t, odeTools::odeVector &X, real step, odeTools::modelStruct const &ode)
{
    // every stage's output vector starts as a copy of the step's INPUT state, as in the reference (odeTools.cpp:51,68,89:
    // F1 = X, F2 = X, ...): a user operator() that leaves components unwritten then sees the same numbers in stages 2-4
    const odeTools::odeVector X0(X);
    X = rk_step(order, t, X0, step, [&](real tt, odeTools::odeVector const &Y) {
        odeTools::odeVector F(X0);
        ode(Y, F, tt);
        return F;
    });
}

// odeTools.cpp:128-146 on the host, for objects without device dynamics: t by t += dt, last step clamped to tf - t, no step
// for a zero-length or backward segment.  `seen` (may be null) is shown (X, t) at t0 and after every step (:103-123).
template <class Seen>
void host_integrate(odeTools::modelStruct const &ode, odeTools::odeVector &X, double t0, double tf, double dt, Seen &&seen)
{
    warn_host_path_once();
    real t = t0;
    seen(X, t);
    while (t < (tf - dt / 2)) {
        step_ode(4, t, X, (t + dt > tf) ? (tf - t) : dt, ode);
        t += dt;
        seen(X, t);
    }
}
}  // namespace

// The one-step helpers of the reference API (odeTools.hpp:107-165) run on the host: they take arbitrary host callbacks
// (interceptor.cpp:117 calls RK4 with a function pointer).  They are utilities for user code, not the integration path of
// models with device dynamics, which goes through integrate() -> the GPU.
odeTools::odeVector odeTools::RK1(real const &t, odeVector const &X, real const &step, odeVector (*function)(real const &, odeVector const &, void *), void *context) { return step_fn(1, t, X, step, function, context); }
void odeTools::RK1(real const &t, odeVector &X, real const &step, modelStruct const &ode) { step_ode(1, t, X, step, ode); }
odeTools::odeVector odeTools::RK2(real const &t, odeVector const &X, real const &step, odeVector (*function)(real const &, odeVector const &, void *), void *context) { return step_fn(2, t, X, step, function, context); }
void odeTools::RK2(real const &t, odeVector &X, real const &step, modelStruct const &ode) { step_ode(2, t, X, step, ode); }
odeTools::odeVector odeTools::RK4(real const &t, odeVector const &X, real const &step, odeVector (*function)(real const &, odeVector const &, void *), void *context) { return step_fn(4, t, X, step, function, context); }
void odeTools::RK4(real const &t, odeVector &X, real const &step, modelStruct const &ode) { step_ode(4, t, X, step, ode); }

// odeTools.cpp:128-146.  The device kernel derives dt = (tf - t0)/stepNbr itself exactly as
// model::ModelInt does (model.hpp:398); a caller-chosen dt that differs from that is not
// representable and is rejected rather than silently replaced.
void odeTools::integrate(modelStruct const &_model, odeVector &X, double const &t0, double const &tf, double const &dt)
{
    model *m = device_model(_model.m_ode);
    if (!m) {
        // no device twin: the reference's own loop over the host virtual (fixed-step only; the adaptive integrator is a device kernel)
        if (AdaptiveIntegrator()) throw std::runtime_error("odeTools::integrate: the adaptive integrator needs device dynamics");
        host_integrate(_model, X, t0, tf, dt, [](odeVector const &, real) {});
        return;
    }
    const double dt_model = (tf - t0) / m->DeviceStepNumber();
    if (!(dt == dt_model))
        throw std::runtime_error("odeTools::integrate: dt must equal (tf - t0)/DeviceStepNumber() for the device path");
    socp_ctx *ctx = m->DeviceContext();
    odeVector Xf(X.size());
    if (socp_integrate_batch(ctx, 1, &t0, &tf, nullptr, X.data(), Xf.data(), _model.m_isJac) != SOCP_OK)
        throw std::runtime_error(std::string("odeTools::integrate: ") + socp_last_error(ctx));
    X.swap(Xf);
}

// odeTools.cpp:103-123: same segment, the observer sees the state at t0 and after every step
void odeTools::integrate(modelStruct const &_model, odeVector &X, double const &t0, double const &tf, double const &dt, observerStruct const &_observer)
{
    model *m = device_model(_model.m_ode);
    if (!m) {
        if (AdaptiveIntegrator()) throw std::runtime_error("odeTools::integrate: the adaptive integrator needs device dynamics");
        host_integrate(_model, X, t0, tf, dt, [&](odeVector const &Xs, real t) { _observer(Xs, t); });
        return;
    }
    const double dt_model = (tf - t0) / m->DeviceStepNumber();
    if (!(dt == dt_model))
        throw std::runtime_error("odeTools::integrate: dt must equal (tf - t0)/DeviceStepNumber() for the device path");
    if (_model.m_isJac) throw std::runtime_error("odeTools::integrate: tracing the variational state is not supported");
    socp_ctx *ctx = m->DeviceContext();
    const int S = (int)X.size();
    // fixed step: the device loop stops after at most stepNbr + 8 steps (integrator.hpp); adaptive: the number of accepted steps is
    // only known afterwards -- start with room for 512 and ask again when the trajectory took more
    int cap = AdaptiveIntegrator() ? 512 : m->DeviceStepNumber() + 10;
    std::vector<double> dense, times;
    int rows = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        dense.assign((size_t)cap * S, 0.0);
        times.assign(cap, 0.0);
        if (socp_integrate_dense(ctx, t0, tf, nullptr, X.data(), dense.data(), times.data(), cap, &rows) != SOCP_OK)
            throw std::runtime_error(std::string("odeTools::integrate: ") + socp_last_error(ctx));
        if (rows <= cap) break;
        if (!AdaptiveIntegrator()) throw std::runtime_error("odeTools::integrate: the device reported more rows than the step guard allows");
        cap = rows;
    }
    for (int k = 0; k < rows; k++) {
        odeVector row(dense.begin() + (size_t)k * S, dense.begin() + (size_t)(k + 1) * S);
        _observer(row, times[k]);
        if (k == rows - 1) X = row;
    }
}
