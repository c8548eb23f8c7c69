// odeTools.cpp -- host side of the ODE toolbox mirror: segments of models with device dynamics go to the GPU; the one-step
// helpers and the integration of user classes WITHOUT a device twin (only the reference's host virtuals) run here.
#include "odeTools.hpp"

#include <cfloat>
#include <cmath>
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

// The adaptive branch of the reference's integrate() (odeTools.cpp:103-123 with an observer, :129-134 without) on the host, for
// objects without device dynamics: integrate_adaptive(make_dense_output<runge_kutta_dopri5>(tol, tol), ode, X, t0, tf, dt).
// [ext] Boost.Odeint is not vendored: this is its published algorithm as the device kernels state it (integrator.hpp:
// Lane::dopri5_try / integrate_dopri5 -- same tableau, same left-to-right stage sums, same controller, same step budget), so a
// class with and without a device twin walks the same steps.  PARITY UNPINNED.  `seen` is shown (X, t) at t0
// and after every accepted step.
template <class Seen>
void host_integrate_adaptive(odeTools::modelStruct const &ode, odeTools::odeVector &X, double t0, double tf, double dt, double tol, Seen &&seen)
{
    warn_host_path_once();
    typedef odeTools::odeVector vec;
    const size_t n = X.size();
    const double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
    const double b21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40, b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9,
                 b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729,
                 b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176, b65 = -5103.0 / 18656,
                 c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
    const double dc1 = 35.0 / 384 - 5179.0 / 57600, dc3 = 500.0 / 1113 - 7571.0 / 16695, dc4 = 125.0 / 192 - 393.0 / 640,
                 dc5 = -2187.0 / 6784 - -92097.0 / 339200, dc6 = 11.0 / 84 - 187.0 / 2100, dc7 = -1.0 / 40;
    const double eps = DBL_EPSILON;
    auto rhs = [&](double t, vec const &Y, vec &F) { F = Y; ode(Y, F, t); };          // (the output starts as a copy of its input: step_ode)
    double t = t0, h = dt;
    seen(X, t);
    if (!(h > 0)) return;                                    // zero-length / backward segment: no step
    vec k1(n), k2(n), k3(n), k4(n), k5(n), k6(n), kn(n), xn(n), y(n);
    bool have_k1 = false;
    long budget = 50000;                                     // trial steps per segment (integrator.hpp: kAdaptiveStepBudget)
    auto poison = [&]() { for (size_t i = 0; i < n; i++) X[i] = std::nan(""); };
    while (tf - t > eps && budget > 0) {
        while (t + h - tf <= eps && budget > 0) {
            if (!have_k1) { rhs(t, X, k1); have_k1 = true; }
            int tries = 0;
            bool ok = false;
            do {
                const double hh = h, tt = t;
                for (size_t i = 0; i < n; i++) y[i] = 1.0 * X[i] + hh * b21 * k1[i];
                rhs(tt + hh * a2, y, k2);
                for (size_t i = 0; i < n; i++) y[i] = 1.0 * X[i] + hh * b31 * k1[i] + hh * b32 * k2[i];
                rhs(tt + hh * a3, y, k3);
                for (size_t i = 0; i < n; i++) y[i] = 1.0 * X[i] + hh * b41 * k1[i] + hh * b42 * k2[i] + hh * b43 * k3[i];
                rhs(tt + hh * a4, y, k4);
                for (size_t i = 0; i < n; i++) y[i] = 1.0 * X[i] + hh * b51 * k1[i] + hh * b52 * k2[i] + hh * b53 * k3[i] + hh * b54 * k4[i];
                rhs(tt + hh * a5, y, k5);
                for (size_t i = 0; i < n; i++)
                    y[i] = 1.0 * X[i] + hh * b61 * k1[i] + hh * b62 * k2[i] + hh * b63 * k3[i] + hh * b64 * k4[i] + hh * b65 * k5[i];
                rhs(tt + hh, y, k6);
                for (size_t i = 0; i < n; i++)
                    xn[i] = 1.0 * X[i] + hh * c1 * k1[i] + hh * c3 * k3[i] + hh * c4 * k4[i] + hh * c5 * k5[i] + hh * c6 * k6[i];
                rhs(tt + hh, xn, kn);
                double err = 0;
                for (size_t i = 0; i < n; i++) {
                    double e = hh * dc1 * k1[i] + hh * dc3 * k3[i] + hh * dc4 * k4[i] + hh * dc5 * k5[i] + hh * dc6 * k6[i] + hh * dc7 * kn[i];
                    e = std::fabs(e) / (tol + tol * (1.0 * std::fabs(X[i]) + 1.0 * hh * std::fabs(k1[i])));
                    if (e > err || e != e) err = e;
                }
                budget--;
                if (!(err <= 1.0)) {                         // reject (also on NaN)
                    double f = 0.9 * std::pow(err, -1.0 / 3.0);
                    if (!(f > 0.2)) f = 0.2;
                    h = hh * f;
                } else {
                    t = tt + hh;
                    if (err < 0.5) {
                        const double floor5 = 1.0 / 3125.0;
                        h = hh * (0.9 * std::pow(err > floor5 ? err : floor5, -1.0 / 5.0));
                    }
                    ok = true;
                }
            } while (!ok && ++tries < 500);
            if (!ok) { poison(); return; }                   // odeint throws step_adjustment_error: the result is NaN here, as on the device
            X = xn;
            k1 = kn;
            seen(X, t);
        }
        h = tf - t;
        have_k1 = false;
    }
    if (budget <= 0 && tf - t > eps) poison();
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
        // no device twin: the reference's own loop over the host virtual, fixed step or adaptive (odeTools.cpp:129-146)
        if (AdaptiveIntegrator()) host_integrate_adaptive(_model, X, t0, tf, dt, odeIntTol, [](odeVector const &, real) {});
        else host_integrate(_model, X, t0, tf, dt, [](odeVector const &, real) {});
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
        if (AdaptiveIntegrator()) host_integrate_adaptive(_model, X, t0, tf, dt, odeIntTol, [&](odeVector const &Xs, real t) { _observer(Xs, t); });
        else host_integrate(_model, X, t0, tf, dt, [&](odeVector const &Xs, real t) { _observer(Xs, t); });
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
