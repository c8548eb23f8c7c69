// odeTools.cpp -- host side of the ODE toolbox mirror: segments go to the device.
#include "odeTools.hpp"

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
[[noreturn]] void no_host_rk(const char *name)
{
    throw std::logic_error(std::string("odeTools::") + name +
                           ": one-step host helpers take arbitrary host callbacks and are not part of the device "
                           "path; integrate whole segments with integrate()/model::ComputeTraj");
}
model *device_model(odeTools *ode)
{
    model *m = dynamic_cast<model *>(ode);
    if (!m || m->DeviceModelId() == 0)
        throw std::runtime_error("odeTools::integrate: this object has no device dynamics (model::DeviceModelId() == 0); "
                                 "socp_amd has no CPU integration path");
    return m;
}
}  // namespace

odeTools::odeVector odeTools::RK1(real const &, odeVector const &, real const &, odeVector (*)(real const &, odeVector const &, void *), void *) { no_host_rk("RK1"); }
void odeTools::RK1(real const &, odeVector &, real const &, modelStruct const &) { no_host_rk("RK1"); }
odeTools::odeVector odeTools::RK2(real const &, odeVector const &, real const &, odeVector (*)(real const &, odeVector const &, void *), void *) { no_host_rk("RK2"); }
void odeTools::RK2(real const &, odeVector &, real const &, modelStruct const &) { no_host_rk("RK2"); }
odeTools::odeVector odeTools::RK4(real const &, odeVector const &, real const &, odeVector (*)(real const &, odeVector const &, void *), void *) { no_host_rk("RK4"); }
void odeTools::RK4(real const &, odeVector &, real const &, modelStruct const &) { no_host_rk("RK4"); }

// odeTools.cpp:128-146.  The device kernel derives dt = (tf - t0)/stepNbr itself exactly as
// model::ModelInt does (model.hpp:398); a caller-chosen dt that differs from that is not
// representable and is rejected rather than silently replaced.
void odeTools::integrate(modelStruct const &_model, odeVector &X, double const &t0, double const &tf, double const &dt)
{
    model *m = device_model(_model.m_ode);
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
    const double dt_model = (tf - t0) / m->DeviceStepNumber();
    if (!(dt == dt_model))
        throw std::runtime_error("odeTools::integrate: dt must equal (tf - t0)/DeviceStepNumber() for the device path");
    if (_model.m_isJac) throw std::runtime_error("odeTools::integrate: tracing the variational state is not supported");
    if (AdaptiveIntegrator()) throw std::runtime_error("odeTools::integrate: trace replay is available with the fixed-step integrator only");
    socp_ctx *ctx = m->DeviceContext();
    const int S = (int)X.size();
    const int cap = m->DeviceStepNumber() + 10;      // the device loop stops after at most stepNbr + 8 steps (integrator.hpp)
    std::vector<double> dense((size_t)cap * S), times(cap);
    int rows = 0;
    if (socp_integrate_dense(ctx, t0, tf, nullptr, X.data(), dense.data(), times.data(), cap, &rows) != SOCP_OK)
        throw std::runtime_error(std::string("odeTools::integrate: ") + socp_last_error(ctx));
    if (rows > cap) throw std::runtime_error("odeTools::integrate: the device reported more rows than the step guard allows");
    for (int k = 0; k < rows; k++) {
        odeVector row(dense.begin() + (size_t)k * S, dense.begin() + (size_t)(k + 1) * S);
        _observer(row, times[k]);
        if (k == rows - 1) X = row;
    }
}
