// api_conventions.cpp -- the reference's error and ownership conventions, kept by the C++ mirror:
//   api_conventions bad_multi | bad_thread   constructor with a count < 1: message on stderr, exit(1) (shooting.cpp:62-77)
//   api_conventions checks                   everything that returns: prints one "ok <name>" line per convention
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "socp/shooting.hpp"
#include "models/goddard/goddard.hpp"

namespace {
// a user model without device dynamics: accepted by the API, rejected at the first integration (no CPU path)
class hostOnly : public model
{
public:
    hostOnly() : model(2) {}
    mstate Model(real const &, mstate const &X, int) const override { return mstate(X.size(), 1.0); }
    mcontrol Control(real const &, mstate const &) const override { return mcontrol(1, 0.0); }
    mstate Hamiltonian(real const &, mstate const &, int) const override { return mstate(1, 0.0); }
};

void ok(const char *name) { std::printf("ok %s\n", name); std::fflush(stdout); }
}  // namespace

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "checks";
    goddard g("");
    if (mode == "bad_multi") { shooting s(g, 0, 1); return 0; }
    if (mode == "bad_thread") { shooting s(g, 1, 0); return 0; }

    g.SetParameterDataName("mu2", 1.0);
    shooting sh(g, 1, 1);
    std::vector<int> mode_X(g.GetDim(), 0);
    mode_X[3] = mode_X[4] = mode_X[5] = mode_X[6] = 1;
    sh.SetMode(0, mode_X);
    model::mstate Xi(14, 0.1), Xf(14, 0.0);
    const double x0[7] = {0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0};
    const double p0[7] = {-8.121947733, 7.775439382e-3, 0.7775438809, -0.4779369965, 5.715013318e-4, 5.715009222e-2, 9.958404873e-2};
    for (int k = 0; k < 7; k++) { Xi[k] = x0[k]; Xi[7 + k] = p0[k] * 1.0005; }
    Xf[0] = 1.01;
    sh.InitShooting(0.0, Xi, 0.2640825, Xf);

    // unknown continuation parameter name: message, 0 = "improper input" (shooting.hpp:120-128)
    if (sh.SolveOCP(0.1, std::string("noSuchParameter"), 1.0) == 0) ok("unknown_parameter_name");

    // std::map::at on an unknown model parameter throws std::out_of_range (goddard.cpp:380-387)
    try { g.GetParameterDataName("noSuchParameter"); } catch (const std::out_of_range &) { ok("out_of_range"); }

    // solver failure is a return code, 1 is the only success value (shooting.cpp:588); nfev/njev are surfaced
    const int info = sh.SolveOCP(0.0);
    const std::vector<int> calls = sh.GetCallNumber();
    if (info == 1 && calls.size() == 2 && calls[0] > 0) ok("solve_returns_1");

    // GetParameters() hands out a new[] the caller frees (shooting.cpp:463-470)
    real *p = sh.GetParameters();
    std::vector<real> v;
    sh.GetParameters(v);
    if (p && v.size() == 14 && std::memcmp(p, v.data(), sizeof(real) * 14) == 0 && p[0] == sh.GetParameters(0)) ok("get_parameters_new_array");
    delete[] p;

    // watchdog overload: a solve that cannot finish in time is aborted through the callback's return value, -1 (shooting.cpp:329-348)
    g.stepNbr = 2000000;
    for (int k = 0; k < 7; k++) Xi[7 + k] = p0[k] * 1.002;
    sh.InitShooting(0.0, Xi, 0.2640825, Xf);
    if (sh.SolveOCP(0.0, 5.0) == -1) ok("timeout_returns_minus_1");
    g.stepNbr = 10;

    // a model class without device dynamics (only the reference's host virtuals): integrated on the host by the reference's
    // own loop (odeTools.cpp:128-146), with a warning -- x' = 1 for every component here, so X(1) = X(0) + 1 up to rounding
    hostOnly h;
    const model::mstate Xh = h.ComputeTraj(0.0, model::mstate(4, 0.5), 1.0, 0, 0);
    if (Xh.size() == 4 && std::fabs(Xh[0] - 1.5) < 1e-14 && std::fabs(Xh[3] - 1.5) < 1e-14) ok("no_device_twin_runs_on_host");
    // ... and with the adaptive integrator too (odeTools.cpp:129-134 on the host: the reference's -D_USE_BOOST build): x' = 1 is
    // integrated exactly by any Runge-Kutta step, whatever the controller makes of the step sizes
    odeTools::UseAdaptiveIntegrator(true);
    const model::mstate Xa = h.ComputeTraj(0.0, model::mstate(4, 0.5), 1.0, 0, 0);
    if (Xa.size() == 4 && std::fabs(Xa[0] - 1.5) < 1e-14 && std::fabs(Xa[3] - 1.5) < 1e-14) ok("no_device_twin_adaptive_runs_on_host");
    odeTools::UseAdaptiveIntegrator(false);
    // the one-step host helpers run host callbacks (odeTools.cpp:46-98; interceptor.cpp:117 uses the function-pointer form)
    struct Cb { static odeTools::odeVector f(real const &, odeTools::odeVector const &X, void *) { return odeTools::odeVector(X.size(), 2.0); } };
    const odeTools::odeVector Y = odeTools::RK4(0.0, odeTools::odeVector(3, 1.0), 0.25, &Cb::f, nullptr);
    if (Y.size() == 3 && Y[0] == 1.5 && Y[2] == 1.5) ok("host_rk_helpers_run");
    return 0;
}
