// hostmodel_flow.cpp -- a USER model written against the reference's plugin surface ONLY: it overrides the host virtuals
// Model / Control / Hamiltonian (odeTools.hpp:82, model.hpp:375,384) and knows nothing of a device (DeviceModelId() == 0).
// It must still solve through shooting::SolveOCP -- on the host, with a warning -- and the reference-style one-step calls
// RK4(t, X, dt, function, context) (interceptor.cpp:117) and RK1/RK2/RK4(t, X, dt, modelStruct) must work.
//   hostmodel_flow <numMulti> [numThread]        (no GPU needed; numThread > 1: segment workers on the host)
//   hostmodel_flow throws <numThread>            a model whose Model() throws inside one segment: the exception must reach the caller
//                                                of ResidualAt -- from a segment worker too (numThread > 1) -- and the shooting
//                                                object must still work afterwards
//   hostmodel_flow residual                      prints the residual of a 3-segment layout with FREE times (H rows, switching
//                                                row) and mixed state modes at its initial guess (shooting::ResidualAt)
// Problem: minimum-energy rest-to-rest transfer of a 1-D double integrator, x' = v, v' = u, cost = int u^2/2 dt;
// u = -p_v, p_x' = 0, p_v' = -p_x.  Analytic solution on [0, 1], x: 0 -> 1:  p_x = -12, p_v(0) = -6, u(0) = 6.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "socp/shooting.hpp"

class lqr_host : public model
{
public:
    lqr_host() : model(2, 0, 20, "") {}
    virtual mstate Model(real const &, mstate const &X, int) const
    {
        mstate d(4);
        d[0] = X[1]; d[1] = -X[3]; d[2] = 0; d[3] = -X[2];
        return d;
    }
    virtual mcontrol Control(real const &, mstate const &X) const { return mcontrol(1, -X[3]); }
    virtual mstate Hamiltonian(real const &, mstate const &X, int) const
    {
        const real u = -X[3];
        return mstate(1, u * u / 2 + X[2] * X[1] + X[3] * u);
    }
};

// the same dynamics, but Model() throws while the state's position is beyond a threshold: reached in ONE segment of the layout below
class lqr_throwing : public lqr_host
{
public:
    real limit = 1e300;
    virtual mstate Model(real const &t, mstate const &X, int isJac) const
    {
        if (X[0] > limit) throw std::out_of_range("lqr_throwing: position beyond the limit");
        return lqr_host::Model(t, X, isJac);
    }
};

static odeTools::odeVector oscillator(real const &t, odeTools::odeVector const &X, void *context)
{
    const real w = *static_cast<real *>(context);
    odeTools::odeVector d(2);
    d[0] = X[1];
    d[1] = -w * w * X[0] + t;
    return d;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 64;
    if (std::string(argv[1]) == "residual") {
        // F(z0) of a layout that exercises every row kind of the host assembly: FREE interior time (SwitchingTimesFunction
        // row), FREE final time (FinalHFunction's H row), FIXED / CONTINUOUS interior state modes, FREE final state component.
        lqr_host m;
        m.stepNbr = 7;
        const int M = 3;
        shooting sh(m, M, 1);
        std::vector<int> mode_t(M + 1, model::CONTINUOUS);
        mode_t[0] = model::FIXED; mode_t[1] = model::FREE; mode_t[M] = model::FREE;
        std::vector<std::vector<int> > mode_X(M + 1, std::vector<int>(2, model::CONTINUOUS));
        mode_X[0] = std::vector<int>(2, model::FIXED);
        mode_X[1][0] = model::FIXED;                       // way-point: position pinned, velocity continuous
        mode_X[M][0] = model::FIXED; mode_X[M][1] = model::FREE;
        sh.SetMode(mode_t, mode_X);
        std::vector<real> vt(M + 1);
        std::vector<model::mstate> vX(M + 1, model::mstate(4));
        for (int i = 0; i <= M; i++) {
            vt[i] = 0.4 * i + 0.01 * i * i;
            vX[i][0] = 0.3 * i; vX[i][1] = 0.1 + 0.05 * i; vX[i][2] = -1.0 - 0.1 * i; vX[i][3] = -0.7 + 0.2 * i;
        }
        sh.InitShooting(vt, vX);
        std::vector<real> z;
        sh.GetParameters(z);
        std::vector<real> F = sh.ResidualAt(z);
        std::printf("{\"n\": %d, \"z\": [", (int)z.size());
        for (size_t k = 0; k < z.size(); k++) std::printf("%s%.17g", k ? ", " : "", z[k]);
        std::printf("], \"F\": [");
        for (size_t k = 0; k < F.size(); k++) std::printf("%s%.17g", k ? ", " : "", F[k]);
        std::printf("]}\n");
        return 0;
    }
    if (std::string(argv[1]) == "throws") {
        // six segments whose node positions are 0, 1, ..., 5; Model() throws beyond 3.5: only the segments starting at nodes 4 and 5
        // throw, i.e. tasks that the worker threads take when there are several
        const int M = 6, threads = argc > 2 ? std::atoi(argv[2]) : 1;
        lqr_throwing m;
        shooting sh(m, M, threads);
        sh.SetMode(model::FIXED, std::vector<int>(2, model::FIXED));
        std::vector<real> vt(M + 1);
        std::vector<model::mstate> vX(M + 1, model::mstate(4, 0.0));
        for (int i = 0; i <= M; i++) { vt[i] = 0.1 * i; vX[i][0] = 1.0 * i; }
        sh.InitShooting(vt, vX);
        std::vector<real> z;
        sh.GetParameters(z);
        const std::vector<real> F0 = sh.ResidualAt(z);                 // no limit yet: the residual evaluates
        m.limit = 3.5;
        int caught = 0;
        try { sh.ResidualAt(z); } catch (const std::out_of_range &) { caught = 1; }
        int caught_again = 0;
        try { sh.ResidualAt(z); } catch (const std::out_of_range &) { caught_again = 1; }      // the pool survived the first throw
        m.limit = 1e300;
        const std::vector<real> F1 = sh.ResidualAt(z);                 // ... and still computes the same numbers
        std::printf("{\"threads\": %d, \"caught\": %d, \"caught_again\": %d, \"same_after\": %d}\n", threads, caught, caught_again, (int)(F0 == F1));
        return (caught && caught_again && F0 == F1) ? 0 : 2;
    }
    const int M = std::atoi(argv[1]);
    const int threads = argc > 2 ? std::atoi(argv[2]) : 1;
    const bool free_tf = false;

    // ---- one-step helpers, function-pointer form (odeTools.cpp:46-87) against the formulas written out by hand
    real w = 1.5;
    const real t = 0.25, h = 0.1;
    odeTools::odeVector X(2);
    X[0] = 1.0; X[1] = -0.5;
    auto f = [&](real tt, real x0, real x1, real &d0, real &d1) { d0 = x1; d1 = -w * w * x0 + tt; };
    real a0, a1, b0, b1, c0, c1, e0, e1;
    f(t, X[0], X[1], a0, a1);
    f(t + h / 2.0, X[0] + (h / 2.0) * a0, X[1] + (h / 2.0) * a1, b0, b1);
    f(t + h / 2.0, X[0] + (h / 2.0) * b0, X[1] + (h / 2.0) * b1, c0, c1);
    f(t + h, X[0] + h * c0, X[1] + h * c1, e0, e1);
    const real want4[2] = {X[0] + (h / 6.0) * (a0 + (e0 + 2.0 * (b0 + c0))), X[1] + (h / 6.0) * (a1 + (e1 + 2.0 * (b1 + c1)))};
    const real want2[2] = {X[0] + h * b0, X[1] + h * b1};
    const real want1[2] = {X[0] + h * a0, X[1] + h * a1};
    const odeTools::odeVector r4 = odeTools::RK4(t, X, h, oscillator, &w), r2 = odeTools::RK2(t, X, h, oscillator, &w),
                              r1 = odeTools::RK1(t, X, h, oscillator, &w);
    const bool steps_ok = r4[0] == want4[0] && r4[1] == want4[1] && r2[0] == want2[0] && r2[1] == want2[1] && r1[0] == want1[0] && r1[1] == want1[1];

    // ---- modelStruct form on the user model (odeTools.cpp:51-98): RK4 of the linear system is exact arithmetic to check by hand
    lqr_host m;
    odeTools::odeVector Y(4);
    Y[0] = 0.0; Y[1] = 0.0; Y[2] = -12.0; Y[3] = -6.0;
    odeTools::odeVector Y4 = Y;
    odeTools::RK4(0.0, Y4, 0.5, odeTools::modelStruct(&m, 0));
    // p_v(t) = -6 + 12 t, v(t) = 6 t - 6 t^2, x(t) = 3 t^2 - 2 t^3: cubic at most => RK4 is exact up to rounding
    const bool struct_ok = std::fabs(Y4[3] - 0.0) < 1e-14 && std::fabs(Y4[1] - 1.5) < 1e-14 && std::fabs(Y4[0] - 0.5) < 1e-14 && Y4[2] == -12.0;

    // ---- the whole solve through shooting, host virtuals only
    shooting sh(m, M, threads);
    sh.SetPrecision(1e-12);
    std::vector<int> mode_Xf(2, model::FIXED);
    sh.SetMode(free_tf ? model::FREE : model::FIXED, mode_Xf);
    model::mstate Xi(4, 0.0), Xf(4, 0.0);
    Xi[2] = -1.0; Xi[3] = -1.0;
    Xf[0] = 1.0;
    sh.InitShooting(0.0, Xi, 1.0, Xf);
    const int info = sh.SolveOCP(0.0);
    std::vector<real> z;
    sh.GetParameters(z);
    const model::mstate u0 = m.Control(0.0, model::mstate(z.begin(), z.begin() + 4));
    std::printf("{\"info\": %d, \"nfev\": %d, \"p_x\": %.17g, \"p_v\": %.17g, \"u0\": %.17g, \"n\": %d, \"steps_ok\": %d, \"struct_ok\": %d, "
                "\"trajectories\": %lld, \"tf\": %.17g, \"z\": [",
                info, sh.GetCallNumber()[0], z[2], z[3], u0[0], (int)z.size(), (int)steps_ok, (int)struct_ok, sh.GetTrajectoryCount(),
                free_tf ? z.back() : 1.0);
    for (size_t k = 0; k < z.size(); k++) std::printf("%s%.17g", k ? ", " : "", z[k]);
    std::printf("]}\n");
    return (info == 1 && steps_ok && struct_ok) ? 0 : 2;
}
