// interceptor_flow.cpp -- the workload of the reference's tests/testInterceptor.cpp as a checkable program: same
// set-up and API calls (analytical guess at mu_gft = 0, Newton solve, continuation on mu_gft, continuation on
// the boundary data of a scenario), one JSON line per SolveOCP.
//   interceptor_flow <xtol> [scenario = 1] [trace file]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "socp/shooting.hpp"
#include "models/interceptor/interceptor.hpp"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static void report(const char *stage, int info, const shooting &s)
{
    std::vector<real> z;
    s.GetParameters(z);
    std::vector<int> calls = s.GetCallNumber();
    std::printf("{\"stage\": \"%s\", \"info\": %d, \"nfev\": %d, \"n\": %d, \"trajectories\": %lld, \"z\": [", stage, info,
                calls[0], (int)z.size(), s.GetTrajectoryCount());
    for (size_t k = 0; k < z.size(); k++) std::printf("%s%.17g", k ? ", " : "", z[k]);
    std::printf("]}\n");
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    const double xtol = argc > 1 ? std::atof(argv[1]) : 1e-8;
    const int scenario = argc > 2 ? std::atoi(argv[2]) : 1;
    const std::string trace = argc > 3 ? argv[3] : "";
    const double RE = 6378145.0;

    // ---- initState() of testInterceptor.cpp:165-218 ------------------------------------------------
    real t0 = 0, t1 = 10;
    model::mstate X0(12, 0.0), X1(12, 0.0);
    X0[0] = 1000; X0[1] = 1000; X0[2] = M_PI / 4; X0[3] = 0.0; X0[4] = 5454661 / RE; X0[5] = 46086 / RE;
    X1[0] = 6000; X1[1] = 1000; X1[2] = 0.01 * M_PI; X1[3] = 0.01 * M_PI; X1[4] = (5454661 + 27829.0) / RE; X1[5] = 46086 / RE;
    int info;
    {
        interceptor ini("");
        shooting sh(ini, 1, 1);
        sh.SetPrecision(xtol);
        std::vector<int> mode_X(ini.GetDim(), 0);
        mode_X[1] = 1;                                  // final velocity free
        sh.SetMode(1, mode_X);                          // final time free
        ini.GetParameterData().mu_gft = 0;              // no gravity, no thrust: the analytical guess is for this model
        ini.InitAnalytical(t0, X0, t1, X1);
        sh.InitShooting(t0, X0, t1, X1);
        info = sh.SolveOCP(0.0);
        report("analytical_guess", info, sh);
        if (info == 1) {
            info = sh.SolveOCP(0.1, ini.GetParameterData().mu_gft, 1);
            report("mu_gft_continuation", info, sh);
        }
        std::vector<real> vt(2);
        std::vector<model::mstate> vX(2);
        sh.GetSolution(vt, vX);
        t0 = vt[0]; t1 = vt[1]; X0 = vX[0]; X1 = vX[1];
    }
    if (info != 1) return 2;

    // ---- solve() for one scenario (testInterceptor.cpp:31-104, :117-160) -----------------------------
    real ti = 0, tf = 20;
    model::mstate Xi(12, 0.0), Xf(12, 0.0);
    Xi[0] = 3000; Xi[1] = 1000; Xi[3] = 0.0; Xi[4] = 5454661 / RE; Xi[5] = 46086 / RE;
    Xf[1] = 1000;
    if (scenario == 1) {
        Xi[2] = -M_PI / 6;
        Xf[0] = 12000; Xf[2] = 0.0; Xf[3] = M_PI / 8; Xf[4] = 5475000 / RE; Xf[5] = 42000 / RE;
    } else if (scenario == 2) {
        Xi[2] = M_PI / 4;
        Xf[0] = 12000; Xf[2] = -M_PI / 4; Xf[3] = -M_PI / 2; Xf[4] = 5485000 / RE; Xf[5] = 36178 / RE;
    } else {
        Xi[2] = 0.0;
        Xf[0] = 3000; Xf[2] = 0.0; Xf[3] = 0.0; Xf[4] = 5485000 / RE; Xf[5] = 46086 / RE;
    }
    interceptor m(trace);
    shooting sh(m, 1, 1);
    sh.SetPrecision(xtol);
    std::vector<int> mode_X(m.GetDim(), 0);
    mode_X[1] = 1;
    sh.SetMode(1, mode_X);
    sh.InitShooting(t0, X0, t1, X1);
    sh.SetDesiredState(ti, Xi, tf, Xf);
    info = sh.SolveOCP(0.1);
    report("scenario_continuation", info, sh);
    if (!trace.empty()) sh.Trace();
    return info == 1 ? 0 : 2;
}
