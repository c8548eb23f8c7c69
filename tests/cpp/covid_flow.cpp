// covid_flow.cpp -- the workload of the reference's tests/testCovid19.cpp as a checkable program:
// same set-up and API calls, one JSON line per SolveOCP.
//   covid_flow <xtol> [stages = 3]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "socp/shooting.hpp"
#include "models/covid19/covid19.hpp"

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
    const int stages = argc > 2 ? std::atoi(argv[2]) : 3;
    covid19 m("");
    const int d = m.GetDim();
    m.GetParameterData().R0 = 3.4;             // testCovid19.cpp:41-43
    m.GetParameterData().Tinf = 14;
    m.GetParameterData().Tinc = 5;
    const int nMulti = 20;
    shooting sh(m, nMulti, 4);
    sh.SetPrecision(xtol);
    std::vector<int> mode_Xf(d, 0);
    mode_Xf[0] = mode_Xf[1] = mode_Xf[2] = 1;  // S, E, I free at tf; R pinned
    sh.SetMode(0, mode_Xf);
    const real ti = 0;
    model::mstate Xi(2 * d, 0.0), Xf(2 * d, 0.0);
    Xi[0] = 0.93; Xi[1] = 0.003; Xi[2] = 0.01; Xi[3] = 0.057; Xi[4] = -0.001; Xi[5] = 0.001;
    real tf = 30;
    Xf[3] = 0.6;
    sh.InitShooting(ti, Xi, tf, Xf);
    int info = sh.SolveOCP(0.0);
    report("solve", info, sh);
    if (stages > 1) {
        Xf[3] = 0.7;
        sh.SetDesiredState(ti, Xi, tf, Xf);
        info = sh.SolveOCP(0.1);
        report("target_continuation", info, sh);
    }
    if (stages > 2) {
        tf = 365;
        sh.SetDesiredState(ti, Xi, tf, Xf);
        info = sh.SolveOCP(0.01);
        report("horizon_continuation", info, sh);
    }
    return info == 1 ? 0 : 2;
}
