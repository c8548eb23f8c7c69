// dint_flow.cpp -- the workloads of the reference's tests/testDoubleIntegrator.cpp ("basic") and
// tests/testDoubleIntegrator_WP.cpp ("wp") as checkable programs: same set-up, same API calls, one
// JSON line per solve.
//   dint_flow basic <modelOrder> <xtol>
//   dint_flow wp    <modelOrder> <xtol> [numMulti]     (numMulti > 2: way-points on the x axis, SURVEY 8d "C3")
// SOCP_FLOW_ADAPTIVE=1: with the adaptive integrator (the reference built with -D_USE_BOOST), variational trajectories included
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "socp/shooting.hpp"
#include "models/doubleIntegrator/doubleIntegrator.hpp"

static void report(const char *stage, int info, const shooting &s)
{
    std::vector<real> z;
    s.GetParameters(z);
    std::vector<int> calls = s.GetCallNumber();
    std::printf("{\"stage\": \"%s\", \"info\": %d, \"nfev\": %d, \"njev\": %d, \"n\": %d, \"trajectories\": %lld, \"z\": [", stage,
                info, calls[0], calls[1], (int)z.size(), s.GetTrajectoryCount());
    for (size_t k = 0; k < z.size(); k++) std::printf("%s%.17g", k ? ", " : "", z[k]);
    std::printf("]}\n");
    std::fflush(stdout);
}

static int basic(int modelOrder, double xtol)
{
    // testDoubleIntegrator.cpp:24-62
    doubleIntegrator m(modelOrder, "");
    m.SetStepNumber(20);
    const int d = m.GetDim();
    shooting sh(m, 1, 1);
    sh.SetPrecision(xtol);
    sh.SetContinuationMinStep(1e-12);
    sh.SetMode(1, std::vector<int>(d, 0));
    const real ti = 0, tf = 10;
    doubleIntegrator::mstate Xi(2 * d, 0.0), Xf(2 * d, 0.0);
    for (int k = d; k < 2 * d; k++) Xi[k] = 0.01;
    Xf[0] = 10.0; Xf[1] = 15.0;
    sh.InitShooting(ti, Xi, tf, Xf);
    int info = sh.SolveOCP(0.0);                                   // :90
    report("solve", info, sh);
    Xf[1] = 20;
    sh.SetDesiredState(ti, Xi, tf, Xf);                            // :116-117
    info = sh.SolveOCP(1.0);                                       // :119
    report("data_continuation", info, sh);
    if (info == 1) info = sh.SolveOCP(1.0, m.GetParameterData().muT, 0.02);   // :143
    report("muT_continuation", info, sh);
    return info == 1 ? 0 : 2;
}

static int wp(int modelOrder, double xtol, int M)
{
    // testDoubleIntegrator_WP.cpp:26-106 (M = 2), refined to M segments as SURVEY 8d describes
    doubleIntegrator m(modelOrder, "");
    const int d = m.GetDim();
    shooting sh(m, M, 2);
    sh.SetPrecision(xtol);
    std::vector<int> mode_t(M + 1, 1);
    mode_t[0] = 0;
    std::vector<std::vector<int> > mode_X(M + 1, std::vector<int>(d, 0));
    for (int i = 1; i < M; i++) mode_X[i][3] = mode_X[i][4] = mode_X[i][5] = 2;
    sh.SetMode(mode_t, mode_X);
    std::vector<real> vt(M + 1);
    std::vector<model::mstate> vX(M + 1, model::mstate(2 * d, 0.0));
    for (int i = 0; i <= M; i++) {
        vt[i] = 60.0 * i / M;
        vX[i][0] = 20.0 * i / M;
        if (i < M) for (int k = d; k < 2 * d; k++) vX[i][k] = 0.001;
    }
    sh.InitShooting(vt, vX);
    int info = sh.SolveOCP(0.0);                                   // :117 (result ignored by the reference)
    report("solve", info, sh);
    if (M == 2) {
        vX[1][1] = 15.0; vX[2][1] = 5.0; vX[2][2] = 10.0;          // :120-125
        sh.SetDesiredState(vt, vX);
        info = sh.SolveOCP(1.0);                                   // :126
        report("data_continuation", info, sh);
        if (info == 1) info = sh.SolveOCP(1.0, m.GetParameterData().muT, 0.02);   // :150
        report("muT_continuation", info, sh);
    }
    return info == 1 ? 0 : 2;
}

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: dint_flow basic|wp <modelOrder> <xtol> [numMulti]\n"); return 64; }
    if (std::getenv("SOCP_FLOW_ADAPTIVE")) odeTools::UseAdaptiveIntegrator(true);      // the reference's -D_USE_BOOST build (odeTools.cpp:129-134)
    const int order = std::atoi(argv[2]);
    const double xtol = std::atof(argv[3]);
    if (std::strcmp(argv[1], "basic") == 0) return basic(order, xtol);
    return wp(order, xtol, argc > 4 ? std::atoi(argv[4]) : 2);
}
