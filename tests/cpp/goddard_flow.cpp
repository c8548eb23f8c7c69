// goddard_flow.cpp -- the workload of the reference's tests/testGoddard.cpp (same problem set-up,
// same four solves, same API calls), written as a checkable program: every stage prints one JSON
// line {stage, info, nfev, n, z[...]} instead of "OK = info".  Run by tests/test_host_flow.py on
// the GPU box; the expected solutions are in tests/golden/.
//
// usage: goddard_flow [stepNbr] [nMulti] [dedup 0|1] [tracefile]
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "socp/shooting.hpp"
#include "models/goddard/goddard.hpp"

static void report(const char *stage, int info, const shooting &s, int n)
{
    std::vector<real> z;
    s.GetParameters(z);
    std::vector<int> calls = s.GetCallNumber();
    std::printf("{\"stage\": \"%s\", \"info\": %d, \"nfev\": %d, \"n\": %d, \"trajectories\": %lld, \"z\": [", stage, info, calls[0], n,
                s.GetTrajectoryCount());
    for (size_t k = 0; k < z.size(); k++) std::printf("%s%.17g", k ? ", " : "", z[k]);
    std::printf("]}\n");
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    const int stepNbr = argc > 1 ? std::atoi(argv[1]) : 10;
    const int nMulti = argc > 2 ? std::atoi(argv[2]) : 6;
    const bool dedup = argc > 3 ? std::atoi(argv[3]) != 0 : true;
    const std::string trace = argc > 4 ? argv[4] : "";

    goddard my_goddard(trace, stepNbr);
    const int dim = my_goddard.GetDim();
    my_goddard.SetParameterDataName("mu2", 1.0);

    shooting my_shooting(my_goddard, nMulti, 1);
    my_shooting.SetPrecision(1e-6);
    my_shooting.SetJacobianDedup(dedup);

    std::vector<real> vt(nMulti + 1);
    std::vector<model::mstate> vX(nMulti + 1);

    const int mode_tf = model::FREE;
    std::vector<int> mode_Xf(dim, model::FIXED);
    mode_Xf[3] = mode_Xf[4] = mode_Xf[5] = mode_Xf[6] = model::FREE;     // final velocity and mass free
    my_shooting.SetMode(mode_tf, mode_Xf);

    const real ti = 0;
    model::mstate Xi(2 * dim);
    Xi[0] = 0.999949994; Xi[1] = 0.0001; Xi[2] = 0.01;
    Xi[3] = Xi[4] = Xi[5] = 1e-10;
    Xi[6] = 1.0;
    for (int k = 7; k < 14; k++) Xi[k] = 0.1;
    real tf = 0.1;
    model::mstate Xf(2 * dim);
    Xf[0] = 1.01;
    my_shooting.InitShooting(ti, Xi, tf, Xf);

    int info = 1;
    my_goddard.SetParameterDataName("KD", 0.0);
    info = my_shooting.SolveOCP(0.0);
    report("no_drag", info, my_shooting, 2 * dim * nMulti + 1);
    if (info == 1) {
        info = my_shooting.SolveOCP(1.0, "KD", 310.0);
        report("drag_continuation", info, my_shooting, 2 * dim * nMulti + 1);
    }
    if (info == 1) {
        info = my_shooting.SolveOCP(1.0, "mu2", 0.2);
        report("mu2_continuation", info, my_shooting, 2 * dim * nMulti + 1);
    }
    if (info != 1 || nMulti != 6) return info == 1 ? 0 : 2;

    // bang - singular - off structure with free switching times (testGoddard.cpp:115-156)
    my_shooting.GetSolution(vt, vX);
    tf = vt[nMulti];
    const real s1 = 0.0227, s2 = 0.08;
    vt[0] = ti; vt[1] = s1 / 2; vt[2] = s1; vt[3] = (s2 + s1) / 2; vt[4] = s2; vt[5] = (s2 + tf) / 2; vt[6] = tf;
    for (int i = 0; i <= nMulti; i++) vX[i] = my_shooting.Move(vt[i]);
    std::vector<int> mode_t(nMulti + 1, model::CONTINUOUS);
    mode_t[0] = model::FIXED; mode_t[2] = model::FREE; mode_t[4] = model::FREE; mode_t[nMulti] = mode_tf;
    std::vector<std::vector<int> > mode_X(nMulti + 1, std::vector<int>(dim, model::CONTINUOUS));
    mode_X[0] = std::vector<int>(dim, model::FIXED);
    mode_X[nMulti] = mode_Xf;
    my_shooting.SetMode(mode_t, mode_X);
    my_shooting.InitShooting(vt, vX);
    my_goddard.SetParameterDataName("mu2", 0.0);
    my_goddard.SetParameterDataName("singularControl", -1);
    info = my_shooting.SolveOCP(0.0);
    report("singular_arc", info, my_shooting, 2 * dim * nMulti + 3);
    if (!trace.empty()) my_shooting.Trace();
    return info == 1 ? 0 : 2;
}
