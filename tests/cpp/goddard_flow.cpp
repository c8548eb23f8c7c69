// goddard_flow.cpp -- the workload of the reference's tests/testGoddard.cpp (same problem set-up,
// same API calls, same four solves) as a checkable program: every solve prints one JSON line
// {stage, info, nfev, n, trajectories, z[...]} instead of "OK = info".
//
//   goddard_flow full  <stepNbr> <dedup> <xtol> [tracefile]      whole flow from the trivial guess
//   goddard_flow stage <k> <stepNbr> <dedup> <xtol> <zfile> [tracefile]   ONE solve (k = 1..4) started from the
//                       85 unknowns in <zfile> (node states + tf), the state testGoddard.cpp is in
//                       just before its k-th SolveOCP call
// Run by tests/test_host_flow.py on the GPU box; expected solutions are in tests/golden/.
//
// The program only uses the reference's public API, so it also compiles against the REFERENCE's own headers and
// sources (-DSOCP_REFERENCE_BUILD, oracle/Makefile target `link`): that binary is the reference's shooting.cpp bound
// to this library's hybrd/hybrj at link level (boundary test, tests/test_link_dropin.py) and the "as shipped" CPU
// baseline B0 of bench.py.  SOCP_FLOW_THREADS = numThread handed to the shooting constructor (default 1).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "socp/shooting.hpp"
#include "models/goddard/goddard.hpp"

namespace {
const int kMulti = 6;
void report_body(const char *stage, int info, const shooting &s);

std::chrono::steady_clock::time_point g_tic = std::chrono::steady_clock::now();

void report(const char *stage, int info, const shooting &s)
{
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_tic).count();
    g_tic = std::chrono::steady_clock::now();
    std::printf("{\"seconds\": %.6f, ", sec);
    report_body(stage, info, s);
}

void report_body(const char *stage, int info, const shooting &s)
{
    std::vector<real> z;
    s.GetParameters(z);
    std::vector<int> calls = s.GetCallNumber();
#ifdef SOCP_REFERENCE_BUILD
    const long long trajectories = -1;               // the reference does not count them
#else
    const long long trajectories = s.GetTrajectoryCount();
#endif
    std::printf("\"stage\": \"%s\", \"info\": %d, \"nfev\": %d, \"n\": %d, \"trajectories\": %lld, \"z\": [", stage, info,
                calls[0], (int)z.size(), trajectories);
    for (size_t k = 0; k < z.size(); k++) std::printf("%s%.17g", k ? ", " : "", z[k]);
    std::printf("]}\n");
    std::fflush(stdout);
}

std::vector<int> final_modes(int dim)
{
    std::vector<int> m(dim, model::FIXED);
    m[3] = m[4] = m[5] = m[6] = model::FREE;     // final velocity and mass free (testGoddard.cpp:43-48)
    return m;
}

// testGoddard.cpp:115-156: re-grid on the bang / singular / off structure, free switching times
int singular_stage(goddard &g, shooting &sh, int dim)
{
    std::vector<real> vt(kMulti + 1);
    std::vector<model::mstate> vX(kMulti + 1);
    sh.GetSolution(vt, vX);
    const real ti = 0, tf = vt[kMulti], s1 = 0.0227, s2 = 0.08;
    vt[0] = ti; vt[1] = s1 / 2; vt[2] = s1; vt[3] = (s2 + s1) / 2; vt[4] = s2; vt[5] = (s2 + tf) / 2; vt[6] = tf;
    for (int i = 0; i <= kMulti; i++) vX[i] = sh.Move(vt[i]);
    std::vector<int> mode_t(kMulti + 1, model::CONTINUOUS);
    mode_t[0] = model::FIXED; mode_t[2] = model::FREE; mode_t[4] = model::FREE; mode_t[kMulti] = model::FREE;
    std::vector<std::vector<int> > mode_X(kMulti + 1, std::vector<int>(dim, model::CONTINUOUS));
    mode_X[0] = std::vector<int>(dim, model::FIXED);
    mode_X[kMulti] = final_modes(dim);
    sh.SetMode(mode_t, mode_X);
    sh.InitShooting(vt, vX);
    g.SetParameterDataName("mu2", 0.0);
    g.SetParameterDataName("singularControl", -1);
    return sh.SolveOCP(0.0);
}
}  // namespace

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "usage: goddard_flow full <stepNbr> <dedup> <xtol> [trace] | stage <k> <stepNbr> <dedup> <xtol> <zfile>\n");
        return 64;
    }
    const bool single = std::strcmp(argv[1], "stage") == 0;
    const int a = single ? 3 : 2;
    const int stage = single ? std::atoi(argv[2]) : 0;
    const int stepNbr = std::atoi(argv[a]);
    const bool dedup = std::atoi(argv[a + 1]) != 0;
    const double xtol = std::atof(argv[a + 2]);
    const std::string extra = argc > a + 3 ? argv[a + 3] : "";

    const std::string trace = single ? (argc > a + 4 ? argv[a + 4] : "") : extra;
#ifndef SOCP_REFERENCE_BUILD
    if (std::getenv("SOCP_FLOW_ADAPTIVE")) odeTools::UseAdaptiveIntegrator(true);     // the -D_USE_BOOST configuration
#endif
    const char *thr = std::getenv("SOCP_FLOW_THREADS");
    const int numThread = thr ? std::atoi(thr) : 1;
    goddard my_goddard(trace, stepNbr);
    const int dim = my_goddard.GetDim();
    my_goddard.SetParameterDataName("mu2", 1.0);
    shooting my_shooting(my_goddard, kMulti, numThread);
    my_shooting.SetPrecision(xtol);
#ifndef SOCP_REFERENCE_BUILD
    my_shooting.SetJacobianDedup(dedup);
    // the arithmetic flavour chosen BY THE PROGRAM (model::SetDeviceVariant) rather than by the SOCP_VARIANT environment variable
    if (const char *fv = std::getenv("SOCP_FLOW_SET_VARIANT")) my_goddard.SetDeviceVariant(std::atoi(fv));
#else
    (void)dedup;
#endif
    my_shooting.SetMode(model::FREE, final_modes(dim));

    model::mstate Xf(2 * dim);
    Xf[0] = 1.01;
    if (!single) {
        model::mstate Xi(2 * dim);
        Xi[0] = 0.999949994; Xi[1] = 0.0001; Xi[2] = 0.01;
        Xi[3] = Xi[4] = Xi[5] = 1e-10;
        Xi[6] = 1.0;
        for (int k = 7; k < 14; k++) Xi[k] = 0.1;
        my_shooting.InitShooting(0.0, Xi, 0.1, Xf);

        my_goddard.SetParameterDataName("KD", 0.0);
        int info = my_shooting.SolveOCP(0.0);
        report("no_drag", info, my_shooting);
        if (info == 1) { info = my_shooting.SolveOCP(1.0, "KD", 310.0); report("drag_continuation", info, my_shooting); }
        if (info == 1) { info = my_shooting.SolveOCP(1.0, "mu2", 0.2); report("mu2_continuation", info, my_shooting); }
        if (info == 1) { info = singular_stage(my_goddard, my_shooting, dim); report("singular_arc", info, my_shooting); }
        if (info == 1 && !trace.empty()) my_shooting.Trace();
        return info == 1 ? 0 : 2;
    }

    // single stage: restore the unknowns the reference program holds before its k-th solve
    std::ifstream in(extra.c_str());
    std::vector<real> z;
    for (real v; in >> v;) z.push_back(v);
    if ((int)z.size() != 2 * dim * kMulti + 1) { std::fprintf(stderr, "zfile must hold %d values\n", 2 * dim * kMulti + 1); return 64; }
    std::vector<real> vt(kMulti + 1);
    std::vector<model::mstate> vX(kMulti + 1, model::mstate(2 * dim));
    const real tf = z.back();
    for (int i = 0; i <= kMulti; i++) vt[i] = 0.0 + i * (tf - 0.0) / kMulti;
    for (int i = 0; i < kMulti; i++) vX[i].assign(z.begin() + 2 * dim * i, z.begin() + 2 * dim * (i + 1));
    vX[kMulti] = Xf;
    my_shooting.InitShooting(vt, vX);
    int info = 0;
    switch (stage) {
    case 1: my_goddard.SetParameterDataName("KD", 0.0); info = my_shooting.SolveOCP(0.0); report("no_drag", info, my_shooting); break;
    case 2: {
        // SOCP_FLOW_KD_GOAL / SOCP_FLOW_STEP / SOCP_FLOW_KD_START: other goals, continuation steps and start values of the same
        // parameter continuation (the sequential twin of the batched chains, tests/test_gpu_chains.py)
        const char *kg = std::getenv("SOCP_FLOW_KD_GOAL"), *ks = std::getenv("SOCP_FLOW_STEP"), *k0 = std::getenv("SOCP_FLOW_KD_START");
        my_goddard.SetParameterDataName("KD", k0 ? std::atof(k0) : 0.0);
        info = my_shooting.SolveOCP(ks ? std::atof(ks) : 1.0, "KD", kg ? std::atof(kg) : 310.0);
        report("drag_continuation", info, my_shooting);
        std::printf("{\"KD_final\": %.17g}\n", my_goddard.GetParameterDataName("KD"));
        break;
    }
    case 3: info = my_shooting.SolveOCP(1.0, "mu2", 0.2); report("mu2_continuation", info, my_shooting); break;
    case 4: my_goddard.SetParameterDataName("mu2", 0.2); info = singular_stage(my_goddard, my_shooting, dim); report("singular_arc", info, my_shooting); break;
    default: return 64;
    }
    if (info == 1 && !trace.empty()) my_shooting.Trace();
    return info == 1 ? 0 : 2;
}
