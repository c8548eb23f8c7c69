// concurrent_solves.cpp -- SURVEY 8b "Threading": neither `shooting` nor `model` is safe for concurrent solves on ONE
// object, but independent (model, shooting) pairs may solve on different threads.  T threads each build their own
// pair, solve the same single-shooting Goddard problem from slightly different starts, and the program prints one
// JSON line per thread; a serial run of the same starts must give the same bits.
//   concurrent_solves <threads> <serial:0|1>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "socp/shooting.hpp"
#include "models/goddard/goddard.hpp"

static std::mutex g_print;

static void solve_one(int k)
{
    goddard g("");
    g.SetParameterDataName("mu2", 1.0);
    g.stepNbr = 200;
    shooting sh(g, 1, 1);
    sh.SetPrecision(1e-10);
    std::vector<int> mode_X(g.GetDim(), 0);
    mode_X[3] = mode_X[4] = mode_X[5] = mode_X[6] = 1;
    sh.SetMode(0, mode_X);
    model::mstate Xi(14, 0.0), Xf(14, 0.0);
    const double x0[7] = {0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0};
    const double p0[7] = {-8.121947733, 7.775439382e-3, 0.7775438809, -0.4779369965, 5.715013318e-4, 5.715009222e-2, 9.958404873e-2};
    for (int i = 0; i < 7; i++) { Xi[i] = x0[i]; Xi[7 + i] = p0[i] * (1.0 + 1e-4 * (k + 1)); }
    Xf[0] = 1.01;
    sh.InitShooting(0.0, Xi, 0.2640825, Xf);
    const int info = sh.SolveOCP(0.0);
    std::vector<real> z;
    sh.GetParameters(z);
    std::lock_guard<std::mutex> lock(g_print);
    std::printf("{\"thread\": %d, \"info\": %d, \"nfev\": %d, \"z\": [", k, info, sh.GetCallNumber()[0]);
    for (size_t i = 0; i < z.size(); i++) std::printf("%s%.17g", i ? ", " : "", z[i]);
    std::printf("]}\n");
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 4;
    const bool serial = argc > 2 && std::atoi(argv[2]) != 0;
    if (serial) {
        for (int k = 0; k < T; k++) solve_one(k);
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < T; k++) th.emplace_back(solve_one, k);
        for (std::thread &t : th) t.join();
    }
    return 0;
}
