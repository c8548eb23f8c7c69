#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#define main minpack_main_unused
#include "../../socp_amd/csrc/minpack.cpp"
#undef main
int main(int argc, char **argv) {
    int n = 832, T = argc > 1 ? atoi(argv[1]) : 1;
    std::vector<double> A((size_t)n*n), B, rd(n), ac(n), wa(n);
    srand48(1); for (auto &v : A) v = drand48() - 0.5;
    B = A;
    auto t0 = std::chrono::steady_clock::now();
    qrfac_nopivot(n, B.data(), n, rd.data(), ac.data(), T);
    auto t1 = std::chrono::steady_clock::now();
    qform(n, B.data(), n, wa.data(), T);
    auto t2 = std::chrono::steady_clock::now();
    double cs = 0; for (auto v : B) cs += v;
    printf("T=%d qrfac %.1f ms qform %.1f ms checksum %.17g\n", T, std::chrono::duration<double,std::milli>(t1-t0).count(), std::chrono::duration<double,std::milli>(t2-t1).count(), cs);
}
