// qr_bench.cpp -- the O(n^3) work of one Jacobian refresh (qrfac + qtf + R + qform) on the host: MINPACK's scalar column
// algorithm (optionally with its columns dealt out to threads) against the columns-in-SIMD-lanes form (minpack.cpp colvec),
// with a bit-for-bit comparison of everything hybrd reads afterwards (Q, R, rdiag, acnorm, qtf).
//   qr_bench [n=832] [threads=1] [reps=3]         prints one JSON line
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../socp_amd/csrc/minpack.cpp"

static double ms(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 832, T = argc > 2 ? atoi(argv[2]) : 1, reps = argc > 3 ? atoi(argv[3]) : 3;
    std::vector<double> A((size_t)n * n), f(n);
    srand48(1);
    for (auto &v : A) v = drand48() - 0.5;
    for (auto &v : f) v = drand48() - 0.5;
    if (n > 40) {                                   // a zero column and a zero sub-column: identity reflectors on the way
        for (int i = 0; i < n; i++) A[i + (size_t)7 * n] = 0;
        for (int i = 20; i < n; i++) A[i + (size_t)20 * n] = 0;
    }
    const size_t lr = (size_t)n * (n + 1) / 2;
    std::vector<double> Qs, rd_s(n), ac_s(n), qtf_s(n), r_s(lr), wa(n), Qv, rd_v(n), ac_v(n), qtf_v(n), r_v(lr);
    double t_qrfac = 1e300, t_qform = 1e300, t_scalar = 1e300, t_vec = 1e300;
    for (int rep = 0; rep < reps; rep++) {
        Qs = A;
        auto t0 = std::chrono::steady_clock::now();
        qrfac_nopivot(n, Qs.data(), n, rd_s.data(), ac_s.data(), T);
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) qtf_s[i] = f[i];                 // as Core::after_jacobian
        for (int j = 0; j < n; j++) {
            const double *aj = Qs.data() + (size_t)j * n;
            if (aj[j] != 0) {
                double sum = 0;
                for (int i = j; i < n; i++) sum += aj[i] * qtf_s[i];
                const double temp = -sum / aj[j];
                for (int i = j; i < n; i++) qtf_s[i] += aj[i] * temp;
            }
        }
        for (int j = 0; j < n; j++) {
            int l = j;
            for (int i = 0; i < j; i++) { r_s[l] = Qs[i + (size_t)j * n]; l += n - i - 1; }
            r_s[l] = rd_s[j];
        }
        auto t2 = std::chrono::steady_clock::now();
        qform(n, Qs.data(), n, wa.data(), T);
        auto t3 = std::chrono::steady_clock::now();
        t_qrfac = std::min(t_qrfac, ms(t0, t1)); t_qform = std::min(t_qform, ms(t2, t3)); t_scalar = std::min(t_scalar, ms(t0, t3));
        Qv = A;
        auto t4 = std::chrono::steady_clock::now();
        { socp::Pool pool(T); colvec::factor(n, Qv.data(), n, f.data(), rd_v.data(), ac_v.data(), qtf_v.data(), r_v.data(), pool); }   // workers started inside the timed region, as a solve's first refresh does
        auto t5 = std::chrono::steady_clock::now();
        t_vec = std::min(t_vec, ms(t4, t5));
    }
    auto same = [](const std::vector<double> &a, const std::vector<double> &b) { return std::memcmp(a.data(), b.data(), sizeof(double) * a.size()) == 0; };
    const bool ok = same(Qs, Qv) && same(rd_s, rd_v) && same(ac_s, ac_v) && same(qtf_s, qtf_v) && same(r_s, r_v);
    printf("{\"n\": %d, \"threads\": %d, \"scalar_ms\": %.3f, \"scalar_qrfac_ms\": %.3f, \"scalar_qform_ms\": %.3f, \"simd_columns_ms\": %.3f, "
           "\"speedup\": %.2f, \"bit_identical\": %s, \"isa\": \"%s\"}\n",
           n, T, t_scalar, t_qrfac, t_qform, t_vec, t_scalar / t_vec, ok ? "true" : "false",
           __builtin_cpu_supports("avx512f") ? "avx512f" : (__builtin_cpu_supports("avx2") ? "avx2" : "baseline"));
    return ok ? 0 : 1;
}
