// sweep_flow.cpp -- a C++ user of the multi-GPU sweep entry points (include/socp_solver.h): no Python, no torch.
//   sweep_flow devices <ndev> <starts.bin> <P> <rk4_steps>     socp_sweep_solve: one process, ndev GPUs (a thread + a context each)
//   sweep_flow samedev <ndev> <starts.bin> <P> <rk4_steps>     the same with every "device" = GPU 0 (a one-GPU box): exercises the threads, the
//                                                               cloned contexts and the slicing of the per-chain arrays; here the chains are
//                                                               KD continuations (SOCP_CHAIN_PARAM) with their own parameters and goals
//   sweep_flow ranksdev <world> ...                            the same with the collective on DEVICE buffers (gather_on_device = 1)
//   sweep_flow ranks   <world> <starts.bin> <P> <rk4_steps>    socp_sweep_solve_rank: `world` ranks emulated by threads that share
//                                                               device 0, gathering through a user collective (here: shared memory
//                                                               + a barrier; a real job passes ncclAllGather / MPI_Allgather)
//                                                               SOCP_SWEEP_INJECT=device_alloc|set_device|copy[:rank] makes that step of the
//                                                               device staging fail (csrc/sweep.cpp): stderr carries "rank_rc <r> <code>"
// Problem: Goddard single shooting, n = 14 (BASELINE configs 2 / 4), throughput flavour.  starts.bin: P x 14 doubles.
// Prints one JSON line: {"n": 14, "z": [[...]], "info": [...], "nfev": [...], "fnorm": [...], "wall_ms": ..., "trajectories": ...}
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "socp_hip.h"
#include "socp_solver.h"

#include <dlfcn.h>
// device-buffer form of the collective (what an RCCL job passes).  The program links no HIP itself: hipMemcpy is taken from the
// runtime libsocp_hip.so has already loaded.  kind: 1 = host to device, 2 = device to host, 4 = decided from the addresses
typedef int (*hip_memcpy_fn)(void *dst, const void *src, size_t bytes, int kind);
static int hipMemcpy(void *dst, const void *src, size_t bytes, int kind)
{
    static hip_memcpy_fn fn = reinterpret_cast<hip_memcpy_fn>(dlsym(RTLD_DEFAULT, "hipMemcpy"));
    return fn ? fn(dst, src, bytes, kind) : 1;
}

static socp_ctx *goddard_ctx(int device, int steps)
{
    socp_ctx *c = nullptr;
    if (socp_ctx_create(&c, SOCP_MODEL_GODDARD, device) != SOCP_OK) { std::fprintf(stderr, "%s\n", socp_last_error(nullptr)); std::exit(3); }
    const double params[8] = {3.5, 7.0, 310.0, 500.0, 1.0, 1.0, 1.0, -1.0};
    socp_ctx_set_params(c, params, 8);
    socp_ctx_set_step_number(c, steps);
    socp_ctx_set_variant(c, SOCP_VARIANT_LANE_FAST);
    const int mode_t[2] = {SOCP_FIXED, SOCP_FIXED};
    int mode_x[14] = {0};
    for (int k = 3; k < 7; k++) mode_x[7 + k] = SOCP_FREE;           // final velocity and mass free
    const double time[2] = {0.0, 0.2640825};
    double X[28] = {0};
    const double x0[7] = {0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0};
    std::memcpy(X, x0, sizeof(x0));
    X[14] = 1.01;
    if (socp_problem_set(c, 1, mode_t, mode_x, time, X) != SOCP_OK) { std::fprintf(stderr, "%s\n", socp_last_error(c)); std::exit(3); }
    return c;
}

// all-gather between threads of one process: every rank deposits its block, the last one to arrive releases the others
struct SharedGather {
    int world;
    std::mutex m;
    std::condition_variable cv;
    std::vector<double> buf;
    int arrived = 0, generation = 0;
};
struct RankGather { SharedGather *g; int rank; };
static int thread_allgather(void *user, const double *send, long count, double *recv)
{
    RankGather *rg = static_cast<RankGather *>(user);
    SharedGather &g = *rg->g;
    std::unique_lock<std::mutex> lk(g.m);
    if (g.buf.size() != (size_t)count * g.world) g.buf.assign((size_t)count * g.world, 0.0);
    std::memcpy(&g.buf[(size_t)rg->rank * count], send, sizeof(double) * count);
    const int gen = g.generation;
    if (++g.arrived == g.world) { g.arrived = 0; g.generation++; g.cv.notify_all(); }
    else g.cv.wait(lk, [&]() { return g.generation != gen; });
    std::memcpy(recv, g.buf.data(), sizeof(double) * count * g.world);
    return 0;
}

// the same collective on DEVICE buffers (gather_on_device = 1), staged through the host the way a test without RCCL can
static int thread_allgather_dev(void *user, const double *send, long count, double *recv)
{
    RankGather *rg = static_cast<RankGather *>(user);
    std::vector<double> hs(count), hr((size_t)count * rg->g->world);
    if (hipMemcpy(hs.data(), send, sizeof(double) * count, 4) != 0) return 1;      // 4 = by address: the buffers may be pinned host memory
    if (thread_allgather(user, hs.data(), count, hr.data()) != 0) return 1;
    return hipMemcpy(recv, hr.data(), sizeof(double) * hr.size(), 4) != 0;
}

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: sweep_flow devices|ranks <count> <starts.bin> <P> <rk4_steps>\n"); return 64; }
    const std::string mode = argv[1];
    const int count = std::atoi(argv[2]), P = std::atoi(argv[4]), steps = std::atoi(argv[5]), n = 14;
    std::vector<double> Z0((size_t)P * n), Z((size_t)P * n), fnorm(P);
    std::vector<int> info(P), nfev(P), nfev_total(P), solves(P);
    FILE *f = std::fopen(argv[3], "rb");
    if (!f || std::fread(Z0.data(), sizeof(double), Z0.size(), f) != Z0.size()) { std::fprintf(stderr, "cannot read the starts\n"); return 3; }
    std::fclose(f);
    socp_chain_options opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.kind = SOCP_CHAIN_PLAIN; opt.xtol = 1e-8; opt.maxfev = 10000; opt.epsfcn = 1e-15; opt.factor = 1.0; opt.dedup = 1; opt.speculate = -1;
    double wall_ms = 0;
    long long trajectories = 0;
    int rc = SOCP_OK;
    if (mode == "samedev") {
        // every chain: KD from 300 to its own goal 310 (1 + 0.01 p), continuation step 0.5 (shooting.cpp:695-778)
        socp_ctx *proto = goddard_ctx(0, steps);
        std::vector<int> devs(count, 0);
        std::vector<double> params((size_t)P * 8), goal(P);
        const double base[8] = {3.5, 7.0, 300.0, 500.0, 1.0, 1.0, 1.0, -1.0};
        for (int p = 0; p < P; p++) { std::memcpy(&params[(size_t)p * 8], base, sizeof(base)); goal[p] = 310.0 * (1.0 + 0.01 * p); }
        opt.kind = SOCP_CHAIN_PARAM; opt.param_index = 2; opt.step = 0.5; opt.step_min = 1e-12;
        socp_sweep_stats st;
        std::vector<double> pf(P), br(P);
        rc = socp_sweep_solve(proto, devs.data(), count, P, &opt, Z0.data(), params.data(), goal.data(), nullptr, nullptr, nullptr, nullptr, Z.data(),
                              info.data(), nfev.data(), nfev_total.data(), solves.data(), br.data(), pf.data(), fnorm.data(), &st);
        wall_ms = st.wall_ms; trajectories = st.trajectories;
        for (int p = 0; p < P && rc == SOCP_OK; p++)
            if (info[p] == 1 && pf[p] != goal[p]) { std::fprintf(stderr, "chain %d ended at KD = %.17g, goal %.17g\n", p, pf[p], goal[p]); rc = SOCP_ERR_ARG; }
        socp_ctx_destroy(proto);
    } else if (mode == "devices") {
        socp_ctx *proto = goddard_ctx(0, steps);
        socp_sweep_stats st;
        rc = socp_sweep_solve(proto, nullptr, count, P, &opt, Z0.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, Z.data(), info.data(),
                              nfev.data(), nfev_total.data(), solves.data(), nullptr, nullptr, fnorm.data(), &st);
        wall_ms = st.wall_ms; trajectories = st.trajectories;
        socp_ctx_destroy(proto);
    } else {
        SharedGather g;
        g.world = count;
        std::vector<int> rcs(count, SOCP_OK);
        std::vector<std::vector<double> > Zr(count, std::vector<double>((size_t)P * n)), Fr(count, std::vector<double>(P));
        std::vector<std::vector<int> > Ir(count, std::vector<int>(P)), Nr(count, std::vector<int>(P));
        std::vector<std::thread> th;
        for (int r = 0; r < count; r++)
            th.emplace_back([&, r]() {
                socp_ctx *c = goddard_ctx(0, steps);
                RankGather rg{&g, r};
                socp_chain_stats st;
                const bool on_device = mode == "ranksdev";
                rcs[r] = socp_sweep_solve_rank(c, r, count, P, &opt, Z0.data(), on_device ? thread_allgather_dev : thread_allgather, &rg, on_device ? 1 : 0,
                                               Zr[r].data(), Ir[r].data(), Nr[r].data(),
                                               nullptr, nullptr, Fr[r].data(), &st);
                socp_ctx_destroy(c);
            });
        for (std::thread &t : th) t.join();
        for (int r = 0; r < count; r++) std::fprintf(stderr, "rank_rc %d %d\n", r, rcs[r]);
        for (int r = 0; r < count && rc == SOCP_OK; r++) if (rcs[r] != SOCP_OK) rc = rcs[r];
        for (int r = 0; r < count && rc == SOCP_OK; r++) {
            // every rank must hold the same full table
            if (Zr[r] != Zr[0] || Ir[r] != Ir[0] || Nr[r] != Nr[0] || Fr[r] != Fr[0]) { std::fprintf(stderr, "rank %d holds a different table\n", r); return 4; }
        }
        Z = Zr[0]; info = Ir[0]; nfev = Nr[0]; fnorm = Fr[0];
    }
    if (rc != SOCP_OK) { std::fprintf(stderr, "sweep failed: %d\n", rc); return 2; }
    std::printf("{\"n\": %d, \"P\": %d, \"wall_ms\": %.3f, \"trajectories\": %lld, \"z\": [", n, P, wall_ms, trajectories);
    for (int p = 0; p < P; p++) {
        std::printf("%s[", p ? ", " : "");
        for (int k = 0; k < n; k++) std::printf("%s%.17g", k ? ", " : "", Z[(size_t)p * n + k]);
        std::printf("]");
    }
    std::printf("], \"info\": [");
    for (int p = 0; p < P; p++) std::printf("%s%d", p ? ", " : "", info[p]);
    std::printf("], \"nfev\": [");
    for (int p = 0; p < P; p++) std::printf("%s%d", p ? ", " : "", nfev[p]);
    std::printf("], \"solves\": [");
    for (int p = 0; p < P; p++) std::printf("%s%d", p ? ", " : "", solves[p]);
    std::printf("], \"fnorm\": [");
    for (int p = 0; p < P; p++) std::printf("%s%.17g", p ? ", " : "", fnorm[p]);
    std::printf("]}\n");
    return 0;
}
