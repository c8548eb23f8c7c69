// sweep_rccl.cpp -- socp_sweep_solve_rank with a REAL RCCL collective: ncclAllGather on device buffers (VERDICT r3 #1b; north_star:
// "shards independent shooting problems across the 8 GPUs of one node with a trivial RCCL gather over xGMI").  One process per GPU:
//   sweep_rccl <starts.bin> <P> <rk4_steps> [<world> <rank> <id-file>]
// world = 1 (default): ncclCommInitRank with a communicator of one -- what a one-GPU box can run; the collective, the device
// staging and the unpacking are the ones an 8-rank job uses.  world > 1: start one process per rank (rank r takes GPU r); rank 0
// writes the ncclUniqueId to <id-file>, the others wait for it.  Partition of the starts: contiguous blocks whose sizes differ by
// at most one (the reference's split of segments over threads, shooting.cpp:1223-1231, applied to problems).
// Links /opt/rocm's librccl and HIP runtime; libsocp_hip.so itself links no communication library.
// Problem: Goddard single shooting, n = 14 (BASELINE configs 2 / 4), throughput flavour.  Output: the JSON line of sweep_flow.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "socp_hip.h"
#include "socp_solver.h"

struct Job {
    ncclComm_t comm;
    hipStream_t stream;
    long calls;
};

// the collective socp_sweep_solve_rank calls exactly once: send / recv are device-visible buffers (gather_on_device = 1)
static int rccl_allgather(void *user, const double *send, long count, double *recv)
{
    Job *j = static_cast<Job *>(user);
    j->calls++;
    if (ncclAllGather(send, recv, (size_t)count, ncclDouble, j->comm, j->stream) != ncclSuccess) return 1;
    return hipStreamSynchronize(j->stream) == hipSuccess ? 0 : 1;
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

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: sweep_rccl <starts.bin> <P> <rk4_steps> [<world> <rank> <id-file>]\n"); return 64; }
    const int P = std::atoi(argv[2]), steps = std::atoi(argv[3]), n = 14;
    const int world = argc >= 7 ? std::atoi(argv[4]) : 1, rank = argc >= 7 ? std::atoi(argv[5]) : 0;
    if (world < 1 || rank < 0 || rank >= world) return 64;
    std::vector<double> Z0((size_t)P * n), Z((size_t)P * n), fnorm(P);
    std::vector<int> info(P), nfev(P), nfev_total(P), solves(P);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(Z0.data(), sizeof(double), Z0.size(), f) != Z0.size()) { std::fprintf(stderr, "cannot read the starts\n"); return 3; }
    std::fclose(f);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no HIP device\n"); return 3; }
    const int device = rank % ndev;
    if (hipSetDevice(device) != hipSuccess) return 3;
    ncclUniqueId id;
    if (world == 1) {
        if (ncclGetUniqueId(&id) != ncclSuccess) { std::fprintf(stderr, "ncclGetUniqueId failed\n"); return 5; }
    } else if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) return 5;
        std::string tmp = std::string(argv[6]) + ".tmp";
        FILE *g = std::fopen(tmp.c_str(), "wb");
        if (!g || std::fwrite(&id, sizeof(id), 1, g) != 1) return 5;
        std::fclose(g);
        std::rename(tmp.c_str(), argv[6]);                            // appears complete or not at all
    } else {
        FILE *g = nullptr;
        for (int tries = 0; tries < 600 && !(g = std::fopen(argv[6], "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!g || std::fread(&id, sizeof(id), 1, g) != 1) { std::fprintf(stderr, "rank %d: no unique id in %s\n", rank, argv[6]); return 5; }
        std::fclose(g);
    }
    Job job;
    job.calls = 0;
    if (ncclCommInitRank(&job.comm, world, id, rank) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitRank failed\n"); return 5; }
    if (hipStreamCreate(&job.stream) != hipSuccess) return 3;

    socp_ctx *ctx = goddard_ctx(device, steps);
    socp_chain_options opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.kind = SOCP_CHAIN_PLAIN; opt.xtol = 1e-8; opt.maxfev = 10000; opt.epsfcn = 1e-15; opt.factor = 1.0; opt.dedup = 1; opt.speculate = -1;
    socp_chain_stats st;
    std::memset(&st, 0, sizeof(st));
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = socp_sweep_solve_rank(ctx, rank, world, P, &opt, Z0.data(), rccl_allgather, &job, 1 /* device buffers */, Z.data(), info.data(),
                                         nfev.data(), nfev_total.data(), solves.data(), fnorm.data(), &st);
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    long long trajectories = 0, launches = 0;
    socp_ctx_counters(ctx, &trajectories, &launches);
    socp_ctx_destroy(ctx);
    (void)hipStreamDestroy(job.stream);
    ncclCommDestroy(job.comm);
    std::fprintf(stderr, "rank_rc %d %d collective_calls %ld\n", rank, rc, job.calls);
    if (rc != SOCP_OK) { std::fprintf(stderr, "sweep failed: %d\n", rc); return 2; }
    if (job.calls != 1) { std::fprintf(stderr, "the collective was called %ld times\n", job.calls); return 4; }
    if (rank != 0) return 0;
    std::printf("{\"n\": %d, \"P\": %d, \"world\": %d, \"wall_ms\": %.3f, \"trajectories\": %lld, \"z\": [", n, P, world, wall_ms, trajectories);
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
