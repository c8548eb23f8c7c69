// sweep.cpp -- multi-GPU sweeps from C++ (include/socp_solver.h: socp_sweep_solve, socp_sweep_solve_rank, socp_sweep_shard).
// Independent shooting problems are dealt out to the GPUs in contiguous blocks; there is no data-path exchange, only the
// gather of result records at the end (SURVEY 8e level 1).  One process with a thread per device shares the output arrays;
// the one-process-per-GPU form gathers through the caller's collective (RCCL, MPI).
#include "../../include/socp_hip.h"
#include "../../include/socp_solver.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" void socp_sweep_shard(int P, int rank, int world, int *lo, int *hi)
{
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    if (rank >= world) rank = world - 1;
    if (P < 0) P = 0;
    const int base = P / world, rem = P % world;
    const int a = rank * base + (rank < rem ? rank : rem);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (rank < rem ? 1 : 0);
}

extern "C" int socp_sweep_solve(const socp_ctx *proto, const int *devices, int ndev, int P, const socp_chain_options *opt, const double *Z0,
                                const double *params, const double *goal, const double *time_prev, const double *x_prev,
                                const double *time_goal, const double *x_goal, double *Zout, int *info, int *nfev_last, int *nfev_total,
                                int *solves, double *b_reached, double *param_final, double *fnorm, socp_sweep_stats *stats)
{
    if (!proto || !opt || ndev < 1 || ndev > 16 || P < 0 || (P > 0 && (!Z0 || !Zout || !info))) return SOCP_ERR_ARG;
    const int n = socp_problem_num_param(proto), nodes = socp_problem_num_nodes(proto), nparams = socp_ctx_num_params(proto);
    int S = 0;
    socp_ctx_dims(proto, nullptr, &S, nullptr);
    if (n <= 0 || nodes < 2) return SOCP_ERR_ARG;
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    auto ms = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    std::vector<int> rc(ndev, SOCP_OK);
    std::vector<socp_chain_stats> st(ndev);
    std::vector<long long> traj(ndev, 0);
    std::vector<double> wall(ndev, 0.0);
    auto work = [&](int k) {
        int lo = 0, hi = 0;
        socp_sweep_shard(P, k, ndev, &lo, &hi);
        socp_ctx *ctx = nullptr;
        rc[k] = socp_ctx_clone(proto, devices ? devices[k] : k, &ctx);
        if (rc[k] != SOCP_OK) return;
        const clk::time_point t = clk::now();
        auto at = [&](const double *a, size_t width) { return a ? a + (size_t)lo * width : nullptr; };
        std::memset(&st[k], 0, sizeof(st[k]));
        rc[k] = socp_chains_solve(ctx, hi - lo, opt, Z0 + (size_t)lo * n, at(params, nparams), at(goal, 1), at(time_prev, nodes),
                                  at(x_prev, (size_t)nodes * S), at(time_goal, nodes), at(x_goal, (size_t)nodes * S), Zout + (size_t)lo * n,
                                  info + lo, nfev_last ? nfev_last + lo : nullptr, nfev_total ? nfev_total + lo : nullptr,
                                  solves ? solves + lo : nullptr, b_reached ? b_reached + lo : nullptr, param_final ? param_final + lo : nullptr,
                                  fnorm ? fnorm + lo : nullptr, &st[k]);
        wall[k] = ms(t);
        long long launches = 0;
        socp_ctx_counters(ctx, &traj[k], &launches);
        socp_ctx_destroy(ctx);
    };
    if (ndev == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < ndev; k++) th.emplace_back(work, k);
        for (std::thread &t : th) t.join();
    }
    int worst = SOCP_OK;
    for (int k = 0; k < ndev; k++) if (rc[k] != SOCP_OK) worst = rc[k];
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->ndev = ndev;
        stats->wall_ms = ms(t0);
        for (int k = 0; k < ndev; k++) { stats->device_wall_ms[k] = wall[k]; stats->device_rounds[k] = st[k].rounds; stats->trajectories += traj[k]; }
    }
    return worst;
}

// Test hook: SOCP_SWEEP_INJECT = "<what>[:<rank>]" makes the staging of socp_sweep_solve_rank fail on that rank (all ranks without
// ":<rank>") -- what = device_alloc (hipMalloc of the staging buffers), set_device (switching to the context's device), copy (the
// host-to-device copy of the message).  tests/cpp/sweep_flow.cpp uses it to show that no failure path skips the collective.
static bool injected(const char *what, int rank)
{
    const char *e = std::getenv("SOCP_SWEEP_INJECT");
    if (!e) return false;
    const size_t len = std::strlen(what);
    if (std::strncmp(e, what, len) != 0) return false;
    if (e[len] == '\0') return true;
    return e[len] == ':' && std::atoi(e + len + 1) == rank;
}

extern "C" int socp_sweep_solve_rank(socp_ctx *ctx, int rank, int world, int P, const socp_chain_options *opt, const double *Z0,
                                     socp_allgather_fn gather, void *user, int gather_on_device, double *Zout, int *info, int *nfev_last,
                                     int *nfev_total, int *solves, double *fnorm, socp_chain_stats *stats)
{
    // argument errors are the same on every rank (the ranks of a job make the same call), so returning before the collective
    // cannot strand anyone; everything that can fail on ONE rank is folded into that rank's message below
    if (!ctx || !opt || !gather || world < 1 || rank < 0 || rank >= world || P < 0 || (P > 0 && (!Z0 || !Zout || !info))) return SOCP_ERR_ARG;
    if (opt->kind != SOCP_CHAIN_PLAIN) return SOCP_ERR_UNSUPPORTED;          // continuation chains: socp_sweep_solve / socp_chains_solve per rank
    const int n = socp_problem_num_param(ctx);
    if (n <= 0) return SOCP_ERR_ARG;
    int lo = 0, hi = 0;
    socp_sweep_shard(P, rank, world, &lo, &hi);
    const int mine = hi - lo, kmax = (P + world - 1) / world;             // equal-size records: short blocks are padded by one row
    const int W = n + 5;
    // a rank's message: kmax records + its own status, so that a rank whose solve failed still takes part in the collective
    // (the others would wait for it for ever) and EVERY rank returns the failure
    const long count = (long)kmax * W + 1;
    const size_t send_bytes = sizeof(double) * (size_t)count, recv_bytes = send_bytes * (size_t)world;

    // Staging for a collective on device buffers is set up BEFORE the solve, and a failure there is not a reason to stay away
    // from the collective: the fallback is pinned host memory (hipHostMalloc: device-visible, so the caller's ncclAllGather /
    // device copy works on it unchanged).  local = this rank's status; it travels in the message.
    int local = SOCP_OK, prev = -1;
    bool switched = false, pinned = false;
    double *dsend = nullptr, *drecv = nullptr;
    if (gather_on_device) {
        bool dev_ok = !injected("set_device", rank) && hipGetDevice(&prev) == hipSuccess && hipSetDevice(socp_ctx_device(ctx)) == hipSuccess;
        switched = dev_ok;
        if (!dev_ok) local = SOCP_ERR_HIP;                                   // the solve may still run (the engine pins its own device); reported anyway
        dev_ok = dev_ok && !injected("device_alloc", rank) && hipMalloc(&dsend, send_bytes) == hipSuccess && hipMalloc(&drecv, recv_bytes) == hipSuccess;
        if (!dev_ok) {
            (void)hipGetLastError();
            if (dsend) { (void)hipFree(dsend); dsend = nullptr; }
            if (drecv) { (void)hipFree(drecv); drecv = nullptr; }
            pinned = hipHostMalloc(&dsend, send_bytes, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&drecv, recv_bytes, hipHostMallocDefault) == hipSuccess;
            if (!pinned) {
                // neither device nor pinned memory: this rank has no buffer a device collective could read.  The one case in
                // which it cannot enter the collective -- include/socp_solver.h tells the caller to abort the communicator.
                if (dsend) (void)hipHostFree(dsend);
                if (switched) (void)hipSetDevice(prev);
                return SOCP_ERR_HIP;
            }
        }
    }

    std::vector<double> z((size_t)mine * n), fn(mine, 0.0), send((size_t)count, 0.0), recv((size_t)world * count, 0.0);
    std::vector<int> inf(mine, 0), nl(mine, 0), nt(mine, 0), so(mine, 0);
    const int rc = socp_chains_solve(ctx, mine, opt, Z0 + (size_t)lo * n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, z.data(), inf.data(),
                                     nl.data(), nt.data(), so.data(), nullptr, nullptr, fn.data(), stats);
    if (rc != SOCP_OK) local = rc;
    send[(size_t)count - 1] = local;
    for (int k = 0; k < mine && rc == SOCP_OK; k++) {
        double *rec = &send[(size_t)k * W];
        std::memcpy(rec, &z[(size_t)k * n], sizeof(double) * n);
        rec[n] = fn[k]; rec[n + 1] = inf[k]; rec[n + 2] = nl[k]; rec[n + 3] = nt[k]; rec[n + 4] = so[k];
    }

    int grc = 0;
    bool read_ok = true;
    if (gather_on_device) {
        const double *src = dsend;
        double *hs = nullptr;
        if (pinned) {
            std::memcpy(dsend, send.data(), send_bytes);
        } else if (injected("copy", rank) || hipMemcpy(dsend, send.data(), send_bytes, hipMemcpyHostToDevice) != hipSuccess) {
            // the message did not reach the device buffer: send it from pinned memory instead (the receive side stays where it is)
            (void)hipGetLastError();
            if (hipHostMalloc(&hs, send_bytes, hipHostMallocDefault) == hipSuccess) {
                std::memcpy(hs, send.data(), send_bytes);
                src = hs;
            } else {
                // no pinned memory either: this rank's table is lost; say so in the status slot (8 bytes) if that much still goes through
                local = SOCP_ERR_HIP;
                const double st = local;
                (void)hipMemcpy(dsend + (count - 1), &st, sizeof(double), hipMemcpyHostToDevice);
            }
        }
        grc = gather(user, src, count, drecv);
        if (hs) (void)hipHostFree(hs);
        if (grc == 0) {
            if (pinned) std::memcpy(recv.data(), drecv, recv_bytes);
            else read_ok = hipMemcpy(recv.data(), drecv, recv_bytes, hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (pinned) { if (dsend) (void)hipHostFree(dsend); (void)hipHostFree(drecv); }
        else { if (dsend) (void)hipFree(dsend); (void)hipFree(drecv); }
        if (switched) (void)hipSetDevice(prev);
    } else {
        grc = gather(user, send.data(), count, recv.data());
    }
    // from here on the collective is behind every rank: returning can no longer strand anyone
    if (grc != 0) return SOCP_ERR_ARG;
    if (!read_ok) return SOCP_ERR_HIP;
    if (local != SOCP_OK) return local;
    for (int r = 0; r < world; r++) {
        const int theirs = (int)recv[(size_t)(r + 1) * count - 1];
        if (theirs != SOCP_OK) return theirs;                                 // another rank failed: no table to report
    }
    for (int r = 0; r < world; r++) {
        int a = 0, b = 0;
        socp_sweep_shard(P, r, world, &a, &b);
        for (int k = 0; k < b - a; k++) {
            const double *rec = &recv[(size_t)r * count + (size_t)k * W];
            std::memcpy(Zout + (size_t)(a + k) * n, rec, sizeof(double) * n);
            if (fnorm) fnorm[a + k] = rec[n];
            info[a + k] = (int)rec[n + 1];
            if (nfev_last) nfev_last[a + k] = (int)rec[n + 2];
            if (nfev_total) nfev_total[a + k] = (int)rec[n + 3];
            if (solves) solves[a + k] = (int)rec[n + 4];
        }
    }
    return SOCP_OK;
}
