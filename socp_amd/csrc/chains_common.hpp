// chains_common.hpp -- what the two lock-step engines share (batchsolve.cpp: solvers on the host; batchsolve_dev.cpp: solvers
// on the device): a chain's homotopy state with the reference's bisection rules (shooting.cpp:598-692 / 695-778) and the
// per-chain parameter / boundary blocks the kernels read.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/socp_hip.h"
#include "../../include/socp_solver.h"

namespace socp {
namespace chains {

inline double blend(double b, double a0, double a1) { return (1 - b) * a0 + b * a1; }

struct ChainCore {
    // homotopy state (shooting.cpp:598-692 / 695-778; the host mirror's `Homotopy`)
    double b = 1, b_prec = 0;
    bool finished = false;
    int info = 0, nfev_last = 0, njev_last = 0, nfev_total = 0, solves = 0;
    double fnorm = 0;
    std::vector<double> committed;      // tab_param: the unknowns of the last converged solve (or the start)
};

// per-chain packed parameters and boundary tables (dev_common.hpp "per-problem blocks"), kept on the host and staged per round
struct Blocks {
    int kind = SOCP_CHAIN_PLAIN, param_index = 0, P = 0, nparams = 0, stride = 0, nodes = 0, S = 0, dim = 0;
    bool pp_params = false, pp_bound = false;
    const double *goal = nullptr, *time_prev = nullptr, *x_prev = nullptr, *time_goal = nullptr, *x_goal = nullptr;
    std::vector<double> pblock, rstart, tblock, xblock;

    void init(int P_, const socp_chain_options &opt, int nparams_, int nodes_, int S_, int dim_, const double *params, const double *shared_params,
              const double *shared_sw, const double *goal_, const double *time_prev_, const double *x_prev_, const double *time_goal_,
              const double *x_goal_)
    {
        P = P_; kind = opt.kind; param_index = opt.param_index; nparams = nparams_; stride = nparams_ + 2; nodes = nodes_; S = S_; dim = dim_;
        goal = goal_; time_prev = time_prev_; x_prev = x_prev_; time_goal = time_goal_; x_goal = x_goal_;
        pp_params = params != nullptr || kind == SOCP_CHAIN_PARAM;
        pp_bound = kind == SOCP_CHAIN_DATA || (time_goal && x_goal);
        pblock.assign(pp_params ? (size_t)P * stride : 0, 0.0);
        rstart.assign(P, 0.0);
        tblock.assign(pp_bound ? (size_t)P * nodes : 0, 0.0);
        xblock.assign(pp_bound ? (size_t)P * nodes * S : 0, 0.0);
        for (int p = 0; p < P; p++) {
            if (pp_params) {
                double *blk = &pblock[(size_t)p * stride];
                std::memcpy(blk, params ? params + (size_t)p * nparams : shared_params, sizeof(double) * nparams);
                blk[nparams] = shared_sw[0]; blk[nparams + 1] = shared_sw[1];
                rstart[p] = kind == SOCP_CHAIN_PARAM ? blk[param_index] : 0.0;
            }
            if (pp_bound && kind != SOCP_CHAIN_DATA) {
                std::memcpy(&tblock[(size_t)p * nodes], time_goal + (size_t)p * nodes, sizeof(double) * nodes);
                std::memcpy(&xblock[(size_t)p * nodes * S], x_goal + (size_t)p * nodes * S, sizeof(double) * nodes * S);
            }
        }
    }
    // the blocks of chain p at homotopy value b (shooting.cpp:609-611 / :704)
    void set(int p, double b)
    {
        if (kind == SOCP_CHAIN_PARAM) pblock[(size_t)p * stride + param_index] = blend(b, rstart[p], goal[p]);
        if (kind == SOCP_CHAIN_DATA) {
            for (int i = 0; i < nodes; i++) {
                tblock[(size_t)p * nodes + i] = blend(b, time_prev[(size_t)p * nodes + i], time_goal[(size_t)p * nodes + i]);
                for (int j = 0; j < dim; j++) {
                    const size_t e = ((size_t)p * nodes + i) * S + j;
                    xblock[e] = blend(b, x_prev[e], x_goal[e]);
                }
            }
        }
    }
    // copy chain p's blocks into slot k of a round's staging arrays (any of which may be null)
    void stage(int p, int k, double *hP, double *hT, double *hX) const
    {
        if (pp_params && hP) std::memcpy(hP + (size_t)k * stride, &pblock[(size_t)p * stride], sizeof(double) * stride);
        if (pp_bound && hT) std::memcpy(hT + (size_t)k * nodes, &tblock[(size_t)p * nodes], sizeof(double) * nodes);
        if (pp_bound && hX) std::memcpy(hX + (size_t)k * nodes * S, &xblock[(size_t)p * nodes * S], sizeof(double) * nodes * S);
    }
};

// Chain logic at the end of one Newton solve: the bisection rules of shooting.cpp:627-660 / 724-760.  x = the solver's final
// iterate.  Returns true when the chain goes on with another solve, started from `next` (blocks already moved).
inline bool after_solve(const socp_chain_options &opt, Blocks &blk, int p, ChainCore &c, const double *x, int n, int info, int nfev,
                        int njev, std::vector<double> &next)
{
    c.info = info;
    c.nfev_last = nfev;
    c.njev_last = njev;
    c.nfev_total += nfev;
    c.solves++;
    if (opt.kind == SOCP_CHAIN_PLAIN) {
        c.committed.assign(x, x + n);                         // multi-start: the final iterate, whatever info says
        c.finished = true;
        return false;
    }
    if (info < 0) { c.finished = true; return false; }        // aborted (round limit): no further homotopy step
    bool running = true;
    if (info != 1) {
        if (std::fabs(c.b - c.b_prec) < opt.step_min) running = false;
        // the halving has stopped moving b (only reachable with step_min = 0, where the reference's loop never ends)
        if (c.b == c.b_prec) running = false;
        c.b = c.b_prec + (c.b - c.b_prec) / 2;
        next = c.committed;
    } else if (c.b == 1) {
        running = false;
        c.committed.assign(x, x + n);
    } else {
        c.b_prec = c.b;
        c.b = std::min(c.b + opt.step, 1.0);
        c.committed.assign(x, x + n);
        next = c.committed;
    }
    blk.set(p, c.b);                                          // the reference also moves Rdata / the boundary data on the failing exit
    if (!running) { c.finished = true; return false; }
    return true;
}

// The stream of a round's residual-type work.  A round's residual launch and Jacobian launch must run CONCURRENTLY (each is one
// trajectory latency long).  HIP multiplexes streams of one priority onto a few hardware queues in creation order, so in a process
// that holds other streams two of ours can land on the same queue and serialise (measured inside bench.py: 40 rounds took 0.90 s
// instead of 0.64 s of launches).  Streams of another priority come from their own queue pool; the residual work is the small,
// latency-critical part of a round, so it gets the high-priority one.
inline hipError_t create_residual_stream(hipStream_t *st)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
        hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest) == hipSuccess)
        return hipSuccess;
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

inline int validate(const socp_chain_options *opt, int nparams, const double *goal, const double *time_prev, const double *x_prev,
                    const double *time_goal, const double *x_goal)
{
    const int kind = opt->kind;
    if (kind != SOCP_CHAIN_PLAIN && kind != SOCP_CHAIN_PARAM && kind != SOCP_CHAIN_DATA) return SOCP_ERR_ARG;
    if (kind == SOCP_CHAIN_PARAM && (!goal || opt->param_index < 0 || opt->param_index >= nparams)) return SOCP_ERR_ARG;
    if (kind == SOCP_CHAIN_DATA && (!time_prev || !x_prev || !time_goal || !x_goal)) return SOCP_ERR_ARG;
    if (kind != SOCP_CHAIN_PLAIN && !(opt->step > 0)) return SOCP_ERR_ARG;
    // continuationStepMin: a negative or NaN value can never end the bisection of a chain whose solves keep failing (0 can: the
    // halving stops moving b after ~55 steps, see after_solve)
    if (kind != SOCP_CHAIN_PLAIN && !(opt->step_min >= 0)) return SOCP_ERR_ARG;
    return SOCP_OK;
}

}  // namespace chains
}  // namespace socp
