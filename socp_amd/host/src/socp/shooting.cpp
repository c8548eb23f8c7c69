// shooting.cpp -- host side of the `shooting` mirror (reference: shooting.cpp:21-1617).
//
// What stays on the host: the bookkeeping of node tables and unknown vector, the continuation
// loops, and the Newton iteration itself (library hybrd/hybrj, O(n^3) factor work).  What goes to
// the GPU: every trajectory.  One residual evaluation = one launch of numMulti trajectories; one
// forward-difference Jacobian = one launch of the n perturbed residuals (or of only the segments
// a column can change), instead of n sequential callbacks each running numMulti integrations.
#include "shooting.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <future>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>

#include "socp_hip.h"
#include "socp_solver.h"
#include "host_pool.hpp"     // socp_amd/csrc: the persistent worker pool the solver library uses too

struct shooting::data_struct {
    // node tables: desired / current (continuation-blended) / last converged (shooting.cpp:22-27)
    std::vector<real> timed, time, time_prec;
    std::vector<model::mstate> Xd, X, X_prec;
    int dim = 0, numMulti = 1, numThread = 1, numParam = 0;
    std::vector<real> tab_param, tab_param_temp;
    std::vector<std::vector<int> > mode_X;
    std::vector<int> mode_t;
    real continuationStepMin = 1e-12;
    // solver knobs, hard-coded in the reference constructor (shooting.cpp:95-105)
    int maxfev = 10000, scalingMode = 1, nprint = 0;
    real xtol = 1e-8, epsfcn = 1e-15, factor = 1;
    int info = 0, nfev = 0, njev = 0;
    std::atomic<int> stopFlag{0};
    bool dedup = true;
    socp_ctx *ctx = nullptr;      // the model's device context, valid during a solve
    long long hostTrajectories = 0;   // segments integrated on the host (models without device dynamics)
    std::unique_ptr<socp::Pool> pool;  // numThread > 1 on the host path: segment workers, started once (not per call)
};

namespace {
void die(const char *msg)
{
    std::cerr << std::endl << msg << std::endl;    // shooting.cpp:62-77: message then exit(1)
    std::exit(1);
}
void check_counts(int numMulti, int numThread)
{
    if (numMulti < 1) die("ERROR : numMulti should be superior or equal to 1");
    if (numThread < 1) die("ERROR : numThread should be superior or equal to 1");
}
// free-time rows of the host residual / Jacobian (models without device dynamics)
struct HostLayout {
    std::vector<int> free_row;     // residual row (= column of the time unknown) of node k's free-time equation, -1: none
    bool ok = false;
};
HostLayout host_layout(int M, int s, std::vector<int> const &mode_t, int numParam)
{
    // extra rows in the reference's nbrParam order (shooting.cpp:947-986): the boundary nodes take one whenever their time mode
    // is not FIXED (the predicate that also selects the H variants), interior nodes when FREE
    HostLayout L;
    L.free_row.assign(M + 1, -1);
    int row_next = s * M;
    for (int k = 0; k <= M; k++) {
        const bool boundary_node = (k == 0 || k == M);
        if (boundary_node ? mode_t[k] != model::FIXED : mode_t[k] == model::FREE) L.free_row[k] = row_next++;
    }
    L.ok = row_next == numParam;
    return L;
}

void device_check(int rc, socp_ctx *ctx, const char *where)
{
    if (rc != SOCP_OK) throw std::runtime_error(std::string(where) + ": " + socp_last_error(ctx));
}
}  // namespace

shooting::shooting(model &m, int numMulti, int numThread) : myModel(m), data(new data_struct)
{
    check_counts(numMulti, numThread);
    data->dim = myModel.GetDim();
    Resize(numMulti, numThread);
    myModel.SetODEIntPrecision(data->xtol);    // shooting.cpp:97
}

shooting::~shooting() { delete data; }

void shooting::Resize(int numMulti, int numThread) const
{
    check_counts(numMulti, numThread);
    data->numMulti = numMulti;
    data->numThread = numThread;
    const size_t nodes = (size_t)numMulti + 1;
    data->time_prec.resize(nodes); data->time.resize(nodes); data->timed.resize(nodes);
    data->X_prec.resize(nodes); data->X.resize(nodes); data->Xd.resize(nodes);
    data->mode_X.resize(nodes); data->mode_t.resize(nodes);
    data->numParam = 2 * data->dim * numMulti + (1 + numMulti);    // upper bound until SetMode (shooting.cpp:86,92)
    data->tab_param.reserve(data->numParam);
    data->tab_param_temp.reserve(data->numParam);
}

// shooting.cpp:165-182: fixed initial time/state, continuity inside, caller-chosen final modes
void shooting::SetMode(int const &mode_tf, std::vector<int> const &mode_Xf) const
{
    const int M = data->numMulti;
    data->mode_t[0] = model::FIXED;
    data->mode_X[0].assign(data->dim, model::FIXED);
    for (int i = 1; i < M; i++) {
        data->mode_t[i] = model::CONTINUOUS;
        data->mode_X[i].assign(data->dim, model::CONTINUOUS);
    }
    data->mode_t[M] = mode_tf;
    data->mode_X[M] = mode_Xf;
    data->numParam = 2 * data->dim * M + mode_tf;
    data->tab_param.resize(data->numParam);
    data->tab_param_temp.resize(data->numParam);
}

// shooting.cpp:185-199
void shooting::SetMode(std::vector<int> const &mode_t, std::vector<std::vector<int> > const &mode_X) const
{
    for (int i = 0; i <= data->numMulti; i++) data->mode_X[i] = mode_X[i];
    data->mode_t = mode_t;
    int nfree = 0;
    for (int m : data->mode_t) nfree += (m == model::FREE);
    data->numParam = 2 * data->dim * data->numMulti + nfree;
    data->tab_param.resize(data->numParam);
    data->tab_param_temp.resize(data->numParam);
}

namespace {
// unknown vector layout (shooting.cpp:228-243): node states 0..M-1, then the FREE times in node order
void pack_guess(int dim, int M, std::vector<model::mstate> const &X, std::vector<real> const &time,
                std::vector<int> const &mode_t, std::vector<real> &z)
{
    const int s = 2 * dim;
    for (int j = 0; j < M; j++)
        for (int i = 0; i < s; i++) z[j * s + i] = X[j][i];
    int k = s * M;
    for (int j = 0; j <= M; j++)
        if (mode_t[j] == model::FREE) z[k++] = time[j];
}
}  // namespace

// shooting.cpp:202-245: uniform time grid, interior node states by integrating the guess from Xi
void shooting::InitShooting(real const &ti, model::mstate const &Xi, real const &tf, model::mstate const &Xf) const
{
    const int M = data->numMulti;
    for (int i = 0; i <= M; i++) {
        const real t = ti + i * (tf - ti) / (M);
        data->time[i] = data->timed[i] = data->time_prec[i] = t;
    }
    data->X[0] = data->Xd[0] = data->X_prec[0] = Xi;
    for (int i = 1; i < M; i++) {
        const model::mstate Xn = Move(ti, Xi, data->time[i]);
        data->X[i] = data->Xd[i] = data->X_prec[i] = Xn;
    }
    data->X[M] = data->Xd[M] = data->X_prec[M] = Xf;
    pack_guess(data->dim, M, data->X, data->time, data->mode_t, data->tab_param);
}

// shooting.cpp:248-291
void shooting::InitShooting(std::vector<real> const &vt, std::vector<model::mstate> const &vX) const
{
    const int M = (int)vt.size() - 1;
    data->time_prec = data->time = data->timed = vt;
    data->X_prec = data->X = data->Xd = vX;
    pack_guess(data->dim, M, data->X, data->time, data->mode_t, data->tab_param);
}

void shooting::SetDesiredState(real const &ti, model::mstate const &Xi, real const &tf, model::mstate const &Xf) const
{
    data->timed[0] = ti; data->Xd[0] = Xi;
    data->timed[data->numMulti] = tf; data->Xd[data->numMulti] = Xf;
}

void shooting::SetDesiredState(std::vector<real> const &vt, std::vector<model::mstate> const &vX) const
{
    for (size_t i = 0; i < vt.size(); i++) { data->timed[i] = vt[i]; data->Xd[i] = vX[i]; }
}

int shooting::SolveOCP(real const &continuationStep) const
{
    return continuationStep <= 0 ? SolveShooting() : SolveShootingContinuation(continuationStep);
}

// shooting.cpp:329-348: the solve runs in a worker; on timeout the stop flag makes the next residual
// callback return -1, which aborts hybrd (shooting.cpp:873).  The worker is joined before returning.
int shooting::SolveOCP(real const &continuationStep, double const &timeoutMS) const
{
    std::packaged_task<int()> task([&]() { return SolveOCP(continuationStep); });
    std::future<int> result = task.get_future();
    std::thread worker(std::move(task));
    int info;
    if (result.wait_for(std::chrono::duration<double, std::milli>(timeoutMS)) == std::future_status::timeout) {
        data->stopFlag = -1;
        worker.join();
        data->stopFlag = 0;
        info = -1;
    } else {
        worker.join();
        info = result.get();
    }
    return info;
}

int shooting::SolveOCP(real const &continuationStep, real &Rdata, real const &Rgoal) const
{
    return SolveShootingContinuation(continuationStep <= 0 ? 1.0 : continuationStep, Rdata, Rgoal);
}

model::mstate shooting::Move(real const &ti, model::mstate const &Xi, real const &tf, int isJac) const
{
    return myModel.ComputeTraj(ti, Xi, tf, 0, isJac);
}

void shooting::Move(real const &ti, model::mstate const &Xi, real const &tf, model::mstate &Xf, int isJac) const
{
    Xf = myModel.ComputeTraj(ti, Xi, tf, 0, isJac);
}

// shooting.cpp:383-437: state at time tf on the trajectory stored in tab_param (clamped to [t0, t_final])
model::mstate shooting::Move(real const &tf, int isJac) const
{
    const int M = data->numMulti, s = 2 * data->dim;
    const real t0 = data->mode_t[0] == model::FIXED ? data->time[0] : data->tab_param[s * M];
    const real t_end = data->mode_t[M] == model::FIXED ? data->time[M] : data->tab_param[data->numParam - 1];
    const real target = (tf >= t0 && tf <= t_end) ? tf : t_end;

    std::vector<real> timeLine(M + 1);
    ComputeTimeLine(data->tab_param, timeLine);
    int seg = 0;
    while (timeLine[seg + 1] < target) seg++;
    model::mstate X1 = data->X[0];
    const int node = (seg > 0 && seg < M) ? seg : 0;       // shooting.cpp:428: only interior segments reload
    for (int i = 0; i < s; i++) X1[i] = data->tab_param[s * node + i];
    return Move(timeLine[seg], X1, target, isJac);
}

void shooting::Move(real const &tf, model::mstate &Xf, int) const { Xf = Move(tf); }   // shooting.cpp:440-444 drops isJac

void shooting::SetPrecision(real const &xtol) const
{
    data->xtol = xtol;
    myModel.SetODEIntPrecision(xtol);
}

void shooting::SetContinuationMinStep(real const &step) const { data->continuationStepMin = step; }
real shooting::GetParameters(int const &k) const { return data->tab_param[k]; }

real *shooting::GetParameters() const
{
    real *p = new real[data->numParam];
    std::copy(data->tab_param.begin(), data->tab_param.begin() + data->numParam, p);
    return p;
}

void shooting::GetParameters(std::vector<real> &v) const { v.assign(data->tab_param.begin(), data->tab_param.begin() + data->numParam); }

void shooting::GetSolution(std::vector<real> &vt, std::vector<model::mstate> &vX) const
{
    UpdateSolution();
    vt = data->time;
    for (int i = 0; i <= data->numMulti; i++) vX[i] = data->X[i];
}

std::vector<int> shooting::GetCallNumber() const { return std::vector<int>({data->nfev, data->njev}); }
model &shooting::GetModel() const { return myModel; }
void shooting::SetJacobianDedup(bool on) const { data->dedup = on; }

long long shooting::GetTrajectoryCount() const
{
    if (myModel.DeviceModelId() == 0) return data->hostTrajectories;
    long long traj = 0, launches = 0;
    socp_ctx_counters(myModel.DeviceContext(), &traj, &launches);
    return traj;
}

// shooting.cpp:496-544: replay the stored solution segment by segment with tracing on
void shooting::Trace() const
{
    UpdateSolution();
    const int M = data->numMulti, s = 2 * data->dim;
    std::vector<real> timeLine(M + 1);
    ComputeTimeLine(data->tab_param, timeLine);
    model::mstate X1 = data->X[0];
    for (int i = 0; i < s; i++) X1[i] = data->tab_param[i];
    for (int i = 0; i < M; i++) {
        myModel.ComputeTraj(timeLine[i], X1, timeLine[i + 1], 1, 0);
        if (i < M - 1)
            for (int j = 0; j < s; j++) X1[j] = data->tab_param[s * (i + 1) + j];
    }
}

// shooting.cpp:1462-1508: node times and states of the stored solution
void shooting::UpdateSolution() const
{
    const int M = data->numMulti, s = 2 * data->dim;
    const real t_end = data->mode_t[M] == model::FIXED ? data->time[M] : data->tab_param[data->numParam - 1];
    std::vector<real> timeLine(M + 1);
    ComputeTimeLine(data->tab_param, timeLine);
    model::mstate X1 = data->X[0];
    for (int i = 0; i < s; i++) X1[i] = data->tab_param[i];
    for (int i = 0; i <= M; i++) {
        data->time[i] = timeLine[i];
        data->X[i] = X1;
        if (i < M - 1) {
            for (int j = 0; j < s; j++) X1[j] = data->tab_param[s * (i + 1) + j];
        } else if (i == M - 1) {
            X1 = Move(t_end, 0);
        }
    }
}

// shooting.cpp:1579-1617.  FIXED and FREE nodes are junctions; CONTINUOUS nodes between two
// junctions are spaced uniformly.  Side effect kept: the model learns the FREE node times with
// index < numMulti as its switching times.
void shooting::ComputeTimeLine(std::vector<real> const &param, std::vector<real> &timeLine) const
{
    const int M = data->numMulti;
    std::vector<real> switching;
    int next_free = 2 * data->dim * M;
    int last = 0;
    for (int j = 0; j <= M; j++) {
        const int mode = data->mode_t[j];
        if (mode != model::FIXED && mode != model::FREE) continue;
        if (mode == model::FIXED) {
            timeLine[j] = data->time[j];
        } else {
            timeLine[j] = param[next_free++];
            if (j < M) switching.push_back(timeLine[j]);
        }
        for (int k = last + 1; k < j; k++)
            timeLine[k] = timeLine[last] + (k - last) * (timeLine[j] - timeLine[last]) / (j - last);
        last = j;
    }
    myModel.SwitchingTimesUpdate(switching);
}

void shooting::PushProblemToDevice() const
{
    const int M = data->numMulti, d = data->dim, s = 2 * d;
    std::vector<int> mx((size_t)(M + 1) * d);
    std::vector<double> xn((size_t)(M + 1) * s, 0.0);
    for (int i = 0; i <= M; i++) {
        for (int j = 0; j < d; j++) mx[(size_t)i * d + j] = data->mode_X[i][j];
        for (size_t j = 0; j < data->X[i].size() && j < (size_t)s; j++) xn[(size_t)i * s + j] = data->X[i][j];
    }
    device_check(socp_problem_set(data->ctx, M, data->mode_t.data(), mx.data(), data->time.data(), xn.data()),
                 data->ctx, "shooting: problem set-up");
    if (socp_problem_num_param(data->ctx) != data->numParam)
        throw std::runtime_error("shooting: device problem size disagrees with numParam");
}

// shooting.cpp:568-595
int shooting::SolveShooting() const
{
    const int n = data->numParam;
    for (int k = 0; k < n; k++) data->tab_param_temp[k] = data->tab_param[k];
    for (int i = 0; i <= data->numMulti; i++) {
        data->time[i] = data->timed[i];
        for (int j = 0; j < data->dim; j++) data->X[i][j] = data->Xd[i][j];
    }
    const int ret = SolveShootingFunction(n, data->tab_param_temp);
    if (ret == 1)
        for (int k = 0; k < n; k++) data->tab_param[k] = data->tab_param_temp[k];
    return ret;
}

namespace {
// bisection state of the discrete continuation loops (shooting.cpp:598-692, 695-778)
struct Homotopy {
    real step, b, b_prec = 0;
    explicit Homotopy(real s) : step(s), b(std::min<real>(s, 1.0)) {}
    void shrink() { b = b_prec + (b - b_prec) / 2; }
    void advance() { b_prec = b; b = std::min<real>(b + step, 1.0); }
};
}  // namespace

// shooting.cpp:598-692: homotopy on the boundary data, (1-b)*previous + b*desired
int shooting::SolveShootingContinuation(real const &continuationStep) const
{
    const int n = data->numParam;
    Homotopy h(continuationStep);
    auto blend = [&](real b) {
        for (int i = 0; i <= data->numMulti; i++) {
            data->time[i] = (1 - b) * data->time_prec[i] + b * data->timed[i];
            for (int j = 0; j < data->dim; j++) data->X[i][j] = (1 - b) * data->X_prec[i][j] + b * data->Xd[i][j];
        }
    };
    blend(h.b);
    for (int k = 0; k < n; k++) data->tab_param_temp[k] = data->tab_param[k];

    int ret = 0;
    for (bool running = true; running;) {
        ret = SolveShootingFunction(n, data->tab_param_temp);
        if (ret != 1) {
            if (std::fabs(h.b - h.b_prec) < data->continuationStepMin) running = false;
            h.shrink();
            for (int k = 0; k < n; k++) data->tab_param_temp[k] = data->tab_param[k];
            blend(h.b);
        } else if (h.b == 1) {
            running = false;
        } else {
            h.advance();
            for (int k = 0; k < n; k++) data->tab_param[k] = data->tab_param_temp[k];
            blend(h.b);
        }
    }
    if (ret == 1) {
        for (int k = 0; k < n; k++) data->tab_param[k] = data->tab_param_temp[k];
        for (int i = 0; i <= data->numMulti; i++) {
            data->time_prec[i] = data->timed[i];
            for (int j = 0; j < data->dim; j++) data->X_prec[i][j] = data->Xd[i][j];
        }
    }
    return ret;
}

// shooting.cpp:695-778: homotopy on one real model parameter, reached through a reference
int shooting::SolveShootingContinuation(real const &continuationStep, real &Rdata, real const &Rgoal) const
{
    const real Rstart = Rdata;
    const int n = data->numParam;
    Homotopy h(continuationStep);
    Rdata = (1 - h.b) * Rstart + h.b * Rgoal;
    for (int k = 0; k < n; k++) data->tab_param_temp[k] = data->tab_param[k];
    for (int i = 0; i <= data->numMulti; i++) {
        data->time[i] = data->timed[i];
        for (int j = 0; j < data->dim; j++) data->X[i][j] = data->Xd[i][j];
    }
    int ret = 0;
    for (bool running = true; running;) {
        ret = SolveShootingFunction(n, data->tab_param_temp);
        if (ret != 1) {
            if (std::fabs(h.b - h.b_prec) < data->continuationStepMin) running = false;
            h.shrink();
            for (int k = 0; k < n; k++) data->tab_param_temp[k] = data->tab_param[k];
            Rdata = (1 - h.b) * Rstart + h.b * Rgoal;
        } else if (h.b == 1) {
            running = false;
        } else {
            h.advance();
            for (int k = 0; k < n; k++) data->tab_param[k] = data->tab_param_temp[k];
            Rdata = (1 - h.b) * Rstart + h.b * Rgoal;
        }
    }
    if (ret == 1)
        for (int k = 0; k < n; k++) data->tab_param[k] = data->tab_param_temp[k];
    return ret;
}

// shooting.cpp:781-856: one Newton solve.  modelOrder 0 -> hybrd with the batched FD stage,
// modelOrder 1 -> hybrj with the variational Jacobian.
int shooting::SolveShootingFunction(int const &numParam, std::vector<real> &param) const
{
    const int n = numParam;
    std::vector<real> xscal(n, 1.0), fvec(n), fjac((size_t)n * n), r((size_t)n * (n + 1) / 2), qtf(n), wa1(n), wa2(n), wa3(n), wa4(n);
    if (myModel.DeviceModelId() == 0) {
        // A user class with the reference's host virtuals only: the reference's own scheme -- hybrd with n sequential
        // residual callbacks per forward-difference Jacobian (shooting.cpp:801-827), every trajectory on the host.
        data->ctx = nullptr;
        if (!host_layout(data->numMulti, 2 * data->dim, data->mode_t, n).ok) {
            std::cerr << std::endl << "ERROR : the first and the last time must be FIXED or FREE" << std::endl;
            return data->info = 0;                     // MINPACK's "improper input parameters"; the reference never throws here
        }
        if (myModel.modelOrder == 0) {
            data->info = hybrd(StaticHostShootingFunction, (void *)this, n, param.data(), fvec.data(), data->xtol, data->maxfev,
                               n - 1, n - 1, data->epsfcn, xscal.data(), data->scalingMode, data->factor, data->nprint, &data->nfev,
                               fjac.data(), n, r.data(), (int)r.size(), qtf.data(), wa1.data(), wa2.data(), wa3.data(), wa4.data());
        } else {
            // modelOrder 1: the class integrates its variational equations in Model(t, X, 1) (shooting.cpp:828-852)
            data->info = hybrj(StaticHostShootingFunctionJacobian, (void *)this, n, param.data(), fvec.data(), fjac.data(), n,
                               data->xtol, data->maxfev, xscal.data(), data->scalingMode, data->factor, data->nprint,
                               &data->nfev, &data->njev, r.data(), (int)r.size(), qtf.data(),
                               wa1.data(), wa2.data(), wa3.data(), wa4.data());
        }
        return data->info;
    }
    data->ctx = myModel.DeviceContext();           // re-packs parameters / step number (continuation mutates them)
    PushProblemToDevice();

    if (myModel.modelOrder == 0) {
        data->info = socp_hybrd_batched(StaticShootingFunction, StaticShootingFdJacobian, (void *)this, n, param.data(),
                                        fvec.data(), data->xtol, data->maxfev, n - 1, n - 1, data->epsfcn, xscal.data(),
                                        data->scalingMode, data->factor, data->nprint, &data->nfev, fjac.data(), n,
                                        r.data(), (int)r.size(), qtf.data(), wa1.data(), wa2.data(), wa3.data(), wa4.data());
    } else {
        data->info = hybrj(StaticShootingFunctionJacobian, (void *)this, n, param.data(), fvec.data(), fjac.data(), n,
                           data->xtol, data->maxfev, xscal.data(), data->scalingMode, data->factor, data->nprint,
                           &data->nfev, &data->njev, r.data(), (int)r.size(), qtf.data(),
                           wa1.data(), wa2.data(), wa3.data(), wa4.data());
    }
    return data->info;
}

// ---- models without device dynamics ------------------------------------------------------------------------------
// shooting.cpp:918-993 with the rows of SURVEY Appendix B: node k's free-time equation sits at 2 d M + (number of free-time
// rows before node k); segment i ends at node i + 1, whose continuity rows are [2d(i+1), 2d(i+2)).  Segments only share
// read-only inputs (param, the timeline) and write disjoint rows, so with numThread > 1 they are dealt out in the reference's
// contiguous blocks (shooting.cpp:1223-1231) to a pool of workers that lives as long as the shooting object -- the
// reference creates and joins numThread std::threads on every call (shooting.cpp:1152-1157).  Like there, the workers call
// ComputeTraj on ONE model object concurrently: the class must be re-entrant.
// the segments of one residual / Jacobian evaluation, serial or over the pool
void shooting::HostForEachSegment(std::function<void(int)> const &segment) const
{
    const int M = data->numMulti, T = std::min(data->numThread, M);
    if (T <= 1) { for (int i = 0; i < M; i++) segment(i); return; }
    if (!data->pool || data->pool->size() != T) data->pool.reset(new socp::Pool(T));
    const int base = M / T, rem = M % T;
    data->pool->run(T, [&](int t) {
        const int count = base + (t < rem ? 1 : 0), start = t * base + std::min(t, rem);     // shooting.cpp:1223-1231
        for (int i = start; i < start + count; i++) segment(i);
    });
}

void shooting::HostShootingFunction(std::vector<real> const &param, std::vector<real> &fvec) const
{
    const int M = data->numMulti, d = data->dim, s = 2 * d;
    std::vector<real> timeLine(M + 1);
    ComputeTimeLine(param, timeLine);
    const HostLayout L = host_layout(M, s, data->mode_t, data->numParam);
    if (!L.ok) {
        // a boundary time that is neither FIXED nor FREE: the reference would write past its rows here.  No solver path throws
        // (SolveShootingFunction refuses such a layout with info = 0 before the first callback); a direct caller gets NaN rows.
        std::fill(fvec.begin(), fvec.end(), std::nan(""));
        return;
    }
    const std::vector<int> &free_row = L.free_row;

    auto boundary = [&](int node, real t, model::mstate const &Xt, int first_row) {
        // Initial[H]Function at node 0, Final[H]Function at node M (model.hpp:90-290); the H variants add the free-time row
        const bool with_h = data->mode_t[node] != model::FIXED;
        std::vector<real> rows(d + (with_h ? 1 : 0));
        if (node == 0) {
            if (with_h) myModel.InitialHFunction(t, Xt, data->X[0], data->mode_X[0], rows, 0);
            else myModel.InitialFunction(t, Xt, data->X[0], data->mode_X[0], rows, 0);
        } else {
            if (with_h) myModel.FinalHFunction(t, Xt, data->X[M], data->mode_X[M], rows, 0);
            else myModel.FinalFunction(t, Xt, data->X[M], data->mode_X[M], rows, 0);
        }
        for (int k = 0; k < d; k++) fvec[first_row + k] = rows[k];
        if (with_h) fvec[free_row[node]] = rows[d];
    };

    HostForEachSegment([&](int i) {
        const real ta = timeLine[i], tb = timeLine[i + 1];
        model::mstate Xstart = data->X[0];
        Xstart.resize(s);
        for (int k = 0; k < s; k++) Xstart[k] = param[s * i + k];
        const model::mstate Xend = Move(ta, Xstart, tb, 0);
        if (i == 0) boundary(0, ta, Xstart, 0);
        if (i < M - 1) {
            model::mstate Xnext(s);
            for (int k = 0; k < s; k++) Xnext[k] = param[s * (i + 1) + k];
            if (data->mode_t[i + 1] == model::FREE) fvec[free_row[i + 1]] = myModel.SwitchingTimesFunction(tb, Xend, Xnext, 0)[0];
            // shooting::MultipleShootingFunction, value form (shooting.cpp:1511-1576)
            for (int j = 0; j < d; j++) {
                const int row = s * (i + 1) + j;
                switch (data->mode_X[i + 1][j]) {
                case model::FIXED:
                    fvec[row] = Xend[j] - data->X[i + 1][j];
                    fvec[row + d] = Xnext[j] - data->X[i + 1][j];
                    break;
                case model::FREE: {
                    model::mstate rows(fvec.begin() + s * (i + 1), fvec.begin() + s * (i + 2));
                    myModel.SwitchingStateFunction(tb, j, Xend, Xnext, data->X[i + 1], rows, 0);
                    for (int k = 0; k < s; k++) fvec[s * (i + 1) + k] = rows[k];
                    break;
                }
                default:
                    fvec[row] = Xend[j] - Xnext[j];
                    fvec[row + d] = Xend[j + d] - Xnext[j + d];
                }
            }
        } else {
            boundary(M, tb, Xend, d);
        }
    });
    data->hostTrajectories += M;
}

// The analytic shooting Jacobian of a model WITHOUT device dynamics whose Model(t, X, 1) integrates the variational equations
// (modelOrder 1: shooting.cpp:996-1130 assembled from the Jacobian forms of the model's virtuals, model.hpp:104-120,149-183,
// 305-326, and of MultipleShootingFunction, shooting.cpp:1511-1576).  fjac: n x n, COLUMN-major as hybrj wants it (the
// reference fills a row-major copy and transposes, :889-893).  Kept as upstream: a segment's start time does not enter, and
// the time term of a FREE interior node is first written one column beyond its block (:1070) before it lands in its own.
void shooting::HostShootingFunctionJacobian(std::vector<real> const &param, std::vector<real> &fjac) const
{
    const int M = data->numMulti, d = data->dim, s = 2 * d, n = data->numParam, w = s + 1, wide = 2 * s + 1;
    std::fill(fjac.begin(), fjac.end(), 0.0);
    std::vector<real> timeLine(M + 1);
    ComputeTimeLine(param, timeLine);
    const HostLayout L = host_layout(M, s, data->mode_t, n);
    if (!L.ok) { std::fill(fjac.begin(), fjac.end(), std::nan("")); return; }
    const std::vector<int> &free_row = L.free_row;
    auto J = [&](int row, int col) -> real & { return fjac[(size_t)row + (size_t)n * col]; };
    // [X ; identity]: the augmented state a segment starts from (shooting.cpp:1003-1005, 1062-1064)
    auto augmented = [&](int node) {
        model::mstate Xa((size_t)w * s, 0.0);
        for (int k = 0; k < s; k++) { Xa[k] = param[s * node + k]; Xa[(size_t)s * (k + 1) + k] = 1; }
        return Xa;
    };
    // rows of a boundary block (value rows, then the H row of a free boundary time) against the unknowns of node `node`
    auto boundary = [&](bool initial, real t, model::mstate const &Xt, model::mstate const &Xother, int first_row, int node) {
        const int bnode = initial ? 0 : M;
        const bool with_h = data->mode_t[bnode] != model::FIXED;
        const int width = with_h ? w : s;
        std::vector<real> blk((size_t)(d + (with_h ? 1 : 0)) * width, 0.0);
        if (initial) {
            if (with_h) myModel.InitialHFunction(t, Xt, Xother, data->mode_X[0], blk, 1);
            else myModel.InitialFunction(t, Xt, Xother, data->mode_X[0], blk, 1);
        } else {
            if (with_h) myModel.FinalHFunction(t, Xt, Xother, data->mode_X[M], blk, 1);
            else myModel.FinalFunction(t, Xt, Xother, data->mode_X[M], blk, 1);
        }
        for (int k = 0; k < d; k++)
            for (int j = 0; j < s; j++) J(first_row + k, s * node + j) = blk[(size_t)width * k + j];
        if (with_h) {
            const int fr = free_row[bnode];
            for (int k = 0; k < d; k++) J(first_row + k, fr) = blk[(size_t)w * k + s];
            for (int j = 0; j < s; j++) J(fr, s * node + j) = blk[(size_t)w * d + j];
            J(fr, fr) = blk[(size_t)w * d + s];
        }
    };

    HostForEachSegment([&](int i) {
        const real ta = timeLine[i], tb = timeLine[i + 1];
        const model::mstate Xa = augmented(i);
        const model::mstate Xend = Move(ta, Xa, tb, 1);
        if (i == 0) boundary(true, ta, Xa, data->mode_t[0] == model::FIXED ? Xa : data->X[0], 0, 0);
        if (i < M - 1) {
            const int index = s * (i + 1), tmode = data->mode_t[i + 1];
            const model::mstate Xp = augmented(i + 1);
            std::vector<real> ms((size_t)s * wide, 0.0);
            HostMultipleShootingJacobian(tb, Xend, Xp, data->X[i + 1], data->mode_X[i + 1], tmode, ms);
            const int cols = tmode == model::FREE ? wide : 2 * s;
            for (int k = 0; k < s; k++)
                for (int j = 0; j < cols; j++) J(index + k, index - s + j) = ms[(size_t)wide * k + j];
            if (tmode == model::FREE) {
                const int fr = free_row[i + 1];
                for (int k = 0; k < s; k++) J(index + k, fr) = ms[(size_t)wide * k + 2 * s];
                const model::mstate sw = myModel.SwitchingTimesFunction(tb, Xend, Xp, 1);
                for (int j = 0; j < 2 * s; j++) J(fr, index - s + j) = sw[j];
                J(fr, fr) = sw[2 * s];
            }
        } else {
            boundary(false, tb, Xend, data->X[M], d, i);
        }
    });
    data->hostTrajectories += M;
}

// shooting::MultipleShootingFunction, Jacobian form (shooting.cpp:1524-1555): rows of stride 4d + 1 = [d/dX(t-) | d/dX+ | d/dt]
void shooting::HostMultipleShootingJacobian(real const &t, model::mstate const &X, model::mstate const &Xp, model::mstate const &Xd,
                                            std::vector<int> const &mode_X, int mode_t, std::vector<real> &rows) const
{
    const int d = data->dim, s = 2 * d, wide = 2 * s + 1;
    model::mstate fxt, fxp;
    if (mode_t == model::FREE) {
        fxt = myModel.Model(t, model::mstate(X.begin(), X.begin() + s), 0);
        fxp = myModel.Model(t, model::mstate(Xp.begin(), Xp.begin() + s), 0);
    }
    for (int j = 0; j < d; j++) {
        real *rs = &rows[(size_t)wide * j], *rc = &rows[(size_t)wide * (j + d)];     // state row, costate (or second) row
        switch (mode_X[j]) {
        case model::FIXED:                       // X_j(t-) = Xd_j and X+_j = Xd_j: both sides pinned
            for (int i = 0; i < s; i++) { rs[i] = X[(size_t)s * (j + 1) + i]; rc[s + i] = Xp[(size_t)s * (j + 1) + i]; }
            if (mode_t == model::FREE) { rs[2 * s] = fxt[j]; rc[2 * s] = fxp[j]; }
            break;
        case model::FREE: {
            model::mstate all(rows.begin(), rows.end());
            myModel.SwitchingStateFunction(t, j, X, Xp, Xd, all, 1);
            std::copy(all.begin(), all.end(), rows.begin());
            break;
        }
        default:                                 // CONTINUOUS: state and costate jumps vanish
            for (int i = 0; i < s; i++) {
                rs[i] = X[(size_t)s * (j + 1) + i];          rs[s + i] = -Xp[(size_t)s * (j + 1) + i];
                rc[i] = X[(size_t)s * (j + d + 1) + i];      rc[s + i] = -Xp[(size_t)s * (j + d + 1) + i];
            }
            if (mode_t == model::FREE) { rs[2 * s] = fxt[j] - fxp[j]; rc[2 * s] = fxt[j + d] - fxp[j + d]; }
        }
    }
}

std::vector<real> shooting::ResidualAt(std::vector<real> const &param) const
{
    const int n = data->numParam;
    if ((int)param.size() != n) throw std::invalid_argument("shooting::ResidualAt: wrong number of unknowns");
    for (int i = 0; i <= data->numMulti; i++) {                       // as SolveShooting: the desired boundary data (shooting.cpp:575-580)
        data->time[i] = data->timed[i];
        for (int j = 0; j < data->dim; j++) data->X[i][j] = data->Xd[i][j];
    }
    std::vector<real> f(n, 0.0);
    if (myModel.DeviceModelId() == 0) {
        HostShootingFunction(param, f);
    } else {
        data->ctx = myModel.DeviceContext();
        PushProblemToDevice();
        if (StaticShootingFunction((void *)this, n, param.data(), f.data(), 1) < 0) throw std::runtime_error("shooting::ResidualAt: residual evaluation failed");
    }
    return f;
}

int shooting::StaticHostShootingFunction(void *userdata, int n, const real *param, real *fvec, int)
{
    const shooting *self = static_cast<const shooting *>(userdata);
    std::vector<real> z(param, param + n), f(n, 0.0);
    self->HostShootingFunction(z, f);
    std::copy(f.begin(), f.end(), fvec);
    return self->data->stopFlag;
}

// shooting.cpp:877-915 for a model without device dynamics: iflag 1 -> residual, 2 -> analytic Jacobian
int shooting::StaticHostShootingFunctionJacobian(void *userdata, int n, const real *param, real *fvec, real *fjac, int ldfjac, int iflag)
{
    const shooting *self = static_cast<const shooting *>(userdata);
    std::vector<real> z(param, param + n);
    if (iflag == 1) {
        std::vector<real> f(n, 0.0);
        self->HostShootingFunction(z, f);
        std::copy(f.begin(), f.end(), fvec);
    } else {
        if (ldfjac != n) return -2;
        std::vector<real> jac((size_t)n * n);
        self->HostShootingFunctionJacobian(z, jac);
        std::copy(jac.begin(), jac.end(), fjac);
    }
    return self->data->stopFlag;
}

// shooting.cpp:859-874: residual callback.  The timeline is also computed on the host so that the
// model sees the same SwitchingTimesUpdate calls as with the reference.
int shooting::StaticShootingFunction(void *userdata, int n, const real *param, real *fvec, int)
{
    const shooting *self = static_cast<const shooting *>(userdata);
    std::vector<real> z(param, param + n), timeLine(self->data->numMulti + 1);
    self->ComputeTimeLine(z, timeLine);
    if (socp_residual_batch(self->data->ctx, 1, param, fvec) != SOCP_OK) {
        std::cerr << "shooting: " << socp_last_error(self->data->ctx) << std::endl;
        return -2;
    }
    return self->data->stopFlag;
}

// the n forward-difference residuals of MINPACK fdjac1 as one device batch
int shooting::StaticShootingFdJacobian(void *userdata, int n, const real *param, const real *fvec, real epsfcn, real *fjac, int ldfjac)
{
    const shooting *self = static_cast<const shooting *>(userdata);
    if (ldfjac != n) return -2;
    if (socp_fd_jacobian(self->data->ctx, param, fvec, epsfcn, fjac, self->data->dedup ? 1 : 0) != SOCP_OK) {
        std::cerr << "shooting: " << socp_last_error(self->data->ctx) << std::endl;
        return -2;
    }
    // fdjac1 leaves the model with the switching times of its last column (x restored except the
    // callbacks' side effect); reproduce the end state: last perturbed vector = z + h e_{n-1}
    std::vector<real> z(param, param + n), timeLine(self->data->numMulti + 1);
    {
        const real eps = std::sqrt(std::max<real>(epsfcn, 2.220446049250313e-16));
        real h = eps * std::fabs(z[n - 1]);
        if (h == 0) h = eps;
        z[n - 1] += h;
    }
    self->ComputeTimeLine(z, timeLine);
    return self->data->stopFlag;
}

// shooting.cpp:877-915: hybrj callback, iflag 1 -> residual, 2 -> variational Jacobian (column-major)
int shooting::StaticShootingFunctionJacobian(void *userdata, int n, const real *param, real *fvec, real *fjac, int ldfjac, int iflag)
{
    const shooting *self = static_cast<const shooting *>(userdata);
    std::vector<real> z(param, param + n), timeLine(self->data->numMulti + 1);
    self->ComputeTimeLine(z, timeLine);
    int rc;
    if (iflag == 1) rc = socp_residual_batch(self->data->ctx, 1, param, fvec);
    else rc = (ldfjac == n) ? socp_var_jacobian(self->data->ctx, param, fjac) : SOCP_ERR_ARG;
    if (rc != SOCP_OK) {
        std::cerr << "shooting: " << socp_last_error(self->data->ctx) << std::endl;
        return -2;
    }
    return self->data->stopFlag;
}
