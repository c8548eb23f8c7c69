/*
 * ref_driver.cpp -- C-ABI shim over the REFERENCE's own objects (test infrastructure).
 *
 * Compiled by oracle/Makefile together with the reference sources where they lie
 * (/root/reference/src/socp/odeTools.cpp, models/goddard/goddard.cpp,
 * models/doubleIntegrator/doubleIntegrator.cpp, models/covid19/covid19.cpp) into oracle/_ref/libsocp_ref.so.
 * Those three files need nothing the image lacks.  No reference source is copied
 * into this repository: this file only calls the reference's public interface
 * (model.hpp:77 ComputeTraj, odeTools.hpp:82 Model, model.hpp:375 Control,
 * model.hpp:384 Hamiltonian, odeTools.cpp:89 RK4).
 *
 * Uses: (1) pin oracle/socp_oracle.c (tests/test_oracle_vs_ref.py, make_golden.py);
 *       (2) bench.py cpu_baseline kind "reference": model::ComputeTraj over a batch,
 *           one model object per std::thread (baseline B1).
 */
#include <pthread.h>
#include <sched.h>

#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "models/goddard/goddard.hpp"
#include "models/doubleIntegrator/doubleIntegrator.hpp"
#include "models/covid19/covid19.hpp"

namespace {
struct RefModel {
    model *m = nullptr;
    goddard *g = nullptr;
    doubleIntegrator *di = nullptr;
    covid19 *cv = nullptr;
};

model::mstate to_vec(const double *X, int len) { return model::mstate(X, X + len); }
}  // namespace

extern "C" {

void *ref_goddard_new(int step_nbr)
{
    RefModel *r = new RefModel;
    r->g = new goddard(std::string(""), step_nbr);
    r->m = r->g;
    return r;
}

void *ref_dint_new(int model_order)
{
    RefModel *r = new RefModel;
    r->di = new doubleIntegrator(model_order, std::string(""));
    r->m = r->di;
    return r;
}

void *ref_covid_new()
{
    RefModel *r = new RefModel;
    r->cv = new covid19(std::string(""));
    r->m = r->cv;
    return r;
}

/* p = {R0, Tinf, Tinc, N, Imax, muI, umin, umax} */
int ref_covid_set(void *h, const double *p)
{
    RefModel *r = static_cast<RefModel *>(h);
    if (!r->cv) return -1;
    covid19::parameters_struct &q = r->cv->GetParameterData();
    q.R0 = p[0]; q.Tinf = p[1]; q.Tinc = p[2]; q.N = p[3]; q.Imax = p[4]; q.muI = p[5]; q.umin = p[6]; q.umax = p[7];
    return 0;
}

void ref_model_free(void *h)
{
    RefModel *r = static_cast<RefModel *>(h);
    delete r->m;
    delete r;
}

int ref_model_dim(void *h) { return static_cast<RefModel *>(h)->m->GetDim(); }
int ref_model_step_nbr(void *h) { return static_cast<RefModel *>(h)->m->stepNbr; }

int ref_goddard_set(void *h, const char *name, double v)
{
    RefModel *r = static_cast<RefModel *>(h);
    if (!r->g) return -1;
    try { r->g->SetParameterDataName(name, v); } catch (...) { return -2; }
    return 0;
}

int ref_dint_set(void *h, double u_max, double a_max, double muT)
{
    RefModel *r = static_cast<RefModel *>(h);
    if (!r->di) return -1;
    r->di->GetParameterData().u_max = u_max;
    r->di->GetParameterData().a_max = a_max;
    r->di->GetParameterData().muT = muT;
    return 0;
}

void ref_model_switching_update(void *h, const double *sw, int n)
{
    static_cast<RefModel *>(h)->m->SwitchingTimesUpdate(std::vector<real>(sw, sw + n));
}

int ref_model_rhs(void *h, double t, const double *X, int len, int is_jac, double *out, int cap)
{
    odeTools *o = static_cast<RefModel *>(h)->m;
    model::mstate d = o->Model(t, to_vec(X, len), is_jac);
    if ((int)d.size() > cap) return -1;
    std::memcpy(out, d.data(), sizeof(double) * d.size());
    return (int)d.size();
}

int ref_model_control(void *h, double t, const double *X, int len, double *out, int cap)
{
    model::mcontrol u = static_cast<RefModel *>(h)->m->Control(t, to_vec(X, len));
    if ((int)u.size() > cap) return -1;
    std::memcpy(out, u.data(), sizeof(double) * u.size());
    return (int)u.size();
}

int ref_model_hamiltonian(void *h, double t, const double *X, int len, int is_jac, double *out, int cap)
{
    model::mstate H = static_cast<RefModel *>(h)->m->Hamiltonian(t, to_vec(X, len), is_jac);
    if ((int)H.size() > cap) return -1;
    std::memcpy(out, H.data(), sizeof(double) * H.size());
    return (int)H.size();
}

void ref_rk4_step(void *h, double t, double *X, int len, double step, int is_jac)
{
    model *m = static_cast<RefModel *>(h)->m;
    model::mstate v = to_vec(X, len);
    odeTools::RK4(t, v, step, odeTools::modelStruct(m, is_jac));
    std::memcpy(X, v.data(), sizeof(double) * len);
}

void ref_model_traj(void *h, double t0, const double *X0, int len, double tf, int is_jac, double *Xf)
{
    model *m = static_cast<RefModel *>(h)->m;
    model::mstate v = m->ComputeTraj(t0, to_vec(X0, len), tf, 0, is_jac);
    std::memcpy(Xf, v.data(), sizeof(double) * len);
}

/* default residual blocks of model.hpp, isJac == 0 (rows a16 of SURVEY 8a) */
void ref_model_final_function(void *h, double tf, const double *Xtf, int len, const double *Xd,
                              const int *mode, int with_h, double *out)
{
    model *m = static_cast<RefModel *>(h)->m;
    int d = m->GetDim();
    std::vector<int> md(mode, mode + d);
    std::vector<real> f(d + (with_h ? 1 : 0));
    if (with_h) m->FinalHFunction(tf, to_vec(Xtf, len), to_vec(Xd, 2 * d), md, f, 0);
    else m->FinalFunction(tf, to_vec(Xtf, len), to_vec(Xd, 2 * d), md, f, 0);
    std::memcpy(out, f.data(), sizeof(double) * f.size());
}

/* Default residual blocks of the reference's header-only model.hpp (:90-328), both forms (rows a16 of SURVEY 8a):
 * which = 0 InitialFunction, 1 InitialHFunction, 2 FinalFunction, 3 FinalHFunction (X = state at the boundary, len 2d or
 * (2d+1)*2d; other = desired boundary state, 2d; mode = d state modes), 4 SwitchingTimesFunction (X, other = states
 * before / after the free interior time, same length; mode unused; goddard overrides it, goddard.cpp:343-370).
 * Returns the number of values written (the block's own size), -1 if cap is too small. */
int ref_model_block(void *h, int which, double t, const double *X, int len, const double *other, int other_len,
                    const int *mode, int is_jac, double *out, int cap)
{
    model *m = static_cast<RefModel *>(h)->m;
    const int d = m->GetDim();
    if (which == 4) {
        model::mstate f = m->SwitchingTimesFunction(t, to_vec(X, len), to_vec(other, other_len), is_jac);
        if ((int)f.size() > cap) return -1;
        std::memcpy(out, f.data(), sizeof(double) * f.size());
        return (int)f.size();
    }
    const bool with_h = (which == 1 || which == 3);
    const int count = is_jac ? (with_h ? (d + 1) * (2 * d + 1) : d * 2 * d) : (with_h ? d + 1 : d);
    if (count > cap) return -1;
    std::vector<int> md(mode, mode + d);
    std::vector<real> f(count, 0.0);
    const model::mstate Xv = to_vec(X, len), Ov = to_vec(other, other_len);
    switch (which) {
    case 0: m->InitialFunction(t, Xv, Ov, md, f, is_jac); break;
    case 1: m->InitialHFunction(t, Xv, Ov, md, f, is_jac); break;
    case 2: m->FinalFunction(t, Xv, Ov, md, f, is_jac); break;
    case 3: m->FinalHFunction(t, Xv, Ov, md, f, is_jac); break;
    default: return -2;
    }
    std::memcpy(out, f.data(), sizeof(double) * count);
    return count;
}

/*
 * CPU baseline B1: B Goddard trajectories through the reference's model::ComputeTraj,
 * split over `threads` std::threads, one goddard object per thread.  Returns seconds.
 * params = {C,b,KD,kr,u_max,mu1,mu2,singularControl}.
 */
double ref_goddard_traj_batch_pinned(int threads, const int *cpus, int step_nbr, const double *params, int B,
                                     const double *t0, const double *tf, const double *X0, double *Xf);

double ref_goddard_traj_batch(int threads, int step_nbr, const double *params, int B,
                              const double *t0, const double *tf, const double *X0, double *Xf)
{
    return ref_goddard_traj_batch_pinned(threads, nullptr, step_nbr, params, B, t0, tf, X0, Xf);
}

/* The same with thread k pinned to logical CPU cpus[k] (NULL: not pinned) -- SURVEY 8d: "P = all physical cores (and P = 1), pinned". */
double ref_goddard_traj_batch_pinned(int threads, const int *cpus, int step_nbr, const double *params, int B,
                                     const double *t0, const double *tf, const double *X0, double *Xf)
{
    static const char *names[8] = {"C", "b", "KD", "kr", "u_max", "mu1", "mu2", "singularControl"};
    if (threads < 1) threads = 1;
    std::vector<goddard *> models(threads);
    for (int k = 0; k < threads; k++) {
        models[k] = new goddard(std::string(""), step_nbr);
        for (int i = 0; i < 8; i++) models[k]->SetParameterDataName(names[i], params[i]);
    }
    auto work = [&](int k) {
        if (cpus) {
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(cpus[k], &set);
            (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
        }
        model *m = models[k];
        for (int b = k; b < B; b += threads) {
            model::mstate v = m->ComputeTraj(t0[b], to_vec(X0 + 14 * (size_t)b, 14), tf[b], 0, 0);
            std::memcpy(Xf + 14 * (size_t)b, v.data(), sizeof(double) * 14);
        }
    };
    cpu_set_t caller;
    const bool have_caller = pthread_getaffinity_np(pthread_self(), sizeof(caller), &caller) == 0;
    auto tic = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int k = 1; k < threads; k++) pool.emplace_back(work, k);
    work(0);
    for (auto &th : pool) th.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - tic).count();
    if (cpus && have_caller) (void)pthread_setaffinity_np(pthread_self(), sizeof(caller), &caller);   /* work(0) ran on the caller's thread */
    for (auto *g : models) delete g;
    return sec;
}

}  // extern "C"
