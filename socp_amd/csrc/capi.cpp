// capi.cpp -- implementation of include/socp_hip.h (context, device tables, launch dispatch).
// Host logic only: every number the caller receives was computed by a gfx950 kernel.
#include "../../include/socp_hip.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>

#include <map>
#include <mutex>

#include "../../include/socp_plugin.h"
#include "launch.hpp"

using namespace socp;

namespace {

// socp_last_error(NULL) reports the calling thread's last creation failure: concurrent socp_ctx_create calls may fail at once
thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

}  // namespace

struct socp_ctx {
    int model_id = 0;
    const ModelLaunchers *vt = nullptr;   // table-driven model (out-of-tree plugin, or in-tree interceptor): its launch table
    const ModelLaunchers *vt_fast = nullptr;   // the same model's throughput flavour, when it has one
    int device = 0;
    int dim = 0, S = 0, nu = 3;
    int nparams = 0;
    int variant = SOCP_VARIANT_AUTO;
    ModelParams P{};
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    // shooting problem (host copy + device tables)
    bool has_problem = false;
    int M = 0, n = 0;
    std::vector<int> mode_t, mode_x, node_kind, lo, hi, ft_row;
    std::vector<double> time, xnode;
    int sw_node0 = -1, sw_node1 = -1;
    DevBuf d_tables;                 // one allocation holding every table
    ProblemDev pb{};
    DevBuf d_pairs_full, d_pairs_dedup;
    int T_full = 0, T_dedup = 0;

    // grow-only staging for the host-pointer entry points
    DevBuf s_t0, s_tf, s_sw, s_in, s_out, s_aux, s_var;

    hipStream_t aux_stream = nullptr; // socp_ctx_aux_stream: created on first use
    bool blocks_smooth_hint = false;  // socp_problem_blocks_all_smooth: every per-problem parameter block has mu2 > 0
    long long n_traj = 0, n_launch = 0;
    std::string err;
};

namespace {

// Registered plugin models.  Contexts keep a pointer to their entry (map nodes do not move; an id registered twice keeps its
// node); registration and look-up may come from different threads.
std::map<int, ModelLaunchers> &plugins()
{
    static std::map<int, ModelLaunchers> table;
    return table;
}
std::mutex &plugins_lock()
{
    static std::mutex m;
    return m;
}

int fail(socp_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

int hip_fail(socp_ctx *c, hipError_t e, const char *what)
{
    return fail(c, SOCP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(c, call)                                                 \
    do {                                                                 \
        hipError_t e__ = (call);                                         \
        if (e__ != hipSuccess) return hip_fail((c), e__, #call);         \
    } while (0)

bool use_fast(const socp_ctx *c)
{
    // AUTO keeps the reference operation order: it is the variant every parity claim is made on.
    return c->variant == SOCP_VARIANT_LANE_FAST;
}

// launch table of a table-driven model for the current variant / integrator
// (a table carries its own adaptive-integrator instantiations, so the throughput table serves both integrators)
const ModelLaunchers *table_of(const socp_ctx *c)
{
    return (c->variant == SOCP_VARIANT_LANE_FAST && c->vt_fast) ? c->vt_fast : c->vt;
}

hipError_t run_traj(socp_ctx *c, int B, const double *t0, const double *tf, const double *sw,
                    const double *X0, double *Xf)
{
    c->n_traj += B; c->n_launch += 1;
    if (c->vt) return table_of(c)->traj(c->stream, c->P, B, t0, tf, sw, X0, Xf);
    return use_fast(c) ? traj_fast(c->model_id, c->stream, c->P, B, t0, tf, sw, X0, Xf)
                       : traj_exact(c->model_id, c->stream, c->P, B, t0, tf, sw, X0, Xf);
}

hipError_t run_residual(socp_ctx *c, int B, const double *Z, double *F)
{
    c->n_traj += (long long)B * c->M; c->n_launch += 1;
    if (c->vt) return table_of(c)->residual(c->stream, c->P, c->pb, B, Z, F);
    return use_fast(c) ? residual_fast(c->model_id, c->stream, c->P, c->pb, B, Z, F)
                       : residual_exact(c->model_id, c->stream, c->P, c->pb, B, Z, F);
}

hipError_t run_fdjac(socp_ctx *c, int np, int T, const int2 *pairs, const double *z, const double *fvec,
                     double eps, double *fjac)
{
    c->n_traj += (long long)np * T; c->n_launch += 1;
    if (c->vt) return table_of(c)->fdjac(c->stream, c->P, c->pb, np, T, pairs, z, fvec, eps, fjac);
    return use_fast(c) ? fdjac_fast(c->model_id, c->stream, c->P, c->pb, np, T, pairs, z, fvec, eps, fjac)
                       : fdjac_exact(c->model_id, c->stream, c->P, c->pb, np, T, pairs, z, fvec, eps, fjac);
}

hipError_t run_fdrows(socp_ctx *c, int np, const double *z, double eps, double *rows)
{
    c->n_traj += (long long)np * (c->n + 1) * c->M; c->n_launch += 1;
    if (c->vt) return table_of(c)->fdrows(c->stream, c->P, c->pb, np, z, eps, rows);
    return use_fast(c) ? fdrows_fast(c->model_id, c->stream, c->P, c->pb, np, z, eps, rows)
                       : fdrows_exact(c->model_id, c->stream, c->P, c->pb, np, z, eps, rows);
}

// variational equations on the device: the in-tree double integrator, or a table-driven model whose table carries them
bool has_var(const socp_ctx *c)
{
    return c->vt ? (c->vt->var_traj && c->vt->var_jacobian && c->vt->var_eval) : c->model_id == SOCP_MODEL_DOUBLE_INTEGRATOR;
}

double fd_eps(double epsfcn) { return std::sqrt(epsfcn > DBL_EPSILON ? epsfcn : DBL_EPSILON); }

}  // namespace

extern "C" {

const char *socp_last_error(const socp_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int socp_ctx_create(socp_ctx **out, int model_id, int device)
{
    if (!out) return fail(nullptr, SOCP_ERR_ARG, "socp_ctx_create: null output pointer");
    *out = nullptr;
    const ModelLaunchers *vt = nullptr;
    if (model_id >= SOCP_PLUGIN_ID_MIN) {
        std::lock_guard<std::mutex> guard(plugins_lock());
        auto it = plugins().find(model_id);
        if (it != plugins().end()) vt = &it->second;
    }
    if (model_id == SOCP_MODEL_INTERCEPTOR) vt = interceptor_launchers();   // in-tree, table-driven (kernels_interceptor.hip)
    if (!vt && model_id != SOCP_MODEL_GODDARD && model_id != SOCP_MODEL_DOUBLE_INTEGRATOR && model_id != SOCP_MODEL_COVID19)
        return fail(nullptr, SOCP_ERR_UNSUPPORTED, "socp_ctx_create: unknown model id (no device dynamics; plugins: socp_plugin_load)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SOCP_ERR_NO_DEVICE,
                    "socp_ctx_create: no HIP device visible -- this library has no CPU path");
    if (device < 0) { e = hipGetDevice(&device); if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDevice"); }
    if (device >= ndev) return fail(nullptr, SOCP_ERR_ARG, "socp_ctx_create: device index out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");

    socp_ctx *c = new socp_ctx;
    c->model_id = model_id;
    c->vt = vt;
    if (model_id == SOCP_MODEL_INTERCEPTOR) c->vt_fast = interceptor_launchers_fast();
    c->device = device;
    if (vt) {
        c->dim = vt->dim; c->nparams = vt->nparams; c->nu = vt->control_dim;
        std::memcpy(c->P.p, vt->default_params, sizeof(double) * kMaxParams);
        c->P.sw0 = c->P.sw1 = 0.0; c->P.step_nbr = vt->default_step_nbr;
        // interceptor: the auxiliary scalars are (stageMode, currentChart); a fresh object has (0, 1) (interceptor.cpp:62-64)
        if (model_id == SOCP_MODEL_INTERCEPTOR) c->P.sw1 = 1.0;
    } else if (model_id == SOCP_MODEL_GODDARD) {
        // goddard.cpp:23-40 defaults
        c->dim = 7; c->nparams = SOCP_GODDARD_NPARAMS;
        const double d[8] = {3.5, 7.0, 310.0, 500.0, 1.0, 1.0, 0.0, -1.0};
        std::memcpy(c->P.p, d, sizeof(d));
        c->P.sw0 = 0.0227; c->P.sw1 = 0.08; c->P.step_nbr = 10;
    } else if (model_id == SOCP_MODEL_COVID19) {
        // covid19.cpp:25-38 defaults; its ModelInt integrates with its own stepNbr = 1000
        c->dim = 4; c->nparams = SOCP_COVID_NPARAMS; c->nu = 1;
        const double d[8] = {4, 10, 5, 1, 0.1, 1, -10, 20};
        std::memcpy(c->P.p, d, sizeof(d));
        c->P.sw0 = c->P.sw1 = 0.0; c->P.step_nbr = 1000;
    } else {
        // doubleIntegrator.cpp:26-34 defaults
        c->dim = 6; c->nparams = SOCP_DINT_NPARAMS;
        c->P.p[0] = 1.0; c->P.p[1] = 1.0; c->P.p[2] = 0.01;
        c->P.sw0 = c->P.sw1 = 0.0; c->P.step_nbr = 30;
    }
    c->S = 2 * c->dim;
    // default arithmetic flavour can be chosen from the environment (host programs that do not call
    // socp_ctx_set_variant): SOCP_VARIANT=exact|fast
    if (const char *v = std::getenv("SOCP_VARIANT")) {
        if (std::strcmp(v, "fast") == 0) c->variant = SOCP_VARIANT_LANE_FAST;
        else if (std::strcmp(v, "exact") == 0) c->variant = SOCP_VARIANT_LANE_EXACT;
    }
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(nullptr, e, "hipStreamCreate"); }
    c->stream = c->own_stream;
    *out = c;
    return SOCP_OK;
}

int socp_ctx_clone(const socp_ctx *proto, int device, socp_ctx **out)
{
    if (!proto || !out) return fail(nullptr, SOCP_ERR_ARG, "socp_ctx_clone: null argument");
    *out = nullptr;
    socp_ctx *c = nullptr;
    int rc = socp_ctx_create(&c, proto->model_id, device);
    if (rc != SOCP_OK) return rc;
    c->P = proto->P;                                   // parameters, switching times, step number, integrator, tolerance
    c->variant = proto->variant;
    if (proto->has_problem)
        rc = socp_problem_set(c, proto->M, proto->mode_t.data(), proto->mode_x.data(), proto->time.data(), proto->xnode.data());
    if (rc != SOCP_OK) { g_create_error = c->err; socp_ctx_destroy(c); return rc; }
    *out = c;
    return SOCP_OK;
}

int socp_ctx_destroy(socp_ctx *c)
{
    if (!c) return SOCP_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->d_tables.release(); c->d_pairs_full.release(); c->d_pairs_dedup.release();
    c->s_t0.release(); c->s_tf.release(); c->s_sw.release(); c->s_in.release(); c->s_out.release(); c->s_aux.release(); c->s_var.release();
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return SOCP_OK;
}

int socp_ctx_set_params(socp_ctx *c, const double *params, int nparams)
{
    if (!c || !params) return fail(c, SOCP_ERR_ARG, "set_params: null argument");
    if (nparams != c->nparams) return fail(c, SOCP_ERR_ARG, "set_params: wrong parameter count for this model");
    std::memcpy(c->P.p, params, sizeof(double) * nparams);
    return SOCP_OK;
}

int socp_ctx_get_params(const socp_ctx *c, double *params, int nparams)
{
    if (!c || !params || nparams != c->nparams) return SOCP_ERR_ARG;
    std::memcpy(params, c->P.p, sizeof(double) * nparams);
    return SOCP_OK;
}

int socp_ctx_num_params(const socp_ctx *c) { return c ? c->nparams : SOCP_ERR_ARG; }

int socp_ctx_set_step_number(socp_ctx *c, int step_nbr)
{
    if (!c) return SOCP_ERR_ARG;
    if (step_nbr < 1) return fail(c, SOCP_ERR_ARG, "set_step_number: step number must be >= 1");
    c->P.step_nbr = step_nbr;
    return SOCP_OK;
}

int socp_ctx_set_integrator(socp_ctx *c, int kind, double tol)
{
    if (!c) return SOCP_ERR_ARG;
    if (kind != SOCP_INT_RK4 && kind != SOCP_INT_DOPRI5) return fail(c, SOCP_ERR_ARG, "set_integrator: unknown integrator");
    if (kind == SOCP_INT_DOPRI5 && !(tol > 0)) return fail(c, SOCP_ERR_ARG, "set_integrator: tolerance must be positive");
    c->P.integrator = kind;
    c->P.tol = tol;
    return SOCP_OK;
}

int socp_ctx_set_switching_times(socp_ctx *c, const double *sw, int nsw)
{
    if (!c || (nsw > 0 && !sw)) return fail(c, SOCP_ERR_ARG, "set_switching_times: null argument");
    // the control law reads entries [0] and [1] only (goddard.cpp:148-151); a missing entry is an
    // out-of-bounds read in the reference -- here it is NaN, so every comparison is false.
    c->P.sw0 = nsw > 0 ? sw[0] : NAN;
    c->P.sw1 = nsw > 1 ? sw[1] : NAN;
    return SOCP_OK;
}

int socp_ctx_get_switching_times(const socp_ctx *c, double *sw2)
{
    if (!c || !sw2) return SOCP_ERR_ARG;
    sw2[0] = c->P.sw0; sw2[1] = c->P.sw1;
    return SOCP_OK;
}

int socp_ctx_set_variant(socp_ctx *c, int variant)
{
    if (!c) return SOCP_ERR_ARG;
    if (variant < SOCP_VARIANT_AUTO || variant > SOCP_VARIANT_LANE_FAST) return fail(c, SOCP_ERR_ARG, "set_variant: unknown variant");
    c->variant = variant;
    return SOCP_OK;
}

int socp_ctx_set_stream(socp_ctx *c, void *hip_stream, int use_own)
{
    if (!c) return SOCP_ERR_ARG;
    // a null hipStream_t is a real stream (the device's default stream), so "own" is explicit
    c->stream = use_own ? c->own_stream : static_cast<hipStream_t>(hip_stream);
    return SOCP_OK;
}

int socp_ctx_get_stream(const socp_ctx *c, void **hip_stream)
{
    if (!c || !hip_stream) return SOCP_ERR_ARG;
    *hip_stream = static_cast<void *>(c->stream);
    return SOCP_OK;
}

int socp_ctx_aux_stream(socp_ctx *c, void **hip_stream)
{
    if (!c || !hip_stream) return SOCP_ERR_ARG;
    if (!c->aux_stream) {
        HIP_TRY(c, hipSetDevice(c->device));
        // Streams of one priority share a few hardware queues, and two streams on one queue run their kernels one after the other
        // (measured: the residual and Jacobian launches of a round took 0.90 s of launches instead of 0.64 s).  Streams of
        // another priority come from their own queue pool.
        int least = 0, greatest = 0;
        hipStream_t st = nullptr;
        if (!(hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
              hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest) == hipSuccess)) {
            (void)hipGetLastError();
            HIP_TRY(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        }
        c->aux_stream = st;
    }
    *hip_stream = static_cast<void *>(c->aux_stream);
    return SOCP_OK;
}

int socp_ctx_warm_up(socp_ctx *c)
{
    if (!c) return SOCP_ERR_ARG;
    void *aux = nullptr;
    const int rc = socp_ctx_aux_stream(c, &aux);
    if (rc != SOCP_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = 256 * 1024;
    void *h = nullptr, *d = nullptr;
    HIP_TRY(c, hipHostMalloc(&h, bytes, hipHostMallocDefault));
    hipError_t e = hipMalloc(&d, bytes);
    if (e == hipSuccess) {
        std::memset(h, 0, bytes);
        e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d);
    }
    (void)hipHostFree(h);
    if (e != hipSuccess) return hip_fail(c, e, "socp_ctx_warm_up");
    return SOCP_OK;
}

int socp_ctx_synchronize(socp_ctx *c)
{
    if (!c) return SOCP_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SOCP_OK;
}

int socp_ctx_dims(const socp_ctx *c, int *dim, int *state_len, int *state_len_jac)
{
    if (!c) return SOCP_ERR_ARG;
    if (dim) *dim = c->dim;
    if (state_len) *state_len = c->S;
    if (state_len_jac) *state_len_jac = (c->S + 1) * c->S;
    return SOCP_OK;
}

int socp_ctx_control_dim(const socp_ctx *c) { return c ? c->nu : SOCP_ERR_ARG; }
int socp_ctx_device(const socp_ctx *c) { return c ? c->device : SOCP_ERR_ARG; }
int socp_ctx_get_variant(const socp_ctx *c) { return c ? c->variant : SOCP_ERR_ARG; }
int socp_ctx_model_id(const socp_ctx *c) { return c ? c->model_id : SOCP_ERR_ARG; }
int socp_ctx_has_variational(const socp_ctx *c) { return c ? (has_var(c) ? 1 : 0) : SOCP_ERR_ARG; }

int socp_ctx_counters(const socp_ctx *c, long long *trajectories, long long *launches)
{
    if (!c) return SOCP_ERR_ARG;
    if (trajectories) *trajectories = c->n_traj;
    if (launches) *launches = c->n_launch;
    return SOCP_OK;
}

// batchsolve.cpp: what a clone of `c` integrated on its behalf (chain groups) counts as c's
void socp_ctx_add_counters(socp_ctx *c, long long trajectories, long long launches)
{
    if (c) { c->n_traj += trajectories; c->n_launch += launches; }
}

/* ---- trajectories ------------------------------------------------------------------------ */

int socp_integrate_batch_dev(socp_ctx *c, int B, const double *d_t0, const double *d_tf,
                             const double *d_sw, const double *d_X0, double *d_Xf, int is_jac)
{
    if (!c) return SOCP_ERR_ARG;
    if (B < 0 || (B > 0 && (!d_t0 || !d_tf || !d_X0 || !d_Xf))) return fail(c, SOCP_ERR_ARG, "integrate_batch: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (is_jac) {
        // variational state: one wavefront per trajectory (doubleIntegrator; goddard has modelOrder 0 only)
        if (!has_var(c))
            return fail(c, SOCP_ERR_UNSUPPORTED, "integrate_batch: this model has no variational equations (modelOrder 0)");
        if (d_Xf == d_X0) return fail(c, SOCP_ERR_ARG, "integrate_batch: is_jac=1 needs distinct input and output");
        c->n_traj += B; c->n_launch += 1;
        HIP_TRY(c, c->vt ? c->vt->var_traj(c->stream, c->P, B, d_t0, d_tf, d_X0, d_Xf)
                         : var_traj(c->model_id, c->stream, c->P, B, d_t0, d_tf, d_X0, d_Xf));
        return SOCP_OK;
    }
    HIP_TRY(c, run_traj(c, B, d_t0, d_tf, d_sw, d_X0, d_Xf));
    return SOCP_OK;
}

int socp_integrate_batch(socp_ctx *c, int B, const double *t0, const double *tf, const double *sw,
                         const double *X0, double *Xf, int is_jac)
{
    if (!c) return SOCP_ERR_ARG;
    if (B < 0 || (B > 0 && (!t0 || !tf || !X0 || !Xf))) return fail(c, SOCP_ERR_ARG, "integrate_batch: null argument");
    if (B == 0) return SOCP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t len = is_jac ? (size_t)(c->S + 1) * c->S : (size_t)c->S;
    const size_t nb = sizeof(double) * len * B;
    HIP_TRY(c, c->s_t0.reserve(sizeof(double) * B));
    HIP_TRY(c, c->s_tf.reserve(sizeof(double) * B));
    HIP_TRY(c, c->s_in.reserve(nb));
    HIP_TRY(c, c->s_out.reserve(nb));
    HIP_TRY(c, hipMemcpyAsync(c->s_t0.p, t0, sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->s_tf.p, tf, sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->s_in.p, X0, nb, hipMemcpyHostToDevice, c->stream));
    const double *dsw = nullptr;
    if (sw) {
        HIP_TRY(c, c->s_sw.reserve(sizeof(double) * 2 * B));
        HIP_TRY(c, hipMemcpyAsync(c->s_sw.p, sw, sizeof(double) * 2 * B, hipMemcpyHostToDevice, c->stream));
        dsw = c->s_sw.as<double>();
    }
    int rc = socp_integrate_batch_dev(c, B, c->s_t0.as<double>(), c->s_tf.as<double>(), dsw,
                                      c->s_in.as<double>(), c->s_out.as<double>(), is_jac);
    if (rc != SOCP_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(Xf, c->s_out.p, nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SOCP_OK;
}

int socp_integrate_dense(socp_ctx *c, double t0, double tf, const double *sw, const double *X0,
                         double *dense, double *times, int cap, int *rows)
{
    return socp_integrate_dense_aux(c, t0, tf, sw, X0, dense, times, nullptr, cap, rows);
}

int socp_integrate_dense_aux(socp_ctx *c, double t0, double tf, const double *sw, const double *X0,
                             double *dense, double *times, double *aux, int cap, int *rows)
{
    if (!c) return SOCP_ERR_ARG;
    if (!X0 || !dense || !times || !rows || cap < 1) return fail(c, SOCP_ERR_ARG, "integrate_dense: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t S = c->S;
    HIP_TRY(c, c->s_in.reserve(sizeof(double) * S));
    HIP_TRY(c, c->s_out.reserve(sizeof(double) * S * cap));
    HIP_TRY(c, c->s_t0.reserve(sizeof(double) * cap));
    HIP_TRY(c, c->s_aux.reserve(sizeof(int) * 4));
    HIP_TRY(c, c->s_tf.reserve(sizeof(double) * 2 * cap));
    double *d_aux = aux ? c->s_tf.as<double>() : nullptr;
    HIP_TRY(c, hipMemcpyAsync(c->s_in.p, X0, sizeof(double) * S, hipMemcpyHostToDevice, c->stream));
    const double s0 = sw ? sw[0] : c->P.sw0, s1 = sw ? sw[1] : c->P.sw1;
    c->n_traj += 1; c->n_launch += 1;
    hipError_t e = c->vt
        ? table_of(c)->dense(c->stream, c->P, t0, tf, s0, s1, c->s_in.as<double>(), c->s_out.as<double>(), c->s_t0.as<double>(), cap, c->s_aux.as<int>(), d_aux)
        : use_fast(c)
        ? dense_fast(c->model_id, c->stream, c->P, t0, tf, s0, s1, c->s_in.as<double>(), c->s_out.as<double>(), c->s_t0.as<double>(), cap, c->s_aux.as<int>(), d_aux)
        : dense_exact(c->model_id, c->stream, c->P, t0, tf, s0, s1, c->s_in.as<double>(), c->s_out.as<double>(), c->s_t0.as<double>(), cap, c->s_aux.as<int>(), d_aux);
    HIP_TRY(c, e);
    HIP_TRY(c, hipMemcpyAsync(rows, c->s_aux.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int kept = *rows < cap ? *rows : cap;
    HIP_TRY(c, hipMemcpy(dense, c->s_out.p, sizeof(double) * S * kept, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(times, c->s_t0.p, sizeof(double) * kept, hipMemcpyDeviceToHost));
    if (aux) HIP_TRY(c, hipMemcpy(aux, d_aux, sizeof(double) * 2 * kept, hipMemcpyDeviceToHost));
    return SOCP_OK;
}

int socp_eval_batch(socp_ctx *c, int what, int B, const double *t, const double *sw,
                    const double *X, int len, double *out, int is_jac)
{
    if (!c) return SOCP_ERR_ARG;
    if (B < 0 || (B > 0 && (!t || !X || !out))) return fail(c, SOCP_ERR_ARG, "eval_batch: null argument");
    if (what < SOCP_EVAL_RHS || what > SOCP_EVAL_HAMILTONIAN) return fail(c, SOCP_ERR_ARG, "eval_batch: unknown quantity");
    const int L = (c->S + 1) * c->S;
    const bool var = is_jac && what != SOCP_EVAL_CONTROL;
    if (var && !has_var(c))
        return fail(c, SOCP_ERR_UNSUPPORTED, "eval_batch: this model has no variational equations (modelOrder 0)");
    if (var && what == SOCP_EVAL_RHS && len != L) return fail(c, SOCP_ERR_ARG, "eval_batch: augmented state length must be (2*dim+1)*2*dim");
    if (!(var && what == SOCP_EVAL_RHS) && len < c->S) return fail(c, SOCP_ERR_ARG, "eval_batch: state length must be at least 2*dim");
    if (B == 0) return SOCP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const int out_len = var ? (what == SOCP_EVAL_RHS ? L : c->S + 1)
                            : (what == SOCP_EVAL_RHS ? c->S : (what == SOCP_EVAL_CONTROL ? c->nu : 1));
    if (!var && len != c->S) {
        // a longer (augmented) vector may be passed for Control / Hamiltonian: only the state part is read
        return fail(c, SOCP_ERR_ARG, "eval_batch: pass the 2*dim state part for is_jac=0 evaluations");
    }
    HIP_TRY(c, c->s_t0.reserve(sizeof(double) * B));
    HIP_TRY(c, c->s_in.reserve(sizeof(double) * (size_t)B * len));
    HIP_TRY(c, c->s_out.reserve(sizeof(double) * (size_t)B * out_len));
    HIP_TRY(c, hipMemcpyAsync(c->s_t0.p, t, sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->s_in.p, X, sizeof(double) * (size_t)B * len, hipMemcpyHostToDevice, c->stream));
    const double *dsw = nullptr;
    if (sw) {
        HIP_TRY(c, c->s_sw.reserve(sizeof(double) * 2 * B));
        HIP_TRY(c, hipMemcpyAsync(c->s_sw.p, sw, sizeof(double) * 2 * B, hipMemcpyHostToDevice, c->stream));
        dsw = c->s_sw.as<double>();
    }
    c->n_launch += 1;
    hipError_t e = var
        ? (c->vt ? c->vt->var_eval(c->stream, c->P, what == SOCP_EVAL_RHS ? 0 : 1, B, c->s_t0.as<double>(), c->s_in.as<double>(), len, c->s_out.as<double>())
                 : var_eval(c->model_id, c->stream, c->P, what == SOCP_EVAL_RHS ? 0 : 1, B, c->s_t0.as<double>(), c->s_in.as<double>(), len, c->s_out.as<double>()))
        : c->vt
        ? table_of(c)->eval(c->stream, c->P, what, B, c->s_t0.as<double>(), dsw, c->s_in.as<double>(), c->s_out.as<double>())
        : use_fast(c)
        ? eval_fast(c->model_id, c->stream, c->P, what, B, c->s_t0.as<double>(), dsw, c->s_in.as<double>(), c->s_out.as<double>())
        : eval_exact(c->model_id, c->stream, c->P, what, B, c->s_t0.as<double>(), dsw, c->s_in.as<double>(), c->s_out.as<double>());
    HIP_TRY(c, e);
    HIP_TRY(c, hipMemcpyAsync(out, c->s_out.p, sizeof(double) * (size_t)B * out_len, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SOCP_OK;
}

/* ---- shooting problem --------------------------------------------------------------------- */

int socp_problem_set(socp_ctx *c, int M, const int *mode_t, const int *mode_x, const double *time, const double *xnode)
{
    if (!c) return SOCP_ERR_ARG;
    if (M < 1 || !mode_t || !mode_x || !time || !xnode) return fail(c, SOCP_ERR_ARG, "problem_set: bad argument");
    if (M + 1 > kMaxNodes) return fail(c, SOCP_ERR_ARG, "problem_set: too many shooting nodes");
    const int d = c->dim, S = c->S;
    for (int j = 0; j <= M; j++)
        if (mode_t[j] < SOCP_FIXED || mode_t[j] > SOCP_CONTINUOUS) return fail(c, SOCP_ERR_ARG, "problem_set: bad time mode");
    // a CONTINUOUS end node would leave the tail of the timeline unset in the reference
    // (shooting.cpp:1586-1613 only closes an interval at a FIXED/FREE node)
    if (mode_t[0] == SOCP_CONTINUOUS || mode_t[M] == SOCP_CONTINUOUS)
        return fail(c, SOCP_ERR_ARG, "problem_set: first and last time must be FIXED or FREE");
    for (int k = 0; k <= M; k++)
        for (int j = 0; j < d; j++) {
            const int m = mode_x[k * d + j];
            if (m < SOCP_FIXED || m > SOCP_CONTINUOUS) return fail(c, SOCP_ERR_ARG, "problem_set: bad state mode");
            // (interior FREE states go to the model's SwitchingStateFunction -- shooting.cpp:1535-1538, model.hpp:339-341 -- i.e. to
            // the optional device trait switching_state; a model without it gets the default hook's zero rows)
            if ((k == 0 || k == M) && m == SOCP_CONTINUOUS)
                return fail(c, SOCP_ERR_ARG, "problem_set: CONTINUOUS state mode at a boundary node");
        }

    c->M = M;
    c->mode_t.assign(mode_t, mode_t + M + 1);
    c->mode_x.assign(mode_x, mode_x + (size_t)(M + 1) * d);
    c->time.assign(time, time + M + 1);
    c->xnode.assign(xnode, xnode + (size_t)(M + 1) * S);
    c->node_kind.assign(M + 1, -2); c->lo.assign(M + 1, 0); c->hi.assign(M + 1, 0); c->ft_row.assign(M + 1, -1);

    // unknown / residual layout (shooting.cpp:228-243, 945-990)
    int nbr = S * M;
    c->sw_node0 = c->sw_node1 = -1;
    for (int j = 0; j <= M; j++) {
        if (mode_t[j] == SOCP_FIXED) c->node_kind[j] = -1;
        if (mode_t[j] == SOCP_FREE) {
            c->node_kind[j] = nbr;           // index of the time unknown in z ...
            c->ft_row[j] = nbr;              // ... and of its residual row in F
            nbr++;
            if (j < M) { if (c->sw_node0 < 0) c->sw_node0 = j; else if (c->sw_node1 < 0) c->sw_node1 = j; }
        }
    }
    c->n = nbr;
    int cur = 0;
    for (int j = 0; j <= M; j++) {
        if (c->node_kind[j] != -2) {
            for (int k = cur + 1; k < j; k++) { c->lo[k] = cur; c->hi[k] = j; }
            c->lo[j] = c->hi[j] = j;
            cur = j;
        }
    }

    // FD column -> segments to integrate
    std::vector<int2> full, dedup;
    for (int j = 0; j < c->n; j++)
        for (int i = 0; i < M; i++) full.push_back(make_int2(j, i));
    for (int j = 0; j < c->n; j++) {
        if (j < S * M) {
            const int k = j / S;
            if (k >= 1) dedup.push_back(make_int2(j, k - 1));
            dedup.push_back(make_int2(j, k));
        } else {
            int q = 0;
            for (int k = 0; k <= M; k++) if (c->node_kind[k] == j) q = k;
            if (q == c->sw_node0 || q == c->sw_node1) {
                for (int i = 0; i < M; i++) dedup.push_back(make_int2(j, i));   // control law reads it everywhere
            } else {
                int a = q, b = q;
                while (a > 0 && c->node_kind[a - 1] == -2) a--;
                if (a > 0) a--;                                // previous junction
                while (b < M && c->node_kind[b + 1] == -2) b++;
                if (b < M) b++;                                // next junction
                for (int i = a; i < b; i++) dedup.push_back(make_int2(j, i));
            }
        }
    }
    c->T_full = (int)full.size();
    c->T_dedup = (int)dedup.size();

    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));     // tables may still be in use by enqueued work
    const size_t nI = (size_t)(M + 1), off_kind = 0, off_lo = nI, off_hi = 2 * nI, off_ft = 3 * nI, off_mx = 4 * nI;
    const size_t n_int = 4 * nI + nI * d;
    const size_t int_bytes = ((n_int * sizeof(int) + 15) / 16) * 16;
    const size_t n_dbl = nI + nI * S;
    HIP_TRY(c, c->d_tables.reserve(int_bytes + n_dbl * sizeof(double)));
    std::vector<char> blob(int_bytes + n_dbl * sizeof(double));
    int *bi = reinterpret_cast<int *>(blob.data());
    double *bd = reinterpret_cast<double *>(blob.data() + int_bytes);
    std::memcpy(bi + off_kind, c->node_kind.data(), nI * sizeof(int));
    std::memcpy(bi + off_lo, c->lo.data(), nI * sizeof(int));
    std::memcpy(bi + off_hi, c->hi.data(), nI * sizeof(int));
    std::memcpy(bi + off_ft, c->ft_row.data(), nI * sizeof(int));
    std::memcpy(bi + off_mx, c->mode_x.data(), nI * d * sizeof(int));
    std::memcpy(bd, c->time.data(), nI * sizeof(double));
    std::memcpy(bd + nI, c->xnode.data(), nI * S * sizeof(double));
    HIP_TRY(c, hipMemcpy(c->d_tables.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, c->d_pairs_full.reserve(sizeof(int2) * full.size()));
    HIP_TRY(c, c->d_pairs_dedup.reserve(sizeof(int2) * dedup.size()));
    HIP_TRY(c, hipMemcpy(c->d_pairs_full.p, full.data(), sizeof(int2) * full.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_pairs_dedup.p, dedup.data(), sizeof(int2) * dedup.size(), hipMemcpyHostToDevice));

    const int *di = c->d_tables.as<int>();
    const double *dd = reinterpret_cast<const double *>(c->d_tables.as<char>() + int_bytes);
    c->pb.dim = d; c->pb.M = M; c->pb.n = c->n;
    c->pb.sw_node0 = c->sw_node0; c->pb.sw_node1 = c->sw_node1;
    c->pb.node_kind = di + off_kind; c->pb.lo = di + off_lo; c->pb.hi = di + off_hi;
    c->pb.ft_row = di + off_ft; c->pb.mode_x = di + off_mx;
    c->pb.time = dd; c->pb.xnode = dd + nI;
    c->pb.pp_params = c->pb.pp_time = c->pb.pp_xnode = nullptr; c->pb.pp_stride = 0; c->pb.pp_smooth = 0;   // a new problem starts without per-problem blocks
    c->blocks_smooth_hint = false;
    c->has_problem = true;
    return SOCP_OK;
}

int socp_problem_set_blocks_dev(socp_ctx *c, const double *d_params, int stride, const double *d_time, const double *d_xnode)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "problem_set_blocks: no problem set");
    if (d_params && stride != c->nparams + 2)
        return fail(c, SOCP_ERR_ARG, "problem_set_blocks: stride must be nparams + 2 (parameters, then two switching times)");
    if (d_params && c->nparams + 2 > kMaxParams + 2) return fail(c, SOCP_ERR_ARG, "problem_set_blocks: too many parameters");
    c->pb.pp_params = d_params;
    c->pb.pp_stride = d_params ? stride : 0;
    c->pb.pp_time = d_time;
    c->pb.pp_xnode = d_xnode;
    c->pb.pp_smooth = (d_params && c->blocks_smooth_hint) ? 1 : 0;
    return SOCP_OK;
}

int socp_problem_blocks_all_smooth(socp_ctx *c, int all_smooth)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "problem_blocks_all_smooth: no problem set");
    c->blocks_smooth_hint = all_smooth != 0;
    c->pb.pp_smooth = (c->pb.pp_params && c->blocks_smooth_hint) ? 1 : 0;
    return SOCP_OK;
}

int socp_residual_batch_blocks(socp_ctx *c, int B, const double *Z, const double *params, int stride, const double *time,
                               const double *xnode, double *F)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "residual_batch_blocks: no problem set");
    if (B < 0 || (B > 0 && (!Z || !F))) return fail(c, SOCP_ERR_ARG, "residual_batch_blocks: null argument");
    // before anything is sized or copied from it: a wrong stride would read past the caller's array
    if (params && stride != c->nparams + 2)
        return fail(c, SOCP_ERR_ARG, "residual_batch_blocks: stride must be nparams + 2 (parameters, then two switching times)");
    if (B == 0) return SOCP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nodes = (size_t)c->M + 1;
    const size_t nbP = params ? sizeof(double) * (size_t)B * stride : 0, nbT = time ? sizeof(double) * B * nodes : 0,
                 nbX = xnode ? sizeof(double) * B * nodes * c->S : 0;
    HIP_TRY(c, c->s_aux.reserve(nbP + nbT + nbX + 64));
    char *base = c->s_aux.as<char>();
    double *dP = reinterpret_cast<double *>(base), *dT = reinterpret_cast<double *>(base + nbP), *dX = reinterpret_cast<double *>(base + nbP + nbT);
    if (params) HIP_TRY(c, hipMemcpyAsync(dP, params, nbP, hipMemcpyHostToDevice, c->stream));
    if (time) HIP_TRY(c, hipMemcpyAsync(dT, time, nbT, hipMemcpyHostToDevice, c->stream));
    if (xnode) HIP_TRY(c, hipMemcpyAsync(dX, xnode, nbX, hipMemcpyHostToDevice, c->stream));
    const ProblemDev saved = c->pb;
    int rc = socp_problem_set_blocks_dev(c, params ? dP : nullptr, stride, time ? dT : nullptr, xnode ? dX : nullptr);
    if (rc == SOCP_OK) rc = socp_residual_batch(c, B, Z, F);
    c->pb = saved;
    return rc;
}

int socp_problem_num_nodes(const socp_ctx *c) { return (c && c->has_problem) ? c->M + 1 : SOCP_ERR_ARG; }
int socp_problem_num_param(const socp_ctx *c) { return (c && c->has_problem) ? c->n : SOCP_ERR_ARG; }

int socp_timeline(socp_ctx *c, const double *z, double *tl)
{
    // Pure index/interpolation logic on host data (no RHS arithmetic): shooting.cpp:1579-1617.
    if (!c || !z || !tl) return fail(c, SOCP_ERR_ARG, "timeline: null argument");
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "timeline: no problem set");
    auto jt = [&](int j) { return c->node_kind[j] >= 0 ? z[c->node_kind[j]] : c->time[j]; };
    for (int k = 0; k <= c->M; k++) {
        if (c->node_kind[k] >= -1) tl[k] = jt(k);
        else {
            const int a = c->lo[k], b = c->hi[k];
            const double ta = jt(a), tb = jt(b);
            tl[k] = ta + (k - a) * (tb - ta) / (b - a);
        }
    }
    return SOCP_OK;
}

int socp_residual_batch_dev(socp_ctx *c, int B, const double *d_Z, double *d_F)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "residual_batch: no problem set");
    if (B < 0 || (B > 0 && (!d_Z || !d_F))) return fail(c, SOCP_ERR_ARG, "residual_batch: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, run_residual(c, B, d_Z, d_F));
    return SOCP_OK;
}

int socp_residual_batch(socp_ctx *c, int B, const double *Z, double *F)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "residual_batch: no problem set");
    if (B < 0 || (B > 0 && (!Z || !F))) return fail(c, SOCP_ERR_ARG, "residual_batch: null argument");
    if (B == 0) return SOCP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nb = sizeof(double) * (size_t)B * c->n;
    HIP_TRY(c, c->s_in.reserve(nb));
    HIP_TRY(c, c->s_out.reserve(nb));
    HIP_TRY(c, hipMemcpyAsync(c->s_in.p, Z, nb, hipMemcpyHostToDevice, c->stream));
    int rc = socp_residual_batch_dev(c, B, c->s_in.as<double>(), c->s_out.as<double>());
    if (rc != SOCP_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(F, c->s_out.p, nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SOCP_OK;
}

int socp_fd_jacobian_multi_dev(socp_ctx *c, int np, const double *d_Z, const double *d_Fvec, double epsfcn,
                               double *d_Fjac, int dedup)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "fd_jacobian: no problem set");
    if (np < 0 || (np > 0 && (!d_Z || !d_Fvec || !d_Fjac))) return fail(c, SOCP_ERR_ARG, "fd_jacobian: null argument");
    if (np == 0) return SOCP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const double eps = fd_eps(epsfcn);
    if (dedup) {
        // rows a column cannot change reproduce fvec bit for bit => exact zeros
        HIP_TRY(c, hipMemsetAsync(d_Fjac, 0, sizeof(double) * (size_t)np * c->n * c->n, c->stream));
        HIP_TRY(c, run_fdjac(c, np, c->T_dedup, c->d_pairs_dedup.as<int2>(), d_Z, d_Fvec, eps, d_Fjac));
    } else {
        HIP_TRY(c, run_fdjac(c, np, c->T_full, c->d_pairs_full.as<int2>(), d_Z, d_Fvec, eps, d_Fjac));
    }
    return SOCP_OK;
}

int socp_fd_jacobian_dev(socp_ctx *c, const double *d_z, const double *d_fvec, double epsfcn, double *d_fjac, int dedup)
{
    return socp_fd_jacobian_multi_dev(c, 1, d_z, d_fvec, epsfcn, d_fjac, dedup);
}

int socp_fd_rows_dev(socp_ctx *c, int np, const double *d_Z, double epsfcn, double *d_Rows)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "fd_rows: no problem set");
    if (np < 0 || (np > 0 && (!d_Z || !d_Rows))) return fail(c, SOCP_ERR_ARG, "fd_rows: null argument");
    if (np == 0) return SOCP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, run_fdrows(c, np, d_Z, fd_eps(epsfcn), d_Rows));
    return SOCP_OK;
}

int socp_fd_diff_dev(socp_ctx *c, int np, const double *d_Z, double epsfcn, const double *d_Rows, double *d_Fjac)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "fd_diff: no problem set");
    if (np < 0 || (np > 0 && (!d_Z || !d_Rows || !d_Fjac))) return fail(c, SOCP_ERR_ARG, "fd_diff: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    c->n_launch += 1;
    HIP_TRY(c, fd_diff(c->stream, c->n, np, d_Z, fd_eps(epsfcn), d_Rows, d_Fjac));
    return SOCP_OK;
}

int socp_fd_rows(socp_ctx *c, int np, const double *Z, double epsfcn, double *Rows)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "fd_rows: no problem set");
    if (np < 0 || (np > 0 && (!Z || !Rows))) return fail(c, SOCP_ERR_ARG, "fd_rows: null argument");
    if (np == 0) return SOCP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->n;
    HIP_TRY(c, c->s_in.reserve(sizeof(double) * n * np));
    HIP_TRY(c, c->s_out.reserve(sizeof(double) * n * (n + 1) * np));
    HIP_TRY(c, hipMemcpyAsync(c->s_in.p, Z, sizeof(double) * n * np, hipMemcpyHostToDevice, c->stream));
    int rc = socp_fd_rows_dev(c, np, c->s_in.as<double>(), epsfcn, c->s_out.as<double>());
    if (rc != SOCP_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(Rows, c->s_out.p, sizeof(double) * n * (n + 1) * np, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SOCP_OK;
}

int socp_fd_jacobian(socp_ctx *c, const double *z, const double *fvec, double epsfcn, double *fjac, int dedup)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "fd_jacobian: no problem set");
    if (!z || !fvec || !fjac) return fail(c, SOCP_ERR_ARG, "fd_jacobian: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->n;
    HIP_TRY(c, c->s_in.reserve(sizeof(double) * n));
    HIP_TRY(c, c->s_aux.reserve(sizeof(double) * n));
    HIP_TRY(c, c->s_out.reserve(sizeof(double) * n * n));
    HIP_TRY(c, hipMemcpyAsync(c->s_in.p, z, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->s_aux.p, fvec, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    int rc = socp_fd_jacobian_dev(c, c->s_in.as<double>(), c->s_aux.as<double>(), epsfcn, c->s_out.as<double>(), dedup);
    if (rc != SOCP_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(fjac, c->s_out.p, sizeof(double) * n * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SOCP_OK;
}

int socp_register_model(int model_id, const void *table, int table_bytes)
{
    if (model_id < SOCP_PLUGIN_ID_MIN) return fail(nullptr, SOCP_ERR_ARG, "register_model: ids below 100 are reserved for in-tree models");
    if (!table || table_bytes != (int)sizeof(ModelLaunchers)) return fail(nullptr, SOCP_ERR_ARG, "register_model: launch table size mismatch (plugin built against other headers)");
    const ModelLaunchers *t = static_cast<const ModelLaunchers *>(table);
    if (t->abi != kPluginAbi || t->dim < 1 || 2 * t->dim > 64 || t->nparams < 0 || t->nparams > kMaxParams || t->default_step_nbr < 1 ||
        !t->traj || !t->residual || !t->fdjac || !t->fdrows || !t->dense || !t->eval)
        return fail(nullptr, SOCP_ERR_ARG, "register_model: malformed launch table");
    std::lock_guard<std::mutex> guard(plugins_lock());
    plugins()[model_id] = *t;
    return SOCP_OK;
}

int socp_plugin_load(const char *path)
{
    if (!path) return fail(nullptr, SOCP_ERR_ARG, "plugin_load: null path");
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(nullptr, SOCP_ERR_ARG, std::string("plugin_load: ") + dlerror());
    typedef int (*reg_fn)(void);
    reg_fn reg = reinterpret_cast<reg_fn>(dlsym(h, "socp_plugin_register"));
    if (!reg) { dlclose(h); return fail(nullptr, SOCP_ERR_ARG, "plugin_load: socp_plugin_register not exported"); }
    return reg();        // the handle stays open: the kernels live in it
}

int socp_var_jacobian_multi_dev(socp_ctx *c, int np, const double *d_Z, double *d_Fjac)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "var_jacobian: no problem set");
    if (np < 0 || (np > 0 && (!d_Z || !d_Fjac))) return fail(c, SOCP_ERR_ARG, "var_jacobian: null argument");
    if (!has_var(c))
        return fail(c, SOCP_ERR_UNSUPPORTED, "var_jacobian: this model has no variational equations (modelOrder 0)");
    if (np == 0) return SOCP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t M = c->M, L = (size_t)(c->S + 1) * c->S, B = (size_t)np * M;
    HIP_TRY(c, c->s_var.reserve(sizeof(double) * (2 * B * L + 2 * B)));
    double *Xaug = c->s_var.as<double>(), *Xtf = Xaug + B * L, *t0 = Xtf + B * L, *tf = t0 + B;
    c->n_traj += (long long)B; c->n_launch += 3;
    HIP_TRY(c, c->vt ? c->vt->var_jacobian(c->stream, c->P, c->pb, np, d_Z, Xaug, Xtf, t0, tf, d_Fjac)
                     : var_jacobian(c->model_id, c->stream, c->P, c->pb, np, d_Z, Xaug, Xtf, t0, tf, d_Fjac));
    return SOCP_OK;
}

int socp_var_jacobian(socp_ctx *c, const double *z, double *fjac)
{
    if (!c) return SOCP_ERR_ARG;
    if (!c->has_problem) return fail(c, SOCP_ERR_ARG, "var_jacobian: no problem set");
    if (!z || !fjac) return fail(c, SOCP_ERR_ARG, "var_jacobian: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->n;
    HIP_TRY(c, c->s_in.reserve(sizeof(double) * n));
    HIP_TRY(c, c->s_out.reserve(sizeof(double) * n * n));
    HIP_TRY(c, hipMemcpyAsync(c->s_in.p, z, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    if (int rc = socp_var_jacobian_multi_dev(c, 1, c->s_in.as<double>(), c->s_out.as<double>())) return rc;
    HIP_TRY(c, hipMemcpyAsync(fjac, c->s_out.p, sizeof(double) * n * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SOCP_OK;
}

}  // extern "C"
