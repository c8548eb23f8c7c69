// launch.hpp -- host-callable launchers, one set per arithmetic flavour (translation unit).
#pragma once
#include <cstdlib>

#include "dev_common.hpp"

namespace socp {

// ---- launch geometry shared by the in-tree flavours (launch_impl.hpp) and table-driven models (plugin_impl.hpp) ----
constexpr int kNumSIMD = 1024;                       // MI355X: 256 CUs x 4 SIMDs

inline unsigned blocks_for(long n) { return (unsigned)((n + 63) / 64); }

// Fill-the-chip placement.  Workgroups here are single waves that keep their trajectory in registers for
// milliseconds.  The dispatcher packs them as deep as registers allow (3 per SIMD at <= 168 VGPRs) and a CU
// does not balance single-wave workgroups over its 4 SIMDs, so a grid of W < 3072 waves would time-slice
// three waves on some SIMDs while others idle (measured: 960 waves took 2x, 1920 waves 3x the single-wave
// time).  Every hot kernel is therefore instantiated with an occupancy cap WPE in {1,2,3}
// (amdgpu_waves_per_eu) and the launcher picks WPE = ceil(W / 1024 SIMDs): up to 1024 waves run one per
// SIMD, up to 2048 two per SIMD, beyond that three.
inline int wpe_for(long waves)
{
    const long k = (waves + kNumSIMD - 1) / kNumSIMD;
    return k < 1 ? 1 : (k > 3 ? 3 : (int)k);
}

// rows per workgroup of the row-owned-tile kernels (integrator.hpp): whole rows, M lanes each, tile <= 32 KiB;
// 0 = direct stores
inline int rows_per_block(int M, int n)
{
    static const bool off = [] { const char *e = std::getenv("SOCP_ROW_TILES"); return e && e[0] == '0'; }();
    if (off || M > 64) return 0;                      // SOCP_ROW_TILES=0: direct stores (A/B measurements)
    int R = 64 / M;
    const long bytes = (long)R * n * 8;
    if (bytes > 32 * 1024) R = (int)(32 * 1024 / ((long)n * 8));
    return R < 1 ? 0 : R;
}

#define SOCP_DECLARE_LAUNCHERS(FLAVOUR)                                                              \
    hipError_t traj_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P, int B,             \
                              const double *t0, const double *tf, const double *sw,                  \
                              const double *X0, double *Xf);                                         \
    hipError_t residual_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P,                \
                                  const ProblemDev &pb, int B, const double *Z, double *F);          \
    hipError_t fdjac_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P,                   \
                               const ProblemDev &pb, int np, int T, const int2 *pairs,               \
                               const double *z, const double *fvec, double eps, double *fjac);       \
    hipError_t fdrows_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P,                  \
                                const ProblemDev &pb, int np, const double *z, double eps,           \
                                double *rows);                                                       \
    hipError_t dense_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P, double t0,         \
                               double tf, double sw0, double sw1, const double *X0, double *dense,   \
                               double *times, int cap, int *rows, double *aux);                      \
    hipError_t eval_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P, int what, int B,   \
                              const double *t, const double *sw, const double *X, double *out);

// Launch table of an out-of-tree model (include/socp_plugin.h, plugin_impl.hpp): what the C-ABI layer calls
// instead of the built-in flavour launchers when a context is created with a registered model id.
constexpr int kPluginAbi = 4;      // 3: ProblemDev carries per-problem blocks; 4: optional variational launchers
struct ModelLaunchers {
    int abi, dim, control_dim, nparams, default_step_nbr;
    double default_params[kMaxParams];
    hipError_t (*traj)(hipStream_t, const ModelParams &, int, const double *, const double *, const double *, const double *, double *);
    hipError_t (*residual)(hipStream_t, const ModelParams &, const ProblemDev &, int, const double *, double *);
    hipError_t (*fdjac)(hipStream_t, const ModelParams &, const ProblemDev &, int, int, const int2 *, const double *, const double *, double, double *);
    hipError_t (*fdrows)(hipStream_t, const ModelParams &, const ProblemDev &, int, const double *, double, double *);
    hipError_t (*dense)(hipStream_t, const ModelParams &, double, double, double, double, const double *, double *, double *, int, int *, double *);
    hipError_t (*eval)(hipStream_t, const ModelParams &, int, int, const double *, const double *, const double *, double *);
    // variational equations (modelOrder 1, hybrj path): all three null when the model has none
    hipError_t (*var_traj)(hipStream_t, const ModelParams &, int, const double *, const double *, const double *, double *);
    hipError_t (*var_jacobian)(hipStream_t, const ModelParams &, const ProblemDev &, int, const double *, double *, double *, double *, double *, double *);
    hipError_t (*var_eval)(hipStream_t, const ModelParams &, int, int, const double *, const double *, int, double *);
};

SOCP_DECLARE_LAUNCHERS(exact)
// flavour-independent: Jacobian from the rows of fdrows (differences and one division per entry)
// variational (hybrj) path: double integrator only; reference operation order
hipError_t var_traj(int model_id, hipStream_t st, const ModelParams &P, int B, const double *t0, const double *tf,
                    const double *X0, double *Xf);
hipError_t var_jacobian(int model_id, hipStream_t st, const ModelParams &P, const ProblemDev &pb, int np, const double *z,
                        double *Xaug, double *Xtf, double *t0, double *tf, double *fjac);
hipError_t var_eval(int model_id, hipStream_t st, const ModelParams &P, int what, int B, const double *t, const double *X, int len,
                    double *out);
hipError_t fd_diff(hipStream_t st, int n, int np, const double *z, double eps, const double *rows, double *fjac);

SOCP_DECLARE_LAUNCHERS(fast)

// in-tree models that live in their own translation unit behind a launch table, like an out-of-tree plugin
const ModelLaunchers *interceptor_launchers();      // kernels_interceptor.hip      (reference operation order)
const ModelLaunchers *interceptor_launchers_fast(); // kernels_interceptor_fast.hip (restructured, contraction on)

}  // namespace socp
