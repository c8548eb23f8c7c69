// launch.hpp -- host-callable launchers, one set per arithmetic flavour (translation unit).
#pragma once
#include "dev_common.hpp"

namespace socp {

#define SOCP_DECLARE_LAUNCHERS(FLAVOUR)                                                              \
    hipError_t traj_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P, int B,             \
                              const double *t0, const double *tf, const double *sw,                  \
                              const double *X0, double *Xf);                                         \
    hipError_t residual_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P,                \
                                  const ProblemDev &pb, int B, const double *Z, double *F);          \
    hipError_t fdjac_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P,                   \
                               const ProblemDev &pb, int T, const int2 *pairs, const double *z,      \
                               const double *fvec, double eps, double *fjac);                        \
    hipError_t eval_##FLAVOUR(int model_id, hipStream_t st, const ModelParams &P, int what, int B,   \
                              const double *t, const double *sw, const double *X, double *out);

SOCP_DECLARE_LAUNCHERS(exact)
SOCP_DECLARE_LAUNCHERS(fast)

}  // namespace socp
