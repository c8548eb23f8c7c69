// kernels_exact.hip -- reference-operation-order kernels.  MUST be compiled with
// -ffp-contract=off (see socp_amd/csrc/Makefile): the parity tests compare these against the
// CPU oracle at the few-ulp level.
#include "models_exact.hpp"
#define SOCP_FLAVOUR exact
#define SOCP_HAVE_DOPRI5 1      // adaptive Dormand-Prince instantiations live here (reference-order RHS)
#define SOCP_DEFINE_COMMON 1   // fd_diff lives in the no-contraction TU
#define SOCP_GODDARD GoddardExact
#define SOCP_GODDARD_SMOOTH GoddardExactSmooth
#define SOCP_COVID CovidExact
#define SOCP_DINT DIntExact
#include "launch_impl.hpp"

// ---- variational path (compiled here: no contraction) -------------------------------------------
#include "variational.hpp"

namespace socp {

hipError_t var_traj(int model_id, hipStream_t st, const ModelParams &P, int B, const double *t0, const double *tf,
                    const double *X0, double *Xf)
{
    if (model_id != 2) return hipErrorInvalidValue;
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(traj_var_wave_kernel<DIntVar>, dim3(B), dim3(64), 0, st, P, t0, tf, X0, Xf, (const double *)nullptr, 0, 1);
    return hipGetLastError();
}

hipError_t var_jacobian(int model_id, hipStream_t st, const ModelParams &P, const ProblemDev &pb, int np, const double *z,
                        double *Xaug, double *Xtf, double *t0, double *tf, double *fjac)
{
    if (model_id != 2) return hipErrorInvalidValue;
    if (np <= 0) return hipSuccess;
    const unsigned B = (unsigned)((long)np * pb.M);                 // one wavefront per (problem, segment)
    hipLaunchKernelGGL(var_prepare_kernel<DIntVar>, dim3(B), dim3(64), 0, st, pb, z, Xaug, t0, tf);
    hipLaunchKernelGGL(traj_var_wave_kernel<DIntVar>, dim3(B), dim3(64), 0, st, P, t0, tf, Xaug, Xtf, pb.pp_params, pb.pp_stride, pb.M);
    hipError_t e = hipMemsetAsync(fjac, 0, sizeof(double) * (size_t)np * pb.n * pb.n, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(var_assemble_kernel<DIntVar>, dim3((B + 63) / 64), dim3(64), 0, st, P, pb, np, z, Xtf, fjac);
    return hipGetLastError();
}

hipError_t var_eval(int model_id, hipStream_t st, const ModelParams &P, int what, int B, const double *X, int len,
                    double *out)
{
    if (model_id != 2) return hipErrorInvalidValue;
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(var_eval_kernel<DIntVar>, dim3((B + 63) / 64), dim3(64), 0, st, P, what, B, X, len, out);
    return hipGetLastError();
}

}  // namespace socp
