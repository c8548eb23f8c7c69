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
#include "models_variational.hpp"

namespace socp {

hipError_t var_traj(int model_id, hipStream_t st, const ModelParams &P, int B, const double *t0, const double *tf,
                    const double *X0, double *Xf)
{
    if (model_id != 2) return hipErrorInvalidValue;      // in-tree: the double integrator; table-driven models bring their own (launch.hpp)
    return varimpl::traj<DIntVar>(st, P, B, t0, tf, X0, Xf);
}

hipError_t var_jacobian(int model_id, hipStream_t st, const ModelParams &P, const ProblemDev &pb, int np, const double *z,
                        double *Xaug, double *Xtf, double *t0, double *tf, double *fjac)
{
    if (model_id != 2) return hipErrorInvalidValue;
    return varimpl::jacobian<DIntVar>(st, P, pb, np, z, Xaug, Xtf, t0, tf, fjac);
}

hipError_t var_eval(int model_id, hipStream_t st, const ModelParams &P, int what, int B, const double *t, const double *X, int len,
                    double *out)
{
    if (model_id != 2) return hipErrorInvalidValue;
    return varimpl::eval<DIntVar>(st, P, what, B, t, X, len, out);
}

}  // namespace socp
