// solver_launch.hpp -- host-callable launchers of the device Powell iteration (kernels_solver.hip) for the lock-step engine
// (batchsolve_dev.cpp).  A pool = P problems of one size n: states[P], workspaces[P][ws_stride] in HBM.
#pragma once
#include <hip/hip_runtime.h>

#include "solver_dev.hpp"

namespace socp {
namespace devsolver {

struct PoolDev {
    Config cfg;
    State *states = nullptr;       // [P]
    double *ws = nullptr;          // [P][ws_stride]
    long ws_stride = 0;
    int P = 0;
};

int threads_for(int n);            // workgroup size of the per-problem kernels: n + 1 columns rounded up to whole waves, <= 1024

// problems list[0 .. count): (re)start from X0[k][n] (k = position in the list)
// the Status of problems list[0 .. count) -> d_out[0 .. count), in list order
hipError_t launch_gather_status(hipStream_t st, const PoolDev &pool, const int *d_list, int count, Status *d_out);
hipError_t launch_start(hipStream_t st, const PoolDev &pool, const int *d_list, int count, const double *d_X0);
// advance the state machines of list[0 .. count); d_flags[k] (may be null = 0): what the pending evaluation returned.
// factor_phase: every problem of the list has just received a Jacobian (its advance is a factorisation)
hipError_t launch_advance(hipStream_t st, const PoolDev &pool, const int *d_list, int count, const int *d_flags, bool factor_phase);
// the Jacobian refresh of list[0 .. count) through the LDS-blocked factor kernel (one wavefront per problem); the following
// launch_advance(..., factor_phase = true) then finds the factors in place.  blocked_factor_applies: whether to use it for size n
bool blocked_factor_applies(int n);
hipError_t read_profile(unsigned long long out[16], bool reset);   // -DSOCP_SOLVER_PROFILE builds: per-phase clock totals
hipError_t launch_factor(hipStream_t st, const PoolDev &pool, const int *d_list, int count);
// the same refresh in the THROUGHPUT flavour (kernels_factor_fast.hip): blocked Householder QR, compact-WY panels of 16, trailing
// updates on the FP64 matrix cores; results equal the order-preserving kernels' to rounding, not bit for bit.  A workgroup of four
// wavefronts per problem; fast_factor_applies: the sizes it is built for (39 <= n <= 256)
bool fast_factor_applies(int n);
hipError_t launch_factor_fast(hipStream_t st, const PoolDev &pool, const int *d_list, int count);
hipError_t read_factor_profile(unsigned long long out[16], bool reset);   // -DSOCP_FACTOR_PROFILE builds: per-phase clock totals of the fast kernel
// the order-preserving factor work alone (solver_dev.hpp: factor), a thread per column: what socp_qr_factor_batch times beside it
hipError_t launch_factor_exact(hipStream_t st, const PoolDev &pool, const int *d_list, int count);
// dst[k][n] = the point problem list[k] asked to be evaluated (x, or the trial point)
hipError_t launch_gather_eval(hipStream_t st, const PoolDev &pool, const int *d_list, int count, double *d_dst);
// the residual of that evaluation back into the problem (fvec, or the trial residual)
// (src_stride: doubles between consecutive problems' results; 0 = n.  (n + 1) n: F is row 0 of a forward-difference batch)
hipError_t launch_scatter_fvec(hipStream_t st, const PoolDev &pool, const int *d_list, int count, const double *d_src, long src_stride = 0);
// dst[dst_idx[k]][0 .. len) = src[src_idx[k]][0 .. len), k < count; a null index list means block k
hipError_t launch_copy_blocks(hipStream_t st, const double *src, const int *d_src_idx, double *dst, const int *d_dst_idx, int count, long len);
// Jacobian requests: dX[k] = x, dF[k] = fvec of problem list[k]
hipError_t launch_gather_jac(hipStream_t st, const PoolDev &pool, const int *d_list, int count, double *d_X, double *d_F);
// column-major Jacobians J[k][n * n] (what the FD / variational kernels write) into the problems' row-major matrices
hipError_t launch_scatter_jac(hipStream_t st, const PoolDev &pool, const int *d_list, int count, const double *d_J);
// out[k] = [x (n) | fvec (n)] of problem list[k] (a finished solve's result)
hipError_t launch_gather_result(hipStream_t st, const PoolDev &pool, const int *d_list, int count, double *d_out);

}  // namespace devsolver
}  // namespace socp
