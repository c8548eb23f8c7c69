// dev_common.hpp -- data shared by the host C-ABI layer and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace socp {

constexpr int kMaxParams = 24;          // the interceptor packs 18 (models_interceptor.hpp)
constexpr int kMaxNodes = 1025;        // M + 1 upper bound for a device-resident problem

// Packed model parameters + default switching times; passed to kernels by value (kernarg).
struct ModelParams {
    double p[kMaxParams];
    double sw0, sw1;                   // goddard data->switchingTimes[0..1] when no per-row value is given
    int step_nbr;                      // model::stepNbr (model.hpp:367)
    int integrator;                    // 0 = fixed-step RK4 (odeTools.cpp:135-145), 1 = adaptive Dormand-Prince (:129-134)
    double tol;                        // odeTools::odeIntTol: abs = rel tolerance of the adaptive integrator
};

// Shooting problem tables (device pointers), built by socp_problem_set.
// node_kind[k]:  >=0 -> FREE junction, value = index into z of its time unknown
//                -1  -> FIXED junction, value time[k]
//                -2  -> CONTINUOUS node, interpolated between junctions lo[k] and hi[k]
struct ProblemDev {
    int dim;                           // d
    int M;                             // numMulti
    int n;                             // number of unknowns
    int sw_node0, sw_node1;            // nodes whose times are the model's switchingTimes[0], [1] (-1: none)
    const int *node_kind;              // [M+1]
    const int *lo;                     // [M+1]
    const int *hi;                     // [M+1]
    const int *ft_row;                 // [M+1] residual row of the free-time equation of node k (-1: none)
    const int *mode_x;                 // [(M+1)*d]
    const double *time;                // [M+1]
    const double *xnode;               // [(M+1)*2d]
    // Per-problem blocks (batched continuation chains, shooting.cpp:598-778 run for many chains at once): when set, row /
    // problem q of a launch reads its OWN packed model parameters and / or boundary tables instead of the shared ones --
    //   pp_params[q][pp_stride]: p[0 .. pp_stride-2) then sw0, sw1      (parameter homotopy: the real& the loop mutates)
    //   pp_time[q][M+1], pp_xnode[q][(M+1)*2d]                          (boundary-data homotopy, shooting.cpp:609-611)
    // Structure (modes, M, n) is shared by all problems of a launch.  Null = shared value.
    const double *pp_params;
    const double *pp_time;
    const double *pp_xnode;
    int pp_stride;
    // Goddard only: the caller guarantees that EVERY block of pp_params has mu2 > 0 (smooth control law, goddard.cpp:137-145), so
    // the launch may take the smooth-law specialisation (156 instead of 254 VGPRs: three waves per SIMD instead of two).  The
    // lock-step engine sets it once per call from its chains' parameters; 0 = unknown -> the general-law kernel.
    int pp_smooth;
};

// the per-problem view of (P, pb) for problem q; every index is static so the blocks stay in registers
__device__ __forceinline__ void load_problem_block(const ProblemDev &pb, long q, ModelParams &Pq, ProblemDev &pq)
{
    if (pb.pp_params) {
        const double *src = pb.pp_params + q * pb.pp_stride;
        const int np = pb.pp_stride - 2;
#pragma unroll
        for (int k = 0; k < kMaxParams; k++)
            if (k < np) Pq.p[k] = src[k];
        Pq.sw0 = src[np];
        Pq.sw1 = src[np + 1];
    }
    if (pb.pp_time) pq.time = pb.pp_time + q * (pb.M + 1);
    if (pb.pp_xnode) pq.xnode = pb.pp_xnode + q * (long)(pb.M + 1) * 2 * pb.dim;
}

// shooting::ComputeTimeLine, one node (shooting.cpp:1586-1613): junction value, or the uniform
// interpolation  tl[cur] + (k-cur)*(tl[j]-tl[cur])/(j-cur)  in that operation order.
__device__ __forceinline__ double junction_time(const ProblemDev &pb, const double *__restrict__ z, int j)
{
    const int kind = pb.node_kind[j];
    return kind >= 0 ? z[kind] : pb.time[j];
}

__device__ __forceinline__ double node_time(const ProblemDev &pb, const double *__restrict__ z, int k)
{
    const int kind = pb.node_kind[k];
    if (kind >= 0) return z[kind];
    if (kind == -1) return pb.time[k];
    const int a = pb.lo[k], b = pb.hi[k];
    const double ta = junction_time(pb, z, a), tb = junction_time(pb, z, b);
    return ta + (k - a) * (tb - ta) / (b - a);
}

}  // namespace socp
