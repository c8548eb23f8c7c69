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
};

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
