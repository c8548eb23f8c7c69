// models_variational.hpp -- in-tree models with variational equations (modelOrder 1): the double integrator's augmented
// right-hand side and dH/dX in the reference's operation order (doubleIntegrator.cpp:113-213, 293-297).
#pragma once
#include "models_exact.hpp"
#include "variational.hpp"

namespace socp {

// ---- double integrator: variational right-hand side and dH/dX ---------------------------------
struct DIntVar : DIntExact {
    // element e of Model(t, Y, isJac = 1) (doubleIntegrator.cpp:113-213).  df/dX is constant and has
    // one nonzero per row (:155-166): rows 0-2 -> +R[row+3], rows 3-5 -> -R[row+6], rows 6-8 -> 0,
    // rows 9-11 -> -R[row-3]; the reference sums the zero terms too, which changes nothing finite.
    __device__ static __forceinline__ double aug_rhs(const ModelParams &P, double /*t*/, int e, const double *Y)
    {
        if (e < S) {
            if (e < 3) return Y[e + 3];
            if (e < 6) {
                double X[S];
#pragma unroll
                for (int k = 0; k < S; k++) X[k] = Y[k];
                double u[3];
                control_only(P, 0, 0, 0, X, u);
                return P.p[DP_AMAX] * u[e - 3];
            }
            if (e < 9) return 0.0;
            return -Y[e - 3];
        }
        const int i = (e - S) / S, j = (e - S) - i * S;
        if (i < 3) return 0.0 + 1.0 * Y[S + S * (i + 3) + j];
        if (i < 6) return 0.0 + (-1.0) * Y[S + S * (i + 6) + j];
        if (i < 9) return 0.0;
        return 0.0 + (-1.0) * Y[S + S * (i - 3) + j];
    }

    // doubleIntegrator.cpp:293-297: {0,0,0, p_x,p_y,p_z, vx,vy,vz, -p_vx,-p_vy,-p_vz, 0}
    __device__ static __forceinline__ void dhamiltonian(const ModelParams &, double /*t*/, const double *X, double (&dH)[S + 1])
    {
        dH[0] = 0; dH[1] = 0; dH[2] = 0;
        dH[3] = X[6]; dH[4] = X[7]; dH[5] = X[8];
        dH[6] = X[3]; dH[7] = X[4]; dH[8] = X[5];
        dH[9] = -X[9]; dH[10] = -X[10]; dH[11] = -X[11];
        dH[12] = 0;
    }
};

}  // namespace socp
