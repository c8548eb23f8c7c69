// lqr1d_plugin.hip -- example of an OUT-OF-TREE device model (include/socp_plugin.h):
// minimum-energy transfer of a 1-D double integrator,  x' = v, v' = u, cost = int u^2/2,
// Pontryagin: u = -p_v, p_x' = 0, p_v' = -p_x, H = u^2/2 + p_x v + p_v u.
// State vector [x, v ; p_x, p_v].  One parameter: a control gain g (u = -g p_v, default 1).
#include "plugin_impl.hpp"

struct Lqr1D {
    static constexpr int D = 2;
    static constexpr int S = 4;
    static constexpr int NU = 1;
    static constexpr bool kRefOrder = true;

    __device__ static void control_only(const socp::ModelParams &P, double, double, double, const double (&X)[S], double (&u)[3])
    {
        u[0] = -P.p[0] * X[3]; u[1] = 0; u[2] = 0;
    }
    __device__ static void rhs(const socp::ModelParams &P, double, double, double, const double (&X)[S], double (&dX)[S])
    {
        dX[0] = X[1];
        dX[1] = -P.p[0] * X[3];
        dX[2] = 0;
        dX[3] = -X[2];
    }
    __device__ static double hamiltonian(const socp::ModelParams &P, double, double, double, const double (&X)[S])
    {
        const double u = -P.p[0] * X[3];
        return u * u / 2 + X[2] * X[1] + X[3] * u;
    }
    // optional trait: variational equations -> modelOrder = 1 / hybrj works for this model.  df/dX is constant:
    // x' = v, v' = -g p_v, p_x' = 0, p_v' = -p_x  =>  dR/dt rows:  R[1], -g R[3], 0, -R[2]
    __device__ static double aug_rhs(const socp::ModelParams &P, double, int e, const double *Y)
    {
        if (e < S) {
            if (e == 0) return Y[1];
            if (e == 1) return -P.p[0] * Y[3];
            if (e == 2) return 0.0;
            return -Y[2];
        }
        const int i = (e - S) / S, j = (e - S) - i * S;
        if (i == 0) return Y[S + S * 1 + j];
        if (i == 1) return -P.p[0] * Y[S + S * 3 + j];
        if (i == 2) return 0.0;
        return -Y[S + S * 2 + j];
    }
    // H = u^2/2 + p_x v + p_v u with u = -g p_v:  dH/d(x, v, p_x, p_v) = (0, p_x, v, g^2 p_v - 2 g p_v), dH/dt = 0
    __device__ static void dhamiltonian(const socp::ModelParams &P, double, const double *X, double (&dH)[S + 1])
    {
        const double g = P.p[0];
        dH[0] = 0; dH[1] = X[2]; dH[2] = X[1]; dH[3] = g * g * X[3] - 2 * g * X[3]; dH[4] = 0;
    }
    // optional trait: a FREE state component at an interior node is a SOFT way-point (the form the reference's one user of the hook
    // has, vtolUAV.cpp:273-284): the state is continuous, the costate jumps by gain x the distance to the way-point
    __device__ static void switching_state(const socp::ModelParams &P, double, int j, const double (&X)[S], const double (&Xp)[S], const double *Xd,
                                           double &f_state, double &f_costate)
    {
        f_state = X[j] - Xp[j];
        f_costate = (X[j + D] - Xp[j + D]) - P.p[0] * (X[j] - Xd[j]);
    }
    // ... and its Jacobian form (isJac = 1): the partial derivatives of the two rows above
    __device__ static void switching_state_jac(const socp::ModelParams &P, double, int j, const double (&)[S], const double (&)[S], const double *,
                                               double (&dfs_dX)[S], double (&dfs_dXp)[S], double (&dfc_dX)[S], double (&dfc_dXp)[S])
    {
        dfs_dX[j] = 1.0; dfs_dXp[j] = -1.0;
        dfc_dX[j + D] = 1.0; dfc_dXp[j + D] = -1.0; dfc_dX[j] = -P.p[0];
    }
    __device__ static double switching_fn(const socp::ModelParams &P, double a, double b, double t, const double (&X)[S], const double (&Xp)[S])
    {
        return hamiltonian(P, a, b, t, X) - hamiltonian(P, a, b, t, Xp);
    }
};

SOCP_DEFINE_MODEL_PLUGIN(1001, Lqr1D, 1, 20, {1.0})
