// hostjac_flow.cpp -- a USER model with modelOrder = 1 and NO device twin: it integrates its own variational equations in
// Model(t, X, 1) and gives dH/dX in Hamiltonian(t, X, 1), exactly what the reference's plugin surface asks of such a class
// (model.hpp:104-120,149-183; shooting.cpp:828-852,996-1130).  The class below restates the 3-D double integrator
// (doubleIntegrator.cpp:49-300) on the host, so the programs of tests/testDoubleIntegrator.cpp and
// tests/testDoubleIntegrator_WP.cpp can run through the host path (hybrj + host Jacobian assembly + numThread segment
// workers) and be compared, bit for bit, with the golden Newton histories of the device path / the oracle.
//   hostjac_flow basic <xtol> <numThread>
//   hostjac_flow wp    <xtol> <numThread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "socp/shooting.hpp"

class dint_host : public model
{
public:
    real u_max, a_max, muT;
    dint_host() : model(6, 1, 30, ""), u_max(1), a_max(1), muT(0.01) {}

    virtual mcontrol Control(real const &, mstate const &X) const
    {
        mcontrol u(3);
        for (int k = 0; k < 3; k++) u[k] = -X[9 + k] / a_max;
        const real nu = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (nu > u_max)
            for (int k = 0; k < 3; k++) u[k] = u[k] / nu * u_max;
        return u;
    }
    virtual mstate Model(real const &t, mstate const &X, int isJac) const
    {
        const int s = 12;
        const mcontrol u = Control(t, X);
        mstate f(s, 0.0);
        for (int k = 0; k < 3; k++) { f[k] = X[3 + k]; f[3 + k] = a_max * u[k]; f[9 + k] = -X[6 + k]; }
        if (!isJac) return f;
        // d/dt R = (df/dX) R with the constant matrix of the unsaturated law: v' = -p_v, p_v' = -p_x, x' = v
        real A[12][12] = {{0}};
        for (int k = 0; k < 3; k++) { A[k][3 + k] = 1; A[3 + k][9 + k] = -1; A[9 + k][6 + k] = -1; }
        mstate out(s, 0.0);
        out = f;
        out.resize((size_t)(s + 1) * s, 0.0);
        for (int i = 0; i < s; i++)
            for (int j = 0; j < s; j++) {
                real acc = 0;
                for (int k = 0; k < s; k++) acc += A[i][k] * X[(size_t)s * (k + 1) + j];
                out[(size_t)s * (i + 1) + j] = acc;
            }
        return out;
    }
    virtual mstate Hamiltonian(real const &t, mstate const &X, int isJac) const
    {
        const mcontrol u = Control(t, X);
        const real nu = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (!isJac)
            return mstate(1, muT + a_max * a_max * nu * nu / 2 + X[6] * X[3] + X[7] * X[4] + X[8] * X[5] +
                                 a_max * (X[9] * u[0] + X[10] * u[1] + X[11] * u[2]));
        mstate dH(13, 0.0);
        for (int k = 0; k < 3; k++) { dH[3 + k] = X[6 + k]; dH[6 + k] = X[3 + k]; dH[9 + k] = -X[9 + k]; }
        return dH;
    }
};

static void report(const char *stage, int info, const shooting &s)
{
    std::vector<real> z;
    s.GetParameters(z);
    std::vector<int> calls = s.GetCallNumber();
    std::printf("{\"stage\": \"%s\", \"info\": %d, \"nfev\": %d, \"njev\": %d, \"n\": %d, \"trajectories\": %lld, \"z\": [", stage,
                info, calls[0], calls[1], (int)z.size(), s.GetTrajectoryCount());
    for (size_t k = 0; k < z.size(); k++) std::printf("%s%.17g", k ? ", " : "", z[k]);
    std::printf("]}\n");
    std::fflush(stdout);
}

static int basic(double xtol, int threads)
{
    // testDoubleIntegrator.cpp:24-143
    dint_host m;
    const int d = m.GetDim();
    shooting sh(m, 1, threads);
    sh.SetPrecision(xtol);
    sh.SetContinuationMinStep(1e-12);
    sh.SetMode(1, std::vector<int>(d, 0));
    const real ti = 0, tf = 10;
    model::mstate Xi(2 * d, 0.0), Xf(2 * d, 0.0);
    for (int k = d; k < 2 * d; k++) Xi[k] = 0.01;
    Xf[0] = 10.0; Xf[1] = 15.0;
    sh.InitShooting(ti, Xi, tf, Xf);
    int info = sh.SolveOCP(0.0);
    report("solve", info, sh);
    Xf[1] = 20;
    sh.SetDesiredState(ti, Xi, tf, Xf);
    info = sh.SolveOCP(1.0);
    report("data_continuation", info, sh);
    if (info == 1) info = sh.SolveOCP(1.0, m.muT, 0.02);
    report("muT_continuation", info, sh);
    return info == 1 ? 0 : 2;
}

static int wp(double xtol, int threads)
{
    // testDoubleIntegrator_WP.cpp:26-150: two segments, FREE interior and final times, way-point position pinned
    dint_host m;
    const int d = m.GetDim(), M = 2;
    shooting sh(m, M, threads);
    sh.SetPrecision(xtol);
    std::vector<int> mode_t(M + 1, 1);
    mode_t[0] = 0;
    std::vector<std::vector<int> > mode_X(M + 1, std::vector<int>(d, 0));
    for (int i = 1; i < M; i++) mode_X[i][3] = mode_X[i][4] = mode_X[i][5] = 2;
    sh.SetMode(mode_t, mode_X);
    std::vector<real> vt(M + 1);
    std::vector<model::mstate> vX(M + 1, model::mstate(2 * d, 0.0));
    for (int i = 0; i <= M; i++) {
        vt[i] = 60.0 * i / M;
        vX[i][0] = 20.0 * i / M;
        if (i < M) for (int k = d; k < 2 * d; k++) vX[i][k] = 0.001;
    }
    sh.InitShooting(vt, vX);
    int info = sh.SolveOCP(0.0);
    report("solve", info, sh);
    vX[1][1] = 15.0; vX[2][1] = 5.0; vX[2][2] = 10.0;
    sh.SetDesiredState(vt, vX);
    info = sh.SolveOCP(1.0);
    report("data_continuation", info, sh);
    if (info == 1) info = sh.SolveOCP(1.0, m.muT, 0.02);
    report("muT_continuation", info, sh);
    return info == 1 ? 0 : 2;
}

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: hostjac_flow basic|wp <xtol> <numThread>\n"); return 64; }
    const double xtol = std::atof(argv[2]);
    const int threads = std::atoi(argv[3]);
    return std::strcmp(argv[1], "basic") == 0 ? basic(xtol, threads) : wp(xtol, threads);
}
