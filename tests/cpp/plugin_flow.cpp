// plugin_flow.cpp -- a USER-DEFINED model class on the reference's plugin surface (class model), whose
// dynamics live in an out-of-tree device plugin (tests/plugin/lqr1d_plugin.hip), solved with shooting.
//   plugin_flow <path/to/liblqr1d_plugin.so> <numMulti> [modelOrder] [free_tf]
// modelOrder 1: the plugin's variational trait (aug_rhs / dhamiltonian) -> hybrj with the device Jacobian (shooting.cpp:828-852)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "socp/shooting.hpp"
#include "socp_hip.h"
#include "socp_plugin.h"

class lqr1d : public model
{
public:
    explicit lqr1d(int order = 0) : model(2, order, 20, "") {}
    real gain = 1.0;
    virtual int DeviceModelId() const { return 1001; }
    virtual int DeviceParams(double *out, int cap) const { if (cap < 1) return 0; out[0] = gain; return 1; }
    virtual mstate Model(real const &t, mstate const &X, int isJac) const { return DeviceEval(SOCP_EVAL_RHS, t, X, isJac); }
    virtual mcontrol Control(real const &t, mstate const &X) const { return DeviceEval(SOCP_EVAL_CONTROL, t, X, 0); }
    virtual mstate Hamiltonian(real const &t, mstate const &X, int isJac) const { return DeviceEval(SOCP_EVAL_HAMILTONIAN, t, X, isJac); }
};

int main(int argc, char **argv)
{
    if (argc < 3) return 64;
    if (socp_plugin_load(argv[1]) != SOCP_OK) { std::fprintf(stderr, "%s\n", socp_last_error(nullptr)); return 3; }
    const int M = std::atoi(argv[2]);
    const int order = argc > 3 ? std::atoi(argv[3]) : 0;
    lqr1d m(order);
    shooting sh(m, M, 1);
    sh.SetPrecision(1e-12);
    sh.SetMode(model::FIXED, std::vector<int>(2, model::FIXED));      // rest-to-rest in fixed time
    model::mstate Xi(4, 0.0), Xf(4, 0.0);
    Xi[2] = -1.0; Xi[3] = -1.0;                                       // costate guess
    Xf[0] = 1.0;
    sh.InitShooting(0.0, Xi, 1.0, Xf);
    const int info = sh.SolveOCP(0.0);
    std::vector<real> z;
    sh.GetParameters(z);
    const model::mstate u0 = m.Control(0.0, model::mstate(z.begin(), z.begin() + 4));
    std::printf("{\"info\": %d, \"nfev\": %d, \"njev\": %d, \"p_x\": %.17g, \"p_v\": %.17g, \"u0\": %.17g, \"n\": %d}\n", info,
                sh.GetCallNumber()[0], sh.GetCallNumber()[1], z[2], z[3], u0[0], (int)z.size());
    return info == 1 ? 0 : 2;
}
