// solver_sim.cpp -- the device Powell iteration (socp_amd/csrc/solver_dev.hpp) compiled for the HOST with one thread per
// problem, behind a callback interface like hybrd's, so that its arithmetic can be compared with the library's host solver
// (socp_amd/csrc/minpack.cpp) without a GPU: tests/test_devsolver_sim.py.  Test infrastructure; the product runs the same
// header inside gfx950 kernels (socp_amd/csrc/kernels_solver.hip).
#include <cstring>
#include <vector>

#define SOCP_SOLVER_HOST 1
#include "../../socp_amd/csrc/solver_dev.hpp"

using namespace socp::devsolver;

extern "C" {
// the device solver's norm (its branch-free usual case and the general three-accumulator loop behind it)
int sim_lazy_capacity(int n) { return lazy_capacity(n); }
double sim_enorm(int n, const double *x, long stride) { return enorm(n, x, stride); }

typedef int (*sim_fcn)(int n, const double *x, double *fvec);
typedef int (*sim_jac)(int n, const double *x, const double *fvec, double *fjac_colmajor);

// returns info; outputs as hybrd leaves them (fjac = Q column-major, r packed by rows).  blocked == 1: the Jacobian refreshes go
// through factor_blocked (panels + column blocks) instead of factor; blocked == 2: Config::lazy_q (Q kept as factorised, Broyden's
// rotations as a list: the throughput flavour -- fjac then comes back as of the last refresh / flush)
int sim_solve(int n, double *x, double *fvec, double xtol, int maxfev, double epsfcn, double factor, int analytic, sim_fcn fcn, sim_jac jac,
              int *nfev, int *njev, double *fjac, double *r, double *qtf, double *diag, int blocked)
{
    Config c;
    c.n = n; c.ld = ld_for(n); c.maxfev = maxfev; c.mode = 1; c.analytic = analytic; c.xtol = xtol; c.epsfcn = epsfcn; c.factor = factor;
    c.lazy_q = blocked == 2 ? 1 : 0;
    std::vector<double> ws((size_t)ws_doubles(n, c.ld), 0.0), J((size_t)n * n), lds((size_t)blocked_lds_doubles(n), 0.0);
    State st;
    std::memset(&st, 0, sizeof(st));
    SerialExec ex;
    start(ex, c, st, ws.data(), x);
    Work w(ws.data(), n, c.ld);
    int flag = 0;
    for (;;) {
        {
            Machine<SerialExec> m(ex, c, st, ws.data());
            if (blocked == 1) { m.blocked_panel = lds.data(); m.blocked_block = lds.data() + (size_t)n * kPanel; }
            m.advance(flag);
        }
        if (st.req == RQ_DONE) break;
        if (st.req == RQ_FVEC) {
            flag = fcn(n, st.eval_sel ? w.wa2 : w.x, st.eval_sel ? w.wa4 : w.fvec);
        } else {
            flag = jac(n, w.x, w.fvec, J.data());
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) w.A[(size_t)i * c.ld + j] = J[i + (size_t)n * j];
        }
    }
    std::memcpy(x, w.x, sizeof(double) * n);
    std::memcpy(fvec, w.fvec, sizeof(double) * n);
    std::memcpy(r, w.r, sizeof(double) * n * (n + 1) / 2);
    std::memcpy(qtf, w.qtf, sizeof(double) * n);
    std::memcpy(diag, w.diag, sizeof(double) * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) fjac[i + (size_t)n * j] = w.A[(size_t)i * c.ld + j];
    *nfev = st.nfev; *njev = st.njev;
    return st.info;
}
}
