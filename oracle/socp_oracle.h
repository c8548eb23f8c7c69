/*
 * socp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the SOCP hot path (batched state+costate integration,
 * shooting residual, finite-difference / variational Jacobian).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (socp_amd/) never links or calls it.
 *
 * Pinning: the reference ships no golden data and no asserting tests
 * (SURVEY.md 8c), so this restatement is pinned against outputs of the
 * reference itself, compiled here from /root/reference into oracle/_ref/
 * (see oracle/Makefile, oracle/ref_driver.cpp) and frozen as fixtures under
 * tests/golden/ (generator: tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line whose arithmetic ORDER it
 * follows; all arithmetic is IEEE double, compiled with -ffp-contract=off.
 */
#ifndef SOCP_ORACLE_H_
#define SOCP_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MODEL_GODDARD = 1, ORC_MODEL_DOUBLE_INTEGRATOR = 2, ORC_MODEL_COVID19 = 3, ORC_MODEL_INTERCEPTOR = 4 };
enum { ORC_FIXED = 0, ORC_FREE = 1, ORC_CONTINUOUS = 2 };   /* model.hpp:34-38 */

#define ORC_MAX_PARAMS 24
#define ORC_MAX_SWITCH 64

/* Goddard parameter slots (goddard.hpp:28-37, goddard.cpp:31-39) */
enum { GP_C = 0, GP_B, GP_KD, GP_KR, GP_UMAX, GP_MU1, GP_MU2, GP_SING };
/* doubleIntegrator parameter slots (doubleIntegrator.hpp:24-28) */
enum { DP_UMAX = 0, DP_AMAX, DP_MUT };
/* covid19 parameter slots (covid19.hpp parameters_struct) */
enum { CP_R0 = 0, CP_TINF, CP_TINC, CP_N, CP_IMAX, CP_MUI, CP_UMIN, CP_UMAX };

/* interceptor parameter slots (interceptor.hpp:28-46 in declaration order, r_2p/t_2p -- never read --
 * left out; then data->R_Earth, data->mu0, data->chartLimit, interceptor.cpp:52-58) */
enum { IP_C0 = 0, IP_HR, IP_D0, IP_ETA, IP_PROP, IP_EMPTY, IP_Q, IP_VE, IP_ALPHA_MAX, IP_UMAX, IP_AMAX,
       IP_MU_GFT, IP_MUT, IP_MUV, IP_MUC, IP_REARTH, IP_MU0, IP_CHART_LIMIT, IP_COUNT };

typedef struct {
    int model_id;                 /* ORC_MODEL_* */
    int dim;                      /* state dimension d (7 / 6) */
    int step_nbr;                 /* model::stepNbr */
    double p[ORC_MAX_PARAMS];     /* packed parameters */
    int nsw;                      /* number of switching times pushed by ComputeTimeLine */
    double sw[ORC_MAX_SWITCH];    /* goddard data->switchingTimes */
    int chart;                    /* interceptor data->currentChart (1 / 2), left as the last trajectory set it */
    int stage;                    /* interceptor data->stageMode (1 = powered), likewise */
    int integrator;               /* 0 = fixed-step RK4 (default), 1 = adaptive Dormand-Prince: what model::ComputeTraj runs
                                     in a -D_USE_BOOST build (odeTools.cpp:129-134) */
    double tol;                   /* odeTools::odeIntTol of that build (abs = rel) */
} orc_model;

typedef struct {
    int dim;                      /* d */
    int num_multi;                /* M */
    const int *mode_t;            /* [M+1] */
    const int *mode_x;            /* [(M+1)*d] */
    const double *time;           /* [M+1]   data->time (current, after continuation blend) */
    const double *xnode;          /* [(M+1)*2d] data->X (only first d of each row is read) */
} orc_problem;

/* ---- model layer ---- */
void orc_model_init(orc_model *m, int model_id);      /* defaults of the reference ctors */
int  orc_state_len(const orc_model *m, int is_jac);   /* 2d or (2d+1)*2d */
int  orc_control_dim(const orc_model *m);              /* 3, 3, 1 */
void orc_control(const orc_model *m, double t, const double *X, double *u3);
void orc_rhs(const orc_model *m, double t, const double *X, int is_jac, double *Xdot);
/* is_jac==0: writes H[0]; is_jac==1: writes dH/dX (2d+1 values) */
void orc_hamiltonian(const orc_model *m, double t, const double *X, int is_jac, double *H);
double orc_goddard_singular_control(const orc_model *m, double t, const double *X);

/* ---- ODE layer ---- */
void orc_rk4_step(const orc_model *m, double t, double *X, double step, int is_jac);
/* returns number of RK4 steps taken */
long orc_integrate(const orc_model *m, double *X, double t0, double tf, double dt, int is_jac);
long orc_model_int(const orc_model *m, double t0, const double *X0, double tf, int is_jac, double *Xf);
/* Adaptive integration as the reference does it when built with -D_USE_BOOST (odeTools.cpp:129-134):
 * boost::numeric::odeint::integrate_adaptive(make_dense_output<runge_kutta_dopri5>(tol, tol), ...).
 * [ext] Boost.Odeint is not vendored and absent offline: restated from its published algorithm
 * (controlled Dormand-Prince 5(4) with FSAL); validated by tolerance only.
 * Returns accepted steps; *rejected (may be NULL) counts rejected trial steps. */
long orc_integrate_dopri5(const orc_model *m, double *X, double t0, double tf, double dt, double tol, long *rejected);
/* ... on the augmented state [X ; dX/dX0] of the hybrj path (is_jac = 1 trajectories: odeTools.cpp:129-134 makes no difference
 * between the two under -D_USE_BOOST); parity unpinned like the state-only form */
long orc_integrate_dopri5_jac(const orc_model *m, double *X, double t0, double tf, double dt, double tol, long *rejected);
/* The same loop with a hook called before every step (may be NULL); a hook that rewrites X (the interceptor's chart
 * change, interceptor.cpp:953-978) returns 1 and the FSAL derivative is recomputed.  At most ORC_ADAPTIVE_BUDGET trial
 * steps per call, then the state is NaN (odeint's step_adjustment_error; the device kernels do the same). */
#define ORC_ADAPTIVE_BUDGET 50000
typedef int (*orc_step_hook)(orc_model *m, double t, double *X);
long orc_integrate_dopri5_hook(orc_model *m, double *X, double t0, double tf, double dt, double tol, long *rejected,
                               orc_step_hook hook);
/* batch of independent trajectories, one per row; aux_sw may be NULL, else [B][2] */
void orc_integrate_batch(const orc_model *m, int B, const double *t0, const double *tf,
                         const double *aux_sw, const double *X0, double *Xf, int is_jac);

/* model::ComputeTraj: ModelInt for every model but the interceptor, which overrides it
 * (two stages + chart switching, interceptor.cpp:162-218) and leaves m->chart / m->stage behind */
void orc_compute_traj(orc_model *m, double t0, const double *X0, double tf, int is_jac, double *Xf);

/* ---- interceptor (interceptor_oracle.c; PARITY UNPINNED, see that file's header) ---- */
typedef void (*orc_interceptor_observer)(void *ctx, double t, const double *X, int chart, int stage);
void orc_interceptor_init(orc_model *m);
void orc_interceptor_rhs(const orc_model *m, double t, const double *X, double *Xdot);
void orc_interceptor_control(const orc_model *m, double t, const double *X, double *u_beta);
double orc_interceptor_hamiltonian(const orc_model *m, double t, const double *X);
double orc_interceptor_hamiltonian_at(const orc_model *m, double t, const double *X, const double *u_beta);
void orc_interceptor_chart12(const orc_model *m, const double *X1, double *X2);
void orc_interceptor_chart21(const orc_model *m, const double *X2, double *X1);
void orc_lu6_solve(const double A[6][6], const double *b, double *x);
void orc_interceptor_compute_traj(orc_model *m, double t0, const double *X0, double tf, double *Xf);
void orc_interceptor_compute_traj_obs(orc_model *m, double t0, const double *X0, double tf, double *Xf,
                                      orc_interceptor_observer obs, void *ctx);
/* ComputeTraj with the adaptive integrator in place of the 50 fixed steps per stage (BASELINE config 5: the reference's
 * interceptor is RK4-only, so this combination is an extrapolation): same stage split, chart re-chosen before every step */
void orc_interceptor_compute_traj_adaptive(orc_model *m, double t0, const double *X0, double tf, double tol, double *Xf);
void orc_interceptor_final_rows(const orc_model *m, const double *Xtf, const double *Xf, const int *mode_x, double *fvec);
void orc_interceptor_init_analytical(const orc_model *m, double ti, double *Xi, double tf, const double *Xf);

/* ---- shooting layer ---- */
/* The default residual blocks of model.hpp:90-328 one at a time, in both forms, so that each can be
 * pinned against the reference's own objects (oracle/ref_driver.cpp: ref_model_block; tests/test_oracle.py):
 * which = 0 InitialFunction, 1 InitialHFunction, 2 FinalFunction, 3 FinalHFunction, 4 SwitchingTimesFunction (the
 * model's own, i.e. goddard.cpp:343-370 for Goddard).  X: state at the node (2d, or (2d+1)*2d when is_jac);
 * other: desired boundary state (2d) resp. the state after the switching time (same length as X); mode: d state modes.
 * Returns the number of values written. */
int  orc_residual_block(const orc_model *m, int which, double t, const double *X, const double *other,
                        const int *mode, int is_jac, double *out);
int  orc_num_param(const orc_problem *p);
void orc_compute_timeline(orc_model *m, const orc_problem *p, const double *z, double *timeline);
void orc_shooting_function(orc_model *m, const orc_problem *p, const double *z, double *fvec);
/* row-major n x n, as the reference assembles it before the hand-over transpose */
void orc_shooting_jacobian(orc_model *m, const orc_problem *p, const double *z, double *fjac_rm);
/* MINPACK fdjac1 as hybrd drives it (dense): fjac column-major, ldfjac = n */
void orc_fdjac1(orc_model *m, const orc_problem *p, const double *z, const double *fvec,
                double epsfcn, double *fjac_cm);
/* residual for a batch of unknown vectors Z[B][n] -> F[B][n] */
void orc_residual_batch(orc_model *m, const orc_problem *p, int B, const double *Z, double *F);

#ifdef __cplusplus
}
#endif
#endif
