/*
 * socp_hip.h -- C-ABI of the MI355X (gfx950) hot path of SOCP.
 *
 * Plain C types only (pointers, sizes, int status); no C++ or torch types cross this line.
 * Each entry point names the reference interface it replaces (file:line into bherisse/socp).
 * Host programs reach it either directly (ctypes / dlopen) or through the C++ mirror of the
 * reference's own classes in socp_amd/host/ (model, goddard, doubleIntegrator, shooting).
 *
 * Conventions
 *   - every function returns SOCP_OK (0) or a negative SOCP_ERR_*; socp_last_error() gives text.
 *     No exceptions cross the boundary.  There is NO CPU fallback: without a HIP device
 *     socp_ctx_create fails with SOCP_ERR_NO_DEVICE.
 *   - "_dev" variants take DEVICE pointers and only enqueue work on the context's stream
 *     (inputs already resident in HBM); the plain variants take HOST pointers, copy in,
 *     run, copy out and synchronise.
 *   - all reals are IEEE double (commonType.hpp:16, real = double).
 *   - state vectors are [state(d) ; costate(d)], s = 2d doubles (Appendix B of SURVEY.md);
 *     batches are row-major, one trajectory / unknown vector per row.
 */
#ifndef SOCP_HIP_H_
#define SOCP_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SOCP_OK               0
#define SOCP_ERR_ARG         -1   /* bad argument / inconsistent problem description */
#define SOCP_ERR_HIP         -2   /* a HIP runtime call failed */
#define SOCP_ERR_NO_DEVICE   -3   /* no gfx950 device visible: the product does not run on CPU */
#define SOCP_ERR_UNSUPPORTED -4   /* model / mode combination without a device implementation */

/* in-tree device dynamics (twins of src/models/goddard, src/models/doubleIntegrator) */
#define SOCP_MODEL_GODDARD            1   /* dim 7, goddard.cpp:48-295 */
#define SOCP_MODEL_DOUBLE_INTEGRATOR  2   /* dim 6, doubleIntegrator.cpp:49-300 */
#define SOCP_MODEL_COVID19            3   /* dim 4, covid19.cpp:53-165 (control dimension 1) */
#define SOCP_MODEL_INTERCEPTOR        4   /* dim 6, interceptor.cpp:69-998: two charts, two stages, own ComputeTraj and
                                             final rows (control dimension 2: u, beta) */

/* packed parameter block, refreshed before every Newton solve (parameters are mutated by the
 * continuation loop through a raw real&, shooting.cpp:695-707) */
#define SOCP_GODDARD_NPARAMS 8   /* C, b, KD, kr, u_max, mu1, mu2, singularControl (goddard.hpp:28-37) */
#define SOCP_DINT_NPARAMS    3   /* u_max, a_max, muT (doubleIntegrator.hpp:24-28) */
#define SOCP_COVID_NPARAMS   8   /* R0, Tinf, Tinc, N, Imax, muI, umin, umax (covid19.hpp parameters_struct) */
/* interceptor.hpp:28-46 in declaration order without the never-read r_2p/t_2p, then data->R_Earth, data->mu0,
 * data->chartLimit (interceptor.cpp:52-58):  c0, hr, d0, eta, propellant_mass, empty_mass, q, ve, alpha_max, u_max,
 * a_max, mu_gft, muT, muV, muC, R_Earth, mu0, chartLimit */
#define SOCP_INTERCEPTOR_NPARAMS 18
#define SOCP_MAX_NPARAMS         24   /* capacity of a packed parameter block */

/* time / state modes, model.hpp:34-38 */
#define SOCP_FIXED      0
#define SOCP_FREE       1
#define SOCP_CONTINUOUS 2

/* kernel variants behind one call */
#define SOCP_VARIANT_AUTO       0   /* default: the reference-order kernels (= LANE_EXACT), the flavour every parity claim is
                                       made on; the environment variable SOCP_VARIANT=fast|exact changes the default */
#define SOCP_VARIANT_LANE_EXACT 1   /* one trajectory per lane, reference operation order, no FMA contraction: bit-identical
                                       to the CPU path for goddard / doubleIntegrator / covid19 */
#define SOCP_VARIANT_LANE_FAST  2   /* one trajectory per lane, reciprocal/FMA-restructured arithmetic (<= 1e-8 after 1e4 steps) */
/* There is no selectable one-trajectory-per-wavefront variant for the state-only path (north_star sketches one): a wave
 * instruction costs the same issue slots with 1 or 64 active lanes (scripts/probes/probe_lanes.hip), and spreading ONE
 * Goddard trajectory over lanes does not shorten its instruction stream (DESIGN.md section 3 has the count), so it would be
 * the same latency at 1/16 of the throughput.  The variational (is_jac = 1) integration, 156 values per trajectory, IS one
 * wavefront per trajectory with its stage vectors in LDS -- always, not as a variant. */

/* what socp_eval_batch computes */
#define SOCP_EVAL_RHS         0   /* odeTools.hpp:82  Model(t, X, isJac)      -> len(X) values  */
#define SOCP_EVAL_CONTROL     1   /* model.hpp:375    Control(t, X)           -> control dim (3, 3, 1) */
#define SOCP_EVAL_HAMILTONIAN 2   /* model.hpp:384    Hamiltonian(t, X, 0)    -> 1 value        */

typedef struct socp_ctx socp_ctx;

/* ---- context ---------------------------------------------------------------------------- */
/* replaces: construction of a model object + its parameters (goddard.cpp:23-40,
 * doubleIntegrator.cpp:26-34).  device < 0 selects the current HIP device. */
int socp_ctx_create(socp_ctx **ctx, int model_id, int device);
int socp_ctx_destroy(socp_ctx *ctx);
const char *socp_last_error(const socp_ctx *ctx);   /* ctx may be NULL: last creation error */

int socp_ctx_set_params(socp_ctx *ctx, const double *params, int nparams);
int socp_ctx_get_params(const socp_ctx *ctx, double *params, int nparams);
int socp_ctx_num_params(const socp_ctx *ctx);        /* length of the model's packed parameter block (< 0: error) */
int socp_ctx_set_step_number(socp_ctx *ctx, int step_nbr);      /* model::stepNbr, model.hpp:367 */
/* integrator of every later call: SOCP_INT_RK4 = the reference's default fixed-step loop
 * (odeTools.cpp:135-145); SOCP_INT_DOPRI5 = what it runs when built with -D_USE_BOOST (odeTools.cpp:129-134),
 * abs = rel tolerance `tol` = odeTools::odeIntTol (set from the solver's xtol by shooting::SetPrecision,
 * shooting.cpp:447-450); initial step (tf - t0)/stepNbr.  Per-lane step control; reference-order RHS. */
#define SOCP_INT_RK4    0
#define SOCP_INT_DOPRI5 1
int socp_ctx_set_integrator(socp_ctx *ctx, int kind, double tol);
int socp_ctx_set_switching_times(socp_ctx *ctx, const double *sw, int nsw);  /* goddard.cpp:373-377 */
int socp_ctx_get_switching_times(const socp_ctx *ctx, double *sw2);          /* the two values the control law reads */
int socp_ctx_set_variant(socp_ctx *ctx, int variant);
int socp_ctx_get_variant(const socp_ctx *ctx);      /* SOCP_VARIANT_* as set (AUTO = reference order) */
/* enqueue on the caller's hipStream_t (NULL is the device's default stream); use_own != 0 switches
 * back to the context's private non-blocking stream */
int socp_ctx_set_stream(socp_ctx *ctx, void *hip_stream, int use_own);
/* the stream launches currently go to (the context's own stream or the one given to socp_ctx_set_stream) */
int socp_ctx_get_stream(const socp_ctx *ctx, void **hip_stream);
/* a second stream owned by the context (non-blocking, highest priority where the device has priorities: its own hardware
 * queue, so that work on it overlaps work on the context's stream).  Created on the first call -- that takes ~6 ms, which is
 * why the lock-step engines keep it instead of creating one per call -- and destroyed with the context.  Nothing is ever
 * enqueued on it by the context itself: the engines (socp_chains_solve, socp_multistart_solve) use it while they run. */
int socp_ctx_aux_stream(socp_ctx *ctx, void **hip_stream);
/* Pay now what a process otherwise pays inside its first large call: the context's second stream (above) and the start-up of the
 * copy engines -- the first copy of more than 16 KB between pinned host memory and the device takes ~8 ms, once per process
 * (measured: scripts/probes/first_copy.hip).  Optional; for callers whose first solve is latency-critical. */
int socp_ctx_warm_up(socp_ctx *ctx);
int socp_ctx_synchronize(socp_ctx *ctx);
int socp_ctx_dims(const socp_ctx *ctx, int *dim, int *state_len, int *state_len_jac);
int socp_ctx_control_dim(const socp_ctx *ctx);
int socp_ctx_device(const socp_ctx *ctx);            /* HIP device index the context lives on (< 0: error) */
int socp_ctx_model_id(const socp_ctx *ctx);          /* SOCP_MODEL_* / plugin id the context was created with */
/* 1 when the model integrates its variational equations on the device (modelOrder 1: is_jac = 1 trajectories,
 * socp_var_jacobian, hybrj chains) -- the in-tree double integrator, or a plugin with the aug_rhs / dhamiltonian trait
 * (socp_amd/csrc/plugin_impl.hpp); 0 otherwise (model.hpp:104-120,149-183 need Model(t, X, 1) and Hamiltonian(t, X, 1)) */
int socp_ctx_has_variational(const socp_ctx *ctx);

/* counters since creation: trajectories integrated, kernel launches */
int socp_ctx_counters(const socp_ctx *ctx, long long *trajectories, long long *launches);
/* adds to them: what a clone of `ctx` (socp_ctx_clone) integrated on its behalf -- the second group of chains of a large
 * socp_chains_solve call runs on one -- counts as ctx's */
void socp_ctx_add_counters(socp_ctx *ctx, long long trajectories, long long launches);

/* ---- batched trajectory integration ------------------------------------------------------ */
/* replaces: shooting::Move -> model::ComputeTraj -> ModelInt -> odeTools::integrate -> RK4
 * (shooting.cpp:365-372, model.hpp:77-79,395-414, goddard.cpp:298-317, odeTools.cpp:89-98,128-146),
 * once per row.  t0,tf: [B]; sw: NULL (use the context's switching times) or [B][2];
 * X0,Xf: [B][len], len = s (is_jac = 0) or (s+1)*s (is_jac = 1, variational state, identity
 * block supplied by the caller exactly as shooting.cpp:1003-1005 builds it).  Xf may alias X0. */
int socp_integrate_batch(socp_ctx *ctx, int B, const double *t0, const double *tf,
                         const double *sw, const double *X0, double *Xf, int is_jac);
int socp_integrate_batch_dev(socp_ctx *ctx, int B, const double *d_t0, const double *d_tf,
                             const double *d_sw, const double *d_X0, double *d_Xf, int is_jac);

/* replaces: the observer form of odeTools::integrate (odeTools.cpp:103-123) used by the trace replay
 * (shooting.cpp:496-544, model.hpp:401-407): one trajectory, the state after every step kept.
 * dense: [cap][len], times: [cap]; row 0 = (t0, X0), row k = accumulated time and state after k
 * steps; *rows = steps + 1 (rows beyond cap are counted, not stored).  sw: NULL or 2 values.
 * Under SOCP_INT_DOPRI5 (the reference built with -D_USE_BOOST: odeTools.cpp:108, integrate_adaptive with the observer) the rows
 * are the adaptive integrator's own accepted steps -- their number is only known afterwards: call again with cap >= *rows
 * when *rows > cap. */
int socp_integrate_dense(socp_ctx *ctx, double t0, double tf, const double *sw, const double *X0,
                         double *dense, double *times, int cap, int *rows);
/* Same, also returning each row's two per-trajectory auxiliary scalars in aux[cap][2] (NULL: not wanted).  They
 * are the Goddard switching times (constant) or, for the interceptor, (stageMode, currentChart) -- what
 * interceptor::Trace prints beside the state (interceptor.cpp:131-151).  A model with its own ComputeTraj
 * (interceptor.cpp:162-218) reports the rows that function traces -- the start of each stage and every step --
 * followed by ONE extra row: the state ComputeTraj returns (chart 1) with the flags it leaves behind. */
int socp_integrate_dense_aux(socp_ctx *ctx, double t0, double tf, const double *sw, const double *X0,
                             double *dense, double *times, double *aux, int cap, int *rows);

/* replaces: model::Model / Control / Hamiltonian called outside the integrator (trace,
 * free-time rows).  t: [B]; X: [B][len]; out: [B][out_len]. */
int socp_eval_batch(socp_ctx *ctx, int what, int B, const double *t, const double *sw,
                    const double *X, int len, double *out, int is_jac);

/* ---- shooting problem --------------------------------------------------------------------- */
/* replaces: the part of shooting::data_struct the residual reads (shooting.cpp:21-35):
 * numMulti M, mode_t[M+1], mode_X[M+1][d], current node times time[M+1] and node states
 * X[M+1][2d] (only the first d of each row is read).  Must be re-sent when the continuation
 * loop blends the boundary data (shooting.cpp:609-611). */
int socp_problem_set(socp_ctx *ctx, int num_multi, const int *mode_t, const int *mode_x,
                     const double *time, const double *xnode);
int socp_problem_num_param(const socp_ctx *ctx);   /* n = 2 d M + #FREE times (shooting.cpp:179,196) */
int socp_problem_num_nodes(const socp_ctx *ctx);   /* M + 1 of the problem set (< 0: none) */

/* Per-problem blocks for the batch entry points below (batched continuation chains: shooting.cpp:598-692 blends the
 * boundary data of ONE problem per Newton solve, :695-778 moves ONE model parameter through a real&; with many chains in
 * one launch every row needs its own).  Device pointers, or NULL for "shared" (the context's parameters / the tables of
 * socp_problem_set); they stay in force for every later *_dev batch call until replaced:
 *   d_params[q][stride]   stride = nparams + 2: the packed parameters of problem q, then its two default switching times
 *   d_time[q][M+1], d_xnode[q][(M+1)*2d]   current node times / node states of problem q (what socp_problem_set takes)
 * Row q of socp_residual_batch_dev, problem q of socp_fd_jacobian_multi_dev / socp_fd_rows_dev read block q.  Modes, M and n
 * are those of socp_problem_set.  With d_params the Goddard control law is chosen per problem from its own mu2. */
int socp_problem_set_blocks_dev(socp_ctx *ctx, const double *d_params, int stride, const double *d_time,
                                const double *d_xnode);
/* Goddard with d_params: a promise that EVERY parameter block has mu2 > 0 (the smooth control law of goddard.cpp:137-145), which
 * lets the batch launches take the kernel specialised on that law (three waves per SIMD instead of two; with shared parameters
 * the library sees mu2 itself).  Stays in force until socp_problem_set or a call with all_smooth = 0.  A wrong promise computes
 * the smooth law for blocks that asked for the bang / singular / off law. */
int socp_problem_blocks_all_smooth(socp_ctx *ctx, int all_smooth);
/* host-pointer form of socp_residual_batch with per-row blocks (any of params / time / xnode may be NULL) */
int socp_residual_batch_blocks(socp_ctx *ctx, int B, const double *Z, const double *params, int stride,
                               const double *time, const double *xnode, double *F);

/* replaces: shooting::ComputeTimeLine (shooting.cpp:1579-1617) for one unknown vector */
int socp_timeline(socp_ctx *ctx, const double *z, double *timeline);

/* replaces: shooting::StaticShootingFunction -> ShootingFunction[Parallel]
 * (shooting.cpp:859-874,918-993,1133-1158,1215-1307), for B unknown vectors at once:
 * Z[B][n] -> F[B][n].  One trajectory per (row, segment); every trajectory writes its own
 * residual slots, as the reference's threads do. */
int socp_residual_batch(socp_ctx *ctx, int B, const double *Z, double *F);
int socp_residual_batch_dev(socp_ctx *ctx, int B, const double *d_Z, double *d_F);

/* replaces: MINPACK fdjac1 as hybrd drives it (call site shooting.cpp:803-826):
 * J[:,j] = (F(z + h_j e_j) - fvec) / h_j, h_j = sqrt(max(epsfcn, eps_mach)) * |z_j| (or that
 * factor itself when z_j == 0).  fjac is column-major with leading dimension n.  The n
 * perturbed residuals are one batch.  dedup != 0 integrates only the segments a column can
 * change (bit-identical result, fewer trajectories; SURVEY 7 "free win"). */
int socp_fd_jacobian(socp_ctx *ctx, const double *z, const double *fvec, double epsfcn,
                     double *fjac, int dedup);
int socp_fd_jacobian_dev(socp_ctx *ctx, const double *d_z, const double *d_fvec, double epsfcn,
                         double *d_fjac, int dedup);

/* The same for `np` independent unknown vectors of ONE problem structure (multi-start /
 * continuation sweeps): Z[np][n], Fvec[np][n] -> Fjac[np][n*n]; one launch. */
int socp_fd_jacobian_multi_dev(socp_ctx *ctx, int np, const double *d_Z, const double *d_Fvec,
                               double epsfcn, double *d_Fjac, int dedup);

/* The (n+1) residual rows of a forward-difference Jacobian in ONE launch: Rows[np][n+1][n], row 0
 * = F(z), row j+1 = F(z + h_j e_j); base and perturbed trajectories are independent, so nothing
 * waits on the base evaluation.  This is "one Newton step's batch": (n+1)*M trajectories per
 * problem (15 rows).  socp_fd_diff_dev turns rows into the Jacobian. */
int socp_fd_rows_dev(socp_ctx *ctx, int np, const double *d_Z, double epsfcn, double *d_Rows);
int socp_fd_rows(socp_ctx *ctx, int np, const double *Z, double epsfcn, double *Rows);
int socp_fd_diff_dev(socp_ctx *ctx, int np, const double *d_Z, double epsfcn, const double *d_Rows,
                     double *d_Fjac);

/* replaces: shooting::ShootingFunctionJacobian (shooting.cpp:996-1130), variational Jacobian
 * for models with modelOrder == 1 (socp_ctx_has_variational); fjac column-major as handed to hybrj (shooting.cpp:889-893).
 * The variational state follows the context's integrator (socp_ctx_set_integrator): fixed-step RK4, or -- as the reference does
 * for EVERY integrate() call when built with -D_USE_BOOST, odeTools.cpp:129-134 -- adaptive Dormand-Prince on the whole augmented
 * state, one wavefront per trajectory with per-wave step control ([ext] parity unpinned, like the state-only adaptive path). */
int socp_var_jacobian(socp_ctx *ctx, const double *z, double *fjac);
/* The same for `np` unknown vectors of one problem structure, device pointers: Z[np][n] -> Fjac[np][n*n]; one wavefront per
 * (problem, segment); per-problem blocks (socp_problem_set_blocks_dev) apply. */
int socp_var_jacobian_multi_dev(socp_ctx *ctx, int np, const double *d_Z, double *d_Fjac);

#ifdef __cplusplus
}
#endif
#endif /* SOCP_HIP_H_ */
