// plugin_impl.hpp -- build an OUT-OF-TREE device model into a loadable plugin.
//
// A model is one struct with the static interface the in-tree models have (models_exact.hpp):
//
//   struct MyModel {
//       static constexpr int D = ...;            // state dimension d; S = 2 d (state + costate)
//       static constexpr int S = 2 * D;
//       static constexpr int NU = ...;           // control dimension (<= 3)
//       static constexpr bool kRefOrder = true;  // true: RK4 in the reference's association order
//       __device__ static void rhs(const socp::ModelParams &P, double sw0, double sw1, double t,
//                                  const double (&X)[S], double (&dX)[S]);              // model::Model
//       __device__ static void control_only(const socp::ModelParams &P, double sw0, double sw1, double t,
//                                           const double (&X)[S], double (&u)[3]);      // model::Control
//       __device__ static double hamiltonian(const socp::ModelParams &P, double sw0, double sw1, double t,
//                                            const double (&X)[S]);                     // model::Hamiltonian
//       __device__ static double switching_fn(const socp::ModelParams &P, double sw0, double sw1, double t,
//                                             const double (&X)[S], const double (&Xp)[S]);  // SwitchingTimesFunction
//       // OPTIONAL -- model::SwitchingStateFunction (model.hpp:339-341; shooting.cpp:1535-1538): the two residual rows of state
//       // component j of an interior node whose mode is FREE; X: end of the arriving segment, Xp: the node's unknowns, Xd: the
//       // node's desired state.  Without it those rows are zero (the reference's default hook is a no-op).
//       __device__ static void switching_state(const socp::ModelParams &P, double t, int j, const double (&X)[S], const double (&Xp)[S],
//                                              const double *Xd, double &f_state, double &f_costate);
//       // OPTIONAL -- the same hook with isJac = 1 (hybrj path, shooting.cpp:1524-1538): partial derivatives of those two rows with
//       // respect to X and Xp (arrays arrive zeroed; fill the nonzeros).  The library chains them through the sensitivity block and
//       // forms the free-time column like MultipleShootingFunction does for its own rows.  Without it: zero rows.
//       __device__ static void switching_state_jac(const socp::ModelParams &P, double t, int j, const double (&X)[S], const double (&Xp)[S],
//                                                  const double *Xd, double (&dfs_dX)[S], double (&dfs_dXp)[S],
//                                                  double (&dfc_dX)[S], double (&dfc_dXp)[S]);
//       // OPTIONAL -- variational equations, for classes with modelOrder = 1 (hybrj; model.hpp:104-120,149-183):
//       __device__ static double aug_rhs(const socp::ModelParams &P, double t, int e, const double *Y);
//                 // element e of Model(t, Y, isJac = 1), Y = [X(S) ; R(S x S)], R[k][i] at Y[S (k+1) + i]  
//       __device__ static void dhamiltonian(const socp::ModelParams &P, double t, const double *X, double (&dH)[S + 1]);
//                 // Hamiltonian(t, X, isJac = 1): dH/dX (S entries), then dH/dt
//   };
//   SOCP_DEFINE_MODEL_PLUGIN(1001, MyModel, 3 /*nparams*/, 30 /*stepNbr*/, {1.0, 2.0, 3.0})
//
// compiled with   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -I<repo>/socp_amd/csrc
//                       -I<repo>/include my_model.hip -o libmy_model.so
// and loaded with socp_plugin_load("libmy_model.so"); afterwards socp_ctx_create(&ctx, 1001, dev) gives a
// context on which every entry point of socp_hip.h works (trajectories, residual, FD Jacobian, dense output,
// evaluation, adaptive integrator, lock-step multi-start).
#pragma once
#include "integrator.hpp"
#include "launch.hpp"
#include "variational.hpp"
#include "../../include/socp_plugin.h"

namespace socp {
namespace plugin {

// optional model hint kOneWavePerSimd: build and launch only the one-wave-per-SIMD instantiations (heavy
// right-hand sides whose batches never fill the chip three deep)
template <class M, class = void> struct one_wave_per_simd : std::false_type {};
template <class M> struct one_wave_per_simd<M, std::void_t<decltype(M::kOneWavePerSimd)>> : std::bool_constant<M::kOneWavePerSimd> {};

// one launch of a hot kernel: adaptive integrator -> one wave per SIMD; otherwise occupancy cap from the grid.
// PP: the PERPROB template argument (always false for the trajectory kernel)
#define SOCP_PLUGIN_LAUNCH_T(KERNEL, PP, GRID, LDS, ST, ...)                                                        \
    do {                                                                                                               \
        if (P.integrator == 1) hipLaunchKernelGGL((KERNEL<Mdl, 1, 1, PP>), dim3(GRID), dim3(64), LDS, ST, __VA_ARGS__); \
        else if constexpr (one_wave_per_simd<Mdl>::value)                                                              \
            hipLaunchKernelGGL((KERNEL<Mdl, 1, 0, PP>), dim3(GRID), dim3(64), LDS, ST, __VA_ARGS__);                 \
        else switch (wpe_for(GRID)) {                                                                                  \
        case 1: hipLaunchKernelGGL((KERNEL<Mdl, 1, 0, PP>), dim3(GRID), dim3(64), LDS, ST, __VA_ARGS__); break;       \
        case 2: hipLaunchKernelGGL((KERNEL<Mdl, 2, 0, PP>), dim3(GRID), dim3(64), LDS, ST, __VA_ARGS__); break;       \
        default: hipLaunchKernelGGL((KERNEL<Mdl, 3, 0, PP>), dim3(GRID), dim3(64), LDS, ST, __VA_ARGS__); break;      \
        }                                                                                                              \
    } while (0)
#define SOCP_PLUGIN_LAUNCH(KERNEL, GRID, ST, ...) SOCP_PLUGIN_LAUNCH_T(KERNEL, false, GRID, 0, ST, __VA_ARGS__)
// kernels that read a shooting problem `pb`: per-problem blocks (dev_common.hpp) select the PERPROB instantiation
#define SOCP_PLUGIN_LAUNCH_PB_LDS(KERNEL, GRID, LDS, ST, ...)                                                           \
    do {                                                                                                               \
        if (pb.pp_params || pb.pp_time || pb.pp_xnode) SOCP_PLUGIN_LAUNCH_T(KERNEL, true, GRID, LDS, ST, __VA_ARGS__); \
        else SOCP_PLUGIN_LAUNCH_T(KERNEL, false, GRID, LDS, ST, __VA_ARGS__);                             \
    } while (0)
#define SOCP_PLUGIN_LAUNCH_PB(KERNEL, GRID, ST, ...) SOCP_PLUGIN_LAUNCH_PB_LDS(KERNEL, GRID, 0, ST, __VA_ARGS__)

template <class Mdl>
hipError_t traj(hipStream_t st, const ModelParams &P, int B, const double *t0, const double *tf, const double *sw,
                const double *X0, double *Xf)
{
    if (B <= 0) return hipSuccess;
    SOCP_PLUGIN_LAUNCH(traj_lane_kernel, blocks_for(B), st, P, B, t0, tf, sw, X0, Xf);
    return hipGetLastError();
}
template <class Mdl>
hipError_t residual(hipStream_t st, const ModelParams &P, const ProblemDev &pb, int B, const double *Z, double *F)
{
    if (B <= 0) return hipSuccess;
    const int R = rows_per_block(pb.M, pb.n);
    const unsigned grid = R ? (unsigned)((B + R - 1) / R) : blocks_for((long)B * pb.M);
    SOCP_PLUGIN_LAUNCH_PB_LDS(residual_lane_kernel, grid, (unsigned)((long)R * pb.n * 8), st, P, pb, B, Z, F, R);
    return hipGetLastError();
}
template <class Mdl>
hipError_t fdjac(hipStream_t st, const ModelParams &P, const ProblemDev &pb, int np, int T, const int2 *pairs,
                 const double *z, const double *fvec, double eps, double *fjac)
{
    if (np <= 0 || T <= 0) return hipSuccess;
    SOCP_PLUGIN_LAUNCH_PB(fdjac_lane_kernel, blocks_for((long)np * T), st, P, pb, np, T, pairs, z, fvec, eps, fjac);
    return hipGetLastError();
}
template <class Mdl>
hipError_t fdrows(hipStream_t st, const ModelParams &P, const ProblemDev &pb, int np, const double *z, double eps, double *rows)
{
    if (np <= 0) return hipSuccess;
    const long vrows = (long)np * (pb.n + 1);
    const int R = rows_per_block(pb.M, pb.n);
    const unsigned grid = R ? (unsigned)((vrows + R - 1) / R) : blocks_for(vrows * pb.M);
    SOCP_PLUGIN_LAUNCH_PB_LDS(fdrows_lane_kernel, grid, (unsigned)((long)R * pb.n * 8), st, P, pb, np, z, eps, rows, R);
    return hipGetLastError();
}
template <class Mdl>
hipError_t dense(hipStream_t st, const ModelParams &P, double t0, double tf, double sw0, double sw1, const double *X0,
                 double *out, double *times, int cap, int *rows, double *aux)
{
    if (P.integrator == 1) hipLaunchKernelGGL((traj_dense_kernel<Mdl, 1>), dim3(1), dim3(64), 0, st, P, t0, tf, sw0, sw1, X0, out, times, cap, rows, aux);
    else hipLaunchKernelGGL((traj_dense_kernel<Mdl, 0>), dim3(1), dim3(64), 0, st, P, t0, tf, sw0, sw1, X0, out, times, cap, rows, aux);
    return hipGetLastError();
}
template <class Mdl>
hipError_t eval(hipStream_t st, const ModelParams &P, int what, int B, const double *t, const double *sw, const double *X, double *out)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(eval_lane_kernel<Mdl>, dim3(blocks_for(B)), dim3(64), 0, st, P, what, B, t, sw, X, out);
    return hipGetLastError();
}

// optional trait: the model integrates its variational equations (aug_rhs + dhamiltonian) -> the hybrj path works for it
template <class M, class = void> struct has_variational : std::false_type {};
template <class M>
struct has_variational<M, std::void_t<decltype(M::aug_rhs(std::declval<const ModelParams &>(), 0.0, 0, (const double *)nullptr)),
                                      decltype(M::dhamiltonian(std::declval<const ModelParams &>(), 0.0, (const double *)nullptr,
                                                               std::declval<double (&)[M::S + 1]>()))>> : std::true_type {};

template <class Mdl>
ModelLaunchers table(int nparams, int step_nbr, std::initializer_list<double> defaults)
{
    static_assert(Mdl::S == 2 * Mdl::D, "state vector is [state ; costate]");
    static_assert(Mdl::NU >= 1 && Mdl::NU <= 3, "control dimension 1..3");
    ModelLaunchers t{};
    t.abi = kPluginAbi; t.dim = Mdl::D; t.control_dim = Mdl::NU; t.nparams = nparams; t.default_step_nbr = step_nbr;
    int i = 0;
    for (double v : defaults) if (i < kMaxParams) t.default_params[i++] = v;
    t.traj = &traj<Mdl>; t.residual = &residual<Mdl>; t.fdjac = &fdjac<Mdl>; t.fdrows = &fdrows<Mdl>;
    t.dense = &dense<Mdl>; t.eval = &eval<Mdl>;
    if constexpr (has_variational<Mdl>::value) {
        t.var_traj = &varimpl::traj<Mdl>; t.var_jacobian = &varimpl::jacobian<Mdl>; t.var_eval = &varimpl::eval<Mdl>;
    }
    return t;
}

}  // namespace plugin
}  // namespace socp

// defines the entry point socp_plugin_load() looks for
#define SOCP_DEFINE_MODEL_PLUGIN(MODEL_ID, MDL, NPARAMS, STEP_NBR, ...)                                   \
    extern "C" int socp_plugin_register(void)                                                             \
    {                                                                                                     \
        static const socp::ModelLaunchers t = socp::plugin::table<MDL>(NPARAMS, STEP_NBR, __VA_ARGS__);   \
        return socp_register_model(MODEL_ID, &t, (int)sizeof(t));                                         \
    }
