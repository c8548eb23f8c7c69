// launch_impl.hpp -- body of the launchers; included once per flavour with
// SOCP_FLAVOUR, SOCP_GODDARD and SOCP_DINT defined by the including .hip file.
#include <cstdlib>
#include "integrator.hpp"
#include "launch.hpp"

namespace socp {

#define SOCP_CAT_(a, b) a##b
#define SOCP_CAT(a, b) SOCP_CAT_(a, b)

// SOCP_HAVE_DOPRI5: this translation unit also carries the adaptive-integrator instantiations (one wave per
// SIMD: seven stage vectors live in registers)
// PP: the PERPROB template argument (true: per-problem parameter / boundary blocks; always false for the trajectory kernel)
#ifdef SOCP_HAVE_DOPRI5
#define SOCP_LAUNCH_ADAPTIVE(KERNEL, MDL, PP, GRID, LDS, ST, ...) \
    if (P.integrator == 1) { hipLaunchKernelGGL((KERNEL<MDL, 1, 1, PP>), dim3(GRID), dim3(64), LDS, ST, __VA_ARGS__); break; }
#else
#define SOCP_LAUNCH_ADAPTIVE(KERNEL, MDL, PP, GRID, LDS, ST, ...)
#endif

#define SOCP_LAUNCH_MDL(KERNEL, MDL, PP, WAVES, GRID, LDS, ST, ...)                                              \
    do {                                                                                                           \
        SOCP_LAUNCH_ADAPTIVE(KERNEL, MDL, PP, GRID, LDS, ST, __VA_ARGS__)                                       \
        switch (wpe_for(WAVES)) {                                                                                  \
        case 1: hipLaunchKernelGGL((KERNEL<MDL, 1, 0, PP>), dim3(GRID), dim3(64), LDS, ST, __VA_ARGS__); break;   \
        case 2: hipLaunchKernelGGL((KERNEL<MDL, 2, 0, PP>), dim3(GRID), dim3(64), LDS, ST, __VA_ARGS__); break;   \
        default: hipLaunchKernelGGL((KERNEL<MDL, 3, 0, PP>), dim3(GRID), dim3(64), LDS, ST, __VA_ARGS__); break;  \
        }                                                                                                          \
    } while (0)

// model_id 1 = Goddard (smooth-law specialisation when mu2 > 0, parameter slot 6), 2 = double integrator
// With per-problem parameter blocks the control law may differ between problems of one launch: the Goddard smooth-law
// specialisation is then taken only when the caller vouches for every block (ProblemDev::pp_smooth).
#define SOCP_DISPATCH_MODELS(KERNEL, PP, SMOOTH_OK, GRID, LDS, ST, ...)                                  \
    do {                                                                                                \
        if (model_id == 1 && (SMOOTH_OK)) SOCP_LAUNCH_MDL(KERNEL, SOCP_GODDARD_SMOOTH, PP, GRID, GRID, LDS, ST, __VA_ARGS__); \
        else if (model_id == 1) SOCP_LAUNCH_MDL(KERNEL, SOCP_GODDARD, PP, GRID, GRID, LDS, ST, __VA_ARGS__); \
        else if (model_id == 3) SOCP_LAUNCH_MDL(KERNEL, SOCP_COVID, PP, GRID, GRID, LDS, ST, __VA_ARGS__);  \
        else SOCP_LAUNCH_MDL(KERNEL, SOCP_DINT, PP, GRID, GRID, LDS, ST, __VA_ARGS__);                      \
    } while (0)
// trajectory kernel (no shooting problem)
#define SOCP_DISPATCH_HOT(KERNEL, GRID, ST, ...) SOCP_DISPATCH_MODELS(KERNEL, false, P.p[6] > 0, GRID, 0, ST, __VA_ARGS__)
// kernels that read a shooting problem `pb`: per-problem blocks select the PERPROB instantiation
#define SOCP_DISPATCH_PB_LDS(KERNEL, GRID, LDS, ST, ...)                                                \
    do {                                                                                                \
        if (pb.pp_params || pb.pp_time || pb.pp_xnode)                                                  \
            SOCP_DISPATCH_MODELS(KERNEL, true, pb.pp_params ? pb.pp_smooth != 0 : P.p[6] > 0, GRID, LDS, ST, __VA_ARGS__); \
        else                                                                                            \
            SOCP_DISPATCH_MODELS(KERNEL, false, P.p[6] > 0, GRID, LDS, ST, __VA_ARGS__);                \
    } while (0)
#define SOCP_DISPATCH_PB(KERNEL, GRID, ST, ...) SOCP_DISPATCH_PB_LDS(KERNEL, GRID, 0, ST, __VA_ARGS__)

#define SOCP_DISPATCH(KERNEL, GRID, ST, ...)                                                            \
    do {                                                                                                \
        if (model_id == 1 && P.p[6] > 0)                                                                \
            hipLaunchKernelGGL(KERNEL<SOCP_GODDARD_SMOOTH>, dim3(GRID), dim3(64), 0, ST, __VA_ARGS__);  \
        else if (model_id == 1)                                                                         \
            hipLaunchKernelGGL(KERNEL<SOCP_GODDARD>, dim3(GRID), dim3(64), 0, ST, __VA_ARGS__);         \
        else if (model_id == 3)                                                                         \
            hipLaunchKernelGGL(KERNEL<SOCP_COVID>, dim3(GRID), dim3(64), 0, ST, __VA_ARGS__);           \
        else                                                                                            \
            hipLaunchKernelGGL(KERNEL<SOCP_DINT>, dim3(GRID), dim3(64), 0, ST, __VA_ARGS__);            \
    } while (0)

hipError_t SOCP_CAT(traj_, SOCP_FLAVOUR)(int model_id, hipStream_t st, const ModelParams &P, int B,
                                         const double *t0, const double *tf, const double *sw,
                                         const double *X0, double *Xf)
{
    if (B <= 0) return hipSuccess;
    SOCP_DISPATCH_HOT(traj_lane_kernel, blocks_for(B), st, P, B, t0, tf, sw, X0, Xf);
    return hipGetLastError();
}

hipError_t SOCP_CAT(residual_, SOCP_FLAVOUR)(int model_id, hipStream_t st, const ModelParams &P,
                                             const ProblemDev &pb, int B, const double *Z, double *F)
{
    if (B <= 0) return hipSuccess;
    const int R = rows_per_block(pb.M, pb.n);
    const unsigned grid = R ? (unsigned)((B + R - 1) / R) : blocks_for((long)B * pb.M);
    SOCP_DISPATCH_PB_LDS(residual_lane_kernel, grid, (unsigned)((long)R * pb.n * 8), st, P, pb, B, Z, F, R);
    return hipGetLastError();
}

hipError_t SOCP_CAT(fdjac_, SOCP_FLAVOUR)(int model_id, hipStream_t st, const ModelParams &P,
                                          const ProblemDev &pb, int np, int T, const int2 *pairs, const double *z,
                                          const double *fvec, double eps, double *fjac)
{
    if (T <= 0 || np <= 0) return hipSuccess;
    const long total = (long)np * T;
    SOCP_DISPATCH_PB(fdjac_lane_kernel, blocks_for(total), st, P, pb, np, T, pairs, z, fvec, eps, fjac);
    return hipGetLastError();
}

hipError_t SOCP_CAT(fdrows_, SOCP_FLAVOUR)(int model_id, hipStream_t st, const ModelParams &P,
                                           const ProblemDev &pb, int np, const double *z, double eps, double *rows)
{
    if (np <= 0) return hipSuccess;
    const long vrows = (long)np * (pb.n + 1);
    const int R = rows_per_block(pb.M, pb.n);
    const unsigned grid = R ? (unsigned)((vrows + R - 1) / R) : blocks_for(vrows * pb.M);
    SOCP_DISPATCH_PB_LDS(fdrows_lane_kernel, grid, (unsigned)((long)R * pb.n * 8), st, P, pb, np, z, eps, rows, R);
    return hipGetLastError();
}

#ifdef SOCP_DEFINE_COMMON
hipError_t fd_diff(hipStream_t st, int n, int np, const double *z, double eps, const double *rows, double *fjac)
{
    if (np <= 0) return hipSuccess;
    const long total = (long)np * n * n;
    hipLaunchKernelGGL(fd_diff_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, n, np, z, eps, rows, fjac);
    return hipGetLastError();
}
#endif

hipError_t SOCP_CAT(dense_, SOCP_FLAVOUR)(int model_id, hipStream_t st, const ModelParams &P, double t0, double tf,
                                          double sw0, double sw1, const double *X0, double *dense, double *times,
                                          int cap, int *rows, double *aux)
{
#ifdef SOCP_HAVE_DOPRI5
#define SOCP_DENSE_LAUNCH(MDL)                                                                                                              \
    do {                                                                                                                                    \
        if (P.integrator == 1) hipLaunchKernelGGL((traj_dense_kernel<MDL, 1>), dim3(1), dim3(64), 0, st, P, t0, tf, sw0, sw1, X0, dense, times, cap, rows, aux); \
        else hipLaunchKernelGGL((traj_dense_kernel<MDL, 0>), dim3(1), dim3(64), 0, st, P, t0, tf, sw0, sw1, X0, dense, times, cap, rows, aux);               \
    } while (0)
#else
#define SOCP_DENSE_LAUNCH(MDL)                                                                                                              \
    do {                                                                                                                                    \
        if (P.integrator == 1) return hipErrorInvalidValue;      /* the adaptive instantiations live in the reference-order translation unit */ \
        hipLaunchKernelGGL((traj_dense_kernel<MDL, 0>), dim3(1), dim3(64), 0, st, P, t0, tf, sw0, sw1, X0, dense, times, cap, rows, aux);      \
    } while (0)
#endif
    if (model_id == 1) SOCP_DENSE_LAUNCH(SOCP_GODDARD);
    else if (model_id == 3) SOCP_DENSE_LAUNCH(SOCP_COVID);
    else SOCP_DENSE_LAUNCH(SOCP_DINT);
#undef SOCP_DENSE_LAUNCH
    return hipGetLastError();
}

hipError_t SOCP_CAT(eval_, SOCP_FLAVOUR)(int model_id, hipStream_t st, const ModelParams &P, int what, int B,
                                         const double *t, const double *sw, const double *X, double *out)
{
    if (B <= 0) return hipSuccess;
    SOCP_DISPATCH(eval_lane_kernel, blocks_for(B), st, P, what, B, t, sw, X, out);
    return hipGetLastError();
}

}  // namespace socp
