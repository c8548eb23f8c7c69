// probe: ONE Goddard trajectory spread over lanes (north_star's "one trajectory per wavefront ... shuffles for the reductions")
// against the product's one-lane-per-trajectory kernel, same arithmetic (the throughput flavour's restructured right-hand
// side, models_fast.hpp), same RK4 form, 1e4 steps.  Layout of the spread form: a quad of lanes per trajectory, lanes 0..2
// carry the x / y / z components of r, v, p_r, p_v, every lane carries the two scalars (mass, its costate); the five 3-term
// dot products are quad reductions (DPP quad_perm on the two 32-bit halves + add), everything component-wise is one
// instruction for all three components.  16 trajectories per wave instead of 64.  Prints the latency of 15 trajectories (the
// single-problem FD batch) and of 1152 (the 128-unknown Jacobian) in both forms, and the largest relative difference.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -I socp_amd/csrc -I include scripts/probes/probe_wave_traj.hip -o probe_wave_traj
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "integrator.hpp"
#include "models_fast.hpp"

using namespace socp;

template <int CTRL>
__device__ __forceinline__ double quad_swizzle(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sum of p over the four lanes of a quad, in every lane (lane 3 holds 0)
__device__ __forceinline__ double quad_sum(double p)
{
    p += quad_swizzle<0xB1>(p);          // quad_perm:[1,0,3,2]
    p += quad_swizzle<0x4E>(p);          // quad_perm:[2,3,0,1]
    return p;
}

struct QuadState { double r, v, pr, pv, m, pm; };

// GoddardFastT<true>::rhs (models_fast.hpp), component-parallel
__device__ __forceinline__ void rhs_quad(const ModelParams &P, const QuadState &X, QuadState &d)
{
    const double b = P.p[GP_B], C = P.p[GP_C], KD = P.p[GP_KD], kr = P.p[GP_KR];
    const double r2 = quad_sum(X.r * X.r), v2 = quad_sum(X.v * X.v), q2 = quad_sum(X.pv * X.pv);
    const double ir = fast_rsqrt(r2), iv = fast_rsqrt(v2), iq = fast_rsqrt(q2);
    const double r = r2 * ir, v = v2 * iv, norm_pv = q2 * iq;
    const double im = fast_rcp(X.m);
    const double pvdotv = quad_sum(X.pv * X.v), pvdotr = quad_sum(X.pv * X.r);
    const double E = fast_exp(-kr * (r - 1));
    const double ir2 = ir * ir, ir3 = ir2 * ir;
    const double Cm = C * im;
    const double Switch = P.p[GP_MU1] - b * X.pm - Cm * norm_pv;
    const double alpha = __builtin_fmax(-Switch * (0.5 / P.p[GP_MU2]), 0.0);
    const double norm_u = __builtin_fmin(alpha, P.p[GP_UMAX]);
    const double ua = -norm_u * iq, pvdotu = -norm_u * norm_pv;
    const double Dm = KD * E * im, Dv = Dm * v, Tm = Cm * ua;
    d.r = X.v;
    d.v = Tm * X.pv - Dv * X.v - ir3 * X.r;
    d.m = -b * norm_u;
    const double W = -(kr * Dv * pvdotv * ir) - 3.0 * ir3 * ir2 * pvdotr;
    d.pr = W * X.r + ir3 * X.pv;
    const double DG = Dm * (pvdotv * iv);
    d.pv = DG * X.v + (Dv * X.pv - X.pr);
    d.pm = im * (Cm * pvdotu - Dv * pvdotv);
}

#define QOP(dst, expr) do { dst.r = expr(r); dst.v = expr(v); dst.pr = expr(pr); dst.pv = expr(pv); dst.m = expr(m); dst.pm = expr(pm); } while (0)

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void traj_quad_kernel(ModelParams P, int T, double t0, double tf, const double *__restrict__ X0, double *__restrict__ Xf)
{
    const int gl = blockIdx.x * 64 + threadIdx.x;
    const int traj = gl >> 2, c = gl & 3;                 // component lane: 0..2 = x, y, z; 3 = padding (zeros)
    if (traj >= T) return;                                // whole quads leave together
    const double *x0 = X0 + (size_t)traj * 14;
    QuadState X;
    X.r = c < 3 ? x0[c] : 0.0; X.v = c < 3 ? x0[3 + c] : 0.0; X.pr = c < 3 ? x0[7 + c] : 0.0; X.pv = c < 3 ? x0[10 + c] : 0.0;
    X.m = x0[6]; X.pm = x0[13];
    const double dt = (tf - t0) / P.step_nbr;
    const double h2 = 0.5 * dt, h6 = dt * (1.0 / 6.0);
    for (int s = 0; s < P.step_nbr; s++) {                // Lane::rk4, throughput form (acc = F1 + 2 F2 + 2 F3 + F4)
        QuadState A, F, Y;
        rhs_quad(P, X, A);
#define E1(f) X.f + h2 * A.f
        QOP(Y, E1);
        rhs_quad(P, Y, F);
#define E2(f) X.f + h2 * F.f
#define E3(f) A.f + 2.0 * F.f
        QOP(Y, E2); QOP(A, E3);
        rhs_quad(P, Y, F);
#define E4(f) X.f + dt * F.f
        QOP(Y, E4); QOP(A, E3);
        rhs_quad(P, Y, F);
#define E5(f) X.f + h6 * (A.f + F.f)
        QOP(X, E5);
    }
    double *xf = Xf + (size_t)traj * 14;
    if (c < 3) { xf[c] = X.r; xf[3 + c] = X.v; xf[7 + c] = X.pr; xf[10 + c] = X.pv; }
    if (c == 0) { xf[6] = X.m; xf[13] = X.pm; }
}

// the product's form: one lane per trajectory, the library's own RK4 driver and right-hand side
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void traj_lane_probe(ModelParams P, int T, double t0, double tf, const double *__restrict__ X0, double *__restrict__ Xf)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= T) return;
    double X[14];
#pragma unroll
    for (int k = 0; k < 14; k++) X[k] = X0[(size_t)b * 14 + k];
    const double dt = (tf - t0) / P.step_nbr;
    double t = t0;
    for (int s = 0; s < P.step_nbr; s++) { Lane<GoddardFastSmooth>::rk4(P, 0.0, 0.0, t, X, dt); t += dt; }
#pragma unroll
    for (int k = 0; k < 14; k++) Xf[(size_t)b * 14 + k] = X[k];
}

int main()
{
    ModelParams P{};
    const double prm[8] = {3.5, 7.0, 310.0, 500.0, 1.0, 1.0, 1.0, -1.0};
    for (int k = 0; k < 8; k++) P.p[k] = prm[k];
    P.step_nbr = 10000;
    const double x0[14] = {0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0, -8.121947733, 7.775439382e-3, 0.7775438809, -0.4779369965, 5.715013318e-4, 5.715009222e-2, 9.958404873e-2};
    const int Tmax = 1152;
    std::vector<double> X0((size_t)Tmax * 14), A(X0.size()), B(X0.size());
    srand48(3);
    for (int t = 0; t < Tmax; t++)
        for (int k = 0; k < 14; k++) X0[(size_t)t * 14 + k] = x0[k] * (k >= 7 ? 1.0 + 1e-3 * (2 * drand48() - 1) : 1.0);
    double *dX0, *dA, *dB;
    (void)hipMalloc(&dX0, X0.size() * 8); (void)hipMalloc(&dA, X0.size() * 8); (void)hipMalloc(&dB, X0.size() * 8);
    (void)hipMemcpy(dX0, X0.data(), X0.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("{");
    for (int T : {15, 1152}) {
        float ms_lane = 1e9f, ms_quad = 1e9f, ms;
        for (int rep = 0; rep < 6; rep++) {             // minimum over repetitions (the first launches run at ramping clocks)
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(traj_lane_probe, dim3((T + 63) / 64), dim3(64), 0, 0, P, T, 0.0, 0.2640825, dX0, dA);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < ms_lane) ms_lane = ms;
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(traj_quad_kernel, dim3((4 * T + 63) / 64), dim3(64), 0, 0, P, T, 0.0, 0.2640825, dX0, dB);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < ms_quad) ms_quad = ms;
        }
        (void)hipMemcpy(A.data(), dA, (size_t)T * 14 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(B.data(), dB, (size_t)T * 14 * 8, hipMemcpyDeviceToHost);
        double err = 0;
        for (size_t i = 0; i < (size_t)T * 14; i++) err = fmax(err, fabs(A[i] - B[i]) / fmax(1.0, fabs(A[i])));
        printf("\"T%d\": {\"lane_per_trajectory_ms\": %.3f, \"quad_of_lanes_per_trajectory_ms\": %.3f, \"waves_lane\": %d, \"waves_quad\": %d, \"max_rel_diff\": %.2e}%s",
               T, ms_lane, ms_quad, (T + 63) / 64, (4 * T + 63) / 64, err, T == 15 ? ", " : "");
    }
    printf("}\n");
    return 0;
}
