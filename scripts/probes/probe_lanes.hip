// probe: FP64 issue time of ONE wave as a function of which lanes are active (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, int active, int iters)
{
    const int lane = threadIdx.x;
    if (lane >= active) return;
    double a0 = 1.0 + lane * 1e-3, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
    const double m = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; i++) {      // 8 independent FMA chains: issue-bound, not latency-bound
        a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
        a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
    }
    out[lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
    double *d; (void)hipMalloc(&d, 64 * 8);
    const int iters = 2000000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int active : {64, 48, 32, 16, 8, 1}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, active, 1000);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, active, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("active lanes %2d: %.3f ms  -> %.2f ns per FP64 FMA wave-instruction\n", active, ms, ms * 1e6 / (8.0 * iters));
    }
    return 0;
}
