// Probe: does a wavefront with 16 (or 1) active lanes issue FP64 VALU instructions faster than one with 64?
// One wave, ILP-8 FMA stream (issue-bound) and an ILP-1 chain (latency-bound); prints cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int ILP>
__global__ void fma_stream(double *out, int active, int iters, double a, double b)
{
    if ((int)threadIdx.x >= active) return;
    double x[ILP];
    for (int k = 0; k < ILP; ++k) x[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int k = 0; k < ILP; ++k) x[k] = __builtin_fma(x[k], a, b);
        }
    }
    double s = 0;
    for (int k = 0; k < ILP; ++k) s += x[k];
    out[threadIdx.x] = s;
}

template <int ILP>
void run(const char *name, double clock_ghz)
{
    double *out;
    hipMalloc(&out, 64 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 200000;
    const int actives[] = {64, 48, 32, 16, 1};
    for (int active : actives) {
        fma_stream<ILP><<<1, 64>>>(out, active, 1000, 0.999999, 1e-7);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        fma_stream<ILP><<<1, 64>>>(out, active, iters, 0.999999, 1e-7);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * 16 * ILP;
        printf("%s active=%2d  %.3f ms  %.3f ns/instr  %.2f cycles/instr at %.2f GHz\n", name, active, ms, ms * 1e6 / instr, ms * 1e6 / instr * clock_ghz,
               clock_ghz);
    }
    hipFree(out);
}

int main()
{
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    run<8>("ilp8", ghz);
    run<1>("ilp1", ghz);
    run<2>("ilp2", ghz);
    return 0;
}
