// Probe: what does the FIRST device access to a fresh multi-GB hipMalloc cost?  (the device engine's first round showed 7 ms of
// stream time before its first kernel ran, with a 1.7 GB arena allocated 0.05 ms earlier)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(double *p, long stride, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[(long)i * stride] = 1.0; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double *warm;
    hipMalloc(&warm, 1 << 20);
    touch<<<1, 64, 0, st>>>(warm, 1, 64);
    hipStreamSynchronize(st);
    for (double gb : {0.25, 1.0, 1.7, 3.2}) {
        const size_t bytes = (size_t)(gb * (1ull << 30));
        double *p = nullptr;
        double t0 = now();
        hipMalloc(&p, bytes);
        double t1 = now();
        touch<<<1, 64, 0, st>>>(p, 1, 64);                         // one cache line
        hipStreamSynchronize(st);
        double t2 = now();
        touch<<<4096, 64, 0, st>>>(p, (long)(bytes / 8 / (4096 * 64)), 4096 * 64);   // spread over the whole allocation
        hipStreamSynchronize(st);
        double t3 = now();
        touch<<<4096, 64, 0, st>>>(p, (long)(bytes / 8 / (4096 * 64)), 4096 * 64);
        hipStreamSynchronize(st);
        double t4 = now();
        hipFree(p);
        double t5 = now();
        printf("%.2f GB: hipMalloc %.2f ms, first kernel (one line) %.2f ms, first spread access %.2f ms, second %.2f ms, hipFree %.2f ms\n", gb, t1 - t0, t2 - t1, t3 - t2,
               t4 - t3, t5 - t4);
    }
    return 0;
}
