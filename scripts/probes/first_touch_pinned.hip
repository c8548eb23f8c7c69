// Probe: first device access (H2D copy / kernel read) to freshly pinned host memory vs the second one
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
__global__ void rd(const double *p, double *out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = p[i]; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double *d;
    hipMalloc(&d, 64 << 20);
    hipMemsetAsync(d, 0, 64 << 20, st);
    hipStreamSynchronize(st);
    for (int rep = 0; rep < 2; rep++)
        for (size_t mb : {1, 4, 13, 64}) {
            const size_t bytes = mb << 20;
            double *h = nullptr;
            double t0 = now();
            hipHostMalloc(&h, bytes, hipHostMallocDefault);
            double t1 = now();
            memset(h, 1, bytes);
            double t2 = now();
            hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
            double t3 = now();
            hipStreamSynchronize(st);
            double t4 = now();
            hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            double t5 = now();
            rd<<<(int)(bytes / 8 / 256), 256, 0, st>>>(h, d, (int)(bytes / 8));
            hipStreamSynchronize(st);
            double t6 = now();
            hipHostFree(h);
            printf("%3zu MB pinned: hipHostMalloc %.2f ms, cpu fill %.2f, first H2D: call %.2f + wait %.2f ms, second H2D %.2f ms, kernel read %.2f ms\n", mb, t1 - t0, t2 - t1,
                   t3 - t2, t4 - t3, t5 - t4, t6 - t5);
        }
    return 0;
}
