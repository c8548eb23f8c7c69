// Probe: the one-time cost of the first pinned-memory copy of a process, by size and direction
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t first = argc > 1 ? (size_t)atol(argv[1]) : 4096;      // bytes of the first copy
    const int d2h = argc > 2 ? atoi(argv[2]) : 0;
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double *d, *h;
    hipMalloc(&d, 64 << 20);
    hipHostMalloc(&h, 64 << 20, hipHostMallocDefault);
    hipMemsetAsync(d, 0, 64 << 20, st);
    hipStreamSynchronize(st);
    const size_t sizes[] = {first, 64, 4096, 65536, 1 << 20, 4 << 20};
    for (size_t b : sizes) {
        double t0 = now();
        if (d2h) hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st); else hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, st);
        double t1 = now();
        hipStreamSynchronize(st);
        printf("%s %8zu B: call %.3f ms + wait %.3f ms\n", d2h ? "D2H" : "H2D", b, t1 - t0, now() - t1);
    }
    return 0;
}
