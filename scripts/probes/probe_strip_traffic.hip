// probe: what the memory system delivers for the trailing pass's access pattern (gfx950), no arithmetic.
// 2048 "problems" of n = 253 rows x ld = 256 doubles at the solver's workspace stride; every 16-column strip right of column 32 is read
// (253 x 128-byte pieces, one per matrix row) and written back from row 32 on, as qrfac_trail_kernel does for pair 0.
//   pattern 0: a wavefront per STRIP (four wavefronts per problem take strips 16 w, + 64 ...): the product's pattern
//   pattern 1: a wavefront per ROW BLOCK: wave w takes rows 64 w .. 64 w + 63 of FOUR adjacent strips (512 contiguous bytes per row)
//   pattern 2: strips stored contiguously (a tiled layout): a wavefront reads / writes 32 KB runs
// occupancy as the product's: 256-thread workgroups, 75 KB of LDS each (two per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int N = 253, LD = 256;
template <int PATTERN>
__global__ __launch_bounds__(256) void k(double *ws, long stride, int count)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, m = lane & 15;
    for (int b = blockIdx.x; b < count; b += gridDim.x) {
        double *A = ws + (long)b * stride;
        if (PATTERN == 0) {
            for (int c0 = 32 + 16 * wave; c0 <= N; c0 += 64) {
                double S[64];
                const bool cok = c0 + m <= N;
#pragma unroll
                for (int q = 0; q < 64; q++) { const int row = 16 * (q >> 2) + g + 4 * (q & 3); S[q] = (cok && row < N) ? A[(long)row * LD + c0 + m] : 0.0; }
#pragma unroll
                for (int q = 8; q < 64; q++) { const int row = 16 * (q >> 2) + g + 4 * (q & 3); if (cok && row < N) A[(long)row * LD + c0 + m] = S[q] + 1.0; }
            }
        } else if (PATTERN == 1) {
            for (int c0 = 32; c0 <= N; c0 += 64) {
                double S[64];
#pragma unroll
                for (int q = 0; q < 64; q++) { const int row = 64 * wave + 4 * (q >> 2) + g, col = c0 + 16 * (q & 3) + m; S[q] = (col <= N && row < N) ? A[(long)row * LD + col] : 0.0; }
#pragma unroll
                for (int q = 0; q < 64; q++) { const int row = 64 * wave + 4 * (q >> 2) + g, col = c0 + 16 * (q & 3) + m; if (col <= N && row < N && row >= 32) A[(long)row * LD + col] = S[q] + 1.0; }
            }
        } else {
            for (int s = 2 + wave; 16 * s <= N; s += 4) {
                double S[64];
                double *T = A + (long)s * (N * 16);                         // strip s stored contiguously: [row][16]
#pragma unroll
                for (int q = 0; q < 64; q++) { const int row = 4 * q + g; S[q] = row < N ? T[row * 16 + m] : 0.0; }
#pragma unroll
                for (int q = 8; q < 64; q++) { const int row = 4 * q + g; if (row < N) T[row * 16 + m] = S[q] + 1.0; }
            }
        }
    }
    if (lds[0] == 12345.678) ws[0] = 0;
}
int main()
{
    const int count = 2048;
    const long stride = ((long)N * LD + (long)N * (N + 1) + 16L * N + 7) / 8 * 8;
    double *ws; if (hipMalloc(&ws, sizeof(double) * stride * count) != hipSuccess) return 1;
    (void)hipMemset(ws, 0, sizeof(double) * stride * count);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double bytes = (double)count * (14.0 * 16 * N * 8 + 14.0 * 16 * (N - 32) * 8);     // 14 strips read whole, written from row 32 on
    auto run = [&](auto kern, const char *name) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        for (int rep = 0; rep < 4; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(count), dim3(256), 75 * 1024, 0, ws, stride, count);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.3f ms  %.2f TB/s (read + write, %.2f GB)\n", name, ms, bytes / (ms * 1e-3) / 1e12, bytes / 1e9);
        }
    };
    run(k<0>, "pattern 0 (wavefront per strip: 128-byte pieces, 2 KB apart)");
    run(k<1>, "pattern 1 (wavefront per 64 rows of four strips: 512-byte pieces)");
    run(k<2>, "pattern 2 (strips stored contiguously: 32 KB runs)");
    return 0;
}
