// probe_reduce16.hip -- development probe (on the GPU box):  hipcc --offload-arch=gfx950 -O3 probe_reduce16.hip -o /tmp/pr16 && /tmp/pr16
// Pins the lane semantics of gfx950's v_permlane32_swap / v_permlane16_swap and of the row_mirror / row_half_mirror DPP controls
// as kernels_factor_fast.hip: reduce16 uses them: 16 values per lane -> the wave-wide sum of value (lane >> 2) in every lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

#define SOCP_REDUCE16_ONLY
#include "../../socp_amd/csrc/wave_reduce.hpp"

__global__ void k(const double *in, double *out)
{
    double x[16];
    for (int j = 0; j < 16; j++) x[j] = in[threadIdx.x * 16 + j];
    out[threadIdx.x] = socp::devsolver::reduce16(x, threadIdx.x);
}

int main()
{
    double h[64 * 16], *d, *o, r[64];
    for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) h[l * 16 + j] = std::sin(1.0 + l * 0.37 + j * 1.91) * (1 + j);
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        double want = 0;
        for (int s = 0; s < 64; s++) want += h[s * 16 + (l >> 2)];
        if (std::fabs(r[l] - want) > 1e-12 * (1 + std::fabs(want))) { bad++; if (bad < 8) printf("lane %d: got %.15g want %.15g\n", l, r[l], want); }
    }
    printf("reduce16: %s (%d lanes off)\n", bad ? "MISMATCH" : "ok", bad);
    return bad != 0;
}
