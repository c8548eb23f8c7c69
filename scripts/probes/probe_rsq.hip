// probe: relative error of v_rsq_f64 / v_rcp_f64 estimates and of 1 / 2 Newton refinements (gfx950)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y0 = __builtin_amdgcn_rsq(v);
    double h = 0.5 * v;
    double e = __builtin_fma(-h * y0, y0, 0.5);
    double y1 = __builtin_fma(y0, e, y0);
    e = __builtin_fma(-h * y1, y1, 0.5);
    double y2 = __builtin_fma(y1, e, y1);
    double r0 = __builtin_amdgcn_rcp(v);
    double f = __builtin_fma(-v, r0, 1.0);
    double r1 = __builtin_fma(r0, f, r0);
    f = __builtin_fma(-v, r1, 1.0);
    double r2 = __builtin_fma(r1, f, r1);
    o[6 * i] = y0; o[6 * i + 1] = y1; o[6 * i + 2] = y2; o[6 * i + 3] = r0; o[6 * i + 4] = r1; o[6 * i + 5] = r2;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(6 * n);
    for (int i = 0; i < n; i++) x[i] = 0.25 + 7.75 * (i + 0.5) / n;
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        long double rs = 1.0L / sqrtl((long double)x[i]), rc = 1.0L / (long double)x[i];
        for (int j = 0; j < 3; j++) m[j] = fmax(m[j], (double)fabsl((o[6 * i + j] - rs) / rs));
        for (int j = 3; j < 6; j++) m[j] = fmax(m[j], (double)fabsl((o[6 * i + j] - rc) / rc));
    }
    printf("rsq: raw %.3e  newton1 %.3e  newton2 %.3e\nrcp: raw %.3e  newton1 %.3e  newton2 %.3e\n", m[0], m[1], m[2], m[3], m[4], m[5]);
    return 0;
}
