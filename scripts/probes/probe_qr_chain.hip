// probe: the DEPENDENCY CHAIN of MINPACK's column Householder QR on one gfx950 wavefront ("why not on
// the device").  Keeping MINPACK's per-column operation order means: the norm of column j (one serial accumulation), then the
// dot product of reflector j with column j+1 (one serial accumulation) before column j+1 can become reflector j+1.  Whatever
// the other 1023 SIMDs do for the remaining columns, this chain runs on one lane of one wave.  Same for qform: column n-1 of
// Q goes through reflectors n-1 .. 0, one serial dot product each.  The probe runs exactly those chains for n = 832 with the
// vectors in LDS and the element-parallel parts (scaling, axpy) spread over the 64 lanes, checks the result against the same
// operations on the host bit for bit, and prints the time: a LOWER BOUND for any order-preserving device factorisation.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off probe_qr_chain.hip -o probe_qr_chain && ./probe_qr_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NMAX 1024

// MINPACK enorm, the mid-range branch only is taken for this data (|x| in (3.8e-20, 1.3e19/n)): s2 += x*x in order
__host__ __device__ inline double enorm_mid(int n, const double *x)
{
    double s2 = 0;
    for (int i = 0; i < n; i++) s2 += x[i] * x[i];
    return sqrt(s2);
}

__global__ __launch_bounds__(64) void qrfac_chain(int n, const double *__restrict__ A, double *__restrict__ rdiag, double *__restrict__ last)
{
    __shared__ double v[NMAX], w[NMAX];
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) v[i] = A[i];                       // column 0
    __syncthreads();
    for (int j = 0; j < n - 1; j++) {
        const int len = n - j;
        __shared__ double s_norm, s_temp;
        if (lane == 0) {
            double a = enorm_mid(len, v);
            if (v[0] < 0) a = -a;
            s_norm = a;
        }
        __syncthreads();
        const double ajnorm = s_norm;
        for (int i = lane; i < len; i += 64) v[i] /= ajnorm;
        __syncthreads();
        if (lane == 0) { v[0] += 1; rdiag[j] = -ajnorm; }
        for (int i = lane; i < len; i += 64) w[i] = A[(size_t)(j + 1) * n + j + i];   // column j+1, rows j .. n-1
        __syncthreads();
        if (lane == 0) {
            double sum = 0;
            for (int i = 0; i < len; i++) sum += v[i] * w[i];
            s_temp = sum / v[0];
        }
        __syncthreads();
        const double temp = s_temp;
        for (int i = lane; i < len; i += 64) w[i] -= temp * v[i];
        __syncthreads();
        for (int i = lane; i < len - 1; i += 64) v[i] = w[i + 1];          // rows j+1 .. n-1 of column j+1: the next reflector
        __syncthreads();
    }
    if (lane == 0) last[0] = v[0];
}

// column n-1 of Q through reflectors n-1 .. 0 (packed vectors V, v_k at off[k], length n-k)
__global__ __launch_bounds__(64) void qform_chain(int n, const double *__restrict__ V, const long *__restrict__ off, double *__restrict__ qcol)
{
    __shared__ double q[NMAX], v[NMAX];
    __shared__ double s_temp;
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) q[i] = 0;
    __syncthreads();
    if (lane == 0) q[n - 1] = 1;
    for (int k = n - 1; k >= 0; k--) {
        const int len = n - k;
        for (int i = lane; i < len; i += 64) v[i] = V[off[k] + i];
        __syncthreads();
        if (lane == 0) {
            double sum = 0;
            for (int i = 0; i < len; i++) sum += q[k + i] * v[i];
            s_temp = sum / v[0];
        }
        __syncthreads();
        const double temp = s_temp;
        for (int i = lane; i < len; i += 64) q[k + i] -= temp * v[i];
        __syncthreads();
    }
    for (int i = lane; i < n; i += 64) qcol[i] = q[i];
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 832;
    std::vector<double> A((size_t)n * n);
    srand48(1);
    for (auto &x : A) x = drand48() - 0.5;
    // host twin of qrfac_chain
    std::vector<double> v(A.begin(), A.begin() + n), w(n), rd(n, 0.0);
    for (int j = 0; j < n - 1; j++) {
        const int len = n - j;
        double a = enorm_mid(len, v.data());
        if (v[0] < 0) a = -a;
        for (int i = 0; i < len; i++) v[i] /= a;
        v[0] += 1; rd[j] = -a;
        for (int i = 0; i < len; i++) w[i] = A[(size_t)(j + 1) * n + j + i];
        double sum = 0;
        for (int i = 0; i < len; i++) sum += v[i] * w[i];
        const double temp = sum / v[0];
        for (int i = 0; i < len; i++) w[i] -= temp * v[i];
        for (int i = 0; i < len - 1; i++) v[i] = w[i + 1];
    }
    // packed "reflectors" for the qform chain: any vectors with non-zero pivot do
    std::vector<long> off(n);
    long total = 0;
    for (int k = 0; k < n; k++) { off[k] = total; total += n - k; }
    std::vector<double> V(total);
    for (auto &x : V) x = drand48() - 0.5;
    for (int k = 0; k < n; k++) V[off[k]] = 1.0 + drand48();
    std::vector<double> q(n, 0.0);
    q[n - 1] = 1;
    for (int k = n - 1; k >= 0; k--) {
        double sum = 0;
        for (int i = 0; i < n - k; i++) sum += q[k + i] * V[off[k] + i];
        const double temp = sum / V[off[k]];
        for (int i = 0; i < n - k; i++) q[k + i] -= temp * V[off[k] + i];
    }

    double *dA, *drd, *dlast, *dV, *dq;
    long *doff;
    (void)hipMalloc(&dA, sizeof(double) * n * n); (void)hipMalloc(&drd, sizeof(double) * n); (void)hipMalloc(&dlast, 8);
    (void)hipMalloc(&dV, sizeof(double) * total); (void)hipMalloc(&dq, sizeof(double) * n); (void)hipMalloc(&doff, sizeof(long) * n);
    (void)hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(dV, V.data(), sizeof(double) * total, hipMemcpyHostToDevice);
    (void)hipMemcpy(doff, off.data(), sizeof(long) * n, hipMemcpyHostToDevice);
    (void)hipMemset(drd, 0, sizeof(double) * n);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms_qr = 0, ms_qf = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(qrfac_chain, dim3(1), dim3(64), 0, 0, n, dA, drd, dlast);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms_qr, e0, e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(qform_chain, dim3(1), dim3(64), 0, 0, n, dV, doff, dq);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms_qf, e0, e1);
    }
    std::vector<double> rd_g(n), q_g(n);
    (void)hipMemcpy(rd_g.data(), drd, sizeof(double) * n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(q_g.data(), dq, sizeof(double) * n, hipMemcpyDeviceToHost);
    const bool ok = std::memcmp(rd_g.data(), rd.data(), sizeof(double) * (n - 1)) == 0 && std::memcmp(q_g.data(), q.data(), sizeof(double) * n) == 0;
    printf("{\"n\": %d, \"qrfac_dependency_chain_ms\": %.3f, \"qform_last_column_chain_ms\": %.3f, \"bit_identical_to_host_twin\": %s}\n", n, ms_qr, ms_qf,
           ok ? "true" : "false");
    return ok ? 0 : 1;
}
