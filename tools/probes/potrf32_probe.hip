// probe: Cholesky + triangular inverse of a 32 x 32 block held in registers by one wavefront (lane l = row l of L,
// column l of the inverse), values broadcast with v_readlane.  Checks against a host factorisation and reports cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__device__ __forceinline__ double rl(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void potrf32(const double *A, double *Lout, double *Xout, unsigned long long *cyc) {
    const int l = threadIdx.x & 31;
    double row[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) row[k] = A[l + k * 32];   // symmetric input: row l
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double rinv[32];   // 1 / L[j][j], uniform
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const double d = rl(row[j], j);
        const double piv = sqrt(d);
        double ri = 1.0 / piv;
        asm volatile("" : "+v"(ri));   // keep the 32 reciprocals in VGPRs (as uniform values they would fill the SGPR file)
        rinv[j] = ri;
        row[j] = (l == j) ? piv : row[j] * ri;
#pragma unroll
        for (int k = j + 1; k < 32; ++k) {
            const double u = rl(row[j], k);
            row[k] -= row[j] * u;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    // inverse: lane c holds column c of X = L^-1.  (The broadcasts L[i][k] below are the very values the factor loop
    // already read with readlane; left to CSE, all 496 of them stay live in SGPRs from there to here and spill.)
#pragma unroll
    for (int k = 0; k < 32; ++k) asm volatile("" : "+v"(row[k]));
    double x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        double s = (l == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const double Lik = rl(row[k], i);
            s -= Lik * x[k];
        }
        x[i] = (l <= i) ? s * rinv[i] : 0.0;
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x < 32) {
#pragma unroll
        for (int k = 0; k < 32; ++k) { Lout[l + k * 32] = (k <= l) ? row[k] : 0.0; Xout[k + l * 32] = x[k]; }
    }
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}

int main() {
    const int n = 32;
    std::vector<double> A(n * n), L(n * n), X(n * n), Lr(n * n, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i + j * n] = (i == j ? 40.0 : 0.0) + cos(0.3 * (i + 1) * (j + 1)) + cos(0.3 * (j + 1) * (i + 1));
    // host reference
    std::vector<double> W = A;
    for (int j = 0; j < n; ++j) {
        double d = W[j + j * n]; for (int k = 0; k < j; ++k) d -= Lr[j + k * n] * Lr[j + k * n];
        Lr[j + j * n] = sqrt(d);
        for (int i = j + 1; i < n; ++i) { double s = W[i + j * n]; for (int k = 0; k < j; ++k) s -= Lr[i + k * n] * Lr[j + k * n]; Lr[i + j * n] = s / Lr[j + j * n]; }
    }
    double *dA, *dL, *dX; unsigned long long *dc, hc[2];
    hipMalloc((void **) &dA, n * n * 8); hipMalloc((void **) &dL, n * n * 8); hipMalloc((void **) &dX, n * n * 8); hipMalloc((void **) &dc, 16);
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        potrf32<<<1, 64>>>(dA, dL, dX, dc);
        hipDeviceSynchronize();
    }
    hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, n * n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    double eL = 0.0, eX = 0.0;
    for (int i = 0; i < n * n; ++i) eL = fmax(eL, fabs(L[i] - Lr[i]));
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += Lr[i + k * n] * X[k + j * n]; eX = fmax(eX, fabs(s - (i == j))); }
    printf("factor %llu cycles, inverse %llu cycles; max |L - Lref| = %.2e, max |L X - I| = %.2e\n", hc[0], hc[1], eL, eX);
    return 0;
}
