// diagnostic (not part of the library): does s_setprio decide which of the two waves of a SIMD gets the fp64 matrix pipe?
// Two workgroups of 256 threads per CU run the same register-only MFMA loop; workgroups in the second half of the grid (the second
// slot of every CU, if the dispatcher fills CUs round-robin) set priority P1, the first half P0.  Reported: mean run time of the
// two halves (s_memrealtime, 100 MHz) -- equal priorities: how uneven the sharing is by age alone; unequal: whether priority wins.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/prio_probe.hip -o tools/probes/prio_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(ACC, A, B) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
template <int P0, int P1>
__global__ __launch_bounds__(256, 2) void probe(unsigned long long *t, int iters) {
    d4 acc[16];
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < 16; ++i) acc[i] = (d4){1e-3 * (gid & 15), 0.5, 0.25, 0.125};
    double a = 1.0 + (gid & 63) * 1e-9, b = 1.0 - (gid & 31) * 1e-9;
    const bool second = blockIdx.x >= gridDim.x / 2;
    if (second) __builtin_amdgcn_s_setprio(P1); else __builtin_amdgcn_s_setprio(P0);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) MFMA(acc[m], a, b);
    }
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = t1 + (s == 12345.678 ? 1 : 0); }
}
template <int P0, int P1> static void run(int iters) {
    const int blocks = 512;
    unsigned long long *d;
    hipMalloc((void **) &d, blocks * 16);
    hipLaunchKernelGGL((probe<P0, P1>), dim3(blocks), dim3(256), 0, 0, d, iters / 8);
    hipLaunchKernelGGL((probe<P0, P1>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
    double s0 = 0, s1 = 0, e0 = 0, e1 = 0;
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < blocks; ++b) tmin = h[2 * b] < tmin ? h[2 * b] : tmin;
    for (int b = 0; b < blocks; ++b) {
        const double dur = (double) (h[2 * b + 1] - h[2 * b]) / 100.0, end = (double) (h[2 * b + 1] - tmin) / 100.0;
        if (b < blocks / 2) { s0 += dur; e0 += end; } else { s1 += dur; e1 += end; }
    }
    printf("priority first half %d, second half %d: mean run time %.1f us / %.1f us, mean finish %.1f us / %.1f us\n", P0, P1,
           s0 / (blocks / 2), s1 / (blocks / 2), e0 / (blocks / 2), e1 / (blocks / 2));
    hipFree(d);
}
int main() {
    const int iters = 20000;
    run<0, 0>(iters); run<0, 3>(iters); run<3, 0>(iters); run<1, 1>(iters); run<0, 0>(iters);
    return 0;
}
