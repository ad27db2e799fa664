// diagnostic (not part of the library): do fp64 vector FMAs execute beside fp64 MFMAs, or do they share the multipliers?
// gfx950 quotes the same 78.6 TFLOP/s for fp64 vector and fp64 matrix work.  If the two pipes were independent, a GEMM
// could feed both and pass the "matrix peak".  Modes, each a register-only loop (no memory traffic), 2 workgroups of 256
// threads per CU like the GEMM kernels:
//   0  MFMA only              16 v_mfma_f64_16x16x4_f64 per iteration on 16 accumulators
//   1  vector only            64 v_fma_f64 per iteration (the flops of 4 MFMAs... per lane: 64 x 2 flops)
//   2  both in one wave       16 MFMAs interleaved with NV v_fma_f64 (NV = 16, 32, 64)
//   3  split by wave          waves 0,1 of a workgroup run mode 0, waves 2,3 run mode 1 (partners on one SIMD differ per
//                             workgroup placement; with 2 workgroups per CU every SIMD sees both kinds)
// Reported: time, MFMA TFLOP/s, vector TFLOP/s and their sum.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/coexec_probe.hip -o tools/probes/coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ __launch_bounds__(256, 2) void probe(double *out, int iters) {
    d4 acc[16];
    double v[16];
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < 16; ++i) {
        acc[i] = (d4){1e-3 * (gid & 15), 0.5, 0.25, 0.125};
        v[i] = 1.0 + 1e-9 * ((gid + i) & 1023);
    }
    double a = 1.0 + (gid & 63) * 1e-9, b = 1.0 - (gid & 31) * 1e-9, c = 0.999999 + 1e-12 * (gid & 7);
    const int wave = threadIdx.x >> 6;
    const bool do_m = (MODE == 0) || (MODE == 2) || (MODE == 3 && wave < 2);
    const bool do_v = (MODE == 1) || (MODE == 2) || (MODE == 3 && wave >= 2);
    if (MODE == 3) {
        if (do_m) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < NV / 16; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], c, b);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (do_m) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
                if (do_v) {
#pragma unroll
                    for (int r = 0; r < NV / 16; ++r) v[(i + 4 * r) & 15] = __builtin_fma(v[(i + 4 * r) & 15], c, b);
                }
            }
        }
    }
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    if (s == 12345.678) out[0] = s;
}

template <int MODE, int NV>
static void run(const char *label, int iters) {
    double *out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 512;
    hipLaunchKernelGGL((probe<MODE, NV>), dim3(grid), dim3(256), 0, 0, out, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NV>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = grid * 4.0;
    double mw = waves, vw = waves;
    if (MODE == 0) vw = 0;
    if (MODE == 1) mw = 0;
    if (MODE == 3) { mw = waves / 2; vw = waves / 2; }
    const double mflop = mw * iters * 16.0 * (16 * 16 * 4 * 2);
    const double vflop = vw * iters * (double) NV * 64 * 2;
    printf("%-44s %8.2f ms   mfma %6.2f TF   vector %6.2f TF   sum %6.2f TF\n", label, ms, mflop / ms / 1e9, vflop / ms / 1e9,
           (mflop + vflop) / ms / 1e9);
    hipFree(out);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    run<0, 16>("mfma only (16 per iteration)", iters);
    run<1, 64>("vector only (64 v_fma_f64 per iteration)", iters);
    run<2, 16>("one wave: 16 mfma + 16 v_fma_f64", iters);
    run<2, 32>("one wave: 16 mfma + 32 v_fma_f64", iters);
    run<2, 64>("one wave: 16 mfma + 64 v_fma_f64", iters);
    run<2, 128>("one wave: 16 mfma + 128 v_fma_f64", iters);
    run<3, 64>("split: waves 0,1 mfma / waves 2,3 vector(64)", iters);
    run<3, 256>("split: waves 0,1 mfma / waves 2,3 vector(256)", iters);
    return 0;
}
