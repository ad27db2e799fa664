// probes.hip -- raw kernels exported for unit tests and micro-benchmarks, and the measurement probes (fp64 MFMA peak / issue
// rate, diagonal-block kernel, envelope Cholesky).  Its own translation unit: nothing here is on the product path; it reaches
// the engine only through the C ABI (HMiStream) and the kernel objects' headers.  Split out of engine.hip in round 3.
#include "../../include/hdsdp_mi355x.h"
#include "chol.h"
#include "hdm_common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
// the engine's stream (HMiStream brings the context up) and two events of this file's own
struct ProbeCtx { hipStream_t stream = nullptr; hipEvent_t ev[8] = {}; bool init = false; };
ProbeCtx g;
int ensure_ctx() {
    if (g.init) return 0;
    g.stream = (hipStream_t) HMiStream();
    if (!g.stream) return 1;
    for (int i = 6; i < 8; ++i) if (hipEventCreate(&g.ev[i]) != hipSuccess) return 1;
    g.init = true;
    return 0;
}
}  // namespace

__device__ __forceinline__ double mi_hash_unit(unsigned x) {  // pseudo-random in (-1, 1)
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (double) (int) x / 2147483648.0;
}
// same loop on full-range pseudo-random operands (data-dependent power -> sustained clock)
__global__ __launch_bounds__(256, 2) void mi_mfma_probe_rand_kernel(double *out, int iters) {
    hdm_d4 acc[4][4];
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 4; ++r) acc[j][i][r] = mi_hash_unit(gid * 64 + j * 16 + i * 4 + r);
    double fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = mi_hash_unit(gid * 8 + i + 1000003u); fb[i] = mi_hash_unit(gid * 8 + 4 + i + 7000001u); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[j], fb[i], acc[j][i], 0, 0, 0);
    }
    double s = 0.0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) s += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
    if (s == 12345.678) out[0] = s;
}

__global__ void mi_mfma_probe_kernel(double *out, int iters) {
    hdm_d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (hdm_d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;  // keep the loop alive
}

// GEMM-shaped MFMA issue probe: 16 accumulators fed by 4 + 4 operand registers exactly like the GEMM inner loop,
// no memory traffic at all.  NW = waves per workgroup.
template <int MODE>
__global__ __launch_bounds__(256, 2) void mi_mfma_probe2_kernel(double *out, int iters) {
    hdm_d4 acc[4][4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) acc[j][i] = (hdm_d4){0.0, 0.0, 0.0, 0.0};
    double fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = 1.0 + threadIdx.x * 1e-9 * (i + 1); fb[i] = 1.0 - threadIdx.x * 1e-9 * (i + 2); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[j], fb[i], acc[j][i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], fb[0], acc[j][i], 0, 0, 0);
        } else {  // 8 accumulators only (2 x 4), GEMM operand pattern
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[j], fb[i], acc[j][i], 0, 0, 0);
        }
    }
    double s = 0.0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) s += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
    if (s == 12345.678) out[0] = s;
}

template <int NACC, int LB>
__global__ __launch_bounds__(256, LB) void mi_mfma_probe3_kernel(double *out, int iters) {
    hdm_d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (hdm_d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}


extern "C" {

// ---------------------------------------------------------------- raw kernels for tests
int HMiGemmNT(const double *A, int64_t lda, int aKMajor, const double *B, int64_t ldb, int bKMajor, double *C,
              int64_t ldc, int M, int N, int K, double alpha, double beta, int kLimit, int lowerOnly) {
    if (ensure_ctx()) return 1;
    HdmGemmArgs q = {};
    q.A = A; q.lda = lda; q.a_kmajor = aKMajor; q.B = B; q.ldb = ldb; q.b_kmajor = bKMajor; q.C = C; q.ldc = ldc;
    q.M = M; q.N = N; q.K = K; q.alpha = alpha; q.beta = beta; q.klimit = kLimit; q.lower_only = lowerOnly;
    q.batch = 1; q.epilogue = HDM_EPI_STORE;
    if (hdm_launch_gemm(q, g.stream)) return 1;
    HDM_HIP_CHECK(hipStreamSynchronize(g.stream));
    return 0;
}

// blocked Cholesky + solve of a host matrix (lower triangle, column-major, leading dimension n) whose structural zeros are
// described by a block envelope: first[i] = first 128-block column with an entry in block row i (NULL = dense)
int HMiCholEnvelopeSolve(const double *A_host, int n, const int *first, const double *b, double *x, double *L_host, int *info) {
    if (ensure_ctx()) return 1;
    HdmChol ch;
    if (ch.init(n)) return 1;
    int rc = 1;
    do {
        if (first && ch.set_envelope(first)) break;
        if (ch.load_host(A_host, n, g.stream)) break;
        if (ch.factor(g.stream, info)) break;
        if (info && *info != 0) { rc = 0; break; }
        if (b && x && ch.solve_host(b, x, 1, 0, g.stream)) break;
        if (L_host && hipMemcpy2D(L_host, sizeof(double) * n, ch.L, sizeof(double) * ch.npad, sizeof(double) * n, n, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (0);
    ch.destroy();
    return rc;
}

// diagnostic: factorisation time (ms, HIP events, graph replay included) of an n x n matrix whose pattern is a band of
// `band` 128-blocks below the diagonal block, once as a dense matrix and once on its block envelope
__global__ void mi_band_spd_kernel(double *A, long ld, int n, int band) {
    const long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long) n * n) return;
    const int i = (int) (e % n), j = (int) (e / n);
    double v = 0.0;
    if (i == j) v = 4.0 * (band + 1) * 128.0;
    else if (i > j && i / 128 - j / 128 <= band) v = ((i * 31 + j * 17) % 13 == 0) ? 1.0 / (1.0 + ((i + j) % 7)) : 0.0;
    A[i + (long) j * ld] = v;
}
int HMiCholEnvelopeProbe(int n, int band, int reps, double *ms_dense, double *ms_env) {
    if (ensure_ctx()) return 1;
    double *A = nullptr;
    if (hipMalloc((void **) &A, sizeof(double) * (size_t) n * n) != hipSuccess) return 1;
    hipLaunchKernelGGL(mi_band_spd_kernel, dim3((unsigned) (((long) n * n + 255) / 256)), dim3(256), 0, g.stream, A, (long) n, n, band);
    int rc = 0;
    for (int pass = 0; pass < 2 && !rc; ++pass) {
        HdmChol ch;
        if (ch.init(n)) { rc = 1; break; }
        std::vector<int> first(ch.nblk);
        for (int b = 0; b < ch.nblk; ++b) first[b] = std::max(0, b - band);
        if (pass == 1 && ch.set_envelope(first.data())) rc = 1;
        float total = 0.f;
        for (int r = -2; r < reps && !rc; ++r) {
            int info = 0;
            if (ch.load_device(A, n, g.stream)) { rc = 1; break; }
            (void) hipEventRecord(g.ev[6], g.stream);
            if (ch.factor(g.stream, &info) || info != 0) { rc = 1; break; }
            (void) hipEventRecord(g.ev[7], g.stream);
            (void) hipEventSynchronize(g.ev[7]);
            float ms = 0.f;
            (void) hipEventElapsedTime(&ms, g.ev[6], g.ev[7]);
            if (r >= 0) total += ms;
        }
        *(pass == 0 ? ms_dense : ms_env) = total / std::max(1, reps);
        ch.destroy();
    }
    (void) hipFree(A);
    return rc;
}

int HMiPotrf(double *A_dev, int n, int64_t lda, int *info) {
    if (ensure_ctx()) return 1;
    HdmChol ch;
    if (ch.init(n)) return 1;
    if (ch.load_device(A_dev, lda, g.stream)) return 1;
    if (ch.factor(g.stream, info)) return 1;
    HDM_HIP_CHECK(hipMemcpy2DAsync(A_dev, sizeof(double) * lda, ch.L, sizeof(double) * ch.npad, sizeof(double) * n, n,
                                   hipMemcpyDeviceToDevice, g.stream));
    HDM_HIP_CHECK(hipStreamSynchronize(g.stream));
    ch.destroy();
    return 0;
}

double HMiDiagBlockProbe(int variant, int reps) {
    if (ensure_ctx()) return -1.0;
    return hdm_diag_block_probe(variant, reps, g.stream);
}

double HMiMfmaPeakProbe(int iters) {
    if (ensure_ctx()) return -1.0;
    double *out = nullptr;
    if (hipMalloc((void **) &out, 8) != hipSuccess) return -1.0;
    const int blocks = 256 * 8, threads = 256;
    hipLaunchKernelGGL(mi_mfma_probe_kernel, dim3(blocks), dim3(threads), 0, g.stream, out, 16);
    (void) hipEventRecord(g.ev[6], g.stream);
    hipLaunchKernelGGL(mi_mfma_probe_kernel, dim3(blocks), dim3(threads), 0, g.stream, out, iters);
    (void) hipEventRecord(g.ev[7], g.stream);
    (void) hipEventSynchronize(g.ev[7]);
    float ms = 0;
    (void) hipEventElapsedTime(&ms, g.ev[6], g.ev[7]);
    (void) hipFree(out);
    const double flops = (double) blocks * (threads / 64) * (double) iters * 8 * 2.0 * 16 * 16 * 4;
    return flops / (ms * 1e-3) / 1e12;
}

// mode 0: GEMM operand pattern, mode 1: one operand pair; wgPerCu workgroups of 256 threads per CU
double HMiMfmaIssueProbe(int mode, int wgPerCu, int iters) {
    if (ensure_ctx()) return -1.0;
    double *out = nullptr;
    if (hipMalloc((void **) &out, 8) != hipSuccess) return -1.0;
    const int blocks = 256 * wgPerCu, threads = 256;
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) (void) hipEventRecord(g.ev[6], g.stream);
        if (mode == 300) {
            hipLaunchKernelGGL(mi_mfma_probe_rand_kernel, dim3(blocks), dim3(threads), 0, g.stream, out, iters);
        } else if (mode == 200) {
            hipLaunchKernelGGL(mi_mfma_probe_kernel, dim3(blocks), dim3(threads), 0, g.stream, out, iters * 2);
        } else if (mode >= 100) {
            const int it3 = iters * 16 / (mode % 100);
            switch (mode) {
                case 104: hipLaunchKernelGGL((mi_mfma_probe3_kernel<4, 1>), dim3(blocks), dim3(threads), 0, g.stream, out, it3); break;
                case 108: hipLaunchKernelGGL((mi_mfma_probe3_kernel<8, 1>), dim3(blocks), dim3(threads), 0, g.stream, out, it3); break;
                case 112: hipLaunchKernelGGL((mi_mfma_probe3_kernel<12, 1>), dim3(blocks), dim3(threads), 0, g.stream, out, it3); break;
                case 116: hipLaunchKernelGGL((mi_mfma_probe3_kernel<16, 1>), dim3(blocks), dim3(threads), 0, g.stream, out, it3); break;
                case 124: hipLaunchKernelGGL((mi_mfma_probe3_kernel<24, 1>), dim3(blocks), dim3(threads), 0, g.stream, out, it3); break;
                default: break;
            }
        } else if (mode == 0) hipLaunchKernelGGL(mi_mfma_probe2_kernel<0>, dim3(blocks), dim3(threads), 0, g.stream, out, iters);
        else if (mode == 1) hipLaunchKernelGGL(mi_mfma_probe2_kernel<1>, dim3(blocks), dim3(threads), 0, g.stream, out, iters);
        else hipLaunchKernelGGL(mi_mfma_probe2_kernel<2>, dim3(blocks), dim3(threads), 0, g.stream, out, iters);
    }
    (void) hipEventRecord(g.ev[7], g.stream);
    (void) hipEventSynchronize(g.ev[7]);
    float ms = 0;
    (void) hipEventElapsedTime(&ms, g.ev[6], g.ev[7]);
    (void) hipFree(out);
    const double flops = (double) blocks * (threads / 64) * (double) iters * 16 * 2.0 * 16 * 16 * 4;
    return flops / (ms * 1e-3) / 1e12;
}

}  // extern "C"
