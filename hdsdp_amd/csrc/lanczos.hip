// lanczos.hip -- device-resident ratio test  max{ alpha : S + alpha dS >= 0 }  (SURVEY.md 8(f) row 2).
//
// Reference: sdpDenseConeRatioTestImpl (interface/hdsdp_conic_sdp.c:1640-1686) forms dS and hands the operator
//     x -> L^-1 (-dS) L^-T x      (sdpDenseConeILanczosMultiply, :462-505: dtrsv, dsymv, dtrsv)
// to HLanczosSolve (linalg/hdsdp_lanczos.c:161-292): <= 30 Lanczos steps from a fixed pseudo-random start, a
// Ritz check every third step, and the conservative step 1 / (gamma + lambda_max) with gamma built from two Ritz
// residuals.  Here the recurrence and the acceptance logic are the same, step for step; what changes is where the
// work runs:
//   * the operator is three dense matrix-vector products with the explicit lower-triangular Linv that the Schur
//     build needs anyway (HBM-bound: 2 x 16 + 32 MB at n = 2000) instead of two latency-bound triangular solves,
//   * one fused single-workgroup kernel per step does the three-term recurrence, both reductions and the
//     normalisation, and returns (alpha_k, beta_k) -- two doubles cross PCIe per step,
//   * the (k+1) x (k+1) Ritz problem is solved on the host by cyclic Jacobi (no LAPACK dependency).
// The start vector reproduces glibc's srand()/rand() stream (TYPE_3 additive feedback generator) without touching
// the process-wide libc state, so the device run starts from the reference's own vector.
#include "lanczos.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

// ---- glibc random_r TYPE_3 (r[i] = r[i-3] + r[i-31]), as srand(seed) / rand() use it -------------------------------
struct GlibcRand {
    int f, b;   // front / rear indices of the 31-word state
    int32_t st[31];
    void seed(unsigned int s) {
        if (s == 0) s = 1;
        st[0] = (int32_t) s;
        long word = (int32_t) s;
        for (int i = 1; i < 31; ++i) {
            long hi = word / 127773, lo = word % 127773;
            word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            st[i] = (int32_t) word;
        }
        f = 3; b = 0;
        for (int i = 0; i < 310; ++i) (void) next();
    }
    int next() {
        uint32_t v = (uint32_t) st[f] + (uint32_t) st[b];
        st[f] = (int32_t) v;
        int out = (int) (v >> 1);
        if (++f >= 31) f = 0;
        if (++b >= 31) b = 0;
        return out;
    }
};

// single workgroup: three-term recurrence + normalisation of one Lanczos step (hdsdp_lanczos.c:199-218)
//   w -= hprev * Vprev ;  alp = -<w, Vk> ;  w += alp * Vk ;  nrm = |w| ;  Vnext = vnext = w / nrm  (if nrm > 0)
// w arrives as the `nchunk` partial sums of the operator's last product (hdm_gemv_n_kernel), summed here in chunk order;
// hprev is read from DEVICE memory (the previous step's norm), so the steps between two Ritz checks are queued back to
// back and the host reads their (alpha, beta) pairs with one synchronisation per group instead of one per step.
__global__ __launch_bounds__(1024) void hdm_lanczos_step_kernel(const double *__restrict__ part, int nchunk, double *__restrict__ w,
                                                                const double *__restrict__ Vprev, const double *__restrict__ hprev_dev,
                                                                const double *__restrict__ Vk, double *__restrict__ Vnext,
                                                                double *__restrict__ vnext, int n, double *__restrict__ out) {
    __shared__ double red[16];
    __shared__ double bc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto reduce = [&](double s) {
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += red[q]; bc = t; }
        __syncthreads();
        double r = bc;
        __syncthreads();
        return r;
    };
    const double hprev = Vprev ? *hprev_dev : 0.0;
    double s = 0.0;
    for (int i = tid; i < n; i += 1024) {
        double x = 0.0;
        for (int c = 0; c < nchunk; ++c) x += part[(long) c * n + i];
        if (Vprev) x -= hprev * Vprev[i];
        w[i] = x;
        s += x * Vk[i];
    }
    const double alp = -reduce(s);
    s = 0.0;
    for (int i = tid; i < n; i += 1024) {
        double x = w[i] + alp * Vk[i];
        w[i] = x;
        s += x * x;
    }
    const double nrm = sqrt(reduce(s));
    if (nrm > 0.0) {
        const double inv = 1.0 / nrm;
        for (int i = tid; i < n; i += 1024) { double x = w[i] * inv; Vnext[i] = x; vnext[i] = x; }
    }
    if (tid == 0) { out[0] = alp; out[1] = nrm; }
}

// Small blocks (n16 <= 256): up to three whole Lanczos steps -- operator application and recurrence -- in ONE single-workgroup
// launch.  The reference checks its Ritz values every third step (checkFreq, hdsdp_lanczos.c:187-189) and the host needs
// nothing but (alpha_k, beta_k) in between, so a group of steps is one launch and one synchronisation instead of five
// launches and a synchronisation per step: a ratio test on a 100 x 100 block went from 0.74 to 0.3 ms, and it is more than
// half of what the reference's driver spends below the C ABI on mcp100 / gpp100.  The matrices (<= 512 KB each) are read
// from L2; vectors live in LDS.  Sums are taken in a fixed order.  out: (alpha, beta) per step, then the number of steps done
// (a zero norm ends the group early, as it ends the reference's loop).
#define LZ_FUSED_MAX 256
#define LZ_NCHUNK 32       // column chunks of the operator's last product (partial sums, reduced in chunk order)
__global__ __launch_bounds__(1024) void hdm_lanczos_fused_kernel(const double *__restrict__ Linv, long ldl,
                                                                 const double *__restrict__ dS, long ldd, int n,
                                                                 double *__restrict__ V, long ldv, int k0, int nsteps, double hprev,
                                                                 double *__restrict__ blk, double *__restrict__ out) {
    __shared__ double sv[LZ_FUSED_MAX], st1[LZ_FUSED_MAX], st2[LZ_FUSED_MAX], part[4][LZ_FUSED_MAX], red[16];
    __shared__ double bc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto reduce = [&](double s) {
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += red[q]; bc = t; }
        __syncthreads();
        const double r = bc;
        __syncthreads();
        return r;
    };
    int done = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int k = k0 + s;
        if (tid < n) sv[tid] = V[tid + (long) k * ldv];
        __syncthreads();
        // t1 = Linv^T v: one wavefront per column (contiguous reads), rows above the diagonal skipped
        for (int j = wave; j < n; j += 16) {
            const double *col = Linv + (long) j * ldl;
            double a = 0.0;
            for (int i = (j & ~63) + lane; i < n; i += 64) a += ((i < j) ? 0.0 : col[i]) * sv[i];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) st1[j] = a;
        }
        __syncthreads();
        // t2 = -dS t1 (dS symmetric: column dots again)
        for (int j = wave; j < n; j += 16) {
            const double *col = dS + (long) j * ldd;
            double a = 0.0;
            for (int i = lane; i < n; i += 64) a += col[i] * st1[i];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) st2[j] = -a;
        }
        __syncthreads();
        // w = Linv t2: rows across lanes (coalesced), four column chunks summed in order
        {
            const int i = tid & (LZ_FUSED_MAX - 1), c = tid >> 8;
            const int cw = (n + 3) / 4, j0 = c * cw, j1 = min(n, j0 + cw), jend = min(j1, i + 1);
            double a0 = 0.0, a1 = 0.0;
            if (i < n) {
                int j = j0;
                for (; j + 1 < jend; j += 2) {
                    a0 += Linv[i + (long) j * ldl] * st2[j];
                    a1 += Linv[i + (long) (j + 1) * ldl] * st2[j + 1];
                }
                if (j < jend) a0 += Linv[i + (long) j * ldl] * st2[j];
            }
            part[c][i] = a0 + a1;
        }
        __syncthreads();
        // the three-term recurrence and the normalisation (hdsdp_lanczos.c:199-218): thread i keeps element i
        double x = 0.0, vk = 0.0;
        if (tid < n) {
            x = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
            vk = sv[tid];
            if (k > 0) x -= hprev * V[tid + (long) (k - 1) * ldv];
        }
        const double alp = -reduce((tid < n) ? x * vk : 0.0);
        x += alp * vk;
        const double nrm = sqrt(reduce((tid < n) ? x * x : 0.0));
        if (tid == 0) { out[2 * s] = alp; out[2 * s + 1] = nrm; }
        done = s + 1;
        if (!(nrm > 0.0)) break;                 // (uniform)
        if (tid < n) {
            const double xn = x * (1.0 / nrm);
            V[tid + (long) (k + 1) * ldv] = xn;
            blk[tid] = xn;
        }
        hprev = nrm;
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) out[2 * nsteps] = (double) done;
}

// z = V[:, 0..kc) * coef   (single workgroup; V column stride ldv)
__global__ __launch_bounds__(1024) void hdm_lincomb_kernel(const double *__restrict__ V, long ldv, int kc,
                                                           const double *__restrict__ coef, double *__restrict__ z, int n) {
    for (int i = threadIdx.x; i < n; i += 1024) {
        double s = 0.0;
        for (int c = 0; c < kc; ++c) s += V[i + (long) c * ldv] * coef[c];
        z[i] = s;
    }
}

// out[0] = | a - lam * b |_2   (single workgroup)
__global__ __launch_bounds__(1024) void hdm_resnorm_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                           double lam, int n, double *__restrict__ out) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (int i = tid; i < n; i += 1024) { double x = a[i] - lam * b[i]; s += x * x; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += red[q]; out[0] = sqrt(t); }
}

// v <- v / |v| into the block's column 0 and into V[:, 0]   (single workgroup)
__global__ __launch_bounds__(1024) void hdm_normalize_kernel(const double *__restrict__ v, double *__restrict__ V0,
                                                             double *__restrict__ blk, int n) {
    __shared__ double red[16];
    __shared__ double bc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (int i = tid; i < n; i += 1024) s += v[i] * v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += red[q]; bc = sqrt(t); }
    __syncthreads();
    const double inv = bc > 0.0 ? 1.0 / bc : 0.0;
    for (int i = tid; i < n; i += 1024) { double x = v[i] * inv; V0[i] = x; blk[i] = x; }
}

// y[j] = alpha * sum_i A[i + j*ld] x[i]  (transposed product: one wavefront per column, contiguous reads).
// lower != 0: A is lower triangular, rows i < j are skipped (half the traffic).
__global__ __launch_bounds__(256) void hdm_gemv_t_kernel(const double *__restrict__ A, long ld, int n, int lower, double alpha,
                                                         const double *__restrict__ x, double *__restrict__ y) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const double *col = A + (long) j * ld;
    double s0 = 0.0, s1 = 0.0;
    int i = (lower ? (j & ~63) : 0) + lane;
    for (; i + 64 < n; i += 128) {
        const double a0 = col[i], a1 = col[i + 64];
        s0 += ((lower && i < j) ? 0.0 : a0) * x[i];
        s1 += ((lower && i + 64 < j) ? 0.0 : a1) * x[i + 64];
    }
    if (i < n) s0 += ((lower && i < j) ? 0.0 : col[i]) * x[i];
    double s = s0 + s1;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[j] = alpha * s;
}

// part[c][i] = sum over the columns j of chunk c of A[i + j*ld] x[j]  (plain product, rows across lanes: coalesced);
// lower != 0: A is lower triangular, column j only reaches rows i >= j.  Reduced in chunk order by the kernel below.
__global__ __launch_bounds__(256) void hdm_gemv_n_kernel(const double *__restrict__ A, long ld, int n, int lower, int nchunk,
                                                         const double *__restrict__ x, double *__restrict__ part) {
    const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    const int cw = (n + nchunk - 1) / nchunk, j0 = c * cw, j1 = min(n, j0 + cw);
    if (i >= n) return;
    double s0 = 0.0, s1 = 0.0;
    const int jend = lower ? min(j1, i + 1) : j1;
    int j = j0;
    for (; j + 1 < jend; j += 2) {
        s0 += A[i + (long) j * ld] * x[j];
        s1 += A[i + (long) (j + 1) * ld] * x[j + 1];
    }
    if (j < jend) s0 += A[i + (long) j * ld] * x[j];
    part[(long) c * n + i] = s0 + s1;
}
__global__ void hdm_gemv_n_reduce_kernel(const double *__restrict__ part, int n, int nchunk, double alpha, double *__restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[(long) c * n + i];
    y[i] = alpha * s;
}

// upper triangle <- lower triangle of an n x n column-major matrix
__global__ void hdm_mirror_lower_kernel(double *A, long ld, int n) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long) n * n) return;
    int i = (int) (e % n), j = (int) (e / n);
    if (i > j) A[j + (long) i * ld] = A[i + (long) j * ld];
}

// cyclic Jacobi for a small dense symmetric matrix (column-major k x k); eigenvalues ascending in d, vectors in Y
void jacobi_eig(int k, std::vector<double> A, std::vector<double> &d, std::vector<double> &Y) {
    Y.assign((size_t) k * k, 0.0);
    for (int i = 0; i < k; ++i) Y[(size_t) i * k + i] = 1.0;
    auto a = [&](int i, int j) -> double & { return A[(size_t) j * k + i]; };
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < k; ++p)
            for (int q = p + 1; q < k; ++q) off += a(p, q) * a(p, q);
        if (off < 1e-300) break;
        for (int p = 0; p < k; ++p)
            for (int q = p + 1; q < k; ++q) {
                const double apq = a(p, q);
                if (fabs(apq) < 1e-300) continue;
                const double theta = (a(q, q) - a(p, p)) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int r = 0; r < k; ++r) {
                    const double arp = a(r, p), arq = a(r, q);
                    a(r, p) = c * arp - s * arq;
                    a(r, q) = s * arp + c * arq;
                }
                for (int r = 0; r < k; ++r) {
                    const double apr = a(p, r), aqr = a(q, r);
                    a(p, r) = c * apr - s * aqr;
                    a(q, r) = s * apr + c * aqr;
                }
                for (int r = 0; r < k; ++r) {
                    const double yrp = Y[(size_t) p * k + r], yrq = Y[(size_t) q * k + r];
                    Y[(size_t) p * k + r] = c * yrp - s * yrq;
                    Y[(size_t) q * k + r] = s * yrp + c * yrq;
                }
            }
    }
    d.resize(k);
    for (int i = 0; i < k; ++i) d[i] = a(i, i);
    // ascending selection sort of (value, vector)
    for (int i = 0; i < k; ++i) {
        int mn = i;
        for (int j = i + 1; j < k; ++j) if (d[j] < d[mn]) mn = j;
        if (mn != i) {
            std::swap(d[i], d[mn]);
            for (int r = 0; r < k; ++r) std::swap(Y[(size_t) i * k + r], Y[(size_t) mn * k + r]);
        }
    }
}

}  // namespace

void hdm_lanczos_start_vector(int n, double *p) {
    // HLanczosIPrepare (hdsdp_lanczos.c:33-42): srand(n); per entry srand(rand()); sqrt(sqrt(rand() % 1627)) * (rand() % 2 - 0.5)
    // (the reference re-seeds the one libc generator inside the loop, so a single generator object follows it)
    GlibcRand g;
    g.seed((unsigned int) n);
    for (int i = 0; i < n; ++i) {
        g.seed((unsigned int) g.next());
        const int a = g.next() % 1627;
        const int b = g.next() % 2;
        p[i] = sqrt(sqrt((double) a)) * ((double) b - 0.5);
    }
}

// A <- scale * ( sym(A) + diag_add * I )   (n x n, column-major): the two symmetrisation passes of the primal recovery
__global__ void hdm_sym_scale_kernel(double *A, long ld, int n, double diag_add, double scale) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long) n * n) return;
    int i = (int) (e % n), j = (int) (e / n);
    if (i < j) return;
    if (i == j) { A[i + (long) i * ld] = scale * (A[i + (long) i * ld] + diag_add); return; }
    const double v = scale * 0.5 * (A[i + (long) j * ld] + A[j + (long) i * ld]);
    A[i + (long) j * ld] = v;
    A[j + (long) i * ld] = v;
}

int hdm_sym_scale(double *A, long ld, int n, double diag_add, double scale, hipStream_t s) {
    long tot = (long) n * n;
    hipLaunchKernelGGL(hdm_sym_scale_kernel, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, s, A, ld, n, diag_add, scale);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// out = S + step * dS over `count` doubles (out may alias S)
__global__ void hdm_axpy_mat_kernel(double *out, const double *S, const double *dS, double step, long count) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) out[e] = S[e] + step * dS[e];
}

int hdm_axpy_mat(double *out, const double *S, const double *dS, double step, long count, hipStream_t s) {
    hipLaunchKernelGGL(hdm_axpy_mat_kernel, dim3((unsigned) ((count + 255) / 256)), dim3(256), 0, s, out, S, dS, step, count);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int hdm_mirror_lower(double *A, long ld, int n, hipStream_t s) {
    long tot = (long) n * n;
    hipLaunchKernelGGL(hdm_mirror_lower_kernel, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, s, A, ld, n);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int HdmLanczos::init(int n_) {
    n = n_;
    n16 = (n + 15) / 16 * 16;
    const size_t blk = sizeof(double) * (size_t) n16 * 8;
    HDM_HIP_CHECK(hipMalloc((void **) &V, sizeof(double) * (size_t) n16 * (maxdim + 1)));
    HDM_HIP_CHECK(hipMalloc((void **) &bv, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &b1, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &b2, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &bw, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &bz, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &warm, sizeof(double) * (size_t) n16));
    HDM_HIP_CHECK(hipMalloc((void **) &tmp, sizeof(double) * (size_t) n16));
    HDM_HIP_CHECK(hipMalloc((void **) &scal, sizeof(double) * 64));
    HDM_HIP_CHECK(hipMalloc((void **) &part, sizeof(double) * 32 * (size_t) n16));
    for (double *b : {bv, b1, b2, bw, bz}) HDM_HIP_CHECK(hdm_memset_sync(b, 0, blk));
    HDM_HIP_CHECK(hdm_memset_sync(warm, 0, sizeof(double) * (size_t) n16));
    HDM_HIP_CHECK(hdm_memset_sync(tmp, 0, sizeof(double) * (size_t) n16));
    start.resize(n);
    hdm_lanczos_start_vector(n, start.data());
    nComputed = 0;
    return 0;
}

void HdmLanczos::destroy() {
    for (double *b : {V, bv, b1, b2, bw, bz, warm, tmp, scal, part})
        if (b) (void) hipFree(b);
    V = bv = b1 = b2 = bw = bz = warm = tmp = scal = part = nullptr;
}

// out (column 0 of a vector block) = Linv * ( -dS * ( Linv^T * in ) ): three HBM-bound matrix-vector products with
// dedicated kernels (the first version went through the 128 x 128-tile GEMM with an 8-column block: 0.7 ms per
// application at n = 2000, launch- and tile-latency bound)
int HdmLanczos::apply(const double *Linv, long ldl, const double *dS, long ldd, const double *in, double *out, hipStream_t s) {
    const int nchunk = LZ_NCHUNK;
    // t1 = Linv^T v            (column dots of the lower-triangular Linv)
    hipLaunchKernelGGL(hdm_gemv_t_kernel, dim3((n16 + 3) / 4), dim3(256), 0, s, Linv, ldl, n16, 1, 1.0, in, b1);
    // t2 = -dS t1              (dS is symmetric: column dots again)
    hipLaunchKernelGGL(hdm_gemv_t_kernel, dim3((n16 + 3) / 4), dim3(256), 0, s, dS, ldd, n16, 0, -1.0, b1, b2);
    // w = Linv t2              (rows across lanes, 32 column chunks, deterministic two-level sum)
    hipLaunchKernelGGL(hdm_gemv_n_kernel, dim3((n16 + 255) / 256, nchunk), dim3(256), 0, s, Linv, ldl, n16, 1, nchunk, b2, part);
    if (out) hipLaunchKernelGGL(hdm_gemv_n_reduce_kernel, dim3((n16 + 255) / 256), dim3(256), 0, s, part, n16, nchunk, 1.0, out);   // (out == nullptr: the caller sums the partials itself)
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int HdmLanczos::solve(const double *Linv, long ldl, const double *dS, long ldd, hipStream_t s, double *maxStep, int *steps) {
    const int md = maxdim, nh = md + 1;
    std::vector<double> H((size_t) nh * nh, 0.0);
    auto Hm = [&](int i, int j) -> double & { return H[(size_t) j * nh + i]; };
    // starting vector: fresh, or the previous Ritz image + 1e-3 * the same pseudo-random vector (:166-181)
    {
        std::vector<double> v0(n16, 0.0);
        if (nComputed == 0) {
            for (int i = 0; i < n; ++i) v0[i] = start[i];
            HDM_HIP_CHECK(hipMemcpyAsync(tmp, v0.data(), sizeof(double) * n16, hipMemcpyHostToDevice, s));
            HDM_HIP_CHECK(hipStreamSynchronize(s));
        } else {
            HDM_HIP_CHECK(hipMemcpyAsync(v0.data(), warm, sizeof(double) * n16, hipMemcpyDeviceToHost, s));
            HDM_HIP_CHECK(hipStreamSynchronize(s));
            for (int i = 0; i < n; ++i) v0[i] += 1e-03 * start[i];
            for (int i = n; i < n16; ++i) v0[i] = 0.0;
            HDM_HIP_CHECK(hipMemcpyAsync(tmp, v0.data(), sizeof(double) * n16, hipMemcpyHostToDevice, s));
            HDM_HIP_CHECK(hipStreamSynchronize(s));
        }
    }
    HDM_HIP_CHECK(hipMemsetAsync(V, 0, sizeof(double) * (size_t) n16 * (md + 1), s));
    hipLaunchKernelGGL(hdm_normalize_kernel, dim3(1), dim3(1024), 0, s, tmp, V, bv, n16);
    HDM_HIP_CHECK(hipGetLastError());

    int checkFreq = md / 5;
    if (checkFreq > 3) checkFreq = 3;
    double step = 0.0;
    int k = 0;
    double hs[2] = {0.0, 0.0};
    std::vector<double> d, Y;
    static const bool fuse_env = [] { const char *e = getenv("HDM_LANCZOS_FUSED"); return !(e && atoi(e) == 0); }();
    const bool fused = fuse_env && n16 <= LZ_FUSED_MAX && checkFreq >= 1;
    static const bool group_env = [] { const char *e = getenv("HDM_LANCZOS_GROUP"); return !(e && atoi(e) == 0); }();   // 0: one synchronisation per step (A/B)
    double grp[2 * 8 + 1] = {0.0};               // (alpha, beta) of the current group of steps, fused form
    int grp_k0 = -1, grp_n = 0;
    for (k = 0; k < md; ++k) {
        const double hprev = (k > 0) ? Hm(k, k - 1) : 0.0;
        if (fused) {
            if (grp_k0 < 0 || k >= grp_k0 + grp_n) {       // next group: as many steps as lie before the next Ritz check
                grp_k0 = k;
                grp_n = std::min(std::min(checkFreq - (k % checkFreq), md - k), 8);
                hipLaunchKernelGGL(hdm_lanczos_fused_kernel, dim3(1), dim3(1024), 0, s, Linv, ldl, dS, ldd, n16, V, (long) n16, k, grp_n,
                                   hprev, bv, scal + 44);
                HDM_HIP_CHECK(hipGetLastError());
                HDM_HIP_CHECK(hipMemcpyAsync(grp, scal + 44, sizeof(double) * (2 * grp_n + 1), hipMemcpyDeviceToHost, s));   // (scal + 8 .. + 39: Ritz coefficients)
                HDM_HIP_CHECK(hipStreamSynchronize(s));
            }
            hs[0] = grp[2 * (k - grp_k0)]; hs[1] = grp[2 * (k - grp_k0) + 1];
        } else {
            // large blocks: the steps up to the next Ritz check are queued back to back (4 launches each: three products and
            // the recurrence, which takes the previous step's norm from device memory); one copy and one synchronisation
            // per group.  A zero norm inside a group ends the reference's loop at that step: the host stops there too and
            // what the later steps of the group computed is never looked at.
            if (grp_k0 < 0 || k >= grp_k0 + grp_n) {
                grp_k0 = k;
                grp_n = group_env ? std::min(std::min(checkFreq - (k % checkFreq), md - k), 8) : 1;
                for (int q = 0; q < grp_n; ++q) {
                    const int kk = k + q;
                    if (apply(Linv, ldl, dS, ldd, bv, nullptr, s)) return 1;
                    hipLaunchKernelGGL(hdm_lanczos_step_kernel, dim3(1), dim3(1024), 0, s, part, LZ_NCHUNK, bw,
                                       kk > 0 ? V + (size_t) (kk - 1) * n16 : nullptr,
                                       q > 0 ? scal + 44 + 2 * (q - 1) + 1 : scal + 43,
                                       V + (size_t) kk * n16, V + (size_t) (kk + 1) * n16, bv, n16, scal + 44 + 2 * q);
                }
                HDM_HIP_CHECK(hipGetLastError());
                HDM_HIP_CHECK(hipMemcpyAsync(grp, scal + 44, sizeof(double) * (2 * grp_n), hipMemcpyDeviceToHost, s));
                // the last norm of this group is the next group's hprev: keep it where the next group's first step reads it
                HDM_HIP_CHECK(hipMemcpyAsync(scal + 43, scal + 44 + 2 * (grp_n - 1) + 1, sizeof(double), hipMemcpyDeviceToDevice, s));
                HDM_HIP_CHECK(hipStreamSynchronize(s));
            }
            hs[0] = grp[2 * (k - grp_k0)]; hs[1] = grp[2 * (k - grp_k0) + 1];
        }
        const double vAlp = hs[0], normPres = hs[1];
        Hm(k, k) = -vAlp;
        if (normPres > 0.0) Hm(k + 1, k) = Hm(k, k + 1) = normPres;

        if ((k + 1) % checkFreq == 0 || k > md - 1 || normPres == 0.0) {
            const int kp = k + 1;
            std::vector<double> U((size_t) kp * kp);
            for (int j = 0; j < kp; ++j)
                for (int i = 0; i < kp; ++i) U[(size_t) j * kp + i] = 0.5 * (Hm(i, j) + Hm(j, i));
            jacobi_eig(kp, U, d, Y);
            const double eig1 = d[kp - 1], eig2 = kp > 1 ? d[kp - 2] : d[kp - 1];
            const double *y1 = &Y[(size_t) (kp - 1) * kp], *y2 = kp > 1 ? &Y[(size_t) (kp - 2) * kp] : y1;
            const double resiVal = fabs(Hm(kp, k) * y1[k]);
            if (resiVal < 1e-04 || k >= md - 1) {
                double r12[2];
                // z1 = V y1 ; z2 = Op z1 ; warm start <- z2 ; resiVal1 = | z2 - eig1 z1 |
                HDM_HIP_CHECK(hipMemcpyAsync(scal + 8, y1, sizeof(double) * kp, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(hdm_lincomb_kernel, dim3(1), dim3(1024), 0, s, V, (long) n16, kp, scal + 8, bz, n16);
                if (apply(Linv, ldl, dS, ldd, bz, bw, s)) return 1;
                HDM_HIP_CHECK(hipMemcpyAsync(warm, bw, sizeof(double) * n16, hipMemcpyDeviceToDevice, s));
                hipLaunchKernelGGL(hdm_resnorm_kernel, dim3(1), dim3(1024), 0, s, bw, bz, eig1, n16, scal + 2);
                HDM_HIP_CHECK(hipStreamSynchronize(s));   // y1 lives in a host vector that is reused below
                // z2' = V y2 ; resiVal2 = | Op z2' - eig1 z2' |   (the reference uses eig1 here too, :262-266)
                HDM_HIP_CHECK(hipMemcpyAsync(scal + 8, y2, sizeof(double) * kp, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(hdm_lincomb_kernel, dim3(1), dim3(1024), 0, s, V, (long) n16, kp, scal + 8, bz, n16);
                if (apply(Linv, ldl, dS, ldd, bz, bw, s)) return 1;
                hipLaunchKernelGGL(hdm_resnorm_kernel, dim3(1), dim3(1024), 0, s, bw, bz, eig1, n16, scal + 3);
                HDM_HIP_CHECK(hipGetLastError());
                HDM_HIP_CHECK(hipMemcpyAsync(r12, scal + 2, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
                HDM_HIP_CHECK(hipStreamSynchronize(s));
                // after the second apply() the vector block bv still holds v_{k+1}: the recurrence can continue
                const double resiVal1 = r12[0], resiVal2 = r12[1];
                const double resiDiff = eig1 - eig2 - resiVal2;
                double gam = (resiDiff > 0) ? resiDiff : 1e-16;
                const double sq = resiVal1 * resiVal1 / gam;
                gam = resiVal1 < sq ? resiVal1 : sq;
                if (gam < 1e-03 || gam + eig1 <= 0.5) {
                    step = (gam + eig1 <= 0.0) ? INFINITY : 1.0 / (gam + eig1);
                    break;
                } else {
                    if (normPres == 0.0) return 1;
                    step = 1.0 / (gam + eig1);
                }
            }
        }
    }
    nComputed += 1;
    if (maxStep) *maxStep = step;
    if (steps) *steps = k;
    return 0;
}
