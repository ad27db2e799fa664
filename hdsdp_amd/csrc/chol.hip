// chol.hip -- blocked dense Cholesky / triangular inverse / triangular solves on gfx950.
//
// Replaces the reference's LAPACK dense-direct backend (linalg/hdsdp_linsolver.c:1044-1286:
// dpotrf :1096/:1126, dtrsm :1158/:1184, dpotrs :1210, diag :1227-1236, dpotri :1250) and the
// Cholesky fallback of the Schur solver (:1427 dpotrf(M), :1440 dpotrs).
//
// Right-looking blocked factorisation with NB = 128 (= the GEMM workgroup tile):
//   diag kernel  : one workgroup factors the 128x128 diagonal block in LDS and also inverts it
//   panel        : L[k+1:,k] = A[k+1:,k] * inv(L_kk)^T        (fp64 MFMA GEMM, in place)
//   trailing     : A[k+1:,k+1:] -= L[k+1:,k] L[k+1:,k]^T      (fp64 MFMA GEMM, lower tiles)
// The inverted diagonal blocks are kept: triangular solves become block GEMVs and the full
// triangular inverse Linv (needed by the congruence kernels) is assembled from them by GEMMs.
#include "hdm_common.h"
#include "chol.h"
#include "sweep128.h"
#include <algorithm>
#include <atomic>
#include <cstring>

#define NB 128

// ------------------------------------------------------------------------------------------
// diagonal block: Cholesky (lower) + triangular inverse of one 128 x 128 block by one workgroup.
//
// Blocked inside LDS with 32-wide panels so the latency-bound chain is short:
//   * the 32 x 32 diagonal sub-block is factored AND inverted entirely in registers by one wave
//     (lane l holds row l; pivots / multipliers are broadcast with v_readlane, no LDS, no barriers);
//   * panel  P = B * inv(L_pp)^T  and the trailing update  A22 -= P P^T  are small LDS matmuls
//     spread over all 256 threads;
//   * the full 128 x 128 inverse is then assembled block column by block column, right to left:
//     X[j+1:, j] = -X[j+1:, j+1:] * (L[j+1:, j] * X_jj).
// L goes to global memory as soon as a piece is final; the inverse is left in LDS and written once.
// ------------------------------------------------------------------------------------------
#define PB 32
#define LDW 98   // row stride of the inverse-assembly scratch W (98 * 8 B = 196 banks: 16 lanes -> 16 distinct bank pairs)
__device__ __forceinline__ double hdm_readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

typedef double hdm_c4 __attribute__((ext_vector_type(4)));

// Small matrix products between LDS-resident operands on the fp64 MFMA, shared by the four waves of the diagonal-block
// kernel:  C(m, n) (+)= alpha * sum_k A(m, k) * B(k, n),  A(m, k) = pa[m*ars + k*acs],  B(k, n) = pb[k*brs + n*bcs],
// C(m, n) = pc[m*crs + n*ccs];  M, N multiples of 16, K of 4.  Work is dealt in units of one 16-row strip with NT (1 or
// 2) adjacent 16-column tiles; a wave finishes reading a unit's operands before it writes the unit, so an output may
// alias the A operand as long as units own their rows (the in-place panel).  k_by_row: A is lower triangular relative
// to the unit's row offset (k <= row), the K loop stops at the strip's last row.  lower_only: units strictly above the
// diagonal of a square output are skipped.
// v_mfma_f64_16x16x4_f64(P, Q): lane (l15, lq) supplies P[l15][lq], Q[l15][lq]; it returns D[lq + 4r][l15] = sum_k P[lq+4r][k] Q[l15][k].
template <int NT>
__device__ __forceinline__ void hdm_lds_mm(double *pc, int crs, int ccs, const double *pa, int ars, int acs, const double *pb,
                                           int brs, int bcs, int M, int N, int K, double alpha, int accumulate, int k_by_row,
                                           int lower_only, int wave, int lane, double *gout, long gld) {
    const int l15 = lane & 15, lq = lane >> 4;
    const int mt = M >> 4, nt = (N >> 4) / NT, units = mt * nt;
    for (int u = wave; u < units; u += 4) {
        const int ti = u / nt, tj = (u % nt) * NT;
        if (lower_only && tj > ti) continue;
        hdm_c4 acc[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] = (hdm_c4){0.0, 0.0, 0.0, 0.0};
        const int kend = k_by_row ? min(K, (ti + 1) * 16) : K;
        const double *qa = pa + (ti * 16 + l15) * ars + lq * acs;
        const double *qb = pb + lq * brs + (tj * 16 + l15) * bcs;
        for (int k0 = 0; k0 < kend; k0 += 4) {
            const double av = qa[k0 * acs];
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(qb[k0 * brs + q * 16 * bcs], av, acc[q], 0, 0, 0);
        }
        // lane (l15, lq), register r of acc[q]:  C(ti*16 + l15, (tj + q)*16 + lq + 4r)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = ti * 16 + l15, n = (tj + q) * 16 + lq + 4 * r;
                double *c = pc + m * crs + n * ccs;
                const double v = (accumulate ? *c : 0.0) + alpha * acc[q][r];
                *c = v;
                if (gout) gout[m + (long) n * gld] = v;
            }
    }
}

__global__ __launch_bounds__(256) void hdm_potrf_diag_kernel(double *__restrict__ A, long ld, double *__restrict__ Dinv,
                                                              int *__restrict__ info, int col0) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *a = sm;             // [NB][NB] column-major: L below the diagonal blocks, inverses on them
    double *w = sm + NB * NB;   // [PB][LDW] scratch for the inverse assembly: W(k, c) = w[k + c * LDW]
    const int tid = threadIdx.x;
    // block load: 16 independent global loads in flight per thread (a conditional load per iteration would
    // serialise 64 global round trips)
    for (int e0 = tid; e0 < NB * NB; e0 += 256 * 16) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = e0 + q * 256;
            v[q] = A[(e & (NB - 1)) + (long) (e >> 7) * ld];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = e0 + q * 256;
            a[e] = ((e & (NB - 1)) >= (e >> 7)) ? v[q] : 0.0;
        }
    }
    __syncthreads();

    for (int p = 0; p < NB / PB; ++p) {
        const int c0 = p * PB;
        if (tid < 64) {
            // ---- wave 0: Cholesky + inverse of the 32 x 32 diagonal sub-block, REGISTER resident: lane l holds row l
            // of L (then column l of the inverse) in 32 named doubles, and the value every lane needs in a step
            // (L[k][j], a uniform) is broadcast with v_readlane and consumed at once as the scalar operand of an FMA.
            // No LDS traffic and no wave barriers inside the 2 x 496 update steps: 32 K cycles per sub-block against
            // about twice that for the LDS-resident lockstep version this replaces.  Two things keep the SGPR file from
            // overflowing: the 32 reciprocal pivots are forced into VGPRs, and the rows pass through an opaque asm
            // before the inverse so that its broadcasts are not CSE'd with (and kept alive since) the factor's.
            const int l = tid & 31;
            double *blk = a + c0 + c0 * NB;          // blk[i + k*NB] = element (i, k) of the sub-block
            double row[PB], rinv[PB];
#pragma unroll
            for (int k = 0; k < PB; ++k) row[k] = blk[l + k * NB];   // (entries above the diagonal are never used)
            int bad = 0;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                double d = hdm_readlane_f64(row[j], j);
                if (!(d > 0.0)) {                    // uniform; also catches NaN
                    if (!bad) bad = j + 1;
                    d = 1.0;
                }
                const double piv = sqrt(d);
                double ri = 1.0 / piv;
                asm volatile("" : "+v"(ri));
                rinv[j] = ri;
                row[j] = (l == j) ? piv : row[j] * ri;
#pragma unroll
                for (int k = j + 1; k < PB; ++k) row[k] -= row[j] * hdm_readlane_f64(row[j], k);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (bad && tid == 0) atomicCAS(info, 0, col0 + c0 + bad);
            if (tid < PB) {
#pragma unroll
                for (int k = 0; k < PB; ++k)
                    if (k <= l) A[(c0 + l) + (long) (c0 + k) * ld] = row[k];
            }
#pragma unroll
            for (int k = 0; k < PB; ++k) asm volatile("" : "+v"(row[k]));
            double x[PB];                            // lane c: column c of X = L^-1
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                double sacc = (l == i) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) sacc -= hdm_readlane_f64(row[k], i) * x[k];
                x[i] = (l <= i) ? sacc * rinv[i] : 0.0;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tid < PB) {
#pragma unroll
                for (int i = 0; i < PB; ++i) blk[i + l * NB] = x[i];   // X[i][l]; zeros above the diagonal
            }
        }
        __syncthreads();
        const int t0 = c0 + PB, nr = NB - t0;
        if (nr <= 0) break;
        const int wave = tid >> 6, lane = tid & 63;
        // ---- panel, in place:  P = B * X_pp^T  (B = rows t0.. of block column p; X_pp lower triangular with exact zeros
        // above its diagonal), written to LDS and to global memory.  Units are 16-row strips of both column tiles.
        hdm_lds_mm<2>(a + t0 + c0 * NB, 1, NB, a + t0 + c0 * NB, 1, NB, a + c0 + c0 * NB, NB, 1, nr, PB, PB, 1.0, 0, 0, 0, wave,
                      lane, A + t0 + (long) c0 * ld, ld);
        __syncthreads();
        // ---- trailing update:  A22 -= P P^T  on the lower 16 x 16 tiles
        hdm_lds_mm<1>(a + t0 + t0 * NB, 1, NB, a + t0 + c0 * NB, 1, NB, a + t0 + c0 * NB, NB, 1, nr, nr, PB, -1.0, 1, 0, 1, wave,
                      lane, nullptr, 0);
        __syncthreads();
    }
    // ---- inverse assembly (diagonal sub-blocks of `a` already hold their inverses), block column by block column,
    // right to left:  W = L[r0:, c0:c0+32] * X_jj  into the scratch (k-contiguous, row stride LDW: conflict-free as the
    // next product's B operand), then  X[r0:, c0:c0+32] = -X[r0:, r0:] * W  with the K loop cut at the row (lower triangular)
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int j = NB / PB - 2; j >= 0; --j) {
            const int c0 = j * PB, r0 = c0 + PB, nr = NB - r0;
            hdm_lds_mm<2>(w, 1, LDW, a + r0 + c0 * NB, 1, NB, a + c0 + c0 * NB, 1, NB, nr, PB, PB, 1.0, 0, 0, 0, wave, lane,
                          nullptr, 0);
            __syncthreads();
            hdm_lds_mm<2>(a + r0 + c0 * NB, 1, NB, a + r0 + r0 * NB, 1, NB, w, 1, LDW, nr, PB, nr, -1.0, 0, 1, 0, wave, lane,
                          nullptr, 0);
            __syncthreads();
        }
    }
    for (int e = tid; e < NB * NB; e += 256) Dinv[e] = a[e];
}

// The same contract on the register sweep of sweep128.h (round 2): the block lives in the registers of 512 threads, four
// pivots per barrier, the triangular inverse is carried along in the same pass.  113 -> about 55 us per block; the old
// kernel stays for A/B runs (HDM_DIAG_SWEEP=0).  A non-positive pivot is not patched here: everything after it turns into
// NaNs (the caller reads `info` and discards the factor).
#define DIAG_SWEEP_LDS_DOUBLES (SMALL_P * (SMALL_P + 1) + 2 * SMALL_P)
// nv: rows of this block that belong to the matrix (the rest is the identity padding, which needs no pivots: a 21 x 21 block
// -- truss1's largest -- is 6 four-pivot steps instead of 32, and the reference's driver factors such blocks thousands of times)
// SIGNED: the LDL' flavour of the sweep (sweep128.h) -- sgn_out[0 .. 127] receives the pivots' signs (+1 in the padding), *nneg_out
// the number of negative ones (added atomically), and only an exactly zero / non-finite pivot is reported through `info`
template <bool SIGNED = false>
__device__ __forceinline__ void hdm_potrf_diag_sweep_body(double *__restrict__ A, long ld, double *__restrict__ Dinv, int *__restrict__ info,
                                                          int col0, int nv, double *__restrict__ sgn_out = nullptr,
                                                          int *__restrict__ nneg_out = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *img = sm;                                  // 128 x 129 staging image; its head doubles as the sweep's block images
    double *rsv = sm + SMALL_P * (SMALL_P + 1);
    constexpr int ldi = SMALL_P + 1;
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const int ei = tid & (SMALL_P - 1), ej0 = tid >> 7;
#pragma unroll 8
    for (int j = ej0; j < SMALL_P; j += SM_T / SMALL_P) img[ei + j * ldi] = (ei >= j) ? A[ei + (long) j * ld] : 0.0;
    __syncthreads();
    double a[SM_NR][SM_NC], rr[SM_NR][SM_NC];
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) a[r][c] = img[(ty + 16 * r) + (tx + 32 * c) * ldi];
    __syncthreads();
    double unused;
    int nneg = 0;
    const int bad = sm_sweep<false, SIGNED>(nv, a, rr, img, rsv, ty, tx, &unused, &nneg);
    if (bad && tid == 0) atomicCAS(info, 0, col0 + bad);
    if (SIGNED) {
        if (tid < SMALL_P && sgn_out) sgn_out[tid] = (tid < nv && rsv[tid] < 0.0) ? -1.0 : 1.0;
        if (tid == 0 && nneg_out && nneg) atomicAdd(nneg_out, nneg);
    }
    __syncthreads();
    sm_store_one<true>(SMALL_P, a, img, A, ld, ty, tx, tid);
    sm_store_one<false>(SMALL_P, rr, img, Dinv, SMALL_P, ty, tx, tid);
}
__global__ __launch_bounds__(SM_T) void hdm_potrf_diag_sweep_kernel(double *__restrict__ A, long ld, double *__restrict__ Dinv,
                                                                    int *__restrict__ info, int col0, int nv) {
    hdm_potrf_diag_sweep_body(A, ld, Dinv, info, col0, nv);
}
// the same for a list of independent diagonal tiles of a block-sparse matrix (bsparse.hip): workgroup b factors the diagonal
// tile of block column cols[b]; a non-positive pivot is reported as the smallest failing row + 1 over the launch
// sgn != nullptr: the signed (LDL') sweep; sgn[128 k ..] receives block column k's pivot signs, info[1] counts negative pivots
template <bool SIGNED>
__global__ __launch_bounds__(SM_T) void hdm_potrf_diag_sweep_tiles_kernel(double *__restrict__ tiles, const int *__restrict__ diag_tile,
                                                                          const int *__restrict__ cols, double *__restrict__ Winv,
                                                                          int *__restrict__ info, int m, double *__restrict__ sgn) {
    const int k = cols[blockIdx.x];
    int dummy = 0;
    (void) dummy;
    double *A = tiles + ((long) diag_tile[k] << 14);
    // (atomicCAS keeps the FIRST reporter; with several workgroups the smallest row is wanted: report through atomicMin on a
    // word that starts at 0 = "none" is awkward, so a failing tile writes row + 1 only if the word is 0 or larger)
    __shared__ int linfo;
    if (threadIdx.x == 0) linfo = 0;
    __syncthreads();
    hdm_potrf_diag_sweep_body<SIGNED>(A, SMALL_P, Winv + ((long) k << 14), &linfo, k * SMALL_P, min(SMALL_P, m - k * SMALL_P),
                                      SIGNED ? sgn + (long) k * SMALL_P : nullptr, SIGNED ? info + 1 : nullptr);
    __syncthreads();
    if (threadIdx.x == 0 && linfo) {
        int old = atomicCAS(info, 0, linfo);
        while (old != 0 && linfo < old) { const int seen = atomicCAS(info, old, linfo); if (seen == old) break; old = seen; }
    }
}
int hdm_potrf_sweep_configure() {
    HDM_HIP_CHECK(hipFuncSetAttribute((const void *) hdm_potrf_diag_sweep_tiles_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      DIAG_SWEEP_LDS_DOUBLES * (int) sizeof(double)));
    HDM_HIP_CHECK(hipFuncSetAttribute((const void *) hdm_potrf_diag_sweep_tiles_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      DIAG_SWEEP_LDS_DOUBLES * (int) sizeof(double)));
    return 0;
}
// info: two words -- [0] first failing pivot + 1 (Cholesky: not positive; LDL': exactly zero or not a number), [1] negative pivots (LDL')
int hdm_potrf_sweep_batched(double *tiles, const int *diag_tile, const int *cols, int ncols, double *Winv, int *info, int m, hipStream_t s,
                            double *sgn) {
    if (ncols <= 0) return 0;
    if (sgn)
        hipLaunchKernelGGL(hdm_potrf_diag_sweep_tiles_kernel<true>, dim3(ncols), dim3(SM_T), DIAG_SWEEP_LDS_DOUBLES * sizeof(double), s, tiles,
                           diag_tile, cols, Winv, info, m, sgn);
    else
        hipLaunchKernelGGL(hdm_potrf_diag_sweep_tiles_kernel<false>, dim3(ncols), dim3(SM_T), DIAG_SWEEP_LDS_DOUBLES * sizeof(double), s, tiles,
                           diag_tile, cols, Winv, info, m, sgn);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// pad region of an (npad x npad) matrix whose valid part is n x n: identity on the diagonal
__global__ void hdm_pad_identity_kernel(double *A, long ld, int n, int npad) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    long tot = (long) npad * npad;
    if (e >= tot) return;
    int i = (int) (e % npad), j = (int) (e / npad);
    if (i >= n || j >= n) A[i + (long) j * ld] = (i == j) ? 1.0 : 0.0;
}

// W[i][j] = F[n-1-j][n-1-i] for i >= j (F = lower factor of the index-reversed matrix J X J = F F^T), zero above the
// diagonal, identity in the padding: W is lower triangular with W^T W = X.
__global__ void hdm_reverse_factor_kernel(const double *__restrict__ F, double *__restrict__ W, long ld, int n, int npad) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    long tot = (long) npad * npad;
    if (e >= tot) return;
    int i = (int) (e % npad), j = (int) (e / npad);
    double v;
    if (i >= n || j >= n) v = (i == j) ? 1.0 : 0.0;
    else v = (i >= j) ? F[(n - 1 - j) + (long) (n - 1 - i) * ld] : 0.0;
    W[i + (long) j * ld] = v;
}

// Linv diagonal blocks <- the inverted 128 x 128 diagonal blocks (blockIdx.y = block)
__global__ void hdm_copy_diag_blocks_kernel(const double *__restrict__ Dinv, double *__restrict__ Linv, long ld) {
    const int e = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    Linv[(long) k * NB * (ld + 1) + (e & (NB - 1)) + (long) (e >> 7) * ld] = Dinv[(long) k * NB * NB + e];
}

__global__ void hdm_copy_block_kernel(const double *__restrict__ src, long lds_, double *__restrict__ dst, long ldd,
                                      int rows, int cols) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long) rows * cols) return;
    int i = (int) (e % rows), j = (int) (e / rows);
    dst[i + (long) j * ldd] = src[i + (long) j * lds_];
}

__global__ void hdm_get_diag_kernel(const double *__restrict__ A, long ld, int n, double *__restrict__ d) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = A[i + (long) i * ld];
}

// ------------------------------------------------------------------------------------------
// block substitution steps (one launch per 128-block; vectors are short, L is streamed once)
//   forward : y_k = Dinv_k b_k ;  b_i -= L[i,k] y_k  (i > k)
//   backward: x_k = Dinv_k^T y_k ; y_i -= L[k,i]^T x_k (i < k)
// Every workgroup recomputes the small 128x128 matvec (L2-resident) so no inter-workgroup
// hand-off is needed inside a launch.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hdm_trsv_fwd_step(const double *__restrict__ L, long ld,
                                                          const double *__restrict__ Dinv, double *__restrict__ b,
                                                          double *__restrict__ x, int k, int nblk, int nrhs, long ldv) {
    __shared__ double yk[NB];
    __shared__ double part[256];
    const int tid = threadIdx.x;
    const int bi = k + blockIdx.x;  // block row handled by this workgroup (>= k)
    const int rhs = blockIdx.y;
    double *bb = b + (long) rhs * ldv, *xx = x + (long) rhs * ldv;
    const double *D = Dinv + (long) k * NB * NB;
    // y_k[r] = sum_{c<=r} D[r,c] b_k[c]; two threads per row
    {
        int r = tid & (NB - 1), h = tid >> 7;
        double s = 0.0;
        for (int c = h; c <= r; c += 2) s += D[r + c * NB] * bb[k * NB + c];
        part[tid] = s;
        __syncthreads();
        if (tid < NB) yk[tid] = part[tid] + part[tid + NB];
        __syncthreads();
    }
    if (bi == k) {
        if (tid < NB) xx[k * NB + tid] = yk[tid];
        return;
    }
    if (bi >= nblk) return;
    {
        int r = tid & (NB - 1), h = tid >> 7;
        const double *Lb = L + (long) bi * NB + (long) k * NB * ld;
        double s = 0.0;
        for (int c = h * 64; c < h * 64 + 64; ++c) s += Lb[r + (long) c * ld] * yk[c];
        part[tid] = s;
        __syncthreads();
        if (tid < NB) bb[bi * NB + tid] -= part[tid] + part[tid + NB];
    }
}

__global__ __launch_bounds__(256) void hdm_trsv_bwd_step(const double *__restrict__ L, long ld,
                                                          const double *__restrict__ Dinv, double *__restrict__ y,
                                                          double *__restrict__ x, int k, int nrhs, long ldv) {
    __shared__ double xk[NB];
    __shared__ double part[256];
    const int tid = threadIdx.x;
    const int bi = (int) blockIdx.x;  // block handled (<= k); blockIdx.x == k computes/stores x_k
    const int rhs = blockIdx.y;
    double *yy = y + (long) rhs * ldv, *xx = x + (long) rhs * ldv;
    const double *D = Dinv + (long) k * NB * NB;
    {
        // x_k[c] = sum_{r>=c} D[r,c] y_k[r]
        int c = tid & (NB - 1), h = tid >> 7;
        double s = 0.0;
        for (int r = c + h; r < NB; r += 2) s += D[r + c * NB] * yy[k * NB + r];
        part[tid] = s;
        __syncthreads();
        if (tid < NB) xk[tid] = part[tid] + part[tid + NB];
        __syncthreads();
    }
    if (bi == k) {
        if (tid < NB) xx[k * NB + tid] = xk[tid];
        return;
    }
    {
        // y_bi[c] -= sum_r L[k*NB + r, bi*NB + c] * x_k[r] : each wave owns 32 columns, lanes stride r
        const double *Lb = L + (long) k * NB + (long) bi * NB * ld;
        const int lane = tid & 63, wave = tid >> 6;
        for (int cc = 0; cc < 32; ++cc) {
            int c = wave * 32 + cc;
            double s = Lb[lane + (long) c * ld] * xk[lane] + Lb[lane + 64 + (long) c * ld] * xk[lane + 64];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) yy[bi * NB + c] -= s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same block substitution as ONE launch: workgroup i owns block row i and consumes the solved blocks of the others
// as they are published (a flag per block, release/acquire at agent scope), so the chain costs one hand-off per block
// instead of one kernel launch per block (n = 2000: 32 launches -> 1).
//   forward : acc = b_i - sum_{k<i} L[i,k] y_k (k ascending) ;  y_i = Dinv_i acc          -> published in `y`
//   backward: acc = y_i - sum_{k>i} L[k,i]^T x_k (k descending) ;  x_i = Dinv_i^T acc     -> published in `x`
// The accumulation order is fixed, so the result does not depend on timing.  Forward waits only on lower-numbered
// workgroups (dispatched earlier), backward on higher-numbered ones, which the host guarantees to be co-resident
// (nblk * nrhs <= 256 workgroups on 256 CUs); every wait is bounded -- a workgroup that gives up raises *err and the
// host repeats the solve with the per-block launches.  `epoch` distinguishes this launch's flags from older ones.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool hdm_flow_wait(const int *flag, int epoch) {
    for (int it = 0; it < (1 << 22); ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// out[r] = sum_c Mb[r + c*ld] * v[c] over the 128 x 128 block, two threads per row (the halves meet in `part`); 16 loads
// in flight per thread -- a load, wait, multiply loop would spend the whole hand-off in memory latency
__device__ __forceinline__ double hdm_row_dot_half(const double *__restrict__ Mb, long ld, const double *v, int tid) {
    const int r = tid & (NB - 1), h = tid >> 7;
    const double *q = Mb + r + (long) (h * 64) * ld;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 16
    for (int c = 0; c < 64; c += 2) {
        s0 += q[(long) c * ld] * v[h * 64 + c];
        s1 += q[(long) (c + 1) * ld] * v[h * 64 + c + 1];
    }
    return s0 + s1;
}
// out[c] -= (or =) sum_r Mb[r + c*ld] * v[r]: lanes run along r (coalesced), each wave owns 32 columns, eight columns
// (16 loads) in flight, reductions by shuffles
template <bool SUBTRACT>
__device__ __forceinline__ void hdm_col_dots(const double *__restrict__ Mb, long ld, const double *v, double *out, int wave,
                                             int lane) {
    const double v0 = v[lane], v1 = v[lane + 64];
    for (int cc = 0; cc < 32; cc += 8) {
        double sacc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double *col = Mb + (long) (wave * 32 + cc + q) * ld;
            sacc[q] = col[lane] * v0 + col[lane + 64] * v1;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int q = 0; q < 8; ++q) sacc[q] += __shfl_down(sacc[q], off, 64);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (SUBTRACT) out[wave * 32 + cc + q] -= sacc[q];
                else out[wave * 32 + cc + q] = sacc[q];
            }
        }
    }
}

__global__ __launch_bounds__(256) void hdm_trsv_flow_kernel(const double *__restrict__ L, long ld,
                                                             const double *__restrict__ Dinv, const double *b, double *y,
                                                             double *x, int nblk, long ldv, int *flags, int epoch,
                                                             int which, volatile int *err, const int *__restrict__ env) {
    __shared__ double v[NB];
    __shared__ double acc[NB];
    __shared__ double part[256];
    __shared__ int ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x, rhs = blockIdx.y;
    const double *bb = b + (long) rhs * ldv;
    double *yy = y + (long) rhs * ldv, *xx = x + (long) rhs * ldv;
    int *ff = flags + (long) rhs * 2 * nblk, *fb = ff + nblk;
    const double *D = Dinv + (long) i * NB * NB;    // lower triangular, explicit zeros above the diagonal
    if (tid < NB) acc[tid] = bb[i * NB + tid];
    if (tid == 0) ok = 1;
    __syncthreads();
    // (block envelope, if any: block row i has nothing left of block column env[i], block column i nothing below row env[nblk + i])
    const int kfirst = env ? env[i] : 0, klast = env ? env[nblk + i] : nblk - 1;
    if (which != 2) {
        for (int k = kfirst; k < i; ++k) {
            if (tid == 0 && !hdm_flow_wait(ff + k, epoch)) ok = 0;
            __syncthreads();
            if (!ok) { if (tid == 0) *err = 1; return; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (tid < NB) v[tid] = __builtin_nontemporal_load(yy + k * NB + tid);
            __syncthreads();
            part[tid] = hdm_row_dot_half(L + (long) i * NB + (long) k * NB * ld, ld, v, tid);
            __syncthreads();
            if (tid < NB) acc[tid] -= part[tid] + part[tid + NB];
            __syncthreads();
        }
        part[tid] = hdm_row_dot_half(D, NB, acc, tid);     // y_i = Dinv_i acc
        __syncthreads();
        if (tid < NB) {
            const double val = part[tid] + part[tid + NB];
            acc[tid] = val;
            yy[i * NB + tid] = val;
            if (which == 1) xx[i * NB + tid] = val;
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ff + i, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (which == 1) return;
    }
    for (int k = klast; k > i; --k) {
        if (tid == 0 && !hdm_flow_wait(fb + k, epoch)) ok = 0;
        __syncthreads();
        if (!ok) { if (tid == 0) *err = 1; return; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (tid < NB) v[tid] = __builtin_nontemporal_load(xx + k * NB + tid);
        __syncthreads();
        hdm_col_dots<true>(L + (long) k * NB + (long) i * NB * ld, ld, v, acc, wave, lane);   // acc -= L[k,i]^T x_k
        __syncthreads();
    }
    hdm_col_dots<false>(D, NB, acc, part, wave, lane);      // x_i = Dinv_i^T acc
    __syncthreads();
    if (tid < NB) xx[i * NB + tid] = part[tid];
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(fb + i, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// diagnostic: average duration (us, HIP events) of `reps` launches of one diagonal-block kernel on a 128 x 128 SPD block
__global__ void hdm_probe_spd_block_kernel(double *A, long ld) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = e & (NB - 1), j = e >> 7;
    if (j < NB) A[i + (long) j * ld] = (i == j) ? 4.0 + 0.01 * i : 1.0 / (1.0 + abs(i - j));
}
double hdm_diag_block_probe(int variant, int reps, hipStream_t s) {
    double *A = nullptr, *D = nullptr;
    int *info = nullptr;
    hipEvent_t e0, e1;
    if (hipMalloc((void **) &A, sizeof(double) * NB * NB) != hipSuccess || hipMalloc((void **) &D, sizeof(double) * NB * NB) != hipSuccess ||
        hipMalloc((void **) &info, sizeof(int)) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return -1.0;
    (void) hipFuncSetAttribute((const void *) hdm_potrf_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (NB * NB + LDW * PB) * (int) sizeof(double));
    (void) hipFuncSetAttribute((const void *) hdm_potrf_diag_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_SWEEP_LDS_DOUBLES * (int) sizeof(double));
    (void) hipMemsetAsync(info, 0, sizeof(int), s);
    float ms = 0.f, total = 0.f;
    for (int r = -2; r < reps; ++r) {
        hipLaunchKernelGGL(hdm_probe_spd_block_kernel, dim3(NB * NB / 256), dim3(256), 0, s, A, (long) NB);
        (void) hipEventRecord(e0, s);
        if (variant)
            hipLaunchKernelGGL(hdm_potrf_diag_sweep_kernel, dim3(1), dim3(SM_T), DIAG_SWEEP_LDS_DOUBLES * sizeof(double), s, A, (long) NB, D, info, 0, NB);
        else
            hipLaunchKernelGGL(hdm_potrf_diag_kernel, dim3(1), dim3(256), (NB * NB + LDW * PB) * sizeof(double), s, A, (long) NB, D, info, 0);
        (void) hipEventRecord(e1, s);
        (void) hipEventSynchronize(e1);
        (void) hipEventElapsedTime(&ms, e0, e1);
        if (r >= 0) total += ms;
    }
    (void) hipFree(A); (void) hipFree(D); (void) hipFree(info); (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    return total / reps * 1e3;
}

// ------------------------------------------------------------------------------------------
// host drivers
// ------------------------------------------------------------------------------------------
int HdmChol::init(int n_) {
    n = n_;
    npad = (int) hdm_roundup(n, NB);
    nblk = npad / NB;
    size_t mat = sizeof(double) * (size_t) npad * npad;
    HDM_HIP_CHECK(hipMalloc((void **) &L, mat));
    HDM_HIP_CHECK(hipMalloc((void **) &Dinv, sizeof(double) * (size_t) nblk * NB * NB));
    HDM_HIP_CHECK(hipMalloc((void **) &info_dev, sizeof(int)));
    HDM_HIP_CHECK(hipMalloc((void **) &vec, sizeof(double) * (size_t) npad * 4));
    HDM_HIP_CHECK(hdm_memset_sync(L, 0, mat));
    HDM_HIP_CHECK(hipFuncSetAttribute((const void *) hdm_potrf_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (NB * NB + LDW * PB) * (int) sizeof(double)));
    HDM_HIP_CHECK(hipFuncSetAttribute((const void *) hdm_potrf_diag_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      DIAG_SWEEP_LDS_DOUBLES * (int) sizeof(double)));
    return 0;
}

int HdmChol::set_envelope(const int *first) {
    env_first.clear(); env_colh.clear();
    if (env_dev) { (void) hipFree(env_dev); env_dev = nullptr; }
    if (factor_graph) { (void) hipGraphExecDestroy(factor_graph); factor_graph = nullptr; factor_runs = 0; }
    if (!first || nblk <= 1) return 0;
    env_first.assign(first, first + nblk);
    env_colh.assign(nblk, 0);
    for (int i = 0; i < nblk; ++i) {
        env_first[i] = std::max(0, std::min(env_first[i], i));
        if (i > 0) env_first[i] = std::min(env_first[i], i);   // (a row's envelope always reaches its diagonal block)
    }
    // colh[k] = max { i : first[i] <= k }: a running maximum over the rows, taken from the bottom
    for (int k = 0; k < nblk; ++k) env_colh[k] = k;
    for (int i = 0; i < nblk; ++i)
        for (int k = env_first[i]; k <= i; ++k) env_colh[k] = std::max(env_colh[k], i);
    for (int k = 1; k < nblk; ++k) env_colh[k] = std::max(env_colh[k], env_colh[k - 1]);   // (fill: the envelope's lower edge never rises)
    std::vector<int> both(env_first);
    both.insert(both.end(), env_colh.begin(), env_colh.end());
    HDM_HIP_CHECK(hipMalloc((void **) &env_dev, sizeof(int) * both.size()));
    HDM_HIP_CHECK(hdm_memcpy_h2d_sync(env_dev, both.data(), sizeof(int) * both.size()));
    return 0;
}

void HdmChol::destroy() {
    if (env_dev) (void) hipFree(env_dev);
    env_dev = nullptr;
    if (L) (void) hipFree(L);
    if (Linv) (void) hipFree(Linv);
    if (Dinv) (void) hipFree(Dinv);
    if (Z) (void) hipFree(Z);
    if (Zd) (void) hipFree(Zd);
    if (info_dev) (void) hipFree(info_dev);
    if (vec) (void) hipFree(vec);
    if (flow_flags) (void) hipFree(flow_flags);
    if (flow_err) (void) hipHostFree(flow_err);
    flow_flags = nullptr; flow_err = nullptr;
    if (factor_graph) (void) hipGraphExecDestroy(factor_graph);
    for (int i = 0; i < nsolves; ++i) if (solves[i].exec) (void) hipGraphExecDestroy(solves[i].exec);
    factor_graph = nullptr; nsolves = 0;
    L = Linv = Dinv = Z = Zd = vec = nullptr;
    info_dev = nullptr;
}

int HdmChol::load_host(const double *A, long lda, hipStream_t s) {
    // host n x n column-major (lower triangle valid) -> device L buffer, identity padded
    HDM_HIP_CHECK(hipMemcpy2DAsync(L, sizeof(double) * npad, A, sizeof(double) * lda, sizeof(double) * n, n,
                                   hipMemcpyHostToDevice, s));
    return finish_load(s);
}

int HdmChol::load_device(const double *A, long lda, hipStream_t s) {
    HDM_HIP_CHECK(hipMemcpy2DAsync(L, sizeof(double) * npad, A, sizeof(double) * lda, sizeof(double) * n, n,
                                   hipMemcpyDeviceToDevice, s));
    return finish_load(s);
}

int HdmChol::finish_load(hipStream_t s) {
    if (npad != n) {
        long tot = (long) npad * npad;
        hipLaunchKernelGGL(hdm_pad_identity_kernel, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, s, L,
                           (long) npad, n, npad);
        HDM_HIP_CHECK(hipGetLastError());
    }
    factored = false;
    have_inv = false;
    logdet_ok = false;
    return 0;
}

int HdmChol::set_reverse_inverse(hipStream_t s) {
    // valid after factor() of the index-reversed matrix; afterwards Linv holds W (see the kernel) and L is unchanged
    if (!factored) return 1;
    if (!Linv) HDM_HIP_CHECK(hipMalloc((void **) &Linv, sizeof(double) * (size_t) npad * npad));
    long tot = (long) npad * npad;
    hipLaunchKernelGGL(hdm_reverse_factor_kernel, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, s, L, Linv,
                       (long) npad, n, npad);
    HDM_HIP_CHECK(hipGetLastError());
    have_inv = true;
    return 0;
}

// capture `body` (launches on stream s with fixed arguments) into an executable graph; nullptr if that is not possible
template <class F> static hipGraphExec_t hdm_capture(hipStream_t s, F body) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    hdm_gemm_capture_mode(1);
    const int rc = body();
    hdm_gemm_capture_mode(0);
    const hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc || e != hipSuccess || !graph) { (void) hipGetLastError(); if (graph) (void) hipGraphDestroy(graph); return nullptr; }
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void) hipGetLastError(); exec = nullptr; }
    (void) hipGraphDestroy(graph);
    return exec;
}
// HDM_GRAPHS: 0 (default since round 4) = never; 1 = the factorisation chain when it is long enough (>= 4 diagonal blocks);
// 2 = also the block substitutions.  When replay was introduced (round 1) the chain at n = 2000 went 2.86 -> 2.71 ms; since the
// diagonal-block kernel halved (round 2) the chain is sixteen dependent 56 us sweeps plus launch-to-launch latency and replay
// buys nothing (1.859 ms replayed, 1.855 ms eager, n = 4000: 4.006 / 3.975 -- profiles/r04_e_cholesky.txt), the substitutions
// never gained (no gain at 16 blocks, +8 us per solve at one block), and replay has cost twice: a replayed MEMSET node that wrote
// stale bytes (profiles/r04_c_poison.txt) and rocprofiler-sdk 7.2 dying in its queue interceptor on multi-packet submissions
// (profiles/r04_a_headline_segv.txt).  The paths stay (tests run levels 1 and 2).
static int hdm_graph_level() {
    static int lvl = -1;
    if (lvl < 0) { const char *e = getenv("HDM_GRAPHS"); lvl = e ? atoi(e) : 0; }
    return lvl;
}

int HdmChol::factor(hipStream_t s, int *info_host) {
    if (hdm_graph_level() >= 1 && nblk >= 4 && graphs_ok && !factor_graph && factor_runs >= 1) {
        factor_graph = hdm_capture(s, [&]() { return enqueue_factor(s); });
        if (!factor_graph) graphs_ok = false;
    }
    if (factor_graph) HDM_HIP_CHECK(hipGraphLaunch(factor_graph, s));
    else if (enqueue_factor(s)) return 1;
    ++factor_runs;
    int info = 0;
    HDM_HIP_CHECK(hipMemcpyAsync(&info, info_dev, sizeof(int), hipMemcpyDeviceToHost, s));
    HDM_HIP_CHECK(hipStreamSynchronize(s));
    if (info > n) info = 0;  // failures inside the identity padding cannot happen; be safe
    if (info_host) *info_host = info;
    factored = (info == 0);
    have_inv = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The two products of a block column -- panel L_ik = A_ik inv(L_kk)^T and trailing update A_ij -= L_ik L_jk^T -- have K = 128:
// eight stages of the general GEMM's 128 x 128 tile, 14 us of MFMAs for ONE workgroup per tile behind a prologue and an
// epilogue, 22-24 us per launch with most of the chip idle (the first update of a 2000 x 2000 matrix has 120 tiles for 256
// CUs), twice per block column beside the 56 us diagonal sweep.  hdm_k128_kernel cuts the same products into 64 x 64 tiles:
// four waves of 32 x 32 (four accumulators, 128 MFMAs each = 3.4 us), fragments straight from global memory (both operands
// are M-major: an MFMA operand is one 8-byte buffer load per lane, sixteen k-steps in flight per wave, addresses on the scalar
// unit), no LDS, no barrier.  Same products in the same order of k as the general kernel: the factor is bit-identical
// (tests/test_gpu_kernels.py::test_cholesky_small_tile_products_give_the_same_bits).  HDM_CHOL_K128=0: the general kernel.
// ---------------------------------------------------------------------------------------------
typedef unsigned hdm_k2 __attribute__((ext_vector_type(2)));
// UPDATE: C(lower 64 x 64 tiles) -= A B^T, a wave owns 32 x 32 (NJ = 2 column sub-tiles).  Panel (!UPDATE, NJ = 4): C = A B^T
// with 64 x 128 tiles -- a workgroup owns ALL 128 columns of its 64 rows, so the product may overwrite its own A operand
// (C == A): every load of the workgroup is complete before its first store (the barrier below).
template <int NJ, bool UPDATE>
__global__ __launch_bounds__(256) void hdm_k128_kernel(const double *__restrict__ A, long lda, const double *__restrict__ B, long ldb,
                                                       double *C, long ldc) {
    constexpr int KB = (NJ == 2) ? 8 : 4;     // k-steps per block; two blocks in flight
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (UPDATE && ti < tj) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int r0 = ti * 64 + (wave & 1) * 32, c0 = tj * (32 * NJ) + (wave >> 1) * (16 * NJ);
    // operand element (row r0 + 16 i + l15, k + lq): base + lane offset + 128 i bytes, k advanced through the scalar offset
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A + r0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(B + c0), 0, 0x7fffffff, 0x00020000);
    const unsigned va = (unsigned) ((l15 + lq * lda) * 8), vb = (unsigned) ((l15 + lq * ldb) * 8);
    const unsigned sa = (unsigned) (4 * lda * 8), sb = (unsigned) (4 * ldb * 8);     // bytes per k-step
    hdm_c4 acc[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = (hdm_c4){0.0, 0.0, 0.0, 0.0};
    hdm_k2 fa[2][KB][NJ], fb[2][KB][2];   // [set][k-step of the block][sub-tile]
#define K128_LOAD(SET, BLK)                                                                                      \
    _Pragma("unroll") for (int q = 0; q < KB; ++q) {                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            fb[SET][q][i] = __builtin_amdgcn_raw_buffer_load_b64(ra, va + 128 * i, ((BLK) * KB + q) * sa, 0);    \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                           \
            fa[SET][q][j] = __builtin_amdgcn_raw_buffer_load_b64(rb, vb + 128 * j, ((BLK) * KB + q) * sb, 0);    \
    }
#define K128_MMA(SET)                                                                                            \
    _Pragma("unroll") for (int q = 0; q < KB; ++q)                                                               \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                           \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                        \
                acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(__builtin_bit_cast(double, fa[SET][q][j]),      \
                                                                 __builtin_bit_cast(double, fb[SET][q][i]), acc[j][i], 0, 0, 0);
    constexpr int NBLK = 32 / KB;             // K = 128 = 32 k-steps
    K128_LOAD(0, 0) K128_LOAD(1, 1)
#pragma unroll
    for (int b = 0; b < NBLK; b += 2) {
        K128_MMA(0)
        if (b + 2 < NBLK) { K128_LOAD(0, b + 2) }
        K128_MMA(1)
        if (b + 3 < NBLK) { K128_LOAD(1, b + 3) }
    }
#undef K128_LOAD
#undef K128_MMA
    if (!UPDATE) __syncthreads();             // in-place panel: nobody stores before everybody has loaded
    // lane l, register r of acc[j][i]: C[r0 + 16 i + l15][c0 + 16 j + lq + 4 r]
    double *cl = C + (r0 + l15) + (long) (c0 + lq) * ldc;
    const bool diag = UPDATE && (ti == tj);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                double *q = cl + 16 * i + (long) (16 * j + 4 * r) * ldc;
                if (UPDATE) {
                    if (diag && (r0 + 16 * i + l15) < (c0 + 16 * j + lq + 4 * r)) continue;   // above the diagonal: not ours
                    *q = -1.0 * acc[j][i][r] + 1.0 * (*q);
                } else {
                    *q = 1.0 * acc[j][i][r];
                }
            }
}

// the rows x 128 panel times a 128 x 128 block (in place allowed), or the lower tiles of the rows x rows update; rows a multiple of 64
static int hdm_k128_launch(bool update, const double *A, long lda, const double *B, long ldb, double *C, long ldc, int rows,
                           hipStream_t s) {
    if (update) hipLaunchKernelGGL((hdm_k128_kernel<2, true>), dim3(rows / 64, rows / 64), dim3(256), 0, s, A, lda, B, ldb, C, ldc);
    else hipLaunchKernelGGL((hdm_k128_kernel<4, false>), dim3(rows / 64, 1), dim3(256), 0, s, A, lda, B, ldb, C, ldc);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void hdm_zero_word_kernel(int *p) { *p = 0; }

int HdmChol::enqueue_factor(hipStream_t s) {
    // (a kernel, not hipMemsetAsync: this chain is captured into a hipGraph, and under HDM_POISON a replay of the captured
    // MEMSET node was seen to leave 0xFFFFFFFF in the word -- profiles/r04_c_poison.txt; a kernel node behaves like its neighbours)
    hipLaunchKernelGGL(hdm_zero_word_kernel, dim3(1), dim3(1), 0, s, info_dev);
    HDM_HIP_CHECK(hipGetLastError());   // a launch that did not happen would leave the previous chain's info word in place
    const long ld = npad;
    const size_t shm = (NB * NB + LDW * PB) * sizeof(double);
    static const bool diag_sweep = [] { const char *e = getenv("HDM_DIAG_SWEEP"); return !(e && atoi(e) == 0); }();
    for (int k = 0; k < nblk; ++k) {
        double *Akk = L + (long) k * NB * (ld + 1);
        if (diag_sweep)
            hipLaunchKernelGGL(hdm_potrf_diag_sweep_kernel, dim3(1), dim3(SM_T), DIAG_SWEEP_LDS_DOUBLES * sizeof(double), s, Akk, ld,
                               Dinv + (long) k * NB * NB, info_dev, k * NB, std::max(1, std::min(NB, n - k * NB)));
        else
            hipLaunchKernelGGL(hdm_potrf_diag_kernel, dim3(1), dim3(256), shm, s, Akk, ld, Dinv + (long) k * NB * NB,
                               info_dev, k * NB);
        HDM_HIP_CHECK(hipGetLastError());
        if (npad - (k + 1) * NB <= 0) break;
        // (with a block envelope the column ends at block row env_colh[k]: everything below is, and stays, zero)
        const int rows = env_colh.empty() ? npad - (k + 1) * NB : (env_colh[k] - k) * NB;
        if (rows <= 0) continue;
        double *P = Akk + NB;  // panel below the diagonal block
        static const bool k128 = [] { const char *e = getenv("HDM_CHOL_K128"); return !(e && atoi(e) == 0); }();
        if (k128 && rows % 64 == 0 && ld < (1L << 20)) {
            if (hdm_k128_launch(false, P, ld, Dinv + (long) k * NB * NB, NB, P, ld, rows, s)) return 1;
            if (hdm_k128_launch(true, P, ld, P, ld, Akk + (long) NB * (ld + 1), ld, rows, s)) return 1;
            continue;
        }
        HdmGemmArgs g = {};
        g.A = P; g.lda = ld; g.B = Dinv + (long) k * NB * NB; g.ldb = NB; g.C = P; g.ldc = ld;
        g.M = rows; g.N = NB; g.K = NB; g.batch = 1; g.alpha = 1.0; g.beta = 0.0;
        g.epilogue = HDM_EPI_STORE;
        if (hdm_launch_gemm(g, s)) return 1;
        HdmGemmArgs u = {};
        u.A = P; u.lda = ld; u.B = P; u.ldb = ld; u.C = Akk + (long) NB * (ld + 1); u.ldc = ld;
        u.M = rows; u.N = rows; u.K = NB; u.batch = 1; u.alpha = -1.0; u.beta = 1.0;
        u.lower_only = 1; u.epilogue = HDM_EPI_STORE;
        if (hdm_launch_gemm(u, s)) return 1;
    }
    return 0;
}

int HdmChol::invert_factor(hipStream_t s) {
    // Linv = L^{-1} (lower, explicit zeros above the diagonal), block columns right to left:
    //   Linv[k,k] = Dinv_k ;  Linv[k+1:,k] = -Linv[k+1:,k+1:] * (L[k+1:,k] * Dinv_k)
    if (have_inv) return 0;
    const long ld = npad;
    size_t mat = sizeof(double) * (size_t) npad * npad;
    if (!Linv) HDM_HIP_CHECK(hipMalloc((void **) &Linv, mat));
    if (!Z && nblk > 1) HDM_HIP_CHECK(hipMalloc((void **) &Z, sizeof(double) * (size_t) npad * NB));
    HDM_HIP_CHECK(hipMemsetAsync(Linv, 0, mat, s));
    if (nblk >= 2 && (nblk & (nblk - 1)) == 0) {
        // Power-of-two block count: recursive doubling instead of the right-to-left sweep.  With the 128-blocks
        // inverted (Dinv), [[A, 0], [C, B]]^-1 = [[A^-1, 0], [-B^-1 C A^-1, B^-1]] doubles the inverted block size per
        // level, and all pairs of a level are independent: 2 batched GEMM launches per level, log2(nblk) levels
        // (8 launches at n = 2000 instead of 30 dependent ones; 2.65 -> ~0.5 ms).
        if (!Zd) HDM_HIP_CHECK(hipMalloc((void **) &Zd, mat));
        hipLaunchKernelGGL(hdm_copy_diag_blocks_kernel, dim3(NB * NB / 256, nblk), dim3(256), 0, s, Dinv, Linv, ld);
        HDM_HIP_CHECK(hipGetLastError());
        for (long sz = NB; sz < npad; sz *= 2) {
            const int pairs = (int) (npad / (2 * sz));
            const long pstride = 2 * sz * (ld + 1);
            HdmGemmArgs g = {};   // T = C * A^-1   (B operand element (j,k) = A^-1(k,j): K-major)
            g.A = L + sz; g.lda = ld; g.strideA = pstride;
            g.B = Linv; g.ldb = ld; g.b_kmajor = 1; g.strideB = pstride;
            g.C = Zd + sz; g.ldc = ld; g.strideC = pstride;
            g.M = (int) sz; g.N = (int) sz; g.K = (int) sz; g.batch = pairs; g.alpha = 1.0; g.epilogue = HDM_EPI_STORE;
            if (hdm_launch_gemm(g, s)) return 1;
            HdmGemmArgs h = {};   // X = -B^-1 * T   (B^-1 lower triangular: K loop cut by the row tile)
            h.A = Linv + sz * (ld + 1); h.lda = ld; h.strideA = pstride;
            h.B = Zd + sz; h.ldb = ld; h.b_kmajor = 1; h.strideB = pstride;
            h.C = Linv + sz; h.ldc = ld; h.strideC = pstride;
            h.M = (int) sz; h.N = (int) sz; h.K = (int) sz; h.batch = pairs; h.alpha = -1.0;
            h.klimit = HDM_KLIM_BY_M; h.epilogue = HDM_EPI_STORE;
            if (hdm_launch_gemm(h, s)) return 1;
        }
        have_inv = true;
        return 0;
    }
    for (int k = nblk - 1; k >= 0; --k) {
        double *Xkk = Linv + (long) k * NB * (ld + 1);
        hipLaunchKernelGGL(hdm_copy_block_kernel, dim3(NB * NB / 256), dim3(256), 0, s, Dinv + (long) k * NB * NB,
                           (long) NB, Xkk, ld, NB, NB);
        HDM_HIP_CHECK(hipGetLastError());
        const int rows = npad - (k + 1) * NB;
        if (rows <= 0) continue;
        const double *P = L + (long) k * NB * (ld + 1) + NB;
        HdmGemmArgs g = {};  // Z = P * Dinv_k   (B operand K-major: Bop[j,kk] = Dinv_k[kk + j*NB])
        g.A = P; g.lda = ld; g.B = Dinv + (long) k * NB * NB; g.ldb = NB; g.b_kmajor = 1;
        g.C = Z; g.ldc = npad; g.M = rows; g.N = NB; g.K = NB; g.batch = 1; g.alpha = 1.0; g.epilogue = HDM_EPI_STORE;
        if (hdm_launch_gemm(g, s)) return 1;
        HdmGemmArgs h = {};  // X = -W * Z, W lower triangular => K loop cut by the row tile
        h.A = Linv + (long) (k + 1) * NB * (ld + 1); h.lda = ld; h.B = Z; h.ldb = npad; h.b_kmajor = 1;
        h.C = Xkk + NB; h.ldc = ld; h.M = rows; h.N = NB; h.K = rows; h.batch = 1; h.alpha = -1.0;
        h.klimit = HDM_KLIM_BY_M; h.epilogue = HDM_EPI_STORE;
        if (hdm_launch_gemm(h, s)) return 1;
    }
    have_inv = true;
    return 0;
}

int HdmChol::get_diag(double *diag_host, hipStream_t s) {
    hipLaunchKernelGGL(hdm_get_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, s, L, (long) npad, n, vec);
    HDM_HIP_CHECK(hipGetLastError());
    HDM_HIP_CHECK(hipMemcpyAsync(diag_host, vec, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    HDM_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
}

// HDM_TRSV_FLOW=0: always the per-block launches.  Also off while several engine shards share one device (a multi-shard
// rehearsal on one GPU, group_impl.h): the single-launch substitution spins on flags and needs all its workgroups
// co-resident, which nobody can promise when other streams of the same process compete for the CUs.
static std::atomic<int> g_flow_shared_device{0};
void hdm_flow_set_shared_device(int on) { g_flow_shared_device.store(on); }
bool hdm_flow_shared_device() { return g_flow_shared_device.load() != 0; }
static bool hdm_flow_enabled() {
    static int on = -1;
    if (on < 0) { const char *e = getenv("HDM_TRSV_FLOW"); on = (e && atoi(e) == 0) ? 0 : 1; }
    return on == 1 && g_flow_shared_device.load() == 0;
}

int HdmChol::solve_device(double *b_dev, double *x_dev, int nrhs, long ldv, int which, hipStream_t s) {
    // replay the launch chain of an earlier call with the same buffers when there is one (the C-ABI solves always
    // come through the object's own scratch vectors, so this is the common case)
    if (hdm_graph_level() >= 2 && graphs_ok && !(hdm_flow_enabled() && flow_ok)) {   // (a replayed launch would reuse its epoch)
        Replay *r = nullptr;
        for (int i = 0; i < nsolves; ++i)
            if (solves[i].b == b_dev && solves[i].x == x_dev && solves[i].nrhs == nrhs && solves[i].which == which &&
                solves[i].ldv == ldv) r = &solves[i];
        if (!r && solve_runs >= 1 && nsolves < 4) {
            hipGraphExec_t ex = hdm_capture(s, [&]() { return enqueue_solve(b_dev, x_dev, nrhs, ldv, which, s); });
            if (!ex) graphs_ok = false;
            else { r = &solves[nsolves++]; r->b = b_dev; r->x = x_dev; r->nrhs = nrhs; r->which = which; r->ldv = ldv; r->exec = ex; }
        }
        if (r) {
            HDM_HIP_CHECK(hipGraphLaunch(r->exec, s));
            return 0;
        }
    }
    ++solve_runs;
    return enqueue_solve(b_dev, x_dev, nrhs, ldv, which, s);
}

int HdmChol::enqueue_solve(double *b_dev, double *x_dev, int nrhs, long ldv, int which, hipStream_t s) {
    // which: 0 = full solve (L L^T x = b), 1 = forward only (L x = b), 2 = backward only (L^T x = b)
    // b_dev is overwritten (workspace); vectors have npad entries (zero padded)
    const long ld = npad;
    // co-residency bound of the single-launch substitution on the device this object lives on (partitioned GPUs have
    // far fewer CUs); several engine instances sharing one device (multi-shard rehearsals) cannot rely on it at all
    if (flow_cap < 0) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hdm_trsv_flow_kernel, 256, 0) == hipSuccess)
            flow_cap = cus * std::max(1, per_cu);
        else
            flow_cap = 0;
        (void) hipGetLastError();
    }
    if (hdm_flow_enabled() && flow_ok && nrhs <= 2 && (long) nblk * nrhs <= flow_cap) {
        if (!flow_flags) {
            HDM_HIP_CHECK(hipMalloc((void **) &flow_flags, sizeof(int) * 4 * (size_t) nblk));
            HDM_HIP_CHECK(hdm_memset_sync(flow_flags, 0, sizeof(int) * 4 * (size_t) nblk));
            // the give-up word lives in mapped host memory: the host reads it after its usual synchronisation, no extra copy
            HDM_HIP_CHECK(hipHostMalloc((void **) &flow_err, sizeof(int), hipHostMallocMapped));
            *flow_err = 0;
        }
        int *err_dev = nullptr;
        HDM_HIP_CHECK(hipHostGetDevicePointer((void **) &err_dev, flow_err, 0));
        ++flow_epoch;
        hipLaunchKernelGGL(hdm_trsv_flow_kernel, dim3(nblk, nrhs), dim3(256), 0, s, L, ld, Dinv, b_dev, b_dev, x_dev, nblk,
                           ldv, flow_flags, flow_epoch, which, err_dev, (const int *) env_dev);
        HDM_HIP_CHECK(hipGetLastError());
        flow_pending = true;
        return 0;
    }
    double *cur = b_dev;
    if (which == 0 || which == 1) {
        for (int k = 0; k < nblk; ++k) {
            hipLaunchKernelGGL(hdm_trsv_fwd_step, dim3(nblk - k, nrhs), dim3(256), 0, s, L, ld, Dinv, cur, x_dev, k,
                               nblk, nrhs, ldv);
        }
        HDM_HIP_CHECK(hipGetLastError());
        if (which == 1) return 0;
        HDM_HIP_CHECK(hipMemcpyAsync(cur, x_dev, sizeof(double) * ldv * nrhs, hipMemcpyDeviceToDevice, s));
    }
    for (int k = nblk - 1; k >= 0; --k) {
        hipLaunchKernelGGL(hdm_trsv_bwd_step, dim3(k + 1, nrhs), dim3(256), 0, s, L, ld, Dinv, cur, x_dev, k, nrhs,
                           ldv);
    }
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// HDM_TRSV_FLOW_FAIL_ONCE=1 (tests): the first single-launch substitution of the process is treated as timed out and its
// result is thrown away, so that the retry below is exercised
static bool hdm_flow_fail_once() {
    static int left = -1;
    if (left < 0) { const char *e = getenv("HDM_TRSV_FLOW_FAIL_ONCE"); left = (e && atoi(e) != 0) ? 1 : 0; }
    if (left == 1) { left = 0; return true; }
    return false;
}

int HdmChol::solve_host(const double *rhs, double *sol, int nrhs, int which, hipStream_t s) {
    // nrhs columns of length n (column-major, ld = n) on the host.  `sol` may be `rhs` (the reference solves in place
    // almost everywhere, hdsdp_algo.c:452-454): a chunk's solution goes through a host staging buffer and reaches `sol`
    // only once the launch is known to be good, so a retry always starts from the caller's intact right-hand side.
    const int chunk = 2;  // vec holds 4 * npad doubles: chunk rhs + chunk sol
    if (host_stage.size() < (size_t) chunk * n) host_stage.resize((size_t) chunk * n);
    for (int c0 = 0; c0 < nrhs; c0 += chunk) {
        int nc = (nrhs - c0 < chunk) ? nrhs - c0 : chunk;
        double *b = vec, *x = vec + 2L * npad;
        HDM_HIP_CHECK(hipMemsetAsync(vec, 0, sizeof(double) * 4L * npad, s));
        HDM_HIP_CHECK(hipMemcpy2DAsync(b, sizeof(double) * npad, rhs + (long) c0 * n, sizeof(double) * n,
                                       sizeof(double) * n, nc, hipMemcpyHostToDevice, s));
        if (solve_device(b, x, nc, npad, which, s)) return 1;
        HDM_HIP_CHECK(hipMemcpy2DAsync(host_stage.data(), sizeof(double) * n, x, sizeof(double) * npad,
                                       sizeof(double) * n, nc, hipMemcpyDeviceToHost, s));
        HDM_HIP_CHECK(hipStreamSynchronize(s));
        bool gave_up = flow_pending && flow_err && *(volatile int *) flow_err != 0;
        if (flow_pending && hdm_flow_fail_once()) gave_up = true;
        flow_pending = false;
        if (gave_up) {
            // a workgroup of the single-launch substitution gave up waiting (it would mean the workgroups were not
            // co-resident): per-block launches from now on, and this chunk again from the untouched right-hand side
            fprintf(stderr, "[hdsdp_mi355x] single-launch substitution timed out; using per-block launches\n");
            flow_ok = false;
            if (flow_err) *flow_err = 0;
            c0 -= chunk;
            continue;
        }
        memcpy(sol + (long) c0 * n, host_stage.data(), sizeof(double) * (size_t) n * nc);
    }
    return 0;
}

int HdmChol::inverse_full(double *out_dev, long ldo, hipStream_t s) {
    // out = Linv^T * Linv  (full symmetric npad x npad); dpotri + HUtilMatSymmetrize equivalent
    if (invert_factor(s)) return 1;
    HdmGemmArgs g = {};
    g.A = Linv; g.lda = npad; g.a_kmajor = 1; g.B = Linv; g.ldb = npad; g.b_kmajor = 1;
    g.C = out_dev; g.ldc = ldo; g.M = npad; g.N = npad; g.K = npad; g.batch = 1; g.alpha = 1.0;
    g.epilogue = HDM_EPI_STORE;
    return hdm_launch_gemm(g, s);
}

// one kernel of this translation unit (= one code object): what the preload thread asks the runtime about (engine.hip: preload_modules)
const void *hdm_module_handle_chol() { return (const void *) hdm_potrf_diag_kernel; }
