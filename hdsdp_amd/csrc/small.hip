// small.hip -- one launch for a whole Phase-A pass on a small rank-one block (BASELINE configs 2-3: mcp100, gpp100).
//
// Reference path being replaced (one pass = interface/hdsdp_algo.c:1082-1101 on a block whose constraints are all
// A_i = s_i a_i a_i', the reference's all-M2 plans, hdsdp_conic_sdp.c:687-778):
//     S = tau C - sum y_i A_i - Rd I -> Cholesky (PSD check, log det) -> S^-1 -> M_ij = s_i s_j (a_i' S^-1 a_j)^2,
//     ASinv_i = s_i a_i' S^-1 a_i,  ASinvRdSinv_i = Rd s_i |S^-1 a_i|^2,  tr S^-1 -> Cholesky of M -> three solves.
// On the multi-launch path that is about forty launches and six host synchronisations for 0.2 MB of operands: 0.6 ms,
// twice what one host core needs.  Here it is ONE workgroup of 512 threads and one synchronisation.
//
// Layout.  n, m <= 128.  A 128 x 128 matrix lives in REGISTERS, dealt cyclically over a 16 x 32 thread grid: thread
// (ty, tx) holds rows ty + 16 r (r < 8) of columns tx + 32 c (c < 4), 32 doubles per matrix.  The cyclic deal keeps
// every thread busy through a triangular sweep.
//
// The sweep (sm_sweep) is a right-looking Cholesky in which step k also finishes row k of W = L^-1 (forward substitution
// on the identity, carried along as a second register matrix): the column of A and the row of R that step k needs are put
// into LDS by their owners, one barrier, and every thread updates its own elements; rows / columns that a step cannot
// touch are skipped with wave-uniform bounds.  One pass over k = 0 .. n-1 with ONE barrier per step (the two LDS lines are
// double-buffered) yields L, L^-1 and log det S.  S^-1 = W'W follows in n / 4 independent rounds (sm_wtw).  (Carrying X
// along in the same sweep as a third register matrix spilled 118 VGPRs at two waves per SIMD.)
//
// The factors are left where the multi-launch path keeps them (HdmChol::L and ::Dinv of the dual matrix and of the Schur
// system -- for one 128-block, Dinv IS the triangular inverse), so every later call on the same state (HKKTSolve, log
// barrier, ratio test) finds what it expects.
#include "small.h"
#include <algorithm>
#include <cmath>

#include "sweep128.h"

// x = W'W for the lower-triangular W in registers (x_ij = sum_k W_ki W_kj).  Every thread owns exactly one row of each
// block of sixteen rows (row 16 KR + ty, its local row KR), so a block is published by ALL threads at once -- no owner
// test, constant register indices -- and consumed as sixteen outer products: eight rounds of two barriers for the whole
// product.  buf: 16 x 128 doubles of LDS.  Row k of W is zero right of column k: rows / columns beyond the block's last row
// are skipped.
template <int KR>
__device__ __forceinline__ void sm_wtw_block(int n, const double (&w)[SM_NR][SM_NC], double (&x)[SM_NR][SM_NC], double *buf, int ty, int tx) {
    constexpr int KC = KR / 2;
#pragma unroll
    for (int c = 0; c < SM_NC; ++c) buf[ty * SMALL_P + tx + 32 * c] = (16 * KR + ty < n) ? w[KR][c] : 0.0;
    __syncthreads();
    const int qend = min(16, n - 16 * KR);
    for (int q = 0; q < qend; ++q) {
        const double *row = buf + q * SMALL_P;
        double wi[SM_NR], wj[SM_NC];
#pragma unroll
        for (int r = 0; r <= KR; ++r) wi[r] = row[ty + 16 * r];
#pragma unroll
        for (int c = 0; c <= KC; ++c) wj[c] = row[tx + 32 * c];
#pragma unroll
        for (int r = 0; r <= KR; ++r)
#pragma unroll
            for (int c = 0; c <= KC; ++c) x[r][c] += wi[r] * wj[c];
    }
    __syncthreads();
}
__device__ __forceinline__ void sm_wtw(int n, const double (&w)[SM_NR][SM_NC], double (&x)[SM_NR][SM_NC], double *buf, int ty, int tx) {
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) x[r][c] = 0.0;
    if (n > 0) sm_wtw_block<0>(n, w, x, buf, ty, tx);
    if (n > 16) sm_wtw_block<1>(n, w, x, buf, ty, tx);
    if (n > 32) sm_wtw_block<2>(n, w, x, buf, ty, tx);
    if (n > 48) sm_wtw_block<3>(n, w, x, buf, ty, tx);
    if (n > 64) sm_wtw_block<4>(n, w, x, buf, ty, tx);
    if (n > 80) sm_wtw_block<5>(n, w, x, buf, ty, tx);
    if (n > 96) sm_wtw_block<6>(n, w, x, buf, ty, tx);
    if (n > 112) sm_wtw_block<7>(n, w, x, buf, ty, tx);
}

// x = W' (W b) for the lower-triangular W held in registers; b, t: LDS vectors; part: 16 x 128 LDS doubles; out: m results
__device__ __forceinline__ void sm_solve(int m, const double (&w)[SM_NR][SM_NC], const double *b, double *t, double *part,
                                         double *out, int ty, int tx, int tid) {
    double s[SM_NR];
#pragma unroll
    for (int r = 0; r < SM_NR; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) acc += w[r][c] * b[tx + 32 * c];
        for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 32);     // over the 32 threads of one ty
        s[r] = acc;
    }
    if (tx == 0) {
#pragma unroll
        for (int r = 0; r < SM_NR; ++r) t[ty + 16 * r] = s[r];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < SM_NC; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < SM_NR; ++r) acc += w[r][c] * t[ty + 16 * r];
        part[ty * SMALL_P + tx + 32 * c] = acc;
    }
    __syncthreads();
    if (tid < m) {
        double acc = 0.0;
        for (int q = 0; q < 16; ++q) acc += part[q * SMALL_P + tid];
        out[tid] = acc;
    }
    __syncthreads();
}

__global__ __launch_bounds__(SM_T) void hdm_small_phase_a_kernel(HdmSmallArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = p.n, m = p.m;
    const int ld = SMALL_P + 1;
    double *Xl = sm;                                   // S, later X = S^-1, later solve scratch: 128 x 129 (its first 2 x 1024 doubles: the sweeps' block images)
    double *vden = sm + (long) SMALL_P * ld;           // SMALL_NDENSE x 128: X a_d of the dense factors
    double *asinv = vden + SMALL_NDENSE * SMALL_P;     // 128
    double *asinvrd = asinv + SMALL_P;                 // 128
    double *tvec = asinvrd + SMALL_P;                  // 128
    double *bvec = tvec + SMALL_P;                     // 128 (+ 128 behind it: the sweep's deferred scale factors)
    // the rank-one factors' CSR, staged once: every later use is a handful of dependent reads per pair of constraints,
    // which from global memory cost an L2 round trip each (the build took 25-45 us that way)
    double *fvl = bvec + 2 * SMALL_P;                  // SMALL_P * SMALL_SPMAX values of the sparse factors
    double *sgl = fvl + SMALL_P * SMALL_SPMAX;         // 128 signs
    int *fil = (int *) (sgl + SMALL_P);                // SMALL_P * SMALL_SPMAX indices
    int *fpl = fil + SMALL_P * SMALL_SPMAX;            // 129 row starts INTO fvl / fil (dense rows: empty)
    int *fnl = fpl + SMALL_P + 4;                      // 128: entries per row (a dense row: n, its values stay in global memory)
    int *dol = fnl + SMALL_P;                          // 128: dense_of
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    // phase stamps (100 MHz wall clock) behind the results: [0] start, [1] S assembled, [2] factor + inverse of S,
    // [3] S^-1, [4] Schur build, [5] factor + inverse of M, [6] solves
    double *stamp = p.out + 4 + 5 * m;
#define SM_STAMP(i) if (tid == 0) stamp[i] = (double) __builtin_amdgcn_s_memrealtime();
    SM_STAMP(0)
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();

    // ---- S = tau C - sum y_i s_i a_i a_i' + eye I, assembled in LDS (full symmetric)
    // (thread = row tid mod 128 of the columns tid / 128, + 4, ...: no integer division, and the loads of an unrolled
    // group are in flight together -- one load per trip of a divided index ran at a memory round trip per element)
    const int ei = tid & (SMALL_P - 1), ej0 = tid >> 7;
    if (ei < n) {
#pragma unroll 8
        for (int j = ej0; j < n; j += SM_T / SMALL_P)
            Xl[ei + j * ld] = p.tau * p.C[ei + (long) j * p.ldc] + ((ei == j) ? p.eye : 0.0);
    }
    if (tid < SMALL_P) {                               // (entries past m meet zeros of the padded factor in the solves: they must be numbers)
        bvec[tid] = (tid < m) ? p.b[tid] : 0.0;
        asinv[tid] = 0.0; asinvrd[tid] = 0.0; tvec[tid] = 0.0;
    }
    if (tid == 0) {                                    // row starts of the staged (sparse-only) CSR: a serial scan of <= 128 rows
        int pos = 0;
        for (int q = 0; q < m; ++q) {
            const int cnt = p.fp[q + 1] - p.fp[q];
            fpl[q] = pos; fnl[q] = cnt;
            if (cnt <= SMALL_SPMAX) pos += cnt;
        }
        fpl[m] = pos;
    }
    if (tid < m) { sgl[tid] = p.sgn[tid]; dol[tid] = p.dense_of[tid]; }
    __syncthreads();
    for (int q = tid; q < m; q += SM_T) {
        const int f0 = p.fp[q], cnt = fnl[q];
        if (cnt <= SMALL_SPMAX)
            for (int u = 0; u < cnt; ++u) { fvl[fpl[q] + u] = p.fv[f0 + u]; fil[fpl[q] + u] = p.fi[f0 + u]; }
    }
    __syncthreads();
    for (int q = tid; q < m; q += SM_T) {              // sparse factors: a handful of entries each
        if (fnl[q] > SMALL_SPMAX) continue;
        const int f0 = fpl[q], f1 = f0 + fnl[q];
        const double coef = -p.y[q] * sgl[q];
        if (coef == 0.0) continue;
        for (int u = f0; u < f1; ++u)
            for (int v = f0; v < f1; ++v) atomicAdd(&Xl[fil[u] + fil[v] * ld], coef * fvl[u] * fvl[v]);
    }
    __syncthreads();
    for (int d = 0; d < p.ndense; ++d) {               // dense factors: the whole workgroup per row
        const int q = p.dense_rows[d];
        const double coef = -p.y[q] * p.sgn[q];
        const double *av = p.fv + p.fp[q];             // all n entries are listed
        if (coef != 0.0)
            if (ei < n) {
                const double ai = coef * av[ei];
                for (int j = ej0; j < n; j += SM_T / SMALL_P) Xl[ei + j * ld] += ai * av[j];
            }
        __syncthreads();
    }
    if (ei < n) {                                      // the dual matrix itself stays resident for the other cone slots
#pragma unroll 8
        for (int j = ej0; j <= ei; j += SM_T / SMALL_P) p.Sout[ei + (long) j * p.lds] = Xl[ei + j * ld];
    }
    double a[SM_NR][SM_NC], rr[SM_NR][SM_NC];
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            const int i = ty + 16 * r, j = tx + 32 * c;
            a[r][c] = (i < n && j < n) ? Xl[i + j * ld] : ((i == j) ? 1.0 : 0.0);
        }
    __syncthreads();
    SM_STAMP(1)

    // ---- factor S, invert the factor, form S^-1: one sweep
    double logdet = 0.0;
    const int infoS = sm_sweep(n, a, rr, Xl, bvec + SMALL_P, ty, tx, &logdet);
    SM_STAMP(2)
    if (infoS) {
        if (tid == 0) { p.out[0] = (double) infoS; p.out[1] = 0.0; p.out[2] = 0.0; p.out[3] = 0.0; }
        return;
    }
    sm_store_factor(n, a, rr, Xl, p.LS, p.WS, ty, tx, tid);
    __syncthreads();
    sm_wtw(n, rr, a, Xl, ty, tx);                      // (the S image in LDS is free: 16 x 128 doubles of it serve as the row buffer; `a` is free too)
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            const int i = ty + 16 * r, j = tx + 32 * c;
            if (i < n && j < n) Xl[i + j * ld] = a[r][c];
        }
    __syncthreads();

    SM_STAMP(3)
    // ---- the rank-one Schur build on X = S^-1
    for (int d = 0; d < p.ndense; ++d) {               // v_d = X a_d
        const double *av = p.fv + p.fp[p.dense_rows[d]];
        for (int r = tid; r < n; r += SM_T) {
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc += av[k] * Xl[r + k * ld];
            vden[d * SMALL_P + r] = acc;
        }
    }
    __syncthreads();
    for (int q = tid; q < m; q += SM_T) {              // Gamma_qq and |X a_q|^2
        const int dq = dol[q];
        double gii = 0.0, nrm = 0.0;
        if (dq >= 0) {
            const double *v = vden + dq * SMALL_P, *av = p.fv + p.fp[q];
            for (int k = 0; k < n; ++k) { gii += av[k] * v[k]; nrm += v[k] * v[k]; }
        } else {
            const int f0 = fpl[q], f1 = f0 + fnl[q];
            for (int u = f0; u < f1; ++u)
                for (int v = f0; v < f1; ++v) gii += fvl[u] * fvl[v] * Xl[fil[u] + fil[v] * ld];
            for (int r = 0; r < n; ++r) {
                double t = 0.0;
                for (int u = f0; u < f1; ++u) t += fvl[u] * Xl[r + fil[u] * ld];
                nrm += t * t;
            }
        }
        asinv[q] = sgl[q] * gii;
        asinvrd[q] = p.Rd * sgl[q] * nrm;
    }
    {   // M_ji = s_i s_j (a_i' X a_j)^2 for j >= i: thread t keeps ROW j = t mod 128 (its factor in registers) and walks
        // the columns i = t / 128, +4, ... <= j; the column's data are the same address for a whole wave (LDS broadcasts),
        // and the 64 stores of a wave are consecutive rows of one column (full lines)
        const int j = tid & (SMALL_P - 1);
        if (j < m) {
            const int dj = dol[j], j0 = fpl[j], jn = (dj >= 0) ? 0 : fnl[j];
            const double aj0 = (jn > 0) ? fvl[j0] : 0.0;               // the common case: one entry per factor
            const int ij0 = (jn > 0) ? fil[j0] : 0;
            const double sj = sgl[j];
            for (int i = tid >> 7; i <= j; i += 4) {
                const int di = dol[i];
                double gam = 0.0;
                if (di >= 0 && dj >= 0) {                  // both dense (a handful of pairs at most): a_i from global memory
                    const double *v = vden + dj * SMALL_P, *av = p.fv + p.fp[i];
                    for (int k = 0; k < n; ++k) gam += av[k] * v[k];
                } else if (dj >= 0) {
                    const double *v = vden + dj * SMALL_P;
                    for (int u = fpl[i]; u < fpl[i] + fnl[i]; ++u) gam += fvl[u] * v[fil[u]];
                } else if (di >= 0) {
                    const double *v = vden + di * SMALL_P;
                    for (int u = j0; u < j0 + jn; ++u) gam += fvl[u] * v[fil[u]];
                } else {
                    const int i0 = fpl[i], in_ = fnl[i];
                    if (in_ == 1 && jn == 1) gam = fvl[i0] * aj0 * Xl[fil[i0] + ij0 * ld];
                    else
                        for (int u = 0; u < in_; ++u) {
                            const double *xc = Xl + fil[i0 + u];       // row index of X from a_i, column from a_j
                            double t = 0.0;
                            for (int v = j0; v < j0 + jn; ++v) t += fvl[v] * xc[fil[v] * ld];
                            gam += fvl[i0 + u] * t;
                        }
                }
                p.M[j + (long) i * p.ldm] = sgl[i] * sj * gam * gam;
            }
        }
    }
    double trs = 0.0;
    if (tid < 64) {                                    // tr S^-1: one wave, two entries per lane, shuffles
        double t = ((tid < n) ? Xl[tid + tid * ld] : 0.0) + ((tid + 64 < n) ? Xl[(tid + 64) + (tid + 64) * ld] : 0.0);
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        trs = t;
    }
    __threadfence();
    __syncthreads();
    if (tid < m) { p.out[4 + tid] = asinv[tid]; p.out[4 + m + tid] = asinvrd[tid]; }

    SM_STAMP(4)
    // ---- factor M, invert the factor, three solves
    // M comes back through the LDS image (X is dead): full-line loads of the lower triangle, and the rest of the 128 x 128
    // device matrix is zeroed on the way -- the state HKKTBuildUp leaves (hdsdp_schur.c:141-165), without a memset launch
    if (ei < SMALL_P) {
#pragma unroll 8
        for (int j = ej0; j < SMALL_P; j += SM_T / SMALL_P) {
            const bool low = (ei < m && j <= ei);
            double v = 0.0;
            if (low) v = p.M[ei + (long) j * p.ldm];
            else p.M[ei + (long) j * p.ldm] = 0.0;
            Xl[ei + j * ld] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            const int i = ty + 16 * r, j = tx + 32 * c;
            const int hi = max(i, j), lo = min(i, j);
            a[r][c] = (i < m && j < m) ? Xl[hi + lo * ld] : ((i == j) ? 1.0 : 0.0);
        }
    __syncthreads();
    double logdetM = 0.0;
    const int infoM = sm_sweep(m, a, rr, Xl, bvec + SMALL_P, ty, tx, &logdetM);
    if (tid == 0) { p.out[0] = 0.0; p.out[1] = (double) infoM; p.out[2] = logdet; p.out[3] = trs; }
    if (infoM) return;
    sm_store_factor(m, a, rr, Xl, p.LM, p.WM, ty, tx, tid);
    __syncthreads();
    SM_STAMP(5)
    double *part = Xl;                                 // X is no longer needed: 16 x 128 partial sums
    sm_solve(m, rr, bvec, tvec, part, p.out + 4 + 2 * m, ty, tx, tid);
    sm_solve(m, rr, asinv, tvec, part, p.out + 4 + 3 * m, ty, tx, tid);
    sm_solve(m, rr, asinvrd, tvec, part, p.out + 4 + 4 * m, ty, tx, tid);
    SM_STAMP(6)
    if (tid == 0) stamp[7] = (double) (__builtin_amdgcn_s_memtime() - clk0);      // shader-clock cycles of the whole pass
#undef SM_STAMP
}

// ---- interior check of a small block in one launch (see small.h) ------------------------------------------------------
__global__ __launch_bounds__(SM_T) void hdm_small_check_kernel(HdmSmallCheckArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int ld = SMALL_P + 1;
    double *Xl = sm;                                   // S, full symmetric: 128 x 129 (its head: the sweep's block images)
    double *rsv = sm + (long) SMALL_P * ld;            // 128 deferred scale factors
    double *yl = rsv + SMALL_P;                        // the multipliers, staged once
    const int n = p.n, n16 = p.n16, m = p.m;
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    for (int q = tid; q < m; q += SM_T) yl[q] = p.y[q];
    __syncthreads();
    // lower triangle, column by column: thread = row (tid mod 128) of the columns tid / 128, + 4, ...
    const int ei = tid & (SMALL_P - 1), ej0 = tid >> 7;
    if (ei < n) {
        for (int j = ej0; j <= ei; j += SM_T / SMALL_P) {
            const double *a = p.A + ei + (long) j * n16;
            double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // eight loads in flight per thread
            int q = 0;
            for (; q + 7 < m; q += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] -= yl[q + u] * a[(long) (q + u) * p.astride];
            }
            for (; q < m; ++q) acc[0] -= yl[q] * a[(long) q * p.astride];
            double v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            if (ei == j) v *= 2.0;                     // A_L form: half the diagonal is stored
            v += p.tau * p.C[ei + (long) j * n16];
            if (ei == j) v += p.eye;
            Xl[ei + j * ld] = v;
            Xl[j + ei * ld] = v;
            p.Sout[ei + (long) j * n16] = v;
        }
    }
    __syncthreads();
    double a[SM_NR][SM_NC], rr[SM_NR][SM_NC];
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            const int i = ty + 16 * r, j = tx + 32 * c;
            a[r][c] = (i < n && j < n) ? Xl[i + j * ld] : ((i == j) ? 1.0 : 0.0);
        }
    __syncthreads();
    double logdet = 0.0;
    const int info = sm_sweep(n, a, rr, Xl, rsv, ty, tx, &logdet);
    if (tid == 0) { p.out[0] = (double) info; p.out[1] = logdet; }
    if (info) return;
    sm_store_factor(n, a, rr, Xl, p.L, p.W, ty, tx, tid);
}

int hdm_small_check(const HdmSmallCheckArgs &args, hipStream_t s) {
    if (args.n < 1 || args.n > SMALL_P || args.n16 > SMALL_P || args.m < 0 || args.m > 4096) return 1;
    const size_t lds = sizeof(double) * ((size_t) SMALL_P * (SMALL_P + 1) + SMALL_P + (size_t) std::max(1, args.m));
    if (lds > 160 * 1024) return 1;
    static thread_local int configured_dev = -1;
    int dev = 0;
    HDM_HIP_CHECK(hipGetDevice(&dev));
    if (configured_dev != dev) {
        HDM_HIP_CHECK(hipFuncSetAttribute((const void *) hdm_small_check_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured_dev = dev;
    }
    hipLaunchKernelGGL(hdm_small_check_kernel, dim3(1), dim3(SM_T), lds, s, args);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t hdm_small_lds_bytes() {
    return sizeof(double) * ((size_t) SMALL_P * (SMALL_P + 1) + SMALL_NDENSE * SMALL_P + 5 * SMALL_P + SMALL_P * SMALL_SPMAX + SMALL_P) +
           sizeof(int) * ((size_t) SMALL_P * SMALL_SPMAX + (SMALL_P + 4) + 2 * SMALL_P);
}

int hdm_small_phase_a(const HdmSmallArgs &args, hipStream_t s) {
    if (args.n < 1 || args.n > SMALL_P || args.m < 1 || args.m > SMALL_P || args.ndense > SMALL_NDENSE) return 1;
    static thread_local int configured_dev = -1;
    int dev = 0;
    HDM_HIP_CHECK(hipGetDevice(&dev));
    if (configured_dev != dev) {
        HDM_HIP_CHECK(hipFuncSetAttribute((const void *) hdm_small_phase_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int) hdm_small_lds_bytes()));
        configured_dev = dev;
    }
    hipLaunchKernelGGL(hdm_small_phase_a_kernel, dim3(1), dim3(SM_T), hdm_small_lds_bytes(), s, args);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// one kernel of this translation unit (= one code object): what the preload thread asks the runtime about (engine.hip: preload_modules)
const void *hdm_module_handle_small() { return (const void *) hdm_small_phase_a_kernel; }
