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

#define SM_T 512
#define SM_NR 8
#define SM_NC 4

// Register matrices are indexed with compile-time constants only.  Where the wanted local row / column is a run-time but
// wave-uniform number, a switch over the constants selects it: a select chain over the whole array (v = c == kc ? a[r][c] : v)
// makes the compiler keep the array in scratch memory (606 scratch instructions, 1.8 ms per pass instead of 0.1).
#define SM_SWITCH4(kc, BODY)   \
    switch (kc) {              \
        case 0: { constexpr int C_ = 0; BODY } break;  \
        case 1: { constexpr int C_ = 1; BODY } break;  \
        case 2: { constexpr int C_ = 2; BODY } break;  \
        default: { constexpr int C_ = 3; BODY } break; \
    }
#define SM_SWITCH8(kr, BODY)   \
    switch (kr) {              \
        case 0: { constexpr int R_ = 0; BODY } break;  \
        case 1: { constexpr int R_ = 1; BODY } break;  \
        case 2: { constexpr int R_ = 2; BODY } break;  \
        case 3: { constexpr int R_ = 3; BODY } break;  \
        case 4: { constexpr int R_ = 4; BODY } break;  \
        case 5: { constexpr int R_ = 5; BODY } break;  \
        case 6: { constexpr int R_ = 6; BODY } break;  \
        default: { constexpr int R_ = 7; BODY } break; \
    }

// one sweep over k < n: a (full symmetric in) -> lower part = L;  rr -> W = L^-1 (lower).
// colb / rowb: 2 x 128 doubles of LDS each, rsv: 128.  Returns 0, or k + 1 for the first non-positive pivot (all threads alike).
//
// Column k of the trailing matrix and row k of R are never touched again after step k, and what is final about them is
// only a scale factor 1 / sqrt(pivot_k) away: L[i, k] = A_k[i, k] rs_k, W[k, j] = R_k[k, j] rs_k.  So the loop never writes
// a "final" column or row back into the register matrices (an insert at a run-time position costs a phi of the whole
// array per step); the factors rs_k are kept in LDS and applied once at the end.  All LDS reads of a step are issued
// together and unconditionally -- under per-element conditions the compiler put every ds_read behind its own branch and
// its own wait, 49 serialised LDS round trips per step (2.1 us per step instead of 0.3).
// sixteen steps k = 16 KR .. 16 KR + 15 of the sweep.  The local row index of the published row (KR) and the local column
// index of the published column (KR / 2) are template constants: the register matrices are indexed with constants only,
// the bounds of the update loops fold, and a step is one straight-line block between two barriers.
template <int KR>
__device__ __forceinline__ int sm_sweep_chunk(int n, double (&a)[SM_NR][SM_NC], double (&rr)[SM_NR][SM_NC],
                                              double *colb, double *rowb, double *rsv, int ty, int tx) {
    constexpr int KC = KR / 2;
    const int kend = min(n, 16 * KR + 16);
    for (int k = 16 * KR; k < kend; ++k) {
        double *cb = colb + (k & 1) * SMALL_P, *rb = rowb + (k & 1) * SMALL_P;
        if (tx == (k & 31)) {                    // owners of column k publish it (rows ty + 16 r)
#pragma unroll
            for (int r = 0; r < SM_NR; ++r) cb[ty + 16 * r] = a[r][KC];
        }
        if (ty == (k & 15)) {                    // owners of row k of R publish it (columns tx + 32 c)
#pragma unroll
            for (int c = 0; c < SM_NC; ++c) rb[tx + 32 * c] = rr[KR][c];
        }
        __syncthreads();
        const double p = cb[k];
        double cbi[SM_NR], cbj[SM_NC], rbj[SM_NC];
#pragma unroll
        for (int r = KR; r < SM_NR; ++r) cbi[r] = cb[ty + 16 * r];
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) { cbj[c] = (c >= KC) ? cb[tx + 32 * c] : 0.0; rbj[c] = (c <= KC) ? rb[tx + 32 * c] : 0.0; }
        if (!(p > 0.0)) return k + 1;
        // 1 / sqrt(p): the hardware estimate (v_rsq_f64, about 2^-26) and two Newton steps
        double rs = __builtin_amdgcn_rsq(p);
        rs = rs * (1.5 - 0.5 * p * rs * rs);
        rs = rs * (1.5 - 0.5 * p * rs * rs);
        if (ty == 0 && tx == 0) rsv[k] = rs;
        double li[SM_NR], lj[SM_NC], wj[SM_NC];
#pragma unroll
        for (int r = KR; r < SM_NR; ++r) li[r] = cbi[r] * ((ty + 16 * r > k) ? rs : 0.0);
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            lj[c] = cbj[c] * ((tx + 32 * c > k) ? rs : 0.0);
            wj[c] = rbj[c] * ((tx + 32 * c <= k) ? rs : 0.0);
        }
#pragma unroll
        for (int r = KR; r < SM_NR; ++r) {       // rows that can lie below k
#pragma unroll
            for (int c = 0; c < SM_NC; ++c) {
                if (c >= KC) a[r][c] -= li[r] * lj[c];      // trailing update (columns right of k)
                if (c <= KC) rr[r][c] -= li[r] * wj[c];     // forward substitution on the identity (columns up to k)
            }
        }
    }
    return 0;
}

__device__ __forceinline__ int sm_sweep(int n, double (&a)[SM_NR][SM_NC], double (&rr)[SM_NR][SM_NC],
                                        double *colb, double *rowb, double *rsv, int ty, int tx, double *logdet) {
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) rr[r][c] = (ty + 16 * r == tx + 32 * c) ? 1.0 : 0.0;
    int info = 0;
    if (!info && n > 0) info = sm_sweep_chunk<0>(n, a, rr, colb, rowb, rsv, ty, tx);
    if (!info && n > 16) info = sm_sweep_chunk<1>(n, a, rr, colb, rowb, rsv, ty, tx);
    if (!info && n > 32) info = sm_sweep_chunk<2>(n, a, rr, colb, rowb, rsv, ty, tx);
    if (!info && n > 48) info = sm_sweep_chunk<3>(n, a, rr, colb, rowb, rsv, ty, tx);
    if (!info && n > 64) info = sm_sweep_chunk<4>(n, a, rr, colb, rowb, rsv, ty, tx);
    if (!info && n > 80) info = sm_sweep_chunk<5>(n, a, rr, colb, rowb, rsv, ty, tx);
    if (!info && n > 96) info = sm_sweep_chunk<6>(n, a, rr, colb, rowb, rsv, ty, tx);
    if (!info && n > 112) info = sm_sweep_chunk<7>(n, a, rr, colb, rowb, rsv, ty, tx);
    __syncthreads();
    double ld = 0.0;
    if (!info) {
        // apply the deferred scale factors: column j of L and row i of W; log det = 2 sum log L_kk from the owners of the
        // diagonal (the sum is finished by the caller)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            const int j = tx + 32 * c;
            const double sj = (j < n) ? rsv[j] : 1.0;
#pragma unroll
            for (int r = 0; r < SM_NR; ++r) a[r][c] *= sj;
        }
#pragma unroll
        for (int r = 0; r < SM_NR; ++r) {
            const int i = ty + 16 * r;
            const double si = (i < n) ? rsv[i] : 1.0;
#pragma unroll
            for (int c = 0; c < SM_NC; ++c) {
                rr[r][c] *= si;
                if (i == tx + 32 * c && i < n) ld += 2.0 * log(a[r][c]);
            }
        }
    }
    __syncthreads();
    *logdet = ld;
    return info;
}

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

// L (lower, identity padded, ld = 128) and W = L^-1 (lower, zeros above, identity padded) to global memory
__device__ __forceinline__ void sm_store_factor(int n, const double (&a)[SM_NR][SM_NC], const double (&rr)[SM_NR][SM_NC],
                                                double *L, double *W, int ty, int tx) {
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            const int i = ty + 16 * r, j = tx + 32 * c;
            const bool in = (i < n && j < n);
            const double pad = (i == j) ? 1.0 : 0.0;
            L[i + (long) j * SMALL_P] = in ? ((i >= j) ? a[r][c] : 0.0) : pad;
            W[i + (long) j * SMALL_P] = in ? ((i >= j) ? rr[r][c] : 0.0) : pad;
        }
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
    double *Xl = sm;                                   // S, later X = S^-1, later solve scratch: 128 x 129
    double *colb = sm + (long) SMALL_P * ld;           // 2 x 128
    double *rowb = colb + 2 * SMALL_P;                 // 2 x 128
    double *vden = rowb + 2 * SMALL_P;                 // SMALL_NDENSE x 128: X a_d of the dense factors
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

    // ---- S = tau C - sum y_i s_i a_i a_i' + eye I, assembled in LDS (full symmetric)
    for (int e = tid; e < n * n; e += SM_T) {
        const int i = e % n, j = e / n;
        Xl[i + j * ld] = p.tau * p.C[i + (long) j * p.ldc] + ((i == j) ? p.eye : 0.0);
    }
    if (tid < m) bvec[tid] = p.b[tid];
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
            for (int e = tid; e < n * n; e += SM_T) {
                const int i = e % n, j = e / n;
                Xl[i + j * ld] += coef * av[i] * av[j];
            }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += SM_T) {          // the dual matrix itself stays resident for the other cone slots
        const int i = e % n, j = e / n;
        if (i >= j) p.Sout[i + (long) j * p.lds] = Xl[i + j * ld];
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
    const int infoS = sm_sweep(n, a, rr, colb, rowb, bvec + SMALL_P, ty, tx, &logdet);
    if (tid == 0) tvec[0] = 0.0;
    __syncthreads();
    if (logdet != 0.0) atomicAdd(&tvec[0], logdet);
    __syncthreads();
    logdet = tvec[0];
    __syncthreads();
    if (infoS) {
        if (tid == 0) { p.out[0] = (double) infoS; p.out[1] = 0.0; p.out[2] = 0.0; p.out[3] = 0.0; }
        return;
    }
    sm_store_factor(n, a, rr, p.LS, p.WS, ty, tx);
    __syncthreads();
    SM_STAMP(2)
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
    {   // M_ij = s_i s_j (a_i' X a_j)^2 for i >= j: thread t keeps column j = t mod 128 (its factor in registers) and walks
        // the rows i = t / 128, +4, ...; the row's data are the same address for a whole wave (LDS broadcasts)
        const int j = tid & (SMALL_P - 1);
        if (j < m) {
            const int dj = dol[j], j0 = fpl[j], jn = (dj >= 0) ? 0 : fnl[j];
            const double aj0 = (jn > 0) ? fvl[j0] : 0.0;               // the common case: one entry per factor
            const int ij0 = (jn > 0) ? fil[j0] : 0;
            const double sj = sgl[j];
            const int istart = j + ((((tid >> 7) - j) % 4 + 4) % 4);      // first row >= j of this thread's residue class
            for (int i = istart; i < m; i += 4) {
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
                p.M[i + (long) j * p.ldm] = sgl[i] * sj * gam * gam;
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
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            const int i = ty + 16 * r, j = tx + 32 * c;
            const int hi = max(i, j), lo = min(i, j);
            a[r][c] = (i < m && j < m) ? p.M[hi + (long) lo * p.ldm] : ((i == j) ? 1.0 : 0.0);
        }
    double logdetM = 0.0;
    const int infoM = sm_sweep(m, a, rr, colb, rowb, bvec + SMALL_P, ty, tx, &logdetM);
    if (tid == 0) { p.out[0] = 0.0; p.out[1] = (double) infoM; p.out[2] = logdet; p.out[3] = trs; }
    if (infoM) return;
    sm_store_factor(m, a, rr, p.LM, p.WM, ty, tx);
    __syncthreads();
    SM_STAMP(5)
    double *part = Xl;                                 // X is no longer needed: 16 x 128 partial sums
    sm_solve(m, rr, bvec, tvec, part, p.out + 4 + 2 * m, ty, tx, tid);
    sm_solve(m, rr, asinv, tvec, part, p.out + 4 + 3 * m, ty, tx, tid);
    sm_solve(m, rr, asinvrd, tvec, part, p.out + 4 + 4 * m, ty, tx, tid);
    SM_STAMP(6)
#undef SM_STAMP
}

size_t hdm_small_lds_bytes() {
    return sizeof(double) * ((size_t) SMALL_P * (SMALL_P + 1) + 4 * SMALL_P + SMALL_NDENSE * SMALL_P + 5 * SMALL_P + SMALL_P * SMALL_SPMAX + SMALL_P) +
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
