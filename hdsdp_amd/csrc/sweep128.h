// sweep128.h -- factor + triangular inverse of one 128 x 128 block held in the registers of one 512-thread workgroup.
// Shared by the fused small-block pass (small.hip) and the blocked Cholesky's diagonal-block kernel (chol.hip).
//
// Layout.  A 128 x 128 matrix lives in REGISTERS, dealt cyclically over a 16 x 32 thread grid: thread (ty, tx) holds rows
// ty + 16 r (r < 8) of columns tx + 32 c (c < 4), 32 doubles per matrix.  The cyclic deal keeps every thread busy through a
// triangular sweep.  Register matrices are indexed with compile-time constants only (a select chain over a run-time index
// makes the compiler keep the array in scratch memory: 606 scratch instructions, 1.8 ms per pass instead of 0.1).
#pragma once
#include "small.h"

#define SM_T 512
#define SM_NR 8
#define SM_NC 4

// one sweep over k < n: a (full symmetric in) -> lower part = L;  rr -> W = L^-1 (lower).
// blk: 2 x 1024 doubles of LDS, rsv: 128.  Returns 0, or k + 1 for the first non-positive pivot (all threads alike).
//
// Column k of the trailing matrix and row k of R are never touched again after step k, and what is final about them is
// only a scale factor 1 / sqrt(pivot_k) away: L[i, k] = A_k[i, k] rs_k, W[k, j] = R_k[k, j] rs_k.  So the loop never writes
// a "final" column or row back into the register matrices (an insert at a run-time position costs a phi of the whole
// array per step); the factors rs_k are kept in LDS and applied once at the end.  All LDS reads of a step are issued
// together and unconditionally -- under per-element conditions the compiler put every ds_read behind its own branch and
// its own wait, 49 serialised LDS round trips per step (2.1 us per step instead of 0.3).
//
// FOUR pivots per barrier.  One pivot per barrier ran at 0.73 us per step -- write, barrier, read and a reciprocal square
// root, each waiting for the one before, with the 20-odd multiply-adds of a thread at the end.  Now the owners publish the
// four columns k0 .. k0 + 3 of the trailing matrix and the four rows of R as they stand BEFORE pivot k0 (image [i][4]:
// a thread's four values of a row are two ds_read_b128), and every thread redoes the little that couples the four pivots
// -- the 4 x 4 factor of the diagonal block and the in-block corrections of its own rows and columns of the panel -- in
// registers, in exactly the order the one-pivot loop applied them (same multiply-adds, same rounding: the results are
// bit-identical), then applies the four rank-one updates to its elements.  The sixteen steps k = 16 KR .. 16 KR + 15 of
// a chunk share the local row index of the published rows (KR) and the local column index of the published columns
// (KR / 2) as template constants: the register matrices are indexed with constants only, the bounds of the update loops
// fold, and a block is one straight-line piece of code between two barriers.
__device__ __forceinline__ double sm_rsqrt(double p) {
    // 1 / sqrt(p): the hardware estimate (v_rsq_f64, about 2^-26) and two Newton steps
    double rs = __builtin_amdgcn_rsq(p);
    rs = rs * (1.5 - 0.5 * p * rs * rs);
    rs = rs * (1.5 - 0.5 * p * rs * rs);
    return rs;
}

#define SM_BLK 1024        // doubles per block image: [128][4] columns, then [128][4] rows of R

// SIGNED (the tile-form LDL' of a sparse Schur matrix, bsparse.hip): the same sweep as A = L~ S L~', S = diag(+-1) the signs of
// the pivots, L~ = L |D|^1/2 -- an LDL' without pivoting in Cholesky clothing.  Pivot q contributes sigma_q u u' with
// u = column / sqrt|pivot| (so L~ = U S, and U's diagonal is sigma sqrt|pivot|): the corrections of the COLUMN panels carry
// sigma (w = sigma u below), and the deferred column scale is rsv[k] = sigma_k / sqrt|pivot_k|.  The forward substitution on the
// identity builds Y = U^-1 row by row -- row q is (e_q - sum u_qt Y_t) / u_qq, i.e. scaled by the SIGNED rsv[q] wherever a
// finished row enters a later one -- and what is wanted is W = L~^-1 = S Y: the rows leave with the UNSIGNED scale |rsv|.  Only a pivot that is exactly zero (or not a
// number) stops the sweep, as in the reference's sparse direct solver (external/qdldl.c:109, :212); *nneg counts the negative
// ones.  With all pivots positive every sigma is +1.0 and the arithmetic is the unsigned sweep's, bit for bit.
template <int KR, bool SIGNED = false>
__device__ __forceinline__ int sm_sweep_chunk(int n, double (&a)[SM_NR][SM_NC], double (&rr)[SM_NR][SM_NC],
                                              double *blk, double *rsv, int ty, int tx, int *nneg = nullptr) {
    constexpr int KC = KR / 2;
    const int kend = min(n, 16 * KR + 16);
    int info = 0;
#pragma unroll 1      // (with a compile-time n the unrolled blocks of a chunk spilled 1342 VGPRs)
    for (int k0 = 16 * KR; k0 < kend; k0 += 4) {
        double *cb = blk + ((k0 >> 2) & 1) * SM_BLK, *rb = cb + 4 * SMALL_P;
        const int qc = tx - (k0 & 31), qr = ty - (k0 & 15);
        if ((unsigned) qc < 4u) {                // owners of columns k0 .. k0 + 3 publish them (rows ty + 16 r)
#pragma unroll
            for (int r = 0; r < SM_NR; ++r) cb[(ty + 16 * r) * 4 + qc] = a[r][KC];
        }
        if ((unsigned) qr < 4u) {                // owners of rows k0 .. k0 + 3 of R publish them (columns tx + 32 c)
#pragma unroll
            for (int c = 0; c < SM_NC; ++c) rb[(tx + 32 * c) * 4 + qr] = rr[KR][c];
        }
        __syncthreads();
        const double2 *cq = reinterpret_cast<const double2 *>(cb), *rq = reinterpret_cast<const double2 *>(rb);
        double d[4][4], ci[SM_NR][4], cj[SM_NC][4], rj[SM_NC][4];
#define SM_RD4(DST, SRC, IDX) { const double2 u_ = SRC[2 * (IDX)], v_ = SRC[2 * (IDX) + 1]; DST[0] = u_.x; DST[1] = u_.y; DST[2] = v_.x; DST[3] = v_.y; }
#pragma unroll
        for (int q = 0; q < 4; ++q) SM_RD4(d[q], cq, k0 + q)
#pragma unroll
        for (int r = KR; r < SM_NR; ++r) SM_RD4(ci[r], cq, ty + 16 * r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            if (c >= KC) SM_RD4(cj[c], cq, tx + 32 * c)
            if (c <= KC) SM_RD4(rj[c], rq, tx + 32 * c)
        }
#undef SM_RD4
        // the 4 x 4 factor of the diagonal block (every thread, redundantly).  No branch on the pivots' signs: with an early
        // return after each pivot the compiler fetched the diagonal block piece by piece behind the branches -- four LDS
        // round trips and four reciprocal square roots in a row before the panel reads were even issued.  A pivot that is
        // not positive turns everything after it into NaNs, harmlessly; the first such pivot is remembered in `info`.
        const double p0 = d[0][0];
        const double g0 = (SIGNED && p0 < 0.0) ? -1.0 : 1.0;
        const double rs0 = sm_rsqrt(SIGNED ? fabs(p0) : p0);
        const double l10 = d[1][0] * rs0, l20 = d[2][0] * rs0, l30 = d[3][0] * rs0;
        const double w10 = SIGNED ? g0 * l10 : l10, w20 = SIGNED ? g0 * l20 : l20, w30 = SIGNED ? g0 * l30 : l30;
        const double p1 = d[1][1] - l10 * w10;
        const double g1 = (SIGNED && p1 < 0.0) ? -1.0 : 1.0;
        const double rs1 = sm_rsqrt(SIGNED ? fabs(p1) : p1);
        const double l21 = (d[2][1] - l20 * w10) * rs1, l31 = (d[3][1] - l30 * w10) * rs1;
        const double w21 = SIGNED ? g1 * l21 : l21, w31 = SIGNED ? g1 * l31 : l31;
        const double p2 = (d[2][2] - l20 * w20) - l21 * w21;
        const double g2 = (SIGNED && p2 < 0.0) ? -1.0 : 1.0;
        const double rs2 = sm_rsqrt(SIGNED ? fabs(p2) : p2);
        const double l32 = ((d[3][2] - l30 * w20) - l31 * w21) * rs2;
        const double w32 = SIGNED ? g2 * l32 : l32;
        const double p3 = ((d[3][3] - l30 * w30) - l31 * w31) - l32 * w32;
        const double g3 = (SIGNED && p3 < 0.0) ? -1.0 : 1.0;
        const double rs3 = sm_rsqrt(SIGNED ? fabs(p3) : p3);
        const int bad = SIGNED ? (!(fabs(p0) > 0.0) ? 1 : !(fabs(p1) > 0.0) ? 2 : !(fabs(p2) > 0.0) ? 3 : !(fabs(p3) > 0.0) ? 4 : 0)
                               : (!(p0 > 0.0) ? 1 : !(p1 > 0.0) ? 2 : !(p2 > 0.0) ? 3 : !(p3 > 0.0) ? 4 : 0);
        info = (info == 0 && bad != 0) ? k0 + bad : info;
        if (SIGNED && nneg) *nneg += (p0 < 0.0) + (p1 < 0.0) + (p2 < 0.0) + (p3 < 0.0);
        // signed scales: what a finished column of the trailing matrix (and, inside this block, a finished row of R) is multiplied by
        const double q0 = SIGNED ? g0 * rs0 : rs0, q1 = SIGNED ? g1 * rs1 : rs1, q2 = SIGNED ? g2 * rs2 : rs2, q3 = SIGNED ? g3 * rs3 : rs3;
        if (ty == 0 && tx < 4) rsv[k0 + tx] = tx == 0 ? q0 : tx == 1 ? q1 : tx == 2 ? q2 : q3;
        // this thread's rows and columns of the panel: entry q is corrected by the pivots before it, then scaled.
        // (x, y, z, w) in: the published values; out: L[., k0 + q] resp. W[k0 + q, .] (up to the deferred sign, SIGNED)
#define SM_PANEL(V)                                                          \
        V[0] = V[0] * rs0;                                                   \
        V[1] = (V[1] - V[0] * w10) * rs1;                                    \
        V[2] = ((V[2] - V[0] * w20) - V[1] * w21) * rs2;                     \
        V[3] = (((V[3] - V[0] * w30) - V[1] * w31) - V[2] * w32) * rs3;
#define SM_PANEL_R(V)                                                        \
        V[0] = V[0] * q0;                                                    \
        V[1] = (V[1] - V[0] * l10) * q1;                                     \
        V[2] = ((V[2] - V[0] * l20) - V[1] * l21) * q2;                      \
        V[3] = (((V[3] - V[0] * l30) - V[1] * l31) - V[2] * l32) * q3;
#pragma unroll
        for (int r = KR; r < SM_NR; ++r) {
            SM_PANEL(ci[r])
            if (r == KR) {                       // rows up to k0 + q take no part in pivot q
                const int i = ty + 16 * r;
#pragma unroll
                for (int q = 0; q < 4; ++q) ci[r][q] = (i > k0 + q) ? ci[r][q] : 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            if (c >= KC) {
                SM_PANEL(cj[c])
                if (c == KC) {
                    const int j = tx + 32 * c;
#pragma unroll
                    for (int q = 0; q < 4; ++q) cj[c][q] = (j > k0 + q) ? cj[c][q] : 0.0;
                }
                if (SIGNED) { cj[c][0] *= g0; cj[c][1] *= g1; cj[c][2] *= g2; cj[c][3] *= g3; }   // the update is - sigma u u'
            }
            if (c <= KC) {                       // rows of W: zero right of their diagonal by construction
                SM_PANEL_R(rj[c])
            }
        }
#undef SM_PANEL
#undef SM_PANEL_R
#pragma unroll
        for (int r = KR; r < SM_NR; ++r) {       // rows that can lie below k0
#pragma unroll
            for (int c = 0; c < SM_NC; ++c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // trailing update (columns right of the pivot).  Only the lower triangle is ever read again; a thread's
                    // element (r, c) lies strictly above the diagonal for every thread when 16 r + 15 < 32 c
                    if (c >= KC && r >= 2 * c) a[r][c] -= ci[r][q] * cj[c][q];
                    if (c <= KC) rr[r][c] -= ci[r][q] * rj[c][q];    // forward substitution on the identity (columns up to it)
                }
            }
        }
    }
    return info;
}

template <bool LOGDET = true, bool SIGNED = false>
__device__ __forceinline__ int sm_sweep(int n, double (&a)[SM_NR][SM_NC], double (&rr)[SM_NR][SM_NC],
                                        double *blk, double *rsv, int ty, int tx, double *logdet, int *nneg = nullptr) {
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) rr[r][c] = (ty + 16 * r == tx + 32 * c) ? 1.0 : 0.0;
    int info = 0;
    if (!info && n > 0) info = sm_sweep_chunk<0, SIGNED>(n, a, rr, blk, rsv, ty, tx, nneg);
    if (!info && n > 16) info = sm_sweep_chunk<1, SIGNED>(n, a, rr, blk, rsv, ty, tx, nneg);
    if (!info && n > 32) info = sm_sweep_chunk<2, SIGNED>(n, a, rr, blk, rsv, ty, tx, nneg);
    if (!info && n > 48) info = sm_sweep_chunk<3, SIGNED>(n, a, rr, blk, rsv, ty, tx, nneg);
    if (!info && n > 64) info = sm_sweep_chunk<4, SIGNED>(n, a, rr, blk, rsv, ty, tx, nneg);
    if (!info && n > 80) info = sm_sweep_chunk<5, SIGNED>(n, a, rr, blk, rsv, ty, tx, nneg);
    if (!info && n > 96) info = sm_sweep_chunk<6, SIGNED>(n, a, rr, blk, rsv, ty, tx, nneg);
    if (!info && n > 112) info = sm_sweep_chunk<7, SIGNED>(n, a, rr, blk, rsv, ty, tx, nneg);
    __syncthreads();
    double ld = 0.0;
    if (!info) {
        // apply the deferred scale factors: column j of L and row i of W
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
            const double si = (i < n) ? (SIGNED ? fabs(rsv[i]) : rsv[i]) : 1.0;
#pragma unroll
            for (int c = 0; c < SM_NC; ++c) rr[r][c] *= si;
        }
        // log det = 2 sum log L_kk = -2 sum log rs_k: one logarithm per pivot, by the first n threads (the sum is finished
        // by the caller).  (Taking it from the diagonal elements where they live ran the long f64 log routine up to 32
        // times per wave under divergent masks: 8 us.)
        const int t = ty * 32 + tx;
        if (LOGDET && t < SMALL_P) blk[t] = (t < n) ? -2.0 * log(SIGNED ? fabs(rsv[t]) : rsv[t]) : 0.0;      // (the block images are free again)
    }
    if (!LOGDET) { *logdet = 0.0; return info; }
    __syncthreads();
    if (!info) {                                 // fixed-order sum: the same bits on every run
        const int t = ty * 32 + tx;
        if (t < 64) {
            double v = blk[t] + blk[t + 64];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (t == 0) blk[SMALL_P] = v;
        }
    }
    __syncthreads();
    if (!info) ld = blk[SMALL_P];
    __syncthreads();
    *logdet = ld;
    return info;
}

// L (lower, identity padded, ld = 128) and W = L^-1 (lower, zeros above, identity padded) to global memory.  A thread's
// elements of one (r, c) are 1 KB apart from its neighbours' (64 partial lines per store instruction: 14 us for the two
// matrices); they go through an LDS image (stride 129: conflict-free both ways) and leave as full lines.
// img: 128 x 129 doubles of LDS, free on entry; free again on return.
// G: leading dimension ldg; LOWER: leave what G holds above the diagonal alone
template <bool LOWER = false>
__device__ __forceinline__ void sm_store_one(int n, const double (&x)[SM_NR][SM_NC], double *img, double *G, long ldg, int ty, int tx, int tid) {
    constexpr int ld = SMALL_P + 1;
#pragma unroll
    for (int r = 0; r < SM_NR; ++r)
#pragma unroll
        for (int c = 0; c < SM_NC; ++c) {
            const int i = ty + 16 * r, j = tx + 32 * c;
            const bool in = (i < n && j < n);
            img[i + j * ld] = in ? ((i >= j) ? x[r][c] : 0.0) : ((i == j) ? 1.0 : 0.0);
        }
    __syncthreads();
    const int i = tid & (SMALL_P - 1), j0 = tid >> 7;       // 4 columns per pass, 128 consecutive rows each
#pragma unroll 8
    for (int j = j0; j < SMALL_P; j += SM_T / SMALL_P)
        if (!LOWER || i >= j) G[i + (long) j * ldg] = img[i + j * ld];
    __syncthreads();
}
__device__ __forceinline__ void sm_store_factor(int n, const double (&a)[SM_NR][SM_NC], const double (&rr)[SM_NR][SM_NC],
                                                double *img, double *L, double *W, int ty, int tx, int tid) {
    sm_store_one(n, a, img, L, SMALL_P, ty, tx, tid);
    sm_store_one(n, rr, img, W, SMALL_P, ty, tx, tid);
}

