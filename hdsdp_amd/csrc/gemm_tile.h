// gemm_tile.h -- device side of the fp64 MFMA GEMM family (one tile by one workgroup, and the two kernels built on it).
// Included by gemm_f64.hip (one workgroup per tile + the host side) and gemm_persist.hip (persistent workgroups): the two
// translation units instantiate disjoint sets of kernels, so the 4.5 minutes this code takes to compile split in two.
#pragma once
#include "hdm_common.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#define LDM 144             // M-major LDS row stride (doubles)
#define LDK 18              // K-major LDS row stride (doubles)
#define STAGE_DOUBLES 2304  // 16*144 == 128*18

struct HdmGemmDev {
    HdmGemmArgs a;
    const int2 *tiles;
    int ntiles;
    unsigned long long *dbg;  // diagnostic builds only (VAR & 32): 8 words per workgroup
};

// Tile staging: a (128 rows x 16 k) tile is 1024 chunks of 16 bytes; thread t moves chunks t, t+256, t+512,
// t+768, so that the 64 lanes of a wave always touch 1 KiB of consecutive global memory per instruction
// (full 128-byte lines) and every 8-lane ds_write_b128 group writes 128 consecutive LDS bytes = all 32
// write banks once (the first version gave each thread 64 contiguous bytes: 4-way write conflicts,
// SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE in profiles/r01_a_*).
//   M-major tile [k][i]: chunk c -> k = c >> 6, i = (c & 63) * 2
//   K-major tile [i][k]: chunk c -> i = c >> 3, k = (c & 7) * 2
// DUAL: the stream can continue on a second operand (the SYR2K form of congruence step 2, generic launches with A2/B2).
// Single-product kernels (congruence step 1, Gram) carry no second pointer, strides or countdown: seven registers per stager
// that the step-1 kernel was spilling to scratch and reloading inside its K loops.
template <bool KM, bool DUAL = true>
struct Stager {
    const double *p;  // this thread's chunk 0 of the current k block
    long qstride;     // elements between the thread's 4 chunks
    long kstep;       // elements between consecutive k blocks
    unsigned okmask;  // bit q: chunk q lies inside the matrix

    __device__ __forceinline__ void init(const double *X, long ld, long kblk, long seg_rows, long seg_extra, int rows,
                                         int x0, int kt0, int tid) {
        okmask = 0;
        if (KM) {
            const int i = tid >> 3, k2 = (tid & 7) * 2;
            // a 128-row tile never straddles a row segment, so the segment offset is tile-uniform
            const long segoff = seg_rows ? (x0 / seg_rows) * seg_extra : 0;
            p = X + segoff + (long) kt0 * kblk + (long) (x0 + i) * ld + k2;
            qstride = 32 * ld;
            kstep = kblk;
#pragma unroll
            for (int q = 0; q < 4; ++q) okmask |= (x0 + i + 32 * q < rows) ? (1u << q) : 0u;
        } else {
            const int k = tid >> 6, i2 = (tid & 63) * 2;
            p = X + ((long) kt0 * HDM_BK + k) * ld + x0 + i2;
            qstride = 4 * ld;
            kstep = (long) HDM_BK * ld;
            okmask = (x0 + i2 < rows) ? 0xFu : 0u;  // rows is even: the 2-double chunk is fully in or out
        }
        if (DUAL) {
            left = -1;  // single product: never switches
            p2 = p; qstride2 = qstride; kstep2 = kstep;
        }
    }
    const double *p2;  // second operand (dual-product mode): the stream continues there after `left` loads
    long qstride2, kstep2;
    int left;

    __device__ __forceinline__ void chain(const double *X2, long ld2, int x0, int kt0, int tid, int nst) {
        // same tile rows, same row count => the same okmask; only the base pointer and the strides change
        if (KM) {
            p2 = X2 + (long) kt0 * HDM_BK + (long) (x0 + (tid >> 3)) * ld2 + (tid & 7) * 2;
            qstride2 = 32 * ld2;
            kstep2 = HDM_BK;
        } else {
            p2 = X2 + ((long) kt0 * HDM_BK + (tid >> 6)) * ld2 + x0 + (tid & 63) * 2;
            qstride2 = 4 * ld2;
            kstep2 = (long) HDM_BK * ld2;
        }
        left = nst;
    }
    __device__ __forceinline__ void load(double2 (&r)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            r[q] = (okmask >> q) & 1u ? *reinterpret_cast<const double2 *>(p + q * qstride) : make_double2(0.0, 0.0);
        p += kstep;
        if (DUAL) { if (--left == 0) { p = p2; qstride = qstride2; kstep = kstep2; } }
    }
};

template <bool KM>
__device__ __forceinline__ void r2s(double *__restrict__ s, int tid, const double2 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double *dst;
        if (KM) {
            const int i = (tid >> 3) + 32 * q, k2 = (tid & 7) * 2;
            dst = s + i * LDK + k2;
        } else {
            const int k = (tid >> 6) + 4 * q, i2 = (tid & 63) * 2;
            dst = s + k * LDM + i2;
        }
        *reinterpret_cast<double2 *>(dst) = r[q];
    }
}

template <bool KM>
__device__ __forceinline__ double frag(const double *__restrict__ s, int x, int k) {
    return KM ? s[x * LDK + k] : s[k * LDM + x];
}

// ---------------------------------------------------------------------------------------------
// Roles 1-3: a stage body WITHOUT vector ALU instructions.
//
// What the fp64 matrix pipe loses in these kernels is not waiting.  tools/probes/issue_mix_probe.hip (profiles/r04_h_issue_mix.txt)
// puts N instructions of one class between the 16 MFMAs of a register-only loop: LDS reads and writes, buffer loads, scalar ALU
// work and barriers at the K loop's rates leave the pipe at 0.99 of its peak, with one or two waves per SIMD alike -- and EVERY
// vector ALU instruction (a 32-bit add as much as a 64-bit one, a v_cndmask, a v_mul_lo) takes about 11 cycles of matrix-pipe
// time: the fp64 MFMA runs on the vector ALU's multipliers, and nothing issues to them while an address is being added up.  The
// K loop of round 3 carried 28 such instructions per stage of 64 MFMAs -- LDS addresses rebuilt from a run-time buffer index,
// eight 64-bit global pointers, the selects of the operand switch --: 7.5 % of the pipe, which IS the 92.6 % MFMA-busy figure.
// So the stage body's addressing is arranged to need none:
//   * the LDS buffer index is a compile-time constant (stages come in pairs), every fragment read is ONE per-lane base register
//     plus a 16-bit instruction offset.  The reads are volatile so that the load/store optimiser leaves them alone: merging two
//     into a ds_read2_b64 (8-bit offsets) costs a new base per pair -- a v_add per stage, or sixteen more live registers;
//   * global memory is read with buffer loads: the k block's address lives in scalar registers (wave-uniform: an M-major
//     operand's row k + wave, a K-major operand's tile row), the lane's part is one 32-bit offset register that never changes,
//     the four chunks of a lane are scalar offsets; stepping to the next k block, clamping at the end of the range and the
//     switch to the second operand pair (step 2) are scalar adds and selects.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char hdm_lchar;
typedef unsigned hdm_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double lds_ld(const hdm_lchar *p) {
    return *reinterpret_cast<const __attribute__((address_space(3))) volatile double *>(p);
}
__device__ __forceinline__ void lds_st(hdm_lchar *p, const hdm_u4 &v) {
    *reinterpret_cast<__attribute__((address_space(3))) volatile hdm_u4 *>(p) = v;
}
// byte offset of fragment element (row 16 * sub, k = kk) in stage buffer cb, relative to a lane base that holds the lane's
// (row & 15 [+ 16 * wave part], k & 3) share
template <bool KM>
__device__ __forceinline__ constexpr int rd_off(int cb, int sub, int kk) {
    return cb * STAGE_DOUBLES * 8 + (KM ? (sub * 16 * LDK + kk) : (kk * LDM + sub * 16)) * 8;
}
// per-lane read base inside an operand's region (w16: the wave's first row, 0 or 16 for the interleaved quadrants, 0 for cells)
template <bool KM>
__device__ __forceinline__ int rd_lane(int w16, int l15, int lq) {
    return (KM ? ((w16 + l15) * LDK + lq) : (lq * LDM + w16 + l15)) * 8;
}
// per-lane base of the staging writes (chunk q of a lane: + 4608 q bytes in both layouts: 32 rows x LDK = 4 k x LDM)
template <bool KM>
__device__ __forceinline__ int wr_lane(int tid) {
    return (KM ? ((tid >> 3) * LDK + (tid & 7) * 2) : ((tid >> 6) * LDM + (tid & 63) * 2)) * 8;
}
__device__ __forceinline__ void st4(hdm_lchar *w, int cb, const hdm_u4 &r0, const hdm_u4 &r1, const hdm_u4 &r2, const hdm_u4 &r3) {
    lds_st(w + cb * STAGE_DOUBLES * 8, r0);
    lds_st(w + cb * STAGE_DOUBLES * 8 + 4608, r1);
    lds_st(w + cb * STAGE_DOUBLES * 8 + 9216, r2);
    lds_st(w + cb * STAGE_DOUBLES * 8 + 13824, r3);
}

// Scalar stager: the same chunks as Stager (thread t moves chunks t, t + 256, t + 512, t + 768 of a 128 x 16 tile), fetched
// with buffer loads whose address is (scalar base of the k block) + (scalar chunk offset) + (lane offset).  No row mask:
// rows past the matrix edge read whatever follows (the engine's operand buffers carry HDM_OPERAND_PAD bytes of slack,
// hdm_common.h).  Sound because output element (i, j) depends only on row i of one operand and row j of the other, and every
// epilogue masks rows/columns past the edge; it removes the eight exec-mask branches per stage, so the stage body is one basic
// block the scheduler can interleave.  DUAL (M-major only: congruence step 2): after `left` loads the stream continues on a second operand.
template <bool KM, bool DUAL>
struct SStager {
    static_assert(!(KM && DUAL), "the dual-product stream exists for M-major operands only");
    const char *g;            // wave-uniform: this wave's chunk 0 of the current k block, lane part excluded
    unsigned voff;            // the lane's byte offset (never changes)
    unsigned so1, so2, so3;   // wave-uniform byte offsets of chunks 1..3
    long kstep;               // bytes between consecutive k blocks
    int remain;               // k blocks not yet loaded; at <= 0 the loads return zeros (branch-free look-ahead)
    const char *g2;           // second operand
    unsigned t1, t2, t3;
    long kstep2;
    int left;

    __device__ __forceinline__ void init(const double *X, long ld, long kblk, long seg_rows, long seg_extra, int x0, int kt0,
                                         int tid, int wave) {
        if (KM) {
            const long segoff = seg_rows ? (x0 / seg_rows) * seg_extra : 0;
            g = reinterpret_cast<const char *>(X + segoff + (long) kt0 * kblk + (long) x0 * ld);
            voff = (unsigned) (((long) (tid >> 3) * ld + (tid & 7) * 2) * 8);
            so1 = (unsigned) (32 * ld * 8); so2 = 2 * so1; so3 = 3 * so1;
            kstep = kblk * 8;
        } else {
            g = reinterpret_cast<const char *>(X + ((long) kt0 * HDM_BK + wave) * ld + x0);
            voff = (unsigned) ((tid & 63) * 16);
            so1 = (unsigned) (4 * ld * 8); so2 = 2 * so1; so3 = 3 * so1;
            kstep = (long) HDM_BK * ld * 8;
        }
        remain = 1 << 30;
        if (DUAL) { left = -1; g2 = g; t1 = so1; t2 = so2; t3 = so3; kstep2 = kstep; }
    }
    __device__ __forceinline__ void chain(const double *X2, long ld2, int x0, int kt0, int wave, int nst) {
        g2 = reinterpret_cast<const char *>(X2 + ((long) kt0 * HDM_BK + wave) * ld2 + x0);
        t1 = (unsigned) (4 * ld2 * 8); t2 = 2 * t1; t3 = 3 * t1;
        kstep2 = (long) HDM_BK * ld2 * 8;
        left = nst;
    }
    __device__ __forceinline__ void load_nomask(hdm_u4 &r0, hdm_u4 &r1, hdm_u4 &r2, hdm_u4 &r3) {
        // raw buffer over [g, g + 2 GiB): stride 0, 32-bit data format; offsets stay far below the range (launcher check).
        // Past the end of the K range the buffer has NO records: every load is out of range and returns zeros without
        // touching memory -- a look-ahead beyond the last k block needs no branch, and a stage made of such loads adds
        // nothing to the accumulators (odd stage counts are run as pairs with one such stage at the end).
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(g), 0, (remain > 0) ? 0x7fffffff : 0, 0x00020000);
        r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
        r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so1, 0);
        r2 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so2, 0);
        r3 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so3, 0);
        remain -= 1;
        g += kstep;
    }
    // DUAL: the stream continues on the second operand pair.  Congruence step 2's main tiles call this at the one place of
    // their straight-line stage sequence where it belongs (scalar moves); a cell-dealt tile, whose stage count is a run-time
    // value, counts down instead (load_counted).  The countdown is NOT part of load_nomask: written as a condition it becomes a
    // branch around five moves, which cuts the stage body into scheduling regions -- the look-ahead loads then issue in a clump
    // after the barrier instead of between the fourth k-step's MFMAs.
    __device__ __forceinline__ void second() { g = g2; so1 = t1; so2 = t2; so3 = t3; kstep = kstep2; }
    __device__ __forceinline__ void load_counted(hdm_u4 &r0, hdm_u4 &r1, hdm_u4 &r2, hdm_u4 &r3) {
        load_nomask(r0, r1, r2, r3);
        if (DUAL) { if (--left == 0) second(); }
    }
    static constexpr bool dual = DUAL;
};

// Diagonal tiles of a lower-only product need 36 of their 64 16x16 sub-tiles (row block >= column block).  A 2x2
// split into 64x64 wave quadrants would leave one wave idle and give the tile the full 16-MFMA cadence; instead
// the 36 cells are dealt 9 per wave, so a diagonal tile runs at 9/16 of the time of a full tile:
//   wave 0: rows {5,6,7} x cols {0,1,2}   wave 1: rows {5,6,7} x cols {3,4,5}   wave 2: rows {2,3,4} x cols {0,1,2}
//   wave 3: the three 2x2 lower triangles on the diagonal: (0,0)(1,0)(1,1) (3,3)(4,3)(4,4) (6,6)(7,6)(7,7)
template <int W> struct DiagCells;
template <> struct DiagCells<0> { static constexpr int NC = 9, si[16] = {5, 5, 5, 6, 6, 6, 7, 7, 7}, sj[16] = {0, 1, 2, 0, 1, 2, 0, 1, 2}; };
template <> struct DiagCells<1> { static constexpr int NC = 9, si[16] = {5, 5, 5, 6, 6, 6, 7, 7, 7}, sj[16] = {3, 4, 5, 3, 4, 5, 3, 4, 5}; };
template <> struct DiagCells<2> { static constexpr int NC = 9, si[16] = {2, 2, 2, 3, 3, 3, 4, 4, 4}, sj[16] = {0, 1, 2, 0, 1, 2, 0, 1, 2}; };
template <> struct DiagCells<3> { static constexpr int NC = 9, si[16] = {0, 1, 1, 3, 4, 4, 6, 7, 7}, sj[16] = {0, 0, 1, 3, 3, 4, 6, 6, 7}; };
// Bottom-edge tiles (fewer than 128 valid rows: n = 2000 leaves 80 = 5 sub-tile rows, the Gram operand 88 = 6) keep
// all 8 sub-tile columns but only rv < 8 sub-tile rows.  The 2x2 quadrant split would run them at the full 16-MFMA
// cadence (the upper quadrants are full); instead wave w takes sub-tile columns {2w, 2w+1} over RV rows: 2 RV cells
// per wave, so the tile costs RV/8 of a full one.  RV is a compile-time 4..7 (fewer valid rows run as 4: the extra
// rows are zeros from the stager and masked by the epilogue); run-time masks would cut the MFMA stream into branches.
template <int W, int RV> struct EdgeCells {
    static constexpr int NC = 2 * RV;
    static constexpr int si[16] = {0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7};
    static constexpr int sj[16] = {2 * W, 2 * W + 1, 2 * W, 2 * W + 1, 2 * W, 2 * W + 1, 2 * W, 2 * W + 1,
                                   2 * W, 2 * W + 1, 2 * W, 2 * W + 1, 2 * W, 2 * W + 1, 2 * W, 2 * W + 1};
};

// The LAST diagonal tile of a lower-only product is both: diagonal and short (n = 2000: 80 rows = 5 sub-tile rows, the Gram
// operand 88 = 6).  Run as a full diagonal tile it costs 36 cells where RV (RV + 1) / 2 are needed (15 of 36 at RV = 5) -- and
// it is the diagonal tile with the LONGEST K range of step 2 (0.75 % of that kernel's MFMAs at n = 2000).  The needed cells,
// numbered row by row, are dealt round-robin to the four waves: 4 + 4 + 4 + 3 at RV = 5.
struct HdmCellList { int nc; int si[16]; int sj[16]; };
constexpr HdmCellList hdm_diag_edge_cells(int W, int RV) {
    HdmCellList L{};
    L.nc = 0;
    for (int k = 0; k < 16; ++k) { L.si[k] = 0; L.sj[k] = 0; }
    int c = 0;
    for (int i = 0; i < RV; ++i)
        for (int j = 0; j <= i; ++j, ++c)
            if ((c & 3) == W) { L.si[L.nc] = i; L.sj[L.nc] = j; L.nc += 1; }
    return L;
}
template <int W, int RV> struct DiagEdgeCells {
    static constexpr HdmCellList L = hdm_diag_edge_cells(W, RV);
    static constexpr int NC = L.nc;
    static constexpr int si[16] = {L.si[0], L.si[1], L.si[2], L.si[3], L.si[4], L.si[5], L.si[6], L.si[7],
                                   L.si[8], L.si[9], L.si[10], L.si[11], L.si[12], L.si[13], L.si[14], L.si[15]};
    static constexpr int sj[16] = {L.sj[0], L.sj[1], L.sj[2], L.sj[3], L.sj[4], L.sj[5], L.sj[6], L.sj[7],
                                   L.sj[8], L.sj[9], L.sj[10], L.sj[11], L.sj[12], L.sj[13], L.sj[14], L.sj[15]};
};

// tri = 1: diagonal tile of a lower-only product (entries above the diagonal are neither scaled nor stored)
template <class T>
__device__ __forceinline__ void cell_epilogue(const HdmGemmArgs &a, int z, int m0, int n0, int l15, int lq, int rv, int tri,
                                              const hdm_d4 (&acc)[4][4]) {
    if (a.epilogue == HDM_EPI_BLOCKED) {
        const long rs16 = a.blk_row_stride * 16;
        double *lane_base = a.C + (a.blk_row0 + z) * 16 + l15 + (long) lq * rs16;
#pragma unroll
        for (int c = 0; c < T::NC; ++c) {
            const int bi = (m0 >> 4) + T::si[c], bj = (n0 >> 4) + T::sj[c];
            if (T::si[c] >= rv || bi >= a.nblk || bj >= a.nblk) continue;
            const double sc = (bi == bj) ? 1.0 : 1.4142135623730951;
            const long sub = (long) bj * a.nblk - (long) bj * (bj - 1) / 2 + (bi - bj);
            double *q = lane_base + sub * 16 * rs16;
#pragma unroll
            for (int r = 0; r < 4; ++r) q[(long) (4 * r) * rs16] = sc * acc[c >> 2][c & 3][r];
        }
        return;
    }
    double *C = a.C + (a.epilogue == HDM_EPI_SLAB ? (long) z * a.slab_stride : (long) z * a.strideC);
#pragma unroll
    for (int c = 0; c < T::NC; ++c) {
        if (T::si[c] >= rv) continue;
        const int gi = m0 + T::si[c] * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gj = n0 + T::sj[c] * 16 + lq + 4 * r;
            if (gi < a.M && gj < a.N && (!tri || gi >= gj)) {
                double *q = C + gi + (long) gj * a.ldc;
                double v = a.alpha * acc[c >> 2][c & 3][r];
                if (a.beta != 0.0) v += a.beta * (*q);
                *q = v;
            }
        }
    }
}

// REP x { NA instructions of class MA, NB of class MB } for the machine scheduler (classes: 0x8 MFMA, 0x20 VMEM read,
// 0x100 LDS read, 0x200 LDS write)
template <int MA, int NA, int MB, int NB, int REP>
__device__ __forceinline__ void sgb_pairs() {
    if constexpr (REP > 0) {
        __builtin_amdgcn_sched_group_barrier(MA, NA, 0);
        __builtin_amdgcn_sched_group_barrier(MB, NB, 0);
        sgb_pairs<MA, NA, MB, NB, REP - 1>();
    }
}

// NM MFMAs with NO instructions of class MASK spread between them
template <int NM, int MASK, int NO>
__device__ __forceinline__ void sgb_spread() {
    if constexpr (NM <= 0) {
        if constexpr (NO > 0) __builtin_amdgcn_sched_group_barrier(MASK, NO, 0);
    } else if constexpr (NO <= 0) {
        __builtin_amdgcn_sched_group_barrier(0x8, NM, 0);
    } else if constexpr (NM >= NO) {
        constexpr int per = NM / NO;
        sgb_pairs<0x8, per, MASK, 1, NO>();
        if constexpr (NM - per * NO > 0) __builtin_amdgcn_sched_group_barrier(0x8, NM - per * NO, 0);
    } else {
        constexpr int per = (NO + NM - 1) / NM;
        sgb_pairs<0x8, 1, MASK, per, NM>();
    }
}

// whole K loop + epilogue of a cell-dealt tile (one-stage look-ahead, at most 16 accumulators).  Unmasked branch-free
// staging loads as in the main loop (callers: roles 1-3 only); the eight global loads are issued among the first
// MFMAs of a stage and the eight LDS writes among the last.
template <class T> constexpr bool cell_uses_row(int q) {
    for (int c = 0; c < T::NC; ++c) if (T::si[c] == q) return true;
    return false;
}
template <class T> constexpr bool cell_uses_col(int q) {
    for (int c = 0; c < T::NC; ++c) if (T::sj[c] == q) return true;
    return false;
}
template <class T> constexpr int cell_frag_count() {
    int k = 0;
    for (int q = 0; q < 8; ++q) k += (cell_uses_row<T>(q) ? 1 : 0) + (cell_uses_col<T>(q) ? 1 : 0);
    return k;
}

// whole K loop + epilogue of a cell-dealt tile (at most 16 accumulators), same rotated stage as the main loop: the
// barrier sits before the fourth k-step, whose MFMAs cover the LDS reads of the next stage's first k-step and the issue
// of the look-ahead loads; unmasked branch-free staging (callers: roles 1-3 only); stages in pairs, so that the LDS buffer
// index is a compile-time constant and the stage body carries no vector ALU work (SStager, above).
template <class T, bool AKM, bool BKM, class STA, class STB>
__device__ __forceinline__ void cell_tile(const HdmGemmArgs &a, STA &stA, STB &stB, hdm_lchar *sb,
                                          int nst, int tid, int z, int m0, int n0, int l15, int lq, int rv, int tri) {
    hdm_d4 acc[4][4];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c >> 2][c & 3] = (hdm_d4){0.0, 0.0, 0.0, 0.0};
    hdm_u4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    double fr0[8], fc0[8], fr1[8], fc1[8];   // two fragment sets; only the entries this wave's cells name are ever loaded
    constexpr int NC = T::NC;                // MFMAs per k-step
    constexpr int NF = cell_frag_count<T>(); // LDS reads per fragment set
    const hdm_lchar *rdA = sb + rd_lane<AKM>(0, l15, lq);
    const hdm_lchar *rdB = sb + 2 * STAGE_DOUBLES * 8 + rd_lane<BKM>(0, l15, lq);
    hdm_lchar *wrA = sb + wr_lane<AKM>(tid);
    hdm_lchar *wrB = sb + 2 * STAGE_DOUBLES * 8 + wr_lane<BKM>(tid);
#define HDM_CLDF(FR, FC, CB, kk)                                                                            \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                         \
        if (cell_uses_row<T>(q)) FR[q] = lds_ld(rdA + rd_off<AKM>(CB, q, kk));                              \
        if (cell_uses_col<T>(q)) FC[q] = lds_ld(rdB + rd_off<BKM>(CB, q, kk));                              \
    }
#define HDM_CMMA(FR, FC)                                                                                    \
    _Pragma("unroll") for (int c = 0; c < NC; ++c)                                                          \
        acc[c >> 2][c & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(FC[T::sj[c]], FR[T::si[c]], acc[c >> 2][c & 3], 0, 0, 0);
    // one stage that has a successor, on buffer CB
#define HDM_CSTAGE(CB)                                                                                      \
    {                                                                                                       \
        HDM_CLDF(fr1, fc1, CB, 4)                                                                           \
        HDM_CMMA(fr0, fc0)                                                                                  \
        HDM_CLDF(fr0, fc0, CB, 8)                                                                           \
        HDM_CMMA(fr1, fc1)                                                                                  \
        HDM_CLDF(fr1, fc1, CB, 12)                                                                          \
        st4(wrA, (CB) ^ 1, ra0, ra1, ra2, ra3); st4(wrB, (CB) ^ 1, rb0, rb1, rb2, rb3);                     \
        HDM_CMMA(fr0, fc0)                                                                                  \
        sgb_spread<NC, 0x100, NF>();                                                                        \
        sgb_spread<NC, 0x100, NF>();                                                                        \
        sgb_spread<NC / 2, 0x100, NF>();                                                                    \
        sgb_spread<NC - NC / 2, 0x200, 8>();                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __syncthreads();                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        HDM_CLDF(fr0, fc0, (CB) ^ 1, 0)                                                                     \
        stA.load_counted(ra0, ra1, ra2, ra3); stB.load_counted(rb0, rb1, rb2, rb3);   /* stage t+2, or zeros past the end */ \
        HDM_CMMA(fr1, fc1)                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);                                                 \
        sgb_spread<NC, 0x20, 8>();                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
#define HDM_CLAST(CB)                                                                                       \
    {                                                                                                       \
        HDM_CLDF(fr1, fc1, CB, 4)                                                                           \
        HDM_CMMA(fr0, fc0)                                                                                  \
        HDM_CLDF(fr0, fc0, CB, 8)                                                                           \
        HDM_CMMA(fr1, fc1)                                                                                  \
        HDM_CLDF(fr1, fc1, CB, 12)                                                                          \
        HDM_CMMA(fr0, fc0)                                                                                  \
        HDM_CMMA(fr1, fc1)                                                                                  \
    }
    stA.remain = nst; stB.remain = nst;
    if (nst > 0) {
        stA.load_counted(ra0, ra1, ra2, ra3); stB.load_counted(rb0, rb1, rb2, rb3);
        st4(wrA, 0, ra0, ra1, ra2, ra3); st4(wrB, 0, rb0, rb1, rb2, rb3);
        stA.load_counted(ra0, ra1, ra2, ra3); stB.load_counted(rb0, rb1, rb2, rb3);   // stage 1, in flight
    }
    __syncthreads();
    if (nst > 0) {
        HDM_CLDF(fr0, fc0, 0, 0)
        // stages in pairs on buffers 0, 1; an odd count ends with a stage of zeros (SStager::load_nomask)
        const int npairs = (nst + 1) >> 1;
        for (int t = 0; t + 1 < npairs; ++t) { HDM_CSTAGE(0) HDM_CSTAGE(1) }
        HDM_CSTAGE(0) HDM_CLAST(1)
    }
#undef HDM_CLDF
#undef HDM_CMMA
#undef HDM_CSTAGE
#undef HDM_CLAST
    cell_epilogue<T>(a, z, m0, n0, l15, lq, rv, tri, acc);
}

// ---------------------------------------------------------------------------------------------
// Congruence step 2's main tiles without LDS and without barriers ("direct" body).
//
// Both operands of step 2 (U = the step-1 output T, and W = Linv) are M-major: the 16 rows of an MFMA sub-tile are 128
// consecutive bytes, so the fragment the MFMA wants from lane l -- element (row l & 15, k + (l >> 4)) -- is ONE
// global_load_dwordx2 per lane whose 64 lanes cover four full 128-byte lines.  The f64 MFMA is slow enough (16 x 16 x 4 in
// 64 cycles) that a wave needs only 8 such loads per 16 MFMAs, which L1/L2 deliver with room to spare; so each wave
// streams its own fragments straight into registers, two k-steps (32 MFMAs, > 2000 cycles) ahead, and never meets its
// siblings: no LDS image, no __syncthreads.  In the LDS kernel a wave parks at one barrier per 64 MFMAs waiting for the
// slowest of its three siblings, each of which shares its SIMD with a wave of the co-resident workgroup; whenever both
// waves of a SIMD are parked the matrix pipe idles (MFMA busy 87 % in profiles/r02_a).  Here the only thing a wave ever
// waits for is its own loads.  The price is L1/L2 request traffic: a row panel is fetched by the two waves that own
// its sub-tiles instead of once per workgroup.
// Tiles: tm > tn, all 128 rows valid; the diagonal and bottom-edge tiles of the same launch keep the cell-dealt LDS
// paths (hdm_gemm_kernel calls this body where its LDS main loop used to be, VAR & 256).  K loop: both products over k < (tn + 1) * 128 in macro-steps of 8 k; the sixteen macro-steps
// of the last K block (the B-side operand's diagonal block) run only the live column sub-tiles, as in the LDS kernel.
// ---------------------------------------------------------------------------------------------
template <int JLO>
__device__ __forceinline__ void cd_mma(hdm_d4 (&acc)[4][4], const double (&fa)[4], const double (&fb)[4]) {
#pragma unroll
    for (int j = JLO; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[j], fb[i], acc[j][i], 0, 0, 0);
}

__device__ __forceinline__ void cong2_direct_body(const HdmGemmArgs &a, int z, int tm, int tn, int wm, int wn, int l15, int lq) {
    const int m0 = tm * HDM_TILE, n0 = tn * HDM_TILE;
    hdm_d4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (hdm_d4){0.0, 0.0, 0.0, 0.0};

    // product 1: rows of U (tile tm) x rows of W (tile tn); product 2: rows of W (tile tm) x rows of U (tile tn)
    const double *U = a.A + (long) z * a.strideA;      // T_z, M-major, ld = lda
    const double *W = a.B;                             // Linv, M-major, ld = ldb (shared by the batch)
    const long ldu = a.lda, ldw = a.ldb;
    const int nms_full = tn * 16;                      // macro-steps (8 k each) before the last K block, per product

    double fa0[4], fb0[4], fa1[4], fb1[4], fa2[4], fb2[4], fa3[4], fb3[4];   // set A = (0,1): k-steps 0,1 of a macro-step; set B = (2,3)
#define CD_LOAD(FA, FB, pr, pc)                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) FB[i] = (pr)[32 * i];                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) FA[j] = (pc)[32 * j];
    for (int prod = 0; prod < 2; ++prod) {
        const double *R = prod == 0 ? U : W, *C = prod == 0 ? W : U;
        const long ldr = prod == 0 ? ldu : ldw, ldc = prod == 0 ? ldw : ldu;
        // this lane's element of row sub-tile (2i + wm) / column sub-tile (2j + wn) at k = lq: + 32 * i doubles per sub-tile
        const double *pr = R + (m0 + wm * 16 + l15) + (long) lq * ldr;
        const double *pc = C + (n0 + wn * 16 + l15) + (long) lq * ldc;
        const long r4 = 4 * ldr, c4 = 4 * ldc;
        // macro-step 0 into set A
        CD_LOAD(fa0, fb0, pr, pc) CD_LOAD(fa1, fb1, pr + r4, pc + c4)
        pr += 2 * r4; pc += 2 * c4;
        // full macro-steps come in pairs (A then B); nms_full is even
        for (int ms = 0; ms < nms_full; ms += 2) {
            CD_LOAD(fa2, fb2, pr, pc) CD_LOAD(fa3, fb3, pr + r4, pc + c4)
            pr += 2 * r4; pc += 2 * c4;
            __builtin_amdgcn_sched_barrier(0);
            cd_mma<0>(acc, fa0, fb0); cd_mma<0>(acc, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            CD_LOAD(fa0, fb0, pr, pc) CD_LOAD(fa1, fb1, pr + r4, pc + c4)
            pr += 2 * r4; pc += 2 * c4;
            __builtin_amdgcn_sched_barrier(0);
            cd_mma<0>(acc, fa2, fb2); cd_mma<0>(acc, fa3, fb3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // last K block: 16 macro-steps h = 0..15 (k = n0 + 8h ..): column sub-tiles below h / 4 (in the wave's own
        // numbering: 2j + wn < h / 2) are structurally zero rows of the triangular operand
#define CD_PAIR(JA, JB, LAST)                                                             \
        CD_LOAD(fa2, fb2, pr, pc) CD_LOAD(fa3, fb3, pr + r4, pc + c4)                     \
        pr += 2 * r4; pc += 2 * c4;                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                \
        cd_mma<JA>(acc, fa0, fb0); cd_mma<JA>(acc, fa1, fb1);                             \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (!(LAST)) { CD_LOAD(fa0, fb0, pr, pc) CD_LOAD(fa1, fb1, pr + r4, pc + c4) pr += 2 * r4; pc += 2 * c4; } \
        __builtin_amdgcn_sched_barrier(0);                                                \
        cd_mma<JB>(acc, fa2, fb2); cd_mma<JB>(acc, fa3, fb3);                             \
        __builtin_amdgcn_sched_barrier(0);
        // macro-steps (0,1) (2,3) ... (14,15): stage s = h / 2 of the LDS kernel keeps column sub-tiles j >= s / 2
        CD_PAIR(0, 0, false) CD_PAIR(0, 0, false) CD_PAIR(1, 1, false) CD_PAIR(1, 1, false)
        CD_PAIR(2, 2, false) CD_PAIR(2, 2, false) CD_PAIR(3, 3, false) CD_PAIR(3, 3, true)
#undef CD_PAIR
    }
#undef CD_LOAD
    // epilogue: the 16x16-blocked, sqrt(2)-weighted layout of the congruence output (same as the LDS kernel's)
    const long rs16 = a.blk_row_stride * 16;
    double *lane_base = a.C + (a.blk_row0 + z) * 16 + l15 + (long) lq * rs16;
    const double rt2 = 1.4142135623730951;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bi = (m0 >> 4) + 2 * i + wm;
            const int bj = (n0 >> 4) + 2 * j + wn;
            if (bi < bj || bi >= a.nblk) continue;
            const double sc = (bi == bj) ? 1.0 : rt2;
            const long sub = (long) bj * a.nblk - (long) bj * (bj - 1) / 2 + (bi - bj);
            double *q = lane_base + sub * 16 * rs16;
#pragma unroll
            for (int r = 0; r < 4; ++r) q[(long) (4 * r) * rs16] = sc * acc[j][i][r];
        }
    }
}


// one tile (batch entry z, entry t of the tile list) by one workgroup; sidx: this tile's slot in the diagnostic stamp buffer
template <bool AKM, bool BKM, int ROLE, int VAR>
__device__ __forceinline__ void hdm_gemm_tile(const HdmGemmDev &p, const int z, const int t, const long sidx) {
    // (declared here, not handed in as a pointer: through a pointer parameter the LDS addresses became 64-bit run-time
    // values instead of instruction immediates, and step 2's kernel spilled 113 VGPRs)
    __shared__ __attribute__((aligned(16))) double smem[4 * STAGE_DOUBLES];
    double *sA = smem;                      // [2][STAGE]
    double *sB = smem + 2 * STAGE_DOUBLES;  // [2][STAGE]
    hdm_lchar *sb = (hdm_lchar *) smem;     // the same four stage buffers as LDS byte addresses (roles 1-3)
    const HdmGemmArgs &a = p.a;
    const int tm = p.tiles[t].x, tn = p.tiles[t].y;

    const int tid = threadIdx.x;
    unsigned long long t_start = 0, t_pro = 0, t_loop = 0;
    if ((VAR & 32) && p.dbg) t_start = __builtin_amdgcn_s_memtime();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: tile/sub-tile index math runs on the SALU
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;

    const int m0 = tm * HDM_TILE, n0 = tn * HDM_TILE;
    long kbeg = 0, kend = a.K;
    if (a.klimit == HDM_KLIM_BY_M) kend = min((long) a.K, (long) (tm + 1) * HDM_TILE);
    if (a.klimit == HDM_KLIM_BY_N) kend = min((long) a.K, (long) (tn + 1) * HDM_TILE);
    if (a.klimit == HDM_KLIM_BAND) { kbeg = (long) tn * HDM_TILE; kend = min((long) a.K, (long) (tm + 1) * HDM_TILE); }
    const double *A = a.A, *B = a.B;
    if (a.epilogue == HDM_EPI_SLAB) {
        kbeg = a.k_base + (long) z * a.k_chunk;
        kend = min(kend, kbeg + a.k_chunk);
    } else {
        A += (long) z * a.strideA;
        B += (long) z * a.strideB;
    }

    hdm_d4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (hdm_d4){0.0, 0.0, 0.0, 0.0};

    const int kt0 = (int) (kbeg / HDM_BK), kt1 = (int) (kend / HDM_BK);
    // Congruence step 2, full diagonal tiles: their own launch (role HDM_ROLE_CONG2D, hdm_launch_gemm splits role 2's tile
    // list).  There the second product is the transpose of the first (W_t U_t^T = (U_t W_t^T)^T: the launcher holds role 2
    // to A2 == B, B2 == A), so the tile is P + P^T with ONE product P = U_t W_t^T over all 64 cells in the main loop -- 64
    // cell-passes where the 36-cell deal of both products ran 72, at the main loop's MFMA density, with the triangular
    // live ranges of the last K block on BOTH operands (U and W are lower triangular: stage s keeps row and column
    // sub-tiles >= s) -- and the transpose is added in the epilogue through LDS.  A kernel of its own because the two stage
    // sequences in one function cost the step-2 kernel 65 spilled VGPRs (it has none without).
    constexpr bool symdiag = (ROLE == HDM_ROLE_CONG2D);
    // diagnostic builds: the cell-dealt tiles leave a stamp too (start, end, where; d[6] = -kind), so that a timeline sees
    // every workgroup of a launch
#define HDM_CELL_STAMP(KIND)                                                                        \
    if ((VAR & 32) && p.dbg && tid == 0) {                                                          \
        unsigned long long *d = p.dbg + (size_t) sidx * 8;                                    \
        d[0] = t_start; d[1] = t_start; d[2] = t_start; d[3] = __builtin_amdgcn_s_memtime();        \
        d[4] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);                                \
        d[5] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);                                \
        d[6] = (unsigned long long) (long long) -(KIND);                                            \
        d[7] = __builtin_amdgcn_s_memrealtime();                                                    \
    }

    if constexpr (ROLE != HDM_ROLE_GENERIC) {
    // ======================================================================== roles 1-3: unmasked scalar staging, rotated loop
    constexpr bool DUAL = (ROLE == HDM_ROLE_CONG2);
    SStager<AKM, DUAL> stA;
    SStager<BKM, DUAL> stB;
    // VAR & 128: timing-only ablation (wrong results): every tile stages rows 0..127, so all operand traffic hits in L2
    long ldb = a.ldb;
    if (ROLE == HDM_ROLE_CONG1 && a.b_sky) {
        // A_L in skyline storage (hdm_common.h): tile column tn reads panel tn, a plain K-major matrix of leading dimension
        // N - 128 tn whose element (row 128 tn, k 128 tn) sits at the panel's start -- the pointer is moved back so that
        // the stager's absolute (row, k) arithmetic lands there
        ldb = a.N - 128L * tn;
        B += hdm_sky_panel(tn, a.N) - 128L * tn * (ldb + 1);
    }
    stA.init(A, a.lda, a.a_kblk, a.seg_rows, a.seg_extra, (VAR & 128) ? 0 : m0, kt0, tid, wave);
    stB.init(B, ldb, a.b_kblk, a.seg_rows, a.seg_extra, (VAR & 128) ? 0 : n0, kt0, tid, wave);
    int npass = 1;
    if constexpr (DUAL) {
        if (a.A2) {
            npass = 2;
            stA.chain(a.A2 + (long) z * a.strideA2, a.lda2, m0, kt0, wave, kt1 - kt0);
            stB.chain(a.B2 + (long) z * a.strideB2, a.ldb2, n0, kt0, wave, kt1 - kt0);
        }
    }
    // (the cell paths stage with UNMASKED loads, which needs the slack the engine gives those operand buffers and the
    // launcher verifies (hdm_launch_gemm: operand spans); a generic launch -- Cholesky updates, the small products of the
    // rank-one path on buffers of a few KB -- takes the masked loop below, whose epilogue knows about diagonal tiles.  A
    // generic diagonal tile on this path once read 16 KB past a 2 KB operand: a device fault.)
    if (!symdiag && a.lower_only && tm == tn) {   // workgroup-uniform: diagonal tile, 36-cell scheme
        const int nst = (kt1 - kt0) * npass;
        const int rvd = (a.M - m0 + 15) >> 4;                      // valid sub-tile rows (= columns) of this tile
        if (rvd < 8) {                                             // the last, short diagonal tile: only its needed cells
#define HDM_DEDGE(RV)                                                                                                      \
    switch (wave) {                                                                                                        \
        case 0: cell_tile<DiagEdgeCells<0, RV>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, rvd, 1); break;  \
        case 1: cell_tile<DiagEdgeCells<1, RV>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, rvd, 1); break;  \
        case 2: cell_tile<DiagEdgeCells<2, RV>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, rvd, 1); break;  \
        default: cell_tile<DiagEdgeCells<3, RV>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, rvd, 1); break; \
    }
            if (rvd <= 4) { HDM_DEDGE(4) } else if (rvd == 5) { HDM_DEDGE(5) } else if (rvd == 6) { HDM_DEDGE(6) } else { HDM_DEDGE(7) }
#undef HDM_DEDGE
            HDM_CELL_STAMP(3)
            return;
        }
        if constexpr (ROLE != HDM_ROLE_CONG2) {   // (step 2's full diagonal tiles are role HDM_ROLE_CONG2D's)
            switch (wave) {
                case 0: cell_tile<DiagCells<0>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, 8, 1); break;
                case 1: cell_tile<DiagCells<1>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, 8, 1); break;
                case 2: cell_tile<DiagCells<2>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, 8, 1); break;
                default: cell_tile<DiagCells<3>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, 8, 1); break;
            }
            HDM_CELL_STAMP(1)
        }
        return;
    }
    if (!symdiag && tm != tn && m0 + HDM_TILE > a.M && n0 + HDM_TILE <= a.N) {
        // workgroup-uniform: bottom-edge tile below the diagonal, rv < 8 valid sub-tile rows
        const int nst = (kt1 - kt0) * npass;
        const int rv = (a.M - m0 + 15) >> 4;
#define HDM_EDGE(RV)                                                                                                     \
    switch (wave) {                                                                                                      \
        case 0: cell_tile<EdgeCells<0, RV>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, rv, 0); break;  \
        case 1: cell_tile<EdgeCells<1, RV>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, rv, 0); break;  \
        case 2: cell_tile<EdgeCells<2, RV>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, rv, 0); break;  \
        default: cell_tile<EdgeCells<3, RV>, AKM, BKM>(a, stA, stB, sb, nst, tid, z, m0, n0, l15, lq, rv, 0); break; \
    }
        if (rv <= 4) { HDM_EDGE(4) } else if (rv == 5) { HDM_EDGE(5) } else if (rv == 6) { HDM_EDGE(6) } else { HDM_EDGE(7) }
#undef HDM_EDGE
        HDM_CELL_STAMP(2)
        return;
    }

    if constexpr (ROLE == HDM_ROLE_CONG2 && (VAR & 256) != 0) {
        // main tile of congruence step 2 (tm > tn, 128 valid rows; the two cell paths above took the others): fragments
        // straight from global memory, no LDS image, no barrier
        cong2_direct_body(a, z, tm, tn, wm, wn, l15, lq);
        return;
    }
    {
        // Rotated loop: the barrier sits between the third and the fourth k-step of a stage.  By then the wave holds
        // the fourth k-step's fragments in registers, so after the barrier it first issues the LDS reads of the NEXT
        // stage's first k-step and then runs the 16 MFMAs of the fourth k-step, which cover that LDS latency; the look-ahead
        // loads (unmasked, branch-free) and the LDS writes of the next stage are spread over the first three k-steps.
        //
        // Triangular operands.  In the congruence kernels the K block that lies on an operand's diagonal is half
        // zeros: for stage s (16 k's) of such a block whole 16-row sub-tiles of the operand are structurally zero
        //   step 2 (both products, last K block, B side lower triangular): column sub-tiles < s are dead,
        //   step 1, first K block (A_L on the B side: k >= column): column sub-tiles > s are dead,
        //   step 1, last K block (Linv on the A side: k <= row): row sub-tiles < s are dead.
        // The stage body exists in variants that run only the live range [JLO..JHI] x [ILO..3] of a wave's 4 x 4
        // sub-tiles; with the interleaved ownership the live sub-tiles are spread evenly over the four waves.
        hdm_u4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;   // staging registers, named so they are never an alloca
        const int nst = (kt1 - kt0) * npass;
        double fa0[4], fb0[4], fa1[4], fb1[4];   // two fragment sets, named (not indexed) so they stay in registers
        // this lane's fragment bases (operand A: rows 32 i + 16 wm + l15; operand B: rows 32 j + 16 wn + l15) and staging bases
        const hdm_lchar *rdA = sb + rd_lane<AKM>(wm * 16, l15, lq);
        const hdm_lchar *rdB = sb + 2 * STAGE_DOUBLES * 8 + rd_lane<BKM>(wn * 16, l15, lq);
        hdm_lchar *wrA = sb + wr_lane<AKM>(tid);
        hdm_lchar *wrB = sb + 2 * STAGE_DOUBLES * 8 + wr_lane<BKM>(tid);
#define HDM_LDF(FA, FB, CB, kk)                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) FB[i] = lds_ld(rdA + rd_off<AKM>(CB, 2 * i, kk)); \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) FA[j] = lds_ld(rdB + rd_off<BKM>(CB, 2 * j, kk));
#define HDM_MMA(FA, FB, JLO, JHI, ILO)                                                            \
    _Pragma("unroll") for (int j = (JLO); j <= (JHI); ++j)                                        \
        _Pragma("unroll") for (int i = (ILO); i < 4; ++i)                                         \
            acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[j], FB[i], acc[j][i], 0, 0, 0);
        constexpr int NR = 8;   // LDS read instructions per fragment set (one ds_read_b64 per fragment)
        // one stage that has a successor, on buffer CB: 48 MFMAs, barrier, next stage's first fragments + look-ahead loads, 16 MFMAs
#define HDM_STAGE(CB, JLO, JHI, ILO, SW)                                                                             \
    {                                                                                                                \
        constexpr int NK = ((JHI) - (JLO) + 1) * (4 - (ILO));   /* live MFMAs per k-step */                          \
        HDM_LDF(fa1, fb1, CB, 4)                                                                                     \
        HDM_MMA(fa0, fb0, JLO, JHI, ILO)                                                                             \
        HDM_LDF(fa0, fb0, CB, 8)                                                                                     \
        HDM_MMA(fa1, fb1, JLO, JHI, ILO)                                                                             \
        HDM_LDF(fa1, fb1, CB, 12)                                                                                    \
        st4(wrA, (CB) ^ 1, ra0, ra1, ra2, ra3); st4(wrB, (CB) ^ 1, rb0, rb1, rb2, rb3);                              \
        HDM_MMA(fa0, fb0, JLO, JHI, ILO)                                                                             \
        if constexpr (NK == 16) {                                                                                    \
            /* 16 MFMAs with the second k-step's fragment reads, 16 with the third's, then the fourth's reads and */ \
            /* the 8 LDS writes among the last 16 */                                                                 \
            sgb_pairs<0x8, 2, 0x100, 1, NR>();                                                                       \
            sgb_pairs<0x8, 2, 0x100, 1, NR>();                                                                       \
            sgb_pairs<0x8, 1, 0x100, NR / 2, 2>();                                                                   \
            sgb_pairs<0x8, 1, 0x200, 1, 8>();                                                                        \
            __builtin_amdgcn_sched_group_barrier(0x8, 6, 0);                                                         \
        } else {                                                                                                     \
            sgb_spread<NK, 0x100, NR>();                                                                             \
            sgb_spread<NK, 0x100, NR>();                                                                             \
            sgb_spread<NK / 2, 0x100, NR>();                                                                         \
            sgb_spread<NK - NK / 2, 0x200, 8>();                                                                     \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __syncthreads();                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        HDM_LDF(fa0, fb0, (CB) ^ 1, 0)                                                                               \
        stA.load_nomask(ra0, ra1, ra2, ra3); stB.load_nomask(rb0, rb1, rb2, rb3);   /* stage t+2, or zeros past the end */ \
        if constexpr (SW) { stA.second(); stB.second(); }   /* that was the first pair's last k block */               \
        HDM_MMA(fa1, fb1, JLO, JHI, ILO)                                                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);                                                          \
        if constexpr (NK == 16) {                                                                                    \
            sgb_pairs<0x8, 1, 0x20, 1, 8>();                                                                         \
            __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);                                                         \
        } else {                                                                                                     \
            sgb_spread<NK, 0x20, 8>();                                                                               \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
        // the last stage: nothing left to stage
#define HDM_LAST(CB, JLO, JHI, ILO)                                                               \
    {                                                                                             \
        HDM_LDF(fa1, fb1, CB, 4)                                                                  \
        HDM_MMA(fa0, fb0, JLO, JHI, ILO)                                                          \
        HDM_LDF(fa0, fb0, CB, 8)                                                                  \
        HDM_MMA(fa1, fb1, JLO, JHI, ILO)                                                          \
        HDM_LDF(fa1, fb1, CB, 12)                                                                 \
        HDM_MMA(fa0, fb0, JLO, JHI, ILO)                                                          \
        HDM_MMA(fa1, fb1, JLO, JHI, ILO)                                                          \
    }
        stA.remain = nst; stB.remain = nst;
        if (nst > 0) {
            stA.load_nomask(ra0, ra1, ra2, ra3); stB.load_nomask(rb0, rb1, rb2, rb3);
            st4(wrA, 0, ra0, ra1, ra2, ra3); st4(wrB, 0, rb0, rb1, rb2, rb3);
            stA.load_nomask(ra0, ra1, ra2, ra3); stB.load_nomask(rb0, rb1, rb2, rb3);   // stage 1, in flight
        }
        __syncthreads();
        if ((VAR & 32) && p.dbg) t_pro = __builtin_amdgcn_s_memtime();
        if (nst > 0) { HDM_LDF(fa0, fb0, 0, 0) }
        // Stages come in PAIRS on buffers 0, 1 (HDM_RUN2): every K block is eight stages, so every stage sequence of the
        // congruence roles is even and the buffer index of a stage is a compile-time constant.
#define HDM_RUN2(JLO, JHI, ILO) { HDM_STAGE(0, JLO, JHI, ILO, false) HDM_STAGE(1, JLO, JHI, ILO, false) }
#define HDM_END2(JLO, JHI, ILO) { HDM_STAGE(0, JLO, JHI, ILO, false) HDM_LAST(1, JLO, JHI, ILO) }
#define HDM_RUN2SW(JLO, JHI, ILO) { HDM_STAGE(0, JLO, JHI, ILO, false) HDM_STAGE(1, JLO, JHI, ILO, true) }
        // The stage sequence of a tile is straight-line: plain loops over the full stages, and the eight stages of a
        // diagonal K block unrolled with compile-time live ranges (stage s of such a block keeps sub-tiles s/2.. or
        // ..s/2 of the wave's four: with the interleaved ownership that bound is the same for every wave).  A
        // run-time switch between stage bodies INSIDE the loop is not an option: two copies of the body under a branch
        // make the register allocator spill 350-480 VGPRs.
        if constexpr (ROLE == HDM_ROLE_CONG2) {
            // (tm > tn, full 128-row tile: diagonal and bottom-edge tiles left through the cell paths above)
            const int nfull = tn * 4;                       // stage PAIRS before the B operand's diagonal block, per product
            // (stage s issues the loads of stage s + 2: the first pair's last k block is requested in the sixth stage of its
            // diagonal block, and the stagers move on to the second pair right there)
            for (int t = 0; t < nfull; ++t) HDM_RUN2(0, 3, 0)
            HDM_RUN2(0, 3, 0) HDM_RUN2(1, 3, 0) HDM_RUN2SW(2, 3, 0) HDM_RUN2(3, 3, 0)
            for (int t = 0; t < nfull; ++t) HDM_RUN2(0, 3, 0)
            HDM_RUN2(0, 3, 0) HDM_RUN2(1, 3, 0) HDM_RUN2(2, 3, 0) HDM_END2(3, 3, 0)
        } else if constexpr (ROLE == HDM_ROLE_CONG2D) {
            // one product; in the last K block both operands are on their diagonal: stage s keeps row and column
            // sub-tiles >= s, i.e. the wave's own [s/2..3] x [s/2..3]: 60 of the block's 128 sub-tile products
            const int nfull = tn * 4;
            for (int t = 0; t < nfull; ++t) HDM_RUN2(0, 3, 0)
            HDM_RUN2(0, 3, 0) HDM_RUN2(1, 3, 1) HDM_RUN2(2, 3, 2) HDM_END2(3, 3, 3)
        } else if constexpr (ROLE == HDM_ROLE_CONG1) {
            // first K block: A_L on the B side, live column sub-tiles <= s; last K block: Linv on the A side, rows >= s
            HDM_RUN2(0, 0, 0) HDM_RUN2(0, 1, 0) HDM_RUN2(0, 2, 0) HDM_RUN2(0, 3, 0)
            const int nmid = (tm - tn - 1) * 4;
            for (int t = 0; t < nmid; ++t) HDM_RUN2(0, 3, 0)
            HDM_RUN2(0, 3, 0) HDM_RUN2(0, 3, 1) HDM_RUN2(0, 3, 2) HDM_END2(0, 3, 3)
        } else {
            // any stage count (the Gram role's K splits): an odd count ends with a stage of zeros (SStager::load_nomask)
            if (nst > 0) {
                const int npairs = (nst + 1) >> 1;
                for (int t = 0; t + 1 < npairs; ++t) HDM_RUN2(0, 3, 0)
                HDM_END2(0, 3, 0)
            }
        }
#undef HDM_RUN2
#undef HDM_RUN2SW
#undef HDM_END2
#undef HDM_LAST
#undef HDM_STAGE
#undef HDM_LDF
#undef HDM_MMA
    }
    } else {
    // ======================================================================== generic launches: masked staging, plain loop
    Stager<AKM, true> stA;
    Stager<BKM, true> stB;
    stA.init(A, a.lda, a.a_kblk, a.seg_rows, a.seg_extra, a.M, m0, kt0, tid);
    stB.init(B, a.ldb, a.b_kblk, a.seg_rows, a.seg_extra, a.N, n0, kt0, tid);
    const int npass = a.A2 ? 2 : 1;
    if (a.A2) {
        stA.chain(a.A2 + (long) z * a.strideA2, a.lda2, m0, kt0, tid, kt1 - kt0);
        stB.chain(a.B2 + (long) z * a.strideB2, a.ldb2, n0, kt0, tid, kt1 - kt0);
    }
    auto compute = [&](const double *cA, const double *cB) {
#pragma unroll
        for (int kk = 0; kk < HDM_BK; kk += 4) {
            double fb[4], fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i] = frag<AKM>(cA, (2 * i + wm) * 16 + l15, kk + lq);
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[j] = frag<BKM>(cB, (2 * j + wn) * 16 + l15, kk + lq);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[j], fb[i], acc[j][i], 0, 0, 0);
        }
    };
    {
        double2 ra[4], rb[4];
        const int kt1x = kt0 + (kt1 - kt0) * npass;
        if (kt0 < kt1) {
            stA.load(ra);
            stB.load(rb);
            r2s<AKM>(sA, tid, ra);
            r2s<BKM>(sB, tid, rb);
        }
        __syncthreads();
        if ((VAR & 32) && p.dbg) t_pro = __builtin_amdgcn_s_memtime();
        int cur = 0;
        for (int kt = kt0; kt < kt1x; ++kt) {
            const bool more = (kt + 1 < kt1x);
            if (more) {
                stA.load(ra);
                stB.load(rb);
            }
            compute(sA + cur * STAGE_DOUBLES, sB + cur * STAGE_DOUBLES);
            if (more) {
                r2s<AKM>(sA + (cur ^ 1) * STAGE_DOUBLES, tid, ra);
                r2s<BKM>(sB + (cur ^ 1) * STAGE_DOUBLES, tid, rb);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    }

    if ((VAR & 32) && p.dbg) t_loop = __builtin_amdgcn_s_memtime();
    auto stamp_end = [&]() {
        if ((VAR & 32) && p.dbg && tid == 0) {
            unsigned long long *d = p.dbg + (size_t) sidx * 8;
            d[0] = t_start; d[1] = t_pro; d[2] = t_loop; d[3] = __builtin_amdgcn_s_memtime();
            d[4] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID[15:0]: wave, simd, pipe, cu, sh, se
            d[5] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
            d[6] = (unsigned long long) (kt1 - kt0);
            d[7] = __builtin_amdgcn_s_memrealtime();
        }
    };
    // ---------------------------------------------------------------- epilogue
    // Sub-tile ownership is INTERLEAVED: wave (wm, wn) owns row sub-tiles 2i + wm and column sub-tiles 2j + wn (i, j = 0..3),
    // so that when a triangular operand kills the first or last few sub-tiles of a stage the live ones are still spread
    // evenly over the waves.  lane l, reg r of acc[j][i] holds C[m0 + (2i+wm)*16 + l15][n0 + (2j+wn)*16 + lq + 4r]
    if constexpr (symdiag) {
        // acc holds P = U_t W_t^T (all 64 sub-tiles); the tile wanted is P + P^T, lower sub-tiles only.  Every wave parks
        // its sub-tiles on or above the diagonal TRANSPOSED in LDS (36 x 2 KiB: exactly the four stage buffers), region
        // bj (bj + 1) / 2 + bi holding X[p][q ^ p] = P[bi, bj](q, p); the owner of the mirrored sub-tile reads its own
        // element positions back.  The XOR keeps both sides at four lanes per 8-byte bank group -- what a 64-lane b64
        // access costs anyway -- in unpadded 16 x 16 regions.
        __syncthreads();                      // the last stage's fragment reads are done
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int si = 2 * i + wm, sj = 2 * j + wn;
                if (si > sj) continue;        // wave-uniform
                double *x = smem + (sj * (sj + 1) / 2 + si) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) x[(lq + 4 * r) * 16 + (l15 ^ (lq + 4 * r))] = acc[j][i][r];
            }
        }
        __syncthreads();
        const long rs16 = a.blk_row_stride * 16;
        double *lane_base = a.C + (a.blk_row0 + z) * 16 + l15 + (long) lq * rs16;
        const double rt2 = 1.4142135623730951;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int si = 2 * i + wm, sj = 2 * j + wn;
                const int bi = (m0 >> 4) + si, bj = (n0 >> 4) + sj;
                if (si < sj || bi >= a.nblk) continue;   // wave-uniform
                const double *x = smem + (si * (si + 1) / 2 + sj) * 256;
                const double sc = (bi == bj) ? 1.0 : rt2;
                const long sub = (long) bj * a.nblk - (long) bj * (bj - 1) / 2 + (bi - bj);
                double *q = lane_base + sub * 16 * rs16;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    q[(long) (4 * r) * rs16] = sc * (acc[j][i][r] + x[l15 * 16 + ((lq + 4 * r) ^ l15)]);
            }
        }
        stamp_end();
        return;
    }
    if (a.epilogue == HDM_EPI_BLOCKED) {
        // one base pointer per lane, everything else is wave-uniform 64-bit strides
        const long rs16 = a.blk_row_stride * 16;                       // elements between consecutive p-blocks
        double *lane_base = a.C + (a.blk_row0 + z) * 16 + l15 + (long) lq * rs16;
        const double rt2 = 1.4142135623730951;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bi = (m0 >> 4) + 2 * i + wm;
                const int bj = (n0 >> 4) + 2 * j + wn;
                if (bi < bj || bi >= a.nblk) continue;  // (bj <= bi < nblk), wave-uniform
                const double sc = (bi == bj) ? 1.0 : rt2;
                const long sub = (long) bj * a.nblk - (long) bj * (bj - 1) / 2 + (bi - bj);
                double *q = lane_base + sub * 16 * rs16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if ((VAR & 512) && (acc[j][i][r] != 12345.678)) continue;   // timing-only ablation (wrong results): no stores
                    q[(long) (4 * r) * rs16] = sc * acc[j][i][r];
                }
            }
        }
        stamp_end();
        return;
    }
    double *C = a.C;
    if (a.epilogue == HDM_EPI_SLAB) C += (long) z * a.slab_stride;
    else C += (long) z * a.strideC;
    const bool diag_tile = a.lower_only && (tm == tn);
    const bool inside = (m0 + HDM_TILE <= a.M) && (n0 + HDM_TILE <= a.N) && !diag_tile;   // wave-uniform
    if (inside) {
        // interior tile: no per-element predicates, one lane pointer, uniform column strides
        double *lane_c = C + (m0 + wm * 16 + l15) + (long) (n0 + wn * 16 + lq) * a.ldc;
        const long c4 = 4 * a.ldc;
        if (a.beta == 0.0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double *q = lane_c + (long) (j * 8 + r) * c4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) q[i * 32] = a.alpha * acc[j][i][r];
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double *q = lane_c + (long) (j * 8 + r) * c4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) q[i * 32] = a.alpha * acc[j][i][r] + a.beta * q[i * 32];
                }
        }
        stamp_end();
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gi = m0 + (2 * i + wm) * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gj = n0 + (2 * j + wn) * 16 + lq + 4 * r;
                if (gi < a.M && gj < a.N && !(diag_tile && gi < gj)) {
                    double *c = C + gi + (long) gj * a.ldc;
                    double v = a.alpha * acc[j][i][r];
                    if (a.beta != 0.0) v += a.beta * (*c);
                    *c = v;
                }
            }
        }
    }
    stamp_end();
}


// One workgroup per tile.  XCD-aware decode: consecutive workgroup ids are dealt round-robin over the 8 XCDs, so the batch
// index (one constraint matrix / one K split) is the fast index: XCD x walks batch entries x, x+8, ... one after the other
// and, inside an entry, the tile list in order (heaviest first, host-sorted) -- the ~64 workgroups an XCD runs at any time
// work on ONE operand set and on neighbouring tiles, and share their row/column panels in that XCD's L2.  The grid is
// sized for the batch rounded up to a multiple of 8; the surplus workgroups leave at once.
template <bool AKM, bool BKM, int ROLE, int VAR>
__global__ __launch_bounds__(256, 2) void hdm_gemm_kernel(HdmGemmDev p) {
    const int nb = p.a.batch;
    const int wg = blockIdx.x;
    int z, t;
    if (nb >= 8) {
        const int idx = wg >> 3;
        z = (wg & 7) + 8 * (idx / p.ntiles);
        t = idx % p.ntiles;
    } else {
        z = wg % nb;
        t = wg / nb;
    }
    if (t >= p.ntiles || z >= nb) return;
    hdm_gemm_tile<AKM, BKM, ROLE, VAR>(p, z, t, (long) blockIdx.x);
}

// PERSISTENT workgroups (roles 1-3, batch >= 8): the grid is exactly what the chip holds (2 workgroups per CU) and every
// workgroup pulls tiles until none are left.
//
// Why.  Per-workgroup stamps of the one-tile-per-workgroup launch (tools/wg_timeline.py, profiles/r02_e_wg_residency.txt)
// show that a CU spends 16 % of congruence step 2's time with ONE workgroup on it and 3.4 % with none (step 1: 15 % / 1.5 %,
// Gram: 8.5 % / 0): after a workgroup ends, the next one starts on that CU 24 us later on average (a tile lasts 250 us), on
// every XCD alike.  Our reading: workgroups are dealt to the XCDs round-robin and IN ORDER, so a free slot on one XCD is not
// refilled until every XCD before it in the deal has taken its workgroup, and each XCD waits for the slowest of eight at
// every hand-over.  That gap -- not barriers, LDS or the loop -- is most of the matrix-pipe idle time the counters showed
// (9-12 %).
//
// Queues.  One atomic counter per XCD (cnt[8], zeroed by the host before the launch): XCD x's workgroups (x read from
// HW_REG_XCC_ID, nothing assumed about the dispatcher) draw idx = cnt[x]++, which decodes exactly like the workgroup id
// above -- batch entries x, x+8, ..., the tile list inside an entry in order -- so the L2 sharing is the same.  When its own
// queue is dry an XCD's workgroups go on to the next XCD's queue (stealing at tile granularity: the XCDs' clocks differ by
// 2 %, and a partition that shows fewer XCDs still processes every queue).  No workgroup ever waits for another one.
template <bool AKM, bool BKM, int ROLE, int VAR>
__global__ __launch_bounds__(256, 2) void hdm_gemm_persist_kernel(HdmGemmDev p, int *__restrict__ cnt) {
    __shared__ int s_idx;
    const int nb = p.a.batch, ntiles = p.ntiles;
    const int x = (int) __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;   // HW_REG_XCC_ID[3:0]
    // queue_global: one queue (cnt[0]) for all XCDs, entry by entry in order -- every workgroup of the chip is then on the entry
    // at the queue's front or the one after it, so the operand bytes in use at any time are those of ~one entry (a K split of the
    // Gram product), which is what the memory-side cache can hold (engine_cone.h: cone_alloc_gemm_work)
    // (compile-time off for the congruence roles: only the Gram product asks for it -- measured without gain there, profiles/
    // r05_d_* -- and a run-time flag kept live across the tile loop cost the step-1 kernel two SGPR spills inside its stages)
    const bool one_queue = (ROLE == HDM_ROLE_GRAM) && p.a.queue_global != 0;
    for (int dx = 0; dx < (one_queue ? 1 : 8); ++dx) {
        const int xq = one_queue ? 0 : ((x + dx) & 7);
        while (true) {
            __syncthreads();                 // the previous tile's LDS images and s_idx are done with
            if (threadIdx.x == 0) s_idx = atomicAdd(&cnt[xq], 1);
            __syncthreads();
            const int idx = __builtin_amdgcn_readfirstlane(s_idx);
            const int z = one_queue ? idx / ntiles : xq + 8 * (idx / ntiles);
            if (z >= nb) break;
            hdm_gemm_tile<AKM, BKM, ROLE, VAR>(p, z, idx % ntiles, (long) z * ntiles + idx % ntiles);
        }
    }
}


