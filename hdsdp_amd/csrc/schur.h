// schur.h -- launch helpers of schur.hip
#pragma once
#include "hdm_common.h"
#include "bsparse.h"
#include <functional>

int hdm_unpack_sym(const double *packed, long pstride, double *full, long fstride, int n, int ld, int batch, hipStream_t s);
int hdm_synth_fill(double *full, long fstride, int n, int ld, int c0, int batch, hipStream_t s);
int hdm_synth_obj(double *C, int n, int ld, int m, hipStream_t s);
int hdm_blocked_eye(double *dst, long row_stride, long row, int nblk, int n, hipStream_t s);
int hdm_slab_reduce(const double *slabs, long slab_stride, int nsplit, double *out, long total, long R, hipStream_t s);
int hdm_extract(const double *G, long ldg, long R, long pI, const int *rows_seg, const HdmMatView &Mv, double *asinv,
                double *asinvrd, double *asinvc, double *scal, double Rd, int hsd, hipStream_t s);
int hdm_sym_combine(const double *A, long astride, int m, const double *y, const double *C, double tau, double eye,
                    double *S, int n, long lda, long lds_, hipStream_t s);
int hdm_r1_hadamard(const double *Gm, long ldg, const double *sgn, const int *rows_global, int mloc, const HdmMatView &Mv,
                    double *asinv, hipStream_t s);
int hdm_r1_colnorm(const double *V, long ldv, int n, const double *sgn, const int *rows_global, int mloc, double Rd,
                   double *asinvrd, hipStream_t s);
int hdm_sym_dot2(const double *A, long astride, int n, long lda, int count, const double *X, const double *Y, long ldx,
                 double *outx, double *outy, const int *rows_global, double sx, double sy, hipStream_t s);
int hdm_sparse_pairs(const int *rp, const int *ti, const int *tj, const double *tv, const double *X, long ldx, int mloc,
                     const int *rows_global, const HdmMatView &Mv, hipStream_t s);
int hdm_sparse_dot(const int *rp, const int *ti, const int *tj, const double *tv, const double *Y, long ldy, int mloc,
                   const int *rows_global, double scale, double *out, hipStream_t s);
// entries (packed lower index, value) of `batch` matrices, row q's at [beg[q], beg[q + 1]), scattered into zeroed A_L-form skyline storage
int hdm_scatter_low(const int *idx, const double *val, const long *beg, long max_per_row, double *full, long fstride, int n, int ld,
                    int batch, hipStream_t s);
int hdm_synth_fill_low(double *full, long fstride, int n, int ld, int c0, int batch, hipStream_t s);
int hdm_lower_half(const double *full, double *low, int n, long ld, hipStream_t s);
// strict upper triangle of the 128 x 128 diagonal tiles of `batch` column-major n x n matrices (ld = n) <- 0
int hdm_zero_diag_upper(double *T, long tstride, int n, int batch, hipStream_t s);
// one skyline-stored A_L matrix (hdm_common.h) -> square column-major n x n
int hdm_sky_to_square(const double *sky, double *sq, int n, hipStream_t s);

// ---- zero-suppressed copy of a batch of skyline-stored A_L matrices for the HBM-bound sweeps (schur.hip) ----
// The sweeps that assemble S and dS read every stored element of every constraint matrix (32 GB at n = m = 2000) to add
// y_i * a to an accumulator -- also where a = 0, which is most of the storage for anything but a fully dense family (the
// reference calls a matrix DENSE from 30 % of its entries on, hdsdp_sdpdata.c:2321-2345; the synthetic family keeps 40 %).
// The copy stores, per chunk of 1024 consecutive skyline positions and per matrix, a 1024-bit occupancy mask with the
// offset of each 64-bit word's first value (192 bytes) and the non-zero values only; chunk-major, so that the workgroup
// that owns a chunk streams ONE contiguous range of memory for all matrices.  Same sums in the same order as the dense
// sweep (the skipped terms are exact zeros): bit-identical results.
struct HdmZs {
    unsigned long long *meta = nullptr;   // [chunk][matrix][24]: 16 mask words, then 16 uint32 offsets relative to base[chunk]
    double *val = nullptr;                // the non-zero values, chunk by chunk, matrix by matrix, in position order
    unsigned long long *base = nullptr;   // [chunk]: first value of the chunk's range in val
    long nchunk = 0, sky = 0, nnz = 0;
    int m = 0;
};
// builds the copy unless more than max_fill of the positions are non-zero (then out->val stays null: not worth the memory)
int hdm_zs_build(const double *A, long astride, int m, long sky, double max_fill, HdmZs *out, hipStream_t s);
// the same from data that is NOT resident: source(c0, count) returns the device pointer of matrices c0 .. c0 + count - 1 (stride
// astride; count <= batch), valid until its next call -- a streamed cone regenerates them into its batch buffer.  Two passes
// (counts, then values), both in matrix order, so the copy is the one hdm_zs_build makes of the resident data, bit for bit.
int hdm_zs_build_from(const std::function<const double *(int, int)> &source, int batch, long astride, int m, long sky,
                      double max_fill, HdmZs *out, hipStream_t s);
void hdm_zs_free(HdmZs *z);
// matrices c0 .. c0 + count - 1 of the copy back into A_L-form skyline storage at A (stride astride): values, zeros elsewhere
int hdm_zs_expand(const HdmZs &z, int c0, int count, double *A, long astride, hipStream_t s);
// <A_c, X>, <A_c, Y> of all matrices from the copy (what hdm_sym_dot2 computes from the dense storage; other summation order)
int hdm_sym_dot2_zs(const HdmZs &z, int n, long lda, const double *X, const double *Y, long ldx, double *outx, double *outy,
                    const int *rows_global, double sx, double sy, hipStream_t s);
int hdm_sym_combine_zs(const HdmZs &z, const double *y, const double *C, double tau, double eye, double *S, int n, long lda,
                       long lds_, hipStream_t s);
