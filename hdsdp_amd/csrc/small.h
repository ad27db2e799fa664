// small.h -- the fused single-launch Phase-A pass for small rank-one blocks (small.hip)
#pragma once
#include "hdm_common.h"

#define SMALL_P 128        // padded dimension: n, m <= 128
#define SMALL_SPMAX 8      // a rank-one factor with more entries than this is treated as dense
#define SMALL_NDENSE 4     // at most this many dense factors per block

struct HdmSmallArgs {
    int n, m;
    const double *C; long ldc;          // objective, full symmetric n x n (device)
    const int *fp, *fi;                 // rank-one factors as CSR over the m rows (a dense factor lists all n entries)
    const double *fv;
    const double *sgn;                  // m signs
    const int *dense_of;                // m: index into dense_rows, or -1
    int ndense;
    const int *dense_rows;              // ndense row indices (device)
    const double *y, *b;                // m multipliers, m right-hand side of the first solve (mapped host or device)
    double tau, eye, Rd;                // S = tau C - sum y_i A_i + eye I  (eye = -Rd + perturbation)
    double *Sout; long lds;             // the dual matrix, lower triangle (the cone's resident S)
    double *LS, *WS;                    // 128 x 128: Cholesky factor of S and its inverse (HdmChol::L, ::Dinv)
    double *M; long ldm;                // Schur matrix, lower triangle (device)
    double *LM, *WM;                    // 128 x 128: factor of M and its inverse
    double *out;                        // [0] info S, [1] info M, [2] logdet S, [3] tr S^-1, then ASinv, ASinvRdSinv, d1, d2, d3 (m each),
                                        // then 8 phase stamps (s_memrealtime, 100 MHz)
};

// One launch for an interior check of a small block (n <= 128, any device path): S = tau C - sum y_i A_i + eye I from the
// resident A_L-form constraint matrices (n16 x n16 each, n16 <= 128: skyline storage is a plain square there), the Cholesky
// factor with the triangular inverse, the pivot information and log det S.  What the call-by-call form does in seven
// operations and two synchronisations (pinned copy of y, assembly kernel, rectangle copy into the factor object, padding
// kernel, memset, sweep kernel, copy back): 88 us on a 100 x 100 block, 56 us on truss1's 21 x 21 ones, of which the
// reference's driver runs thousands.
struct HdmSmallCheckArgs {
    int n, n16, m;                      // block dimension, its leading dimension, constraint matrices resident
    const double *A; long astride;      // m matrices in A_L form (strict lower + half diagonal), column-major n16 x n16
    const double *C;                    // objective, full symmetric, ld n16
    const double *y;                    // m multipliers (mapped host or device)
    double tau, eye;
    double *Sout;                       // the dual matrix buffer (lower triangle written), ld n16
    double *L, *W;                      // 128 x 128: Cholesky factor and its inverse (HdmChol::L, ::Dinv)
    double *out;                        // [0] info (0, or first non-positive pivot + 1), [1] log det S
};
int hdm_small_check(const HdmSmallCheckArgs &args, hipStream_t s);

size_t hdm_small_lds_bytes();
int hdm_small_phase_a(const HdmSmallArgs &args, hipStream_t s);
