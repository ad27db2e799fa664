// lanczos.h -- device-resident ratio test (see lanczos.hip)
#pragma once
#include <vector>
#include "hdm_common.h"

struct HdmLanczos {
    int n = 0, n16 = 0;
    int maxdim = 30;          // Krylov dimension, hdsdp_conic_sdp.c:1393
    int nComputed = 0;        // calls so far: the second and later calls warm-start (hdsdp_lanczos.c:166-181)
    double *V = nullptr;      // n16 x (maxdim + 1) Lanczos basis
    double *bv = nullptr, *b1 = nullptr, *b2 = nullptr, *bw = nullptr, *bz = nullptr;   // n16 x 8 vector blocks (column 0 used)
    double *warm = nullptr, *tmp = nullptr, *scal = nullptr, *startd = nullptr;   // startd: device copy of `start`, zero padded
    double *part = nullptr;   // 32 x n16 partial sums of the plain matrix-vector product
    double *LT = nullptr;     // n16 x n16 transposed copy of Linv (large blocks, hdm_lanczos_group_kernel), made per test
    unsigned *gsync = nullptr; // its grid barrier: counter, give-up word
    bool big_ok = true; int big_wg = 0;
    unsigned sync_epoch = 0;   // barrier epochs handed out so far
    double *scal_h = nullptr;  // host side of `scal` (mapped pinned memory: `scal` is its device address)
    std::vector<double> start;   // the reference's pseudo-random start vector (host)

    int init(int n);
    void destroy();
    int apply(const double *Linv, long ldl, const double *dS, long ldd, const double *in, double *out, hipStream_t s);
    // max step of  S + alpha dS >= 0  given Linv (S = L L^T) and the full symmetric dS; INFINITY if unbounded
    int solve(const double *Linv, long ldl, const double *dS, long ldd, hipStream_t s, double *maxStep, int *steps);
};

// HLanczosIPrepare's vector (glibc srand/rand stream reproduced without touching libc state); host only
void hdm_lanczos_start_vector(int n, double *p);
int hdm_mirror_lower(double *A, long ld, int n, hipStream_t s);
int hdm_sym_scale(double *A, long ld, int n, double diag_add, double scale, hipStream_t s);   // A <- scale*(sym(A) + diag_add*I)
int hdm_axpy_mat(double *out, const double *S, const double *dS, double step, long count, hipStream_t s);   // out = S + step*dS
int hdm_axpy_mat_eye(double *out, const double *S, const double *dS, double step, double eye, long ld, int n, hipStream_t s);   // out = S + step*dS + eye*I
