// lu.h -- device-resident pivoted factorisation for Schur matrices that are not positive definite (see lu.hip)
#pragma once
#include "hdm_common.h"

struct HdmLu {
    int n = 0, npad = 0;
    double *A = nullptr;     // npad x npad, column-major, full; P*A = L*U in place after factor()
    int *piv = nullptr;      // device, npad: row exchanged with row j at elimination step j
    int *perm = nullptr;     // device, npad: the same exchanges as one gather (b_permuted[i] = b[perm[i]])
    double *vec = nullptr;   // 4 * npad scratch vectors (host-side solves)
    int *info_dev = nullptr;
    bool factored = false;

    int init(int n);
    void destroy();
    int load_host_lower(const double *M, long ldm, hipStream_t s);     // lower triangle valid (the reference's M)
    int load_device_lower(const double *M, long ldm, hipStream_t s);
    int factor(hipStream_t s, int *info_host);                         // info = 0 ok, j+1 = exactly singular at step j
    int solve_device(const double *b_dev, double *x_dev, int nrhs, long ldv, hipStream_t s);
    int solve_host(const double *rhs, double *sol, int nrhs, hipStream_t s);
};
