/* oracle/syev_guard.c -- TEST INFRASTRUCTURE (linked into oracle/_ref/sdpasolve_mi355x and oracle/_ref/sdpasolve_ref_guard).
 *
 * The reference's final solution check (HDSDPCheckSolution, interface/hdsdp.c:811,858) hands fds_syev
 * (linalg/dense_opts.c:56-69) a TWO-element array for the eigenvalues, and fds_syev passes it to LAPACK's dsyevr as W.
 * dsyevr documents W as an array of dimension N ("the first M elements contain the selected eigenvalues"): an
 * implementation is free to use all of it as work space, and MKL does whenever it takes its dsterf branch -- the overrun
 * that ended one run of round 1 in "*** stack smashing detected ***" inside HDSDPCheckSolution.
 *
 * The executable defines fds_syev itself (calls from inside the reference's shared library go through the PLT, so they
 * land here, like HConePresolveData in drop_attach.c): the same dsyevr call, with an eigenvalue array of the documented
 * size N, of which the first M entries are handed back.  It also records what the call did to that array: entries still
 * holding the sentinel afterwards were not written.  The first call and every call that wrote more than the caller's
 * two entries are reported on stderr, so a run shows whether the reference's own buffer would have been overrun.
 * No reference source is contained here; the prototype is the one of linalg/dense_opts.h. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "interface/hdsdp.h"

extern void dsyevr(const char *jobz, const char *range, const char *uplo, const int *n, double *a, const int *lda,
                   const double *vl, const double *vu, const int *il, const int *iu, const double *abstol, int *m,
                   double *w, double *z, const int *ldz, int *isuppz, double *work, const int *lwork, int *iwork,
                   const int *liwork, int *info);

hdsdp_retcode fds_syev(int n, double *U, double *d, double *Y, int m, double *work, int *iwork, int lwork, int liwork) {
    static int calls = 0, overruns = 0;
    const char jobz = 'V', range = 'I', uplo = 'U';
    const double zero = 0.0;
    int isuppz[4] = {0, 0, 0, 0};
    int il = n - m + 1, iu = n, info = 0, mm = m;
    const uint64_t sentinel = 0x7ff8dead0000beefULL;          /* a NaN payload no eigenvalue can equal */
    double *w = (double *) malloc(sizeof(double) * (size_t) (n > 2 ? n : 2));
    if (!w) return HDSDP_RETCODE_MEMORY;
    for (int i = 0; i < n; ++i) memcpy(&w[i], &sentinel, sizeof(double));
    dsyevr(&jobz, &range, &uplo, &n, U, &n, &zero, &zero, &il, &iu, &zero, &mm, w, Y, &n, isuppz, work, &lwork, iwork,
           &liwork, &info);
    int written = 0, last = -1;
    for (int i = 0; i < n; ++i) {
        uint64_t bits;
        memcpy(&bits, &w[i], sizeof(bits));
        if (bits != sentinel) { ++written; last = i; }
    }
    ++calls;
    if (last >= 2) ++overruns;
    if (calls == 1 || last >= 2 || getenv("SYEV_GUARD_VERBOSE"))
        fprintf(stderr, "syev_guard: call %d  dsyevr(n = %d, range 'I', %d eigenvalue(s)) wrote %d entries of W, the last at "
                        "index %d; the reference's caller provides 2 (interface/hdsdp.c:811)%s  [%d overrun(s) so far]\n",
                calls, n, m, written, last, last >= 2 ? "  -> OVERRUN of the reference's buffer" : "", overruns);
    for (int i = 0; i < m; ++i) d[i] = w[i];
    free(w);
    return info == 0 ? HDSDP_RETCODE_OK : HDSDP_RETCODE_FAILED;
}
