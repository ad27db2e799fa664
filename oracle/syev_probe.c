/* oracle/syev_probe.c -- TEST INFRASTRUCTURE (stand-alone; `make -C oracle syev_probe`): how much of W does this box's
 * dsyevr write when ONE eigenvalue is requested (range 'I', il = iu = n, jobz 'V': the call of the reference's final
 * solution check, linalg/dense_opts.c:56-69 from interface/hdsdp.c:858 with a two-element W at :811)?  LAPACK documents W
 * as an array of dimension N.  The probe runs the call on families of symmetric matrices of the block sizes of the
 * `blocks` instance (9, 21, 34) and a few more, with W = N sentinels, and reports the largest index written per family.
 * Any index >= 2 is a write past the reference's buffer, i.e. into HDSDPCheckSolution's stack frame. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern void dsyevr(const char *jobz, const char *range, const char *uplo, const int *n, double *a, const int *lda,
                   const double *vl, const double *vu, const int *il, const int *iu, const double *abstol, int *m,
                   double *w, double *z, const int *ldz, int *isuppz, double *work, const int *lwork, int *iwork,
                   const int *liwork, int *info);

static uint64_t rs = 0x9E3779B97F4A7C15ULL;
static double urand(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double) (rs >> 11) / 9007199254740992.0 * 2.0 - 1.0; }

static int last_written(int n, double *A) {
    const char jobz = 'V', range = 'I', uplo = 'U';
    const double zero = 0.0;
    int isuppz[4] = {0, 0, 0, 0}, il = n, iu = n, info = 0, mm = 1, lwork = 30 * n, liwork = 12 * n;
    const uint64_t sentinel = 0x7ff8dead0000beefULL;
    double *w = malloc(sizeof(double) * n), *z = malloc(sizeof(double) * n * n), *work = malloc(sizeof(double) * lwork);
    int *iwork = malloc(sizeof(int) * liwork);
    for (int i = 0; i < n; ++i) memcpy(&w[i], &sentinel, 8);
    dsyevr(&jobz, &range, &uplo, &n, A, &n, &zero, &zero, &il, &iu, &zero, &mm, w, z, &n, isuppz, work, &lwork, iwork, &liwork, &info);
    int last = -1;
    for (int i = 0; i < n; ++i) { uint64_t b; memcpy(&b, &w[i], 8); if (b != sentinel) last = i; }
    free(w); free(z); free(work); free(iwork);
    return last;
}

int main(void) {
    const int sizes[] = {2, 3, 9, 21, 34, 100};
    const char *fam[] = {"random dense", "diagonal", "block diagonal (3 blocks)", "scaled identity", "rank one", "with a NaN",
                         "with an Inf", "all zero", "tiny off-diagonals (1e-300)", "graded (1e-12 .. 1e12)"};
    int worst = -1;
    for (int f = 0; f < 10; ++f) {
        printf("%-32s", fam[f]);
        for (int s = 0; s < 6; ++s) {
            const int n = sizes[s];
            int mx = -1;
            for (int rep = 0; rep < 20; ++rep) {
                double *A = calloc((size_t) n * n, sizeof(double));
                for (int j = 0; j < n; ++j)
                    for (int i = 0; i <= j; ++i) {
                        double v = urand();
                        if (f == 1 && i != j) v = 0.0;
                        if (f == 2 && (i * 3 / n) != (j * 3 / n)) v = 0.0;
                        if (f == 3) v = (i == j) ? 2.5 : 0.0;
                        if (f == 4) v = (1.0 + 0.1 * i) * (1.0 + 0.1 * j);
                        if (f == 7) v = 0.0;
                        if (f == 8 && i != j) v *= 1e-300;
                        if (f == 9) v *= pow(10.0, -12.0 + 24.0 * (double) (i + j) / (2.0 * n));
                        A[i + (size_t) j * n] = v; A[j + (size_t) i * n] = v;
                    }
                if (f == 5) A[(n / 2) * (n + 1)] = NAN;
                if (f == 6) A[(n / 2) * (n + 1)] = INFINITY;
                const int l = last_written(n, A);
                if (l > mx) mx = l;
                free(A);
            }
            printf("  n=%-3d last=%-3d", n, mx);
            if (mx > worst) worst = mx;
        }
        printf("\n");
    }
    printf("largest index of W written with ONE eigenvalue requested: %d  (the reference's caller provides indices 0..1)%s\n", worst,
           worst >= 2 ? "  -> OVERRUN possible on this box's LAPACK" : "");
    return 0;
}
