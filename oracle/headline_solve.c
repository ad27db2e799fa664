/* oracle/headline_solve.c -- TEST / MEASUREMENT INFRASTRUCTURE (built by `make -C oracle drop` into
 * oracle/_ref/headline_solve_mi355x; needs /root/reference at build time, like sdpasolve_mi355x).
 *
 * A complete solve of the SURVEY.md 8(d) synthetic dense SDP at sizes no SDPA text file can carry comfortably
 * (n = m = 2000: 1.7e9 non-zeros, about 20 GB as a CSC, several times that as text): the instance is generated in memory
 * in the reference's user_data layout (interface/def_hdsdp_user_data.h:16-32: one CSC of shape n(n+1)/2 x (m+1), column 0
 * the objective) and handed to the reference's UNCHANGED driver -- HDSDPCreate / HDSDPInit / HDSDPSetCone /
 * HDSDPSetDualObjective / HDSDPOptimize, exactly the calls of tests/test_file_io.c:203-245 -- linked against the product
 * library, with the engine's cones attached at presolve by drop_attach.c.  What comes out: iterations, the optimum, the
 * driver's own optimisation time (correctors, line searches, S assemblies and all), and, with
 * HDSDP_MI355X_CALL_STATS=1, how much of that time was spent below the C ABI and where.
 *
 *     headline_solve_mi355x n [m]        (environment: HDSDP_MI355X_CALL_STATS=1, HDSDP_MI355X_GPUS=..., HDSDP_DROP_ATTACH)
 *
 * No reference source is contained here; the headers are included from where they lie. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <signal.h>
#include <execinfo.h>
#include <unistd.h>

#include "interface/hdsdp.h"
#include "interface/hdsdp_utils.h"
#include "interface/hdsdp_user_data.h"

static double draw_at(uint64_t t) {              /* draw number t (0-based) of the one splitmix64 stream, SURVEY 8(d) */
    uint64_t z = 0x9E3779B97F4A7C15ULL * (t + 2);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return 2.0 * ((double) (z >> 11) / 9007199254740992.0) - 1.0;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* HEADLINE_SEGV_TRACE=1: on SIGSEGV / SIGBUS / SIGABRT print the faulting thread's frames (module + offset; symbols where
 * the module exports them) and the last lines of /proc/self/maps' module list on stderr, then die by the same signal.  Exists
 * because round 3 recorded a segfault of this program under rocprofv3 with graph replay on and could not name the frame. */
static void segv_trace(int sig) {
    void *fr[64];
    static const char msg[] = "\n[headline_solve] fatal signal; frames of the faulting thread:\n";
    (void) !write(2, msg, sizeof(msg) - 1);
    const int nfr = backtrace(fr, 64);
    backtrace_symbols_fd(fr, nfr, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char **argv) {
    if (getenv("HEADLINE_SEGV_TRACE")) {
        void *warm[4];
        (void) backtrace(warm, 4);                  /* (loads libgcc now, not inside the handler) */
        signal(SIGSEGV, segv_trace); signal(SIGBUS, segv_trace); signal(SIGABRT, segv_trace);
    }
    if (argc < 2) { fprintf(stderr, "usage: %s n [m]\n", argv[0]); return 2; }
    const int n = atoi(argv[1]), m = argc > 2 ? atoi(argv[2]) : n;
    const uint64_t P = (uint64_t) n * (n + 1) / 2;
    double t0 = now_s();
    /* pass 1: entries per column (column c + 1 = constraint c; kept iff diagonal or second draw >= 0.2) */
    long *cnt = calloc((size_t) m + 2, sizeof(long));
    #pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < m; ++c) {
        long k = 0, q = 0;
        for (int j = 0; j < n; ++j)
            for (int i = j; i < n; ++i, ++k) {
                const uint64_t t = 2 * ((uint64_t) c * P + (uint64_t) k);
                if (i == j || draw_at(t + 1) >= 0.2) ++q;
            }
        cnt[c + 1] = q;
    }
    cnt[0] = (long) P;                           /* the objective is stored dense */
    long tot = 0;
    for (int c = 0; c <= m; ++c) tot += cnt[c];
    if (tot > 2147483647L) { fprintf(stderr, "instance does not fit the reference's int32 CSC (%ld non-zeros)\n", tot); return 2; }
    int *beg = malloc(sizeof(int) * ((size_t) m + 2));
    int *idx = malloc(sizeof(int) * (size_t) tot);
    double *val = malloc(sizeof(double) * (size_t) tot);
    double *b = calloc((size_t) m, sizeof(double)), *y0 = malloc(sizeof(double) * (size_t) m);
    if (!beg || !idx || !val || !b || !y0) { fprintf(stderr, "out of host memory\n"); return 2; }
    beg[0] = 0;
    for (int c = 0; c <= m; ++c) beg[c + 1] = beg[c] + (int) cnt[c];
    for (int c = 0; c < m; ++c) y0[c] = draw_at(2 * (uint64_t) m * P + (uint64_t) c);
    #pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < m; ++c) {
        long k = 0, q = beg[c + 1];
        double tr = 0.0;
        for (int j = 0; j < n; ++j)
            for (int i = j; i < n; ++i, ++k) {
                const uint64_t t = 2 * ((uint64_t) c * P + (uint64_t) k);
                const double v = draw_at(t);
                if (i == j || draw_at(t + 1) >= 0.2) { idx[q] = (int) k; val[q] = v; ++q; if (i == j) tr += v; }
            }
        b[c] = tr;                               /* b = A(I): X = I is strictly primal feasible */
    }
    /* C = I + sum_c y0_c A_c (y0 strictly dual feasible), every entry summed in constraint order */
    #pragma omp parallel for schedule(dynamic, 4)
    for (int j = 0; j < n; ++j) {
        long k = (long) (2 * (long) n - j - 1) * j / 2 + j;
        for (int i = j; i < n; ++i, ++k) {
            double acc = (i == j) ? 1.0 : 0.0;
            for (int c = 0; c < m; ++c) {
                const uint64_t t = 2 * ((uint64_t) c * P + (uint64_t) k);
                const double v = draw_at(t);
                if (i == j || draw_at(t + 1) >= 0.2) acc += y0[c] * v;
            }
            idx[k] = (int) k; val[k] = acc;
        }
    }
    printf("headline_solve: n = %d, m = %d, %ld non-zeros (%.1f GB as a CSC) generated in %.1f s\n", n, m, tot,
           (double) tot * 12.0 / 1e9, now_s() - t0);
    fflush(stdout);

    hdsdp_retcode retcode = HDSDP_RETCODE_OK;
    hdsdp *hsolve = NULL;
    user_data *SDPData = NULL;
    t0 = now_s();
    HDSDP_CALL(HDSDPCreate(&hsolve));
    HDSDP_CALL(HDSDPInit(hsolve, m, 1));
    HDSDP_CALL(HUserDataCreate(&SDPData));
    HUserDataSetConeData(SDPData, HDSDP_CONETYPE_DENSE_SDP, m, n, beg, idx, val);
    HDSDP_CALL(HDSDPSetCone(hsolve, 0, SDPData));
    HDSDPSetDualObjective(hsolve, b);
    HDSDP_CALL(HDSDPOptimize(hsolve, 1));
    printf("headline_solve: HDSDPCreate .. HDSDPOptimize took %.1f s of wall time (presolve and device upload included)\n",
           now_s() - t0);
exit_cleanup:
    HUserDataDestroy(&SDPData);
    HDSDPDestroy(&hsolve);
    free(cnt); free(beg); free(idx); free(val); free(b); free(y0);
    return (int) retcode;
}
