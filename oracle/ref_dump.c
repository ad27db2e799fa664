/* oracle/ref_dump.c -- TEST INFRASTRUCTURE (golden-vector generator).
 *
 * Drives the REAL reference (compiled from /root/reference into oracle/_ref/libhdsdp_ref.so by
 * oracle/Makefile) through its own public C API and dumps the inputs/outputs of the Schur hot path
 * as .npy files.  oracle/gen_golden.py packs them into tests/golden/*.npz.  Nothing here is shipped
 * or measured; this file only *calls* the reference (it includes the reference headers from where
 * they lie), it contains none of its source.
 *
 * Call sequence follows the reference's own (dead) unit test tests/test_file_io.c:356-467
 * (test_sdpa_io): HUserDataSetConeData -> HConeCreate/SetData/ProcData/PresolveData ->
 * HConeSetStart(Rd) -> HConeGetLogBarrier(tau, y) -> HKKTCreate/Init -> HKKTBuildUp(type) ->
 * HKKTExport -> HKKTFactorize -> HKKTSolve.
 *
 * usage: ref_dump <outdir> sdpa <file.dat-s> <Rd> <tau> <yscale>
 *        ref_dump <outdir> syn  <n> <m>      <Rd> <tau> <yscale>     (SURVEY.md 8(d) generator)
 *        ref_dump <outdir> mix  <n> <m>      <Rd> <tau> <yscale>     (all five coefficient types)
 *        ref_dump -        bench <n> <m>     <Rd> <tau> <yscale>     (CPU baseline: prints one JSON line
 *                                                                     with stage timings, dumps nothing)
 *        ref_dump <outdir> indef <m> <kind> <shift> 0 0              (Schur-system solver on a matrix that is not
 *                                                                     positive definite: the LDL^T fallback,
 *                                                                     linalg/hdsdp_linsolver.c:1827-1857, 2029-2110)
 */
#include "interface/hdsdp.h"
#include "interface/hdsdp_utils.h"
#include "interface/hdsdp_user_data.h"
#include "interface/hdsdp_file_io.h"
#include "interface/hdsdp_conic.h"
#include "interface/hdsdp_schur.h"
#include "linalg/hdsdp_sdpdata.h"
#include "linalg/hdsdp_linsolver.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>

static const char *g_out = NULL;

static void npy_write(const char *name, const char *descr, int ndim, const long *dims, const void *data, size_t elsz) {
    char path[1024], hdr[256], shape[128] = "";
    snprintf(path, sizeof(path), "%s/%s.npy", g_out, name);
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); exit(2); }
    size_t cnt = 1;
    for (int i = 0; i < ndim; ++i) {
        char t[32]; snprintf(t, sizeof(t), "%ld,", dims[i]); strcat(shape, t); cnt *= (size_t) dims[i];
    }
    int len = snprintf(hdr, sizeof(hdr), "{'descr': '%s', 'fortran_order': False, 'shape': (%s), }", descr, shape);
    int total = 10 + len + 1;
    int pad = (64 - total % 64) % 64;
    unsigned short hlen = (unsigned short) (len + pad + 1);
    fwrite("\x93NUMPY\x01\x00", 1, 8, f);
    fwrite(&hlen, 2, 1, f);
    fwrite(hdr, 1, len, f);
    for (int i = 0; i < pad; ++i) fputc(' ', f);
    fputc('\n', f);
    fwrite(data, elsz, cnt, f);
    fclose(f);
}
static void dump_d(const char *name, const double *x, long n) { npy_write(name, "<f8", 1, &n, x, 8); }
static void dump_i(const char *name, const int *x, long n) { npy_write(name, "<i4", 1, &n, x, 4); }
static void dump_d2(const char *name, const double *x, long r, long c) { long d[2] = {r, c}; npy_write(name, "<f8", 2, d, x, 8); }
static void dump_s(const char *name, double v) { dump_d(name, &v, 1); }

/* splitmix64 -> U(-1,1), SURVEY.md 8(d) */
static uint64_t g_s = 0x9E3779B97F4A7C15ULL;
static double urand(void) {
    uint64_t z = (g_s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return 2.0 * ((double) (z >> 11) / 9007199254740992.0) - 1.0;
}

/* CSC of shape P x (m+1): column 0 = C, column c = A_c, rows = packed-lower col-major index */
typedef struct { int n, m; int *beg, *idx; double *val; double *b; double *y0; } csc_prob;

static void gen_syn(csc_prob *p, int n, int m) {
    long P = (long) n * (n + 1) / 2;
    double *Apk = calloc((size_t) P * m, sizeof(double));
    char *keep = calloc((size_t) P * m, 1);
    for (int c = 0; c < m; ++c) {
        long k = 0;
        for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i, ++k) {
            double v = urand(); int kp = (urand() >= 0.2);
            if (i == j || kp) { Apk[(size_t) c * P + k] = v; keep[(size_t) c * P + k] = 1; }
        }
    }
    double *y0 = malloc(m * sizeof(double));
    for (int i = 0; i < m; ++i) y0[i] = urand();
    double *Cpk = calloc(P, sizeof(double));
    { long k = 0; for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i, ++k) Cpk[k] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < m; ++c) for (long k = 0; k < P; ++k) Cpk[k] += y0[c] * Apk[(size_t) c * P + k];
    long tot = P; for (size_t q = 0; q < (size_t) P * m; ++q) tot += keep[q];
    p->n = n; p->m = m;
    p->beg = calloc(m + 2, sizeof(int)); p->idx = malloc(tot * sizeof(int)); p->val = malloc(tot * sizeof(double));
    p->b = calloc(m, sizeof(double));
    long pos = 0;
    for (long k = 0; k < P; ++k) { p->idx[pos] = (int) k; p->val[pos++] = Cpk[k]; }
    p->beg[1] = (int) pos;
    for (int c = 0; c < m; ++c) {
        for (long k = 0; k < P; ++k) if (keep[(size_t) c * P + k]) { p->idx[pos] = (int) k; p->val[pos++] = Apk[(size_t) c * P + k]; }
        p->beg[c + 2] = (int) pos;
        long k = 0; double tr = 0.0;
        for (int j = 0; j < n; ++j) { tr += Apk[(size_t) c * P + k]; k += n - j; }
        p->b[c] = tr;
    }
    p->y0 = y0;   /* kept: S(y0) = I is the interior point the primal-recovery dump uses */
    free(Apk); free(keep); free(Cpk);
}

/* multi-type mix: row c%6 -> 0 dense, 1 sparse, 2 sparse rank-1, 3 dense rank-1, 4 zero, 5 sparse (diag heavy) */
static void gen_mix(csc_prob *p, int n, int m) {
    long P = (long) n * (n + 1) / 2;
    double *M = calloc((size_t) P * (m + 1), sizeof(double));
    #define PK(i, j) ((long) (2 * n - (j) - 1) * (j) / 2 + (i))
    double *a = malloc(n * sizeof(double));
    for (int c = 1; c <= m; ++c) {
        double *A = M + (size_t) c * P;
        int kind = (c - 1) % 6;
        if (kind == 0) {
            for (long k = 0; k < P; ++k) A[k] = urand();
        } else if (kind == 1) {
            int nz = 2 + (c % 5);
            for (int q = 0; q < nz; ++q) {
                int i = (int) ((urand() + 1.0) * 0.5 * n) % n, j = (int) ((urand() + 1.0) * 0.5 * n) % n;
                if (i < j) { int t = i; i = j; j = t; }
                A[PK(i, j)] += urand();
            }
            A[PK((c * 7) % n, (c * 7) % n)] += 1.0 + 0.5 * urand();
        } else if (kind == 2) {
            memset(a, 0, n * sizeof(double));
            int nz = 1 + (c % 4);
            for (int q = 0; q < nz; ++q) a[(c * 3 + q * 5) % n] = 0.5 + 0.5 * (urand() + 1.0);
            double sgn = (c % 4 == 2) ? -1.7 : 2.3;
            for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i) A[PK(i, j)] = sgn * a[i] * a[j];
        } else if (kind == 3) {
            for (int i = 0; i < n; ++i) a[i] = urand();
            double sgn = (c % 4 == 3) ? -0.9 : 1.1;
            for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i) A[PK(i, j)] = sgn * a[i] * a[j];
        } else if (kind == 4) {
            /* zero row */
        } else {
            for (int i = 0; i < n; i += 3) A[PK(i, i)] = 1.0 + 0.25 * urand();
            A[PK(n - 1, 0)] = 0.3;
        }
    }
    /* C = I + sum y0_c A_c  (strictly dual feasible at y0) */
    double *C = M;
    for (int j = 0; j < n; ++j) C[PK(j, j)] = 1.0;
    for (int c = 1; c <= m; ++c) { double y0 = 0.3 * urand(); for (long k = 0; k < P; ++k) C[k] += y0 * M[(size_t) c * P + k]; }
    long tot = 0; for (size_t q = 0; q < (size_t) P * (m + 1); ++q) tot += (M[q] != 0.0);
    p->n = n; p->m = m;
    p->beg = calloc(m + 2, sizeof(int)); p->idx = malloc((tot + 1) * sizeof(int)); p->val = malloc((tot + 1) * sizeof(double));
    p->b = calloc(m, sizeof(double));
    long pos = 0;
    for (int c = 0; c <= m; ++c) {
        for (long k = 0; k < P; ++k) if (M[(size_t) c * P + k] != 0.0) { p->idx[pos] = (int) k; p->val[pos++] = M[(size_t) c * P + k]; }
        p->beg[c + 1] = (int) pos;
        if (c > 0) { double tr = 0.0; for (int j = 0; j < n; ++j) tr += M[(size_t) c * P + PK(j, j)]; p->b[c - 1] = tr; }
    }
    free(M); free(a);
    #undef PK
}

int main(int argc, char **argv) {
    if (argc < 7) { fprintf(stderr, "usage: see header\n"); return 2; }
    g_out = argv[1];
    const char *mode = argv[2];
    hdsdp_retcode retcode = HDSDP_RETCODE_OK;
    csc_prob pb; memset(&pb, 0, sizeof(pb));
    double Rd, tau, yscale;

    if (!strcmp(mode, "indef")) {
        /* the Schur system object exactly as HKKTIAllocDenseKKT sets it up (hdsdp_schur.c:11-44), fed with a symmetric
           matrix that is not positive definite.  kind 0: uniform(-1,1) entries + shift on the diagonal (strongly
           indefinite for shift 0); kind 1: G G^T / m + shift I with shift < 0 pushing the small eigenvalues below zero
           ("almost indefinite").  Two factor+solve rounds: the second one shows that the object stays switched. */
        const int mm = atoi(argv[3]), kind = atoi(argv[4]);
        const double shift = atof(argv[5]);
        double *Mm = calloc((size_t) mm * mm, sizeof(double)), *bb = calloc(mm, sizeof(double));
        double *x1 = calloc(mm, sizeof(double)), *x2 = calloc(mm, sizeof(double));
        if (kind == 0) {
            for (int j = 0; j < mm; ++j)
                for (int i = j; i < mm; ++i) Mm[i + (size_t) j * mm] = urand();
        } else {
            double *G = calloc((size_t) mm * mm, sizeof(double));
            for (size_t e = 0; e < (size_t) mm * mm; ++e) G[e] = urand();
            for (int j = 0; j < mm; ++j)
                for (int i = j; i < mm; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < mm; ++k) acc += G[i + (size_t) k * mm] * G[j + (size_t) k * mm];
                    Mm[i + (size_t) j * mm] = acc / mm;
                }
            free(G);
        }
        for (int j = 0; j < mm; ++j) Mm[j + (size_t) j * mm] += shift;
        for (int i = 0; i < mm; ++i) bb[i] = urand();
        { int dm[2] = {mm, kind}; dump_i("indef_dims", dm, 2); }
        dump_d2("indef_M", Mm, mm, mm); dump_d("indef_b", bb, mm);
        hdsdp_linsys_fp *lin = NULL;
        HDSDP_CALL(HFpLinsysCreate(&lin, mm, HDSDP_LINSYS_DENSE_ITERATIVE));
        HFpLinsysSetParam(lin, 5.0 * KKT_ACCURACY, KKT_ACCURACY, -1, -1, -1);
        int codes[6] = {0, 0, 0, 0, 0, 0};
        codes[0] = (int) HFpLinsysNumeric(lin, NULL, NULL, Mm);
        codes[1] = (int) HFpLinsysSolve(lin, 1, bb, x1);
        codes[2] = (int) lin->LinType;
        dump_d("indef_x1", x1, mm);
        /* second round on a modified matrix (what the next IPM iteration does) */
        for (int j = 0; j < mm; ++j) Mm[j + (size_t) j * mm] += 0.125;
        codes[3] = (int) HFpLinsysNumeric(lin, NULL, NULL, Mm);
        codes[4] = (int) HFpLinsysSolve(lin, 1, bb, x2);
        codes[5] = (int) lin->LinType;
        dump_d("indef_x2", x2, mm);
        dump_i("indef_codes", codes, 6);
        printf("ref_dump ok: indef m=%d kind=%d codes=%d %d %d | %d %d %d\n", mm, kind, codes[0], codes[1], codes[2],
               codes[3], codes[4], codes[5]);
        return 0;
    }
    if (!strcmp(mode, "sdpam")) {
        /* multi-block SDPA instance (truss1: six 2x2 blocks and a 1x1): one dense-SDP cone per block in ONE Schur
           operator -- HKKTBuildUp loops over the cones and every cone adds its part (hdsdp_schur.c:256-268) */
        int nConstrs = 0, nBlks = 0, *BlkDims = NULL, nCols = 0, nLpCols = 0, nElem = 0;
        int **cBeg = NULL, **cIdx = NULL, *LpBeg = NULL, *LpIdx = NULL;
        double **cElem = NULL, *rowRHS = NULL, *LpElem = NULL;
        HDSDP_CALL(HReadSDPA(argv[3], &nConstrs, &nBlks, &BlkDims, &rowRHS, &cBeg, &cIdx, &cElem,
                             &nCols, &nLpCols, &LpBeg, &LpIdx, &LpElem, &nElem));
        if (nLpCols != 0) { fprintf(stderr, "harness handles SDP blocks only\n"); return 2; }
        const int mm = nConstrs;
        Rd = atof(argv[4]); tau = atof(argv[5]); yscale = atof(argv[6]);
        double *yy = calloc(mm, sizeof(double));
        for (int i = 0; i < mm; ++i) yy[i] = yscale * sin(1.7 * (i + 1));
        { int dm[2] = {nBlks, mm}; dump_i("mb_dims", dm, 2); dump_i("mb_blkdims", BlkDims, nBlks); }
        dump_d("b", rowRHS, mm); dump_d("y", yy, mm); dump_s("Rd", Rd); dump_s("tau", tau);
        hdsdp_cone **cones = calloc(nBlks, sizeof(hdsdp_cone *));
        char nm[64];
        double ldsum = 0.0;
        for (int k = 0; k < nBlks; ++k) {
            user_data *udk = NULL;
            long nz = cBeg[k][mm + 1];
            snprintf(nm, 64, "mb%d_beg", k); dump_i(nm, cBeg[k], mm + 2);
            snprintf(nm, 64, "mb%d_idx", k); dump_i(nm, cIdx[k], nz);
            snprintf(nm, 64, "mb%d_val", k); dump_d(nm, cElem[k], nz);
            HDSDP_CALL(HUserDataCreate(&udk));
            HUserDataSetConeData(udk, HDSDP_CONETYPE_DENSE_SDP, mm, BlkDims[k], cBeg[k], cIdx[k], cElem[k]);
            HDSDP_CALL(HConeCreate(&cones[k], k));
            HDSDP_CALL(HConeSetData(cones[k], udk));
            HDSDP_CALL(HConeProcData(cones[k]));
            HDSDP_CALL(HConePresolveData(cones[k]));
            HConeSetStart(cones[k], Rd);
            int ok = 0; double ld = 0.0;
            HDSDP_CALL(HConeCheckIsInterior(cones[k], tau, yy, &ok));
            if (!ok) { fprintf(stderr, "block %d not interior\n", k); return 3; }
            HDSDP_CALL(HConeGetLogBarrier(cones[k], tau, yy, BUFFER_DUALVAR, &ld));
            ldsum += ld;
        }
        dump_s("logdet", ldsum);
        hdsdp_kkt *kk = NULL;
        HDSDP_CALL(HKKTCreate(&kk));
        HDSDP_CALL(HKKTInit(kk, mm, nBlks, cones));
        /* a sparse Schur operator (isKKTSparse, hdsdp_schur.c:46-139: every block touches few constraints) is dumped as
           what it is: the aggregated lower-triangular CSC pattern and its nnz values */
        { int sp = kk->isKKTSparse; dump_i("kkt_sparse", &sp, 1); }
        const int knz = kk->isKKTSparse ? kk->kktMatBeg[mm] : 0;
        if (kk->isKKTSparse) { dump_i("kkt_beg", kk->kktMatBeg, mm + 1); dump_i("kkt_idx", kk->kktMatIdx, knz); }
        HDSDP_CALL(HKKTBuildUp(kk, KKT_TYPE_HOMOGENEOUS));
        if (kk->isKKTSparse) dump_d("M_hsd", kk->kktMatElem, knz); else dump_d2("M_hsd", kk->kktMatElem, mm, mm);
        dump_d("ASinv_hsd", kk->dASinvVec, mm); dump_d("ASinvRdSinv_hsd", kk->dASinvRdSinvVec, mm);
        dump_d("ASinvCSinv_hsd", kk->dASinvCSinvVec, mm);
        { double sc[4] = { kk->dCSinv, kk->dCSinvCSinv, kk->dCSinvRdSinv, kk->dTraceSinv }; dump_d("hsd_scalars", sc, 4); }
        HDSDP_CALL(HKKTBuildUp(kk, KKT_TYPE_CORRECTOR));
        dump_d("ASinv_cor", kk->dASinvVec, mm); dump_d("ASinvRdSinv_cor", kk->dASinvRdSinvVec, mm);
        HDSDP_CALL(HKKTBuildUp(kk, KKT_TYPE_INFEASIBLE));
        if (kk->isKKTSparse) dump_d("M_inf", kk->kktMatElem, knz); else dump_d2("M_inf", kk->kktMatElem, mm, mm);
        dump_d("ASinv_inf", kk->dASinvVec, mm); dump_d("ASinvRdSinv_inf", kk->dASinvRdSinvVec, mm);
        double *so = calloc(mm, sizeof(double));
        HDSDP_CALL(HKKTFactorize(kk));
        HDSDP_CALL(HKKTSolve(kk, rowRHS, so)); dump_d("sol_b", so, mm);
        if (kk->isKKTSparse && argc > 7) {
            /* the SPARSE operator on a matrix that is not positive definite: the diagonal is lowered through kktDiag (how the
               bound cone and HKKTRegularize write into it) until some eigenvalues are negative; the reference's sparse direct
               solver is an LDL' without pivoting (linalg/hdsdp_linsolver.c:596-626 over external/qdldl.c): it factors, solves,
               and only its PSD check says "not positive definite" */
            const double shift = atof(argv[7]);
            for (int i = 0; i < mm; ++i) *kk->kktDiag[i] -= shift;
            int codes[3] = {-1, -1, -1};
            codes[0] = (int) HKKTFactorize(kk);
            codes[1] = (int) HKKTSolve(kk, rowRHS, so);
            dump_d("indef_sol", so, mm);
            (void) HFpLinsysPsdCheck(kk->kktM, kk->kktMatBeg, kk->kktMatIdx, kk->kktMatElem, &codes[2]);
            dump_i("indef_codes", codes, 3);
            dump_s("indef_shift", shift);
            printf("indefinite round: shift %g, factorize rc %d, solve rc %d, isPsd %d\n", shift, codes[0], codes[1], codes[2]);
        }
        printf("ref_dump ok: blocks=%d m=%d logdet=%.12e\n", nBlks, mm, ldsum);
        return 0;
    }
    if (!strcmp(mode, "sdpa")) {
        int nConstrs = 0, nBlks = 0, *BlkDims = NULL, nCols = 0, nLpCols = 0, nElem = 0;
        int **cBeg = NULL, **cIdx = NULL, *LpBeg = NULL, *LpIdx = NULL;
        double **cElem = NULL, *rowRHS = NULL, *LpElem = NULL;
        HDSDP_CALL(HReadSDPA(argv[3], &nConstrs, &nBlks, &BlkDims, &rowRHS, &cBeg, &cIdx, &cElem,
                             &nCols, &nLpCols, &LpBeg, &LpIdx, &LpElem, &nElem));
        if (nBlks != 1 || nLpCols != 0) { fprintf(stderr, "harness handles single-block SDPs only\n"); return 2; }
        pb.n = BlkDims[0]; pb.m = nConstrs; pb.beg = cBeg[0]; pb.idx = cIdx[0]; pb.val = cElem[0]; pb.b = rowRHS;
        Rd = atof(argv[4]); tau = atof(argv[5]); yscale = atof(argv[6]);
    } else {
        int n = atoi(argv[3]), m = atoi(argv[4]);
        if (argc < 8) { fprintf(stderr, "usage\n"); return 2; }
        if (!strcmp(mode, "syn") || !strcmp(mode, "bench")) gen_syn(&pb, n, m); else gen_mix(&pb, n, m);
        Rd = atof(argv[5]); tau = atof(argv[6]); yscale = atof(argv[7]);
    }
    int n = pb.n, m = pb.m;
    long nnz = pb.beg[m + 1];
    if (!strcmp(mode, "bench")) {
        /* the unit of work of BASELINE.json's metric, timed on the reference's CPU path */
        user_data *ud = NULL; hdsdp_cone *cone = NULL; hdsdp_kkt *kkt = NULL;
        double *y = calloc(m, sizeof(double)), *d2 = calloc(m, sizeof(double)), *d3 = calloc(m, sizeof(double));
        double *sol = calloc(m, sizeof(double));
        int isInt = 0;
        HDSDP_CALL(HUserDataCreate(&ud));
        HUserDataSetConeData(ud, HDSDP_CONETYPE_DENSE_SDP, m, n, pb.beg, pb.idx, pb.val);
        HDSDP_CALL(HConeCreate(&cone, 0));
        HDSDP_CALL(HConeSetData(cone, ud));
        HDSDP_CALL(HConeProcData(cone));
        HDSDP_CALL(HConePresolveData(cone));
        HConeSetStart(cone, Rd);
        HDSDP_CALL(HKKTCreate(&kkt));
        HDSDP_CALL(HKKTInit(kkt, m, 1, &cone));
        double t0 = HUtilGetTimeStamp();
        HDSDP_CALL(HConeCheckIsInterior(cone, tau, y, &isInt));
        double t1 = HUtilGetTimeStamp();
        HDSDP_CALL(HKKTBuildUp(kkt, KKT_TYPE_INFEASIBLE));
        double t2 = HUtilGetTimeStamp();
        HKKTExport(kkt, d2, d3, NULL, NULL, NULL, NULL, NULL);
        HDSDP_CALL(HKKTFactorize(kkt));
        double t3 = HUtilGetTimeStamp();
        HDSDP_CALL(HKKTSolve(kkt, pb.b, sol));
        HDSDP_CALL(HKKTSolve(kkt, d2, NULL));
        HDSDP_CALL(HKKTSolve(kkt, d3, NULL));
        double t4 = HUtilGetTimeStamp();
        double sum_d2 = 0.0;
        for (int i = 0; i < m; ++i) sum_d2 += d2[i];
        printf("{\"n\": %d, \"m\": %d, \"interior\": %d, \"chol_s\": %.6f, \"buildup_s\": %.6f, \"factor_s\": %.6f, \"solve3_s\": %.6f, \"sum_d2\": %.15e}\n",
               n, m, isInt, t1 - t0, t2 - t1, t3 - t2, t4 - t3, sum_d2);
        return 0;
    }
    dump_i("csc_beg", pb.beg, m + 2); dump_i("csc_idx", pb.idx, nnz); dump_d("csc_val", pb.val, nnz);
    dump_d("b", pb.b, m);
    { int dims[2] = {n, m}; dump_i("dims", dims, 2); }
    dump_s("Rd", Rd); dump_s("tau", tau);

    user_data *ud = NULL; hdsdp_cone *cone = NULL; hdsdp_kkt *kkt = NULL;
    double *y = calloc(m, sizeof(double)), *rhs = calloc(m, sizeof(double)), *sol = calloc(m, sizeof(double));
    for (int i = 0; i < m; ++i) y[i] = yscale * sin(1.7 * (i + 1));
    dump_d("y", y, m);
    double logdet = 0.0;
    HDSDP_CALL(HUserDataCreate(&ud));
    HUserDataSetConeData(ud, HDSDP_CONETYPE_DENSE_SDP, m, n, pb.beg, pb.idx, pb.val);
    HDSDP_CALL(HConeCreate(&cone, 0));
    HDSDP_CALL(HConeSetData(cone, ud));
    HDSDP_CALL(HConeProcData(cone));
    HDSDP_CALL(HConePresolveData(cone));
    HConeSetStart(cone, Rd);
    int isInt = 0;
    HDSDP_CALL(HConeCheckIsInterior(cone, tau, y, &isInt));
    if (!isInt) { fprintf(stderr, "state is not interior; pick a more negative Rd\n"); return 3; }
    HDSDP_CALL(HConeGetLogBarrier(cone, tau, y, BUFFER_DUALVAR, &logdet));
    dump_s("logdet", logdet);

    /* ratio test (HConeRatioTest -> sdpDenseConeRatioTestImpl hdsdp_conic_sdp.c:1640-1686 -> HLanczosSolve): two calls,
       the second one warm-started from the first (hdsdp_lanczos.c:166-181).  Dense dual matrices only: with a sparse
       dual matrix (mcp100) the reference's operator returns the zero vector from this harness state and its own
       dsyevr call aborts (k = 0 check), so there is nothing to pin. */
    if (!((hdsdp_cone_sdp_dense *) cone->coneData)->isDualSparse) {
        double *dy = calloc(m, sizeof(double));
        double step = 0.0;
        const double sc = 0.05 * fabs(Rd) / sqrt((double) m);
        for (int i = 0; i < m; ++i) dy[i] = sc * cos(0.7 * i + 0.2);
        HDSDP_CALL(HConeRatioTest(cone, 0.0, dy, 0.0, BUFFER_DUALVAR, &step));
        dump_d("rt_dy1", dy, m); { double pr[2] = {0.0, 0.0}; dump_d("rt_par1", pr, 2); } dump_s("rt_step1", step);
        for (int i = 0; i < m; ++i) dy[i] = 0.5 * sc * sin(1.3 * i + 0.4);
        HDSDP_CALL(HConeRatioTest(cone, -0.05, dy, 1.0, BUFFER_DUALVAR, &step));
        dump_d("rt_dy2", dy, m); { double pr[2] = {-0.05, 1.0}; dump_d("rt_par2", pr, 2); } dump_s("rt_step2", step);
        /* the checker buffer of the line search (hdsdp_conic_sdp.c:2333-2361, :2192-2207, :2252-2291): trial point
           S + 0.5*step*dS factored in the checker, its log-barrier, a ratio test from that trial point (third Lanczos
           call, warm-started), a trial point beyond the boundary, and an expert check with half the residual */
        {
            int ok = 0; double ld = 0.0, st3 = 0.0;
            HDSDP_CALL(HConeAddStepToBufferAndCheck(cone, 0.5 * step, BUFFER_DUALCHECK, &ok));
            HDSDP_CALL(HConeGetLogBarrier(cone, 0.0, NULL, BUFFER_DUALCHECK, &ld));
            HDSDP_CALL(HConeRatioTest(cone, -0.05, dy, 1.0, BUFFER_DUALCHECK, &st3));
            double r1[3] = { (double) ok, ld, st3 };
            HDSDP_CALL(HConeAddStepToBufferAndCheck(cone, 1.5 * step, BUFFER_DUALCHECK, &ok));
            double r2 = (double) ok;
            HDSDP_CALL(HConeCheckIsInteriorExpert(cone, tau, -1.0, y, -0.5 * Rd, BUFFER_DUALCHECK, &ok));
            HDSDP_CALL(HConeGetLogBarrier(cone, 0.0, NULL, BUFFER_DUALCHECK, &ld));
            double r3[2] = { (double) ok, ld };
            dump_d("ck_axpy_half", r1, 3); dump_s("ck_axpy_beyond", r2); dump_d("ck_expert", r3, 2);
        }
        free(dy);
    }

    hdsdp_cone_sdp_dense *dc = (hdsdp_cone_sdp_dense *) cone->coneData;
    /* classification, ordering, strategies (hdsdp_conic_sdp.c:602-676, hdsdp_sdpdata.c:2321-2458) */
    {
        int *types = malloc(m * sizeof(int)), *ranks = malloc(m * sizeof(int)), *nnzs = malloc(m * sizeof(int));
        for (int i = 0; i < m; ++i) {
            types[i] = (int) sdpDataMatGetType(dc->sdpRow[i]);
            ranks[i] = sdpDataMatGetRank(dc->sdpRow[i]);
            nnzs[i] = sdpDataMatGetNnz(dc->sdpRow[i]);
        }
        dump_i("coef_type", types, m); dump_i("coef_rank", ranks, m); dump_i("coef_nnz", nnzs, m);
        dump_i("kkt_perm", dc->sdpConePerm, m); dump_i("kkt_strategy", dc->KKTStrategies, m);
        int ot = (int) sdpDataMatGetType(dc->sdpObj); dump_i("obj_type", &ot, 1);
        int ds = dc->isDualSparse; dump_i("dual_sparse", &ds, 1);
    }
    /* feature detection (HConeDetectFeature -> the cone's getstat slot, hdsdp_conic_sdp.c:2651-2745; read once by the driver,
       hdsdp.c:163): class counts, "no primal interior", implied trace bound, "very dense" */
    {
        int fi[20]; double fd[20];
        memset(fi, 0, sizeof(fi)); memset(fd, 0, sizeof(fd));
        HConeDetectFeature(cone, pb.b, fi, fd);
        dump_i("feat_int", fi, 20); dump_d("feat_dbl", fd, 20);
    }
    /* S as a dense lower-valid n x n column-major array */
    {
        double *S = calloc((size_t) n * n, sizeof(double));
        if (dc->isDualSparse) {
            for (int j = 0; j < n; ++j) for (int k = dc->dualMatBeg[j]; k < dc->dualMatBeg[j + 1]; ++k)
                S[(size_t) j * n + dc->dualMatIdx[k]] = dc->dualMatElem[k];
        } else {
            for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i) S[(size_t) j * n + i] = dc->dualMatElem[(size_t) j * n + i];
        }
        dump_d2("S", S, n, n);
        free(S);
    }
    { double *dg = calloc(n, sizeof(double)); HFpLinsysGetDiag(dc->dualFactor, dg); dump_d("Ldiag", dg, n); free(dg); }

    HDSDP_CALL(HKKTCreate(&kkt));
    HDSDP_CALL(HKKTInit(kkt, m, 1, &cone));
    if (kkt->isKKTSparse) { fprintf(stderr, "sparse Schur: harness dumps dense M only\n"); return 4; }

    const int types[3] = { KKT_TYPE_INFEASIBLE, KKT_TYPE_HOMOGENEOUS, KKT_TYPE_CORRECTOR };
    const char *tn[3] = { "inf", "hsd", "cor" };
    char nm[64];
    for (int t = 0; t < 3; ++t) {
        HDSDP_CALL(HKKTBuildUp(kkt, types[t]));
        if (t == 0) dump_d2("Sinv", kkt->invBuffer, n, n);
        if (types[t] != KKT_TYPE_CORRECTOR) { snprintf(nm, 64, "M_%s", tn[t]); dump_d2(nm, kkt->kktMatElem, m, m); }
        snprintf(nm, 64, "ASinv_%s", tn[t]); dump_d(nm, kkt->dASinvVec, m);
        snprintf(nm, 64, "ASinvRdSinv_%s", tn[t]); dump_d(nm, kkt->dASinvRdSinvVec, m);
        if (types[t] == KKT_TYPE_HOMOGENEOUS) {
            dump_d("ASinvCSinv_hsd", kkt->dASinvCSinvVec, m);
            double sc[4] = { kkt->dCSinv, kkt->dCSinvCSinv, kkt->dCSinvRdSinv, kkt->dTraceSinv };
            dump_d("hsd_scalars", sc, 4);
        }
        if (types[t] == KKT_TYPE_INFEASIBLE) {
            dump_s("TraceSinv_inf", kkt->dTraceSinv);
            /* cross-strategy invariant (reference HUtilKKTCheck, hdsdp_utils.c:536-707): fixed M3 and M4 */
        }
    }
    /* rebuild INFEASIBLE, then the three Phase-A solves (hdsdp_algo.c:1099-1101) */
    HDSDP_CALL(HKKTBuildUp(kkt, KKT_TYPE_INFEASIBLE));
    double *d2 = calloc(m, sizeof(double)), *d3 = calloc(m, sizeof(double));
    HKKTExport(kkt, d2, d3, NULL, NULL, NULL, NULL, NULL);
    /* the mix family has ZERO rows (singular M): add what the y-box cone would add to diag(M)
       (hdsdp_conic_bound.c:201-229 writes through kkt->kktDiag[] exactly like this) */
    double diag_add = !strcmp(mode, "mix") ? 1e-2 : 0.0;
    for (int i = 0; i < m; ++i) *kkt->kktDiag[i] += diag_add;
    dump_s("diag_add", diag_add);
    HDSDP_CALL(HKKTFactorize(kkt));
    HDSDP_CALL(HKKTSolve(kkt, pb.b, sol)); dump_d("sol_b", sol, m);
    HDSDP_CALL(HKKTSolve(kkt, d2, NULL)); dump_d("sol_ASinv", d2, m);
    HDSDP_CALL(HKKTSolve(kkt, d3, NULL)); dump_d("sol_ASinvRdSinv", d3, m);
    /* fixed-strategy rebuilds: every strategy must give the same M (reference's own invariant) */
    HDSDP_CALL(HKKTBuildUpFixed(kkt, KKT_TYPE_INFEASIBLE, KKT_M3)); dump_d2("M_inf_fixedM3", kkt->kktMatElem, m, m);
    HDSDP_CALL(HKKTBuildUpFixed(kkt, KKT_TYPE_INFEASIBLE, KKT_M4)); dump_d2("M_inf_fixedM4", kkt->kktMatElem, m, m);
    /* primal recovery HConeGetPrimal -> sdpDenseConeGetPrimal (hdsdp_conic_sdp.c:2393-2446): X = mu * L^-T (sym(L^-1 dS L^-T)
       + I) L^-1 with S = C - sum y_i A_i (no residual term) factored in the checker and dS = sum dy_i A_i.  The point
       y must make that S positive definite: y0 of the synthetic family (S = I), else the first of two simple
       candidates the reference's own expert check accepts; no candidate -> nothing dumped. */
    {
        double *yr = calloc(m, sizeof(double)), *dyr = calloc(m, sizeof(double));
        int found = 0;
        for (int cand = 0; cand < 3 && !found; ++cand) {
            if (cand == 0) { if (!pb.y0) continue; memcpy(yr, pb.y0, m * sizeof(double)); }
            if (cand == 1) for (int i = 0; i < m; ++i) yr[i] = -100.0;
            if (cand == 2) { memset(yr, 0, m * sizeof(double)); yr[0] = -100.0; }
            int ok = 0;
            HDSDP_CALL(HConeCheckIsInteriorExpert(cone, 1.0, -1.0, yr, 0.0, BUFFER_DUALCHECK, &ok));
            found = ok;
        }
        if (found) {
            const double mu = 0.37;
            for (int i = 0; i < m; ++i) dyr[i] = 0.02 * sin(0.9 * i + 0.1);
            double *X = calloc((size_t) n * n, sizeof(double)), *aux = calloc((size_t) n * n, sizeof(double));
            HConeGetPrimal(cone, mu, yr, dyr, X, aux);
            dump_d("pr_y", yr, m); dump_d("pr_dy", dyr, m); dump_s("pr_mu", mu);
            dump_d2("pr_X", X, n, n);
            double tr = 0.0, sm = 0.0, wsum = 0.0;
            for (int j = 0; j < n; ++j) { tr += X[j + (size_t) j * n]; for (int i = 0; i < n; ++i) { sm += X[i + (size_t) j * n]; wsum += X[i + (size_t) j * n] * cos(0.013 * i + 0.007 * j); } }
            double cs[3] = { tr, sm, wsum };
            dump_d("pr_checks", cs, 3);
            /* the recovery re-factored the checker and overwrote dualStep: restore the state the rest of the dump expects */
            HDSDP_CALL(HConeCheckIsInterior(cone, tau, y, &isInt));
            /* the remaining cone utilities on that X (hdsdp_conic_sdp.c:1558-1614, :2470-2560) */
            {
                double *ax = calloc(m, sizeof(double));
                for (int i = 0; i < m; ++i) ax[i] = 0.25 * i;
                HConeComputeATimesXpy(cone, X, ax);
                dump_d("ut_atimesx", ax, m);
                double sc[7];
                sc[0] = HConeComputeTraceCX(cone, X); sc[1] = HConeComputeXDotS(cone, X);
                sc[2] = HConeGetCoeffNorm(cone, ABS_NORM); sc[3] = HConeGetCoeffNorm(cone, FRO_NORM);
                sc[4] = HConeGetObjNorm(cone, ABS_NORM); sc[5] = HConeGetObjNorm(cone, FRO_NORM);
                HConeScalByConstant(cone, 0.5);
                sc[6] = HConeGetObjNorm(cone, FRO_NORM);
                HConeScalByConstant(cone, 2.0);
                dump_d("ut_scalars", sc, 7);
                free(ax);
            }
            free(X); free(aux);
        }
        free(yr); free(dyr);
    }
    /* KKT_TYPE_PRIMAL (hdsdp_conic_sdp.c:1745-1753, driver hdsdp_psdp.c:156,203): the same builder on a registered
       primal matrix X instead of S^-1.  X is a fixed closed form (tests/util.py:primal_X regenerates it). */
    {
        double *X = calloc((size_t) n * n, sizeof(double));
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
                X[i + (size_t) j * n] = (i == j) ? 2.0 + 0.01 * (i % 7)
                                                 : 0.5 / n * cos(0.37 * (double) (i + j) + 0.11 * (double) i * (double) j);
        double *Xs[1] = { X };
        HKKTRegisterPSDP(kkt, Xs);
        HDSDP_CALL(HKKTBuildUp(kkt, KKT_TYPE_PRIMAL));
        dump_d2("M_pri", kkt->kktMatElem, m, m);
        dump_d("ASinv_pri", kkt->dASinvVec, m);
        dump_d("ASinvRdSinv_pri", kkt->dASinvRdSinvVec, m);
        dump_s("TraceSinv_pri", kkt->dTraceSinv);
        free(X);
    }
    /* state "C": the two cone slots a solve moves between builds -- a dual perturbation (HConeSetPerturb,
       hdsdp_conic_sdp.c:2237-2241: added to the diagonal of S and of the checker, never of the step, :383, :441) and a
       reduced residual (HConeReduceResi, :2225-2229; Phase A shrinks it every iteration, hdsdp_algo.c:244; Phase B starts
       with residual 0 and perturbation -10 x residual, hdsdp_algo.c:1698-1704).  c1: perturbation on top of the state
       above; c2: the residual reduced as well; c3: Phase B's form, residual 0 and the whole shift carried by the
       perturbation (the dual matrix is then the one of the main dump). */
    {
        const double perts[3] = { 0.02 * fabs(Rd), 0.02 * fabs(Rd), fabs(Rd) };
        const double resis[3] = { Rd, 0.6 * Rd, 0.0 };
        double par[6];
        for (int k = 0; k < 3; ++k) { par[2 * k] = perts[k]; par[2 * k + 1] = resis[k]; }
        dump_d("c_par", par, 6);
        for (int k = 0; k < 3; ++k) {
            HConeSetPerturb(cone, perts[k]);
            HConeReduceResi(cone, resis[k]);
            int ok = 0; double ld = 0.0;
            HDSDP_CALL(HConeCheckIsInterior(cone, tau, y, &ok));
            snprintf(nm, 64, "c%d_interior", k + 1); dump_i(nm, &ok, 1);
            if (!ok) continue;
            HDSDP_CALL(HConeGetLogBarrier(cone, tau, y, BUFFER_DUALVAR, &ld));
            snprintf(nm, 64, "c%d_logdet", k + 1); dump_s(nm, ld);
            HDSDP_CALL(HKKTBuildUp(kkt, KKT_TYPE_HOMOGENEOUS));
            snprintf(nm, 64, "c%d_ASinvCSinv_hsd", k + 1); dump_d(nm, kkt->dASinvCSinvVec, m);
            { double sc[4] = { kkt->dCSinv, kkt->dCSinvCSinv, kkt->dCSinvRdSinv, kkt->dTraceSinv };
              snprintf(nm, 64, "c%d_hsd_scalars", k + 1); dump_d(nm, sc, 4); }
            HDSDP_CALL(HKKTBuildUp(kkt, KKT_TYPE_INFEASIBLE));
            snprintf(nm, 64, "c%d_M_inf", k + 1); dump_d2(nm, kkt->kktMatElem, m, m);
            snprintf(nm, 64, "c%d_ASinv_inf", k + 1); dump_d(nm, kkt->dASinvVec, m);
            snprintf(nm, 64, "c%d_ASinvRdSinv_inf", k + 1); dump_d(nm, kkt->dASinvRdSinvVec, m);
            snprintf(nm, 64, "c%d_TraceSinv_inf", k + 1); dump_s(nm, kkt->dTraceSinv);
            for (int i = 0; i < m; ++i) *kkt->kktDiag[i] += diag_add;
            HKKTExport(kkt, d2, NULL, NULL, NULL, NULL, NULL, NULL);
            HDSDP_CALL(HKKTFactorize(kkt));
            HDSDP_CALL(HKKTSolve(kkt, d2, NULL));
            snprintf(nm, 64, "c%d_sol_ASinv", k + 1); dump_d(nm, d2, m);
            HDSDP_CALL(HKKTBuildUp(kkt, KKT_TYPE_CORRECTOR));
            snprintf(nm, 64, "c%d_ASinv_cor", k + 1); dump_d(nm, kkt->dASinvVec, m);
            snprintf(nm, 64, "c%d_ASinvRdSinv_cor", k + 1); dump_d(nm, kkt->dASinvRdSinvVec, m);
            /* the checker takes the perturbation too (:441), the step buffer does not (:383-385): trial point through the
               expert check with the caller's own diagonal term */
            HDSDP_CALL(HConeCheckIsInteriorExpert(cone, tau, -1.0, y, -0.5 * resis[k], BUFFER_DUALCHECK, &ok));
            HDSDP_CALL(HConeGetLogBarrier(cone, 0.0, NULL, BUFFER_DUALCHECK, &ld));
            { double r3[2] = { (double) ok, ld }; snprintf(nm, 64, "c%d_ck_expert", k + 1); dump_d(nm, r3, 2); }
        }
    }
    printf("ref_dump ok: n=%d m=%d nnz=%ld logdet=%.12e\n", n, m, nnz, logdet);
exit_cleanup:
    if (retcode != HDSDP_RETCODE_OK) fprintf(stderr, "ref_dump: reference returned %d\n", (int) retcode);
    return (int) retcode;
}
