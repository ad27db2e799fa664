/* hdsdp_oracle.c -- TEST INFRASTRUCTURE ONLY (see hdsdp_oracle.h).
 *
 * Plain-C restatement of the reference CPU algorithm for the Schur hot path: same data classes, same
 * strategy plan, same level-1/2 loop structure (the reference deliberately avoids level-3 BLAS here,
 * linalg/hdsdp_sdpdata.c:1161-1168), written from the reference's behaviour with our own helpers.
 * Every routine names the reference code it follows.  Parity is pinned by tests/test_oracle.py against
 * the golden vectors generated from the compiled reference.
 */
#include "hdsdp_oracle.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { T_ZERO = 0, T_SPARSE = 1, T_DENSE = 2, T_SPR1 = 3, T_DSR1 = 4 };
enum { S_M1 = 0, S_M2 = 1, S_M3 = 2, S_M4 = 3, S_M5 = 4 };

typedef struct {
    int type, n, nnz, rank;
    int nt, *ri, *ci; double *x;   /* sparse: lower triplets sorted by packed index */
    double *pk;                    /* dense: packed lower, column-major */
    double sign; int k, *kidx; double *kval, *fac; /* rank one: sign * a a', support list + dense a */
} coef;

struct orc_block_s {
    int n, m;
    coef obj, *rows;
    int *perm, *strategy;
};

#define PK(n, i, j) ((long) (2 * (n) - (j) - 1) * (j) / 2 + (i))
#define FE(B, n, i, j) ((B)[(long) (j) * (n) + (i)])

/* ---------------------------------------------------------------- small level-1/2 helpers */
static double vdot(int n, const double *x, int ix, const double *y, int iy) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += x[(long) i * ix] * y[(long) i * iy];
    return s;
}
static void vaxpy(int n, double a, const double *x, int ix, double *y, int iy) {
    for (int i = 0; i < n; ++i) y[(long) i * iy] += a * x[(long) i * ix];
}
/* y = A x for packed-lower symmetric A (the dspmv the reference calls at hdsdp_sdpdata.c:1199-1203) */
static void packed_symv(int n, const double *pk, const double *x, double *y) {
    memset(y, 0, sizeof(double) * n);
    const double *col = pk;
    for (int j = 0; j < n; ++j) {
        y[j] += col[0] * x[j];
        for (int i = j + 1; i < n; ++i) {
            double a = col[i - j];
            y[i] += a * x[j];
            y[j] += a * x[i];
        }
        col += n - j;
    }
}
/* y = S x for a full symmetric S (fds_symv, dense_opts.c:49) */
static void full_symv(int n, const double *S, const double *x, double *y) {
    for (int i = 0; i < n; ++i) y[i] = 0.0;
    for (int j = 0; j < n; ++j) vaxpy(n, x[j], S + (long) j * n, 1, y, 1);
}

/* ---------------------------------------------------------------- data classes */
static void free_coef(coef *c) {
    free(c->ri); free(c->ci); free(c->x); free(c->pk); free(c->kidx); free(c->kval); free(c->fac);
    memset(c, 0, sizeof(*c));
}

/* rank-one test on sorted lower triplets (linalg/sparse_opts.c:453-516) */
static int triplet_rank_one(int n, int nt, const int *ri, const int *ci, const double *x, double *sgn, double *a) {
    if (ri[0] != ci[0]) return 0;
    int i = ri[0];
    if (nt == 1) { *sgn = x[0]; a[i] = 1.0; return 1; }
    double s = x[0] > 0 ? 1.0 : -1.0, v = sqrt(fabs(x[0]));
    int k, cnt = 0;
    for (k = 0; k < nt; ++k) {
        if (ci[k] > i) break;
        a[ri[k]] = x[k] / v;
        cnt++;
    }
    if (nt != cnt * (cnt + 1) / 2) return 0;
    if (k == n) return 0;
    double eps = 0.0;
    for (k = 0; k < nt; ++k) eps += fabs(x[k] - s * a[ri[k]] * a[ci[k]]);
    if (eps > 1e-10) return 0;
    *sgn = s;
    return 1;
}

/* rank-one test on a packed dense matrix (linalg/dense_opts.c:233-285) */
static int packed_rank_one(int n, const double *pk, double *sgn, double *a) {
    int i; long k = 0;
    for (i = 0; i < n; ++i) { if (pk[k] != 0) break; k += n - i; }
    if (i == n) return 0;
    double s = pk[k] > 0 ? 1.0 : -1.0, v = sqrt(fabs(pk[k]));
    for (int r = 0; r < n; ++r) a[r] = pk[PK(n, 0, i) + r] / v;
    double eps = 0.0; long id = 0;
    for (int c = 0; c < n; ++c) {
        for (int r = 0; r < n - c; ++r) eps += fabs(pk[id + r] - s * a[c] * a[c + r]);
        id += n - c;
        if (eps > 1e-10) return 0;
    }
    *sgn = s;
    return 1;
}

/* classification + rank-one conversion: hdsdp_sdpdata.c:2321-2345 and :2373-2458, normalisation :880-899 */
static void make_coef(coef *c, int n, int nnz, const int *idx, const double *val) {
    memset(c, 0, sizeof(*c));
    c->n = n;
    const long P = (long) n * (n + 1) / 2;
    if (nnz == 0) { c->type = T_ZERO; return; }
    double *a = calloc(n, sizeof(double)), sgn = 0.0;
    int r1 = 0;
    if (nnz > 0.3 * P) {
        c->type = T_DENSE; c->nnz = (int) P; c->rank = n;
        c->pk = calloc(P, sizeof(double));
        for (int k = 0; k < nnz; ++k) c->pk[idx[k]] = val[k];
        r1 = packed_rank_one(n, c->pk, &sgn, a);
    } else {
        c->type = T_SPARSE; c->nnz = nnz; c->rank = n; c->nt = nnz;
        c->ri = malloc(sizeof(int) * nnz); c->ci = malloc(sizeof(int) * nnz); c->x = malloc(sizeof(double) * nnz);
        int j = 0; long thresh = n;   /* walk the packed columns (sparse_opts.c:427-441) */
        for (int k = 0; k < nnz; ++k) {
            while (idx[k] >= thresh) { j++; thresh += n - j; }
            c->ri[k] = (int) (idx[k] - thresh + n); c->ci[k] = j; c->x[k] = val[k];
        }
        r1 = triplet_rank_one(n, nnz, c->ri, c->ci, c->x, &sgn, a);
    }
    if (r1) {
        int fn = 0;
        for (int r = 0; r < n; ++r) fn += fabs(a[r]) > 1e-10;
        free(c->ri); free(c->ci); free(c->x); free(c->pk);
        c->ri = c->ci = NULL; c->x = c->pk = NULL; c->nt = 0;
        c->rank = 1;
        c->fac = calloc(n, sizeof(double));
        if (fn > 0.5 * n) {
            c->type = T_DSR1; c->nnz = (int) P; c->k = n;
            c->kidx = malloc(sizeof(int) * n); c->kval = malloc(sizeof(double) * n);
            for (int r = 0; r < n; ++r) { c->kidx[r] = r; c->kval[r] = a[r]; c->fac[r] = a[r]; }
        } else {
            c->type = T_SPR1; c->nnz = fn * (fn + 1) / 2; c->k = fn;
            c->kidx = malloc(sizeof(int) * (fn + 1)); c->kval = malloc(sizeof(double) * (fn + 1));
            int q = 0;
            for (int r = 0; r < n; ++r) if (fabs(a[r]) > 1e-10) { c->kidx[q] = r; c->kval[q] = a[r]; c->fac[r] = a[r]; q++; }
        }
        double nrm = sqrt(vdot(c->k, c->kval, 1, c->kval, 1));
        c->sign = sgn * nrm * nrm;
        for (int q = 0; q < c->k; ++q) c->kval[q] /= nrm;
        for (int r = 0; r < n; ++r) c->fac[r] /= nrm;
    }
    free(a);
}

/* descending sort with the reference's partition scheme so ties land identically (hdsdp_utils.c:93-112) */
static int part_desc(int *ind, int *val, int l, int h) {
    int t0 = l, p = val[l], t;
    while (l < h) {
        while (l < h && val[h] <= p) --h;
        while (l < h && val[l] >= p) ++l;
        if (l < h) { t = val[l]; val[l] = val[h]; val[h] = t; t = ind[l]; ind[l] = ind[h]; ind[h] = t; }
    }
    t = val[l]; val[l] = val[t0]; val[t0] = t; t = ind[l]; ind[l] = ind[t0]; ind[t0] = t;
    return l;
}
static void sort_desc(int *ind, int *val, int lo, int up) {
    if (lo < up) { int p = part_desc(ind, val, lo, up); sort_desc(ind, val, lo, p - 1); sort_desc(ind, val, p + 1, up); }
}

/* per-position strategy score (hdsdp_conic_sdp.c:539-600) */
static int pick_strategy(const int *ranks, const int *sp, const int *perm, int m, int n, int pos) {
    const double kap = 1.5, n3 = (double) n * n * n;
    double after = 0.0, f = sp[pos], best = INFINITY;
    for (int i = pos; i < m; ++i) after += sp[i];
    int r = ranks[perm[pos]], pick = S_M1;
    double s2 = r * (f * n + 3 * kap * after);
    double s3 = (double) n * kap * f + n3 + kap * after + n3 / m;
    double s4 = (double) n * kap * f + kap * (n + 1) * after + n3 / m;
    double s5 = kap * (2.0 * kap * f + 1) * after + n3 / m;
    if (s2 <= best) { pick = S_M2; best = s2; }
    if (s3 < best) { pick = S_M3; best = s3; }
    if (s4 < best) { pick = S_M4; best = s4; }
    if (s5 < best) { pick = S_M5; best = s5; }
    return pick;
}

orc_block *orc_block_create(int n, int m, const int *beg, const int *idx, const double *val) {
    orc_block *b = calloc(1, sizeof(*b));
    b->n = n; b->m = m;
    b->rows = calloc(m, sizeof(coef));
    make_coef(&b->obj, n, beg[1] - beg[0], idx + beg[0], val + beg[0]);
    for (int i = 0; i < m; ++i) make_coef(&b->rows[i], n, beg[i + 2] - beg[i + 1], idx + beg[i + 1], val + beg[i + 1]);
    b->perm = malloc(sizeof(int) * m); b->strategy = malloc(sizeof(int) * m);
    int *ranks = malloc(sizeof(int) * m), *sp = malloc(sizeof(int) * m);
    for (int i = 0; i < m; ++i) { b->perm[i] = i; ranks[i] = b->rows[i].rank; sp[i] = b->rows[i].nnz; }
    sort_desc(b->perm, sp, 0, m - 1);                       /* hdsdp_conic_sdp.c:651 */
    for (int p = 0; p < m; ++p) b->strategy[p] = pick_strategy(ranks, sp, b->perm, m, n, p);
    free(ranks); free(sp);
    return b;
}

void orc_block_free(orc_block *b) {
    if (!b) return;
    free_coef(&b->obj);
    for (int i = 0; i < b->m; ++i) free_coef(&b->rows[i]);
    free(b->rows); free(b->perm); free(b->strategy); free(b);
}

void orc_get_presolve(const orc_block *b, int *type, int *rank, int *nnz, int *perm, int *strategy, int *objType) {
    for (int i = 0; i < b->m; ++i) {
        if (type) type[i] = b->rows[i].type;
        if (rank) rank[i] = b->rows[i].rank;
        if (nnz) nnz[i] = b->rows[i].nnz;
        if (perm) perm[i] = b->perm[i];
        if (strategy) strategy[i] = b->strategy[i];
    }
    if (objType) *objType = b->obj.type;
}

/* B(lower) += alpha * A   (the add2buffer family, hdsdp_sdpdata.c:589-683) */
static void add_to_lower(const coef *c, double alpha, double *B) {
    const int n = c->n;
    if (alpha == 0.0) return;
    switch (c->type) {
        case T_SPARSE: for (int k = 0; k < c->nt; ++k) FE(B, n, c->ri[k], c->ci[k]) += alpha * c->x[k]; break;
        case T_DENSE: { const double *col = c->pk;
            for (int j = 0; j < n; ++j) { for (int i = j; i < n; ++i) FE(B, n, i, j) += alpha * col[i - j]; col += n - j; } break; }
        case T_SPR1: case T_DSR1:
            for (int q = 0; q < c->k; ++q) for (int p = q; p < c->k; ++p)
                FE(B, n, c->kidx[p], c->kidx[q]) += alpha * c->sign * c->kval[p] * c->kval[q];
            break;
        default: break;
    }
}

void orc_assemble_S(const orc_block *b, double tau, const double *y, double Rd, double *S) {
    const int n = b->n;
    memset(S, 0, sizeof(double) * (size_t) n * n);
    for (int i = 0; i < b->m; ++i) add_to_lower(&b->rows[i], -1.0 * y[i], S);   /* hdsdp_conic_sdp.c:374-376 */
    add_to_lower(&b->obj, tau, S);                                             /* :379 */
    for (int j = 0; j < n; ++j) FE(S, n, j, j) += -Rd;                         /* :386-399 */
}

int orc_potrf(int n, double *A) {
    for (int j = 0; j < n; ++j) {
        double d = FE(A, n, j, j) - vdot(j, A + j, n, A + j, n);
        if (!(d > 0.0)) return j + 1;
        d = sqrt(d);
        FE(A, n, j, j) = d;
        for (int i = j + 1; i < n; ++i)
            FE(A, n, i, j) = (FE(A, n, i, j) - vdot(j, A + i, n, A + j, n)) / d;
    }
    return 0;
}

void orc_potri_sym(int n, const double *L, double *Sinv) {
    /* X = L^-1 column by column, then Sinv = X^T X, mirrored to both triangles */
    double *X = calloc((size_t) n * n, sizeof(double));
    for (int c = 0; c < n; ++c) {
        FE(X, n, c, c) = 1.0 / FE(L, n, c, c);
        for (int i = c + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s += FE(L, n, i, k) * FE(X, n, k, c);
            FE(X, n, i, c) = -s / FE(L, n, i, i);
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i) {
            double s = vdot(n - i, X + (long) i * n + i, 1, X + (long) j * n + i, 1);
            FE(Sinv, n, i, j) = s; FE(Sinv, n, j, i) = s;
        }
    free(X);
}

double orc_logdet(int n, const double *L) {           /* hdsdp_conic_sdp.c:2277-2287 */
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += log(FE(L, n, i, i));
    return 2.0 * s;
}

/* ---------------------------------------------------------------- M2 kernels (hdsdp_sdpdata.c:1003-1118) */
static void r1_solve(const coef *c, const double *Sinv, double *v) {
    const int n = c->n;
    if (c->type == T_SPR1 && c->k < 0.3 * n) {          /* :1018-1025 sparse combination of columns */
        memset(v, 0, sizeof(double) * n);
        for (int q = 0; q < c->k; ++q) vaxpy(n, c->kval[q], Sinv + (long) c->kidx[q] * n, 1, v, 1);
    } else {
        /* :1013-1016 solves with the factor (dpotrs), :1035-1039 dsymv with S^-1: the same vector */
        full_symv(n, Sinv, c->fac, v);
    }
}
static double r1_trace_asinv(const coef *c, const double *v) {  /* :1066-1082 */
    double s = 0.0;
    for (int q = 0; q < c->k; ++q) s += v[c->kidx[q]] * c->kval[q];
    return c->sign * s;
}
static double quad_form(const coef *c, const double *v, double *aux) {  /* :1085-1118 */
    const int n = c->n;
    double s = 0.0;
    switch (c->type) {
        case T_SPARSE:   /* sparse_opts.c:565 */
            for (int k = 0; k < c->nt; ++k) s += (c->ri[k] == c->ci[k] ? 0.5 : 1.0) * c->x[k] * v[c->ri[k]] * v[c->ci[k]];
            return 2.0 * s;
        case T_DENSE: packed_symv(n, c->pk, v, aux); return vdot(n, aux, 1, v, 1);  /* dense_opts.c:287 */
        case T_SPR1: case T_DSR1:  /* r1_opts.c:43,64 */
            for (int q = 0; q < c->k; ++q) s += c->kval[q] * v[c->kidx[q]];
            return c->sign * s * s;
        default: return 0.0;
    }
}

/* ---------------------------------------------------------------- M3 kernels (hdsdp_sdpdata.c:1122-1359) */
static double sinv_a_sinv(const coef *c, const double *Sinv, double *aux, double *B) {
    const int n = c->n;
    double tr = 0.0;
    switch (c->type) {
        case T_SPARSE:   /* :1127-1182: aux = S^-1 A by column combinations, then strided dots */
            memset(aux, 0, sizeof(double) * (size_t) n * n);
            for (int k = 0; k < c->nt; ++k) {
                int r = c->ri[k], q = c->ci[k];
                vaxpy(n, c->x[k], Sinv + (long) r * n, 1, aux + (long) q * n, 1);
                if (r != q) vaxpy(n, c->x[k], Sinv + (long) q * n, 1, aux + (long) r * n, 1);
            }
            for (int r = 0; r < n; ++r) {
                tr += FE(aux, n, r, r);
                for (int q = 0; q <= r; ++q) FE(B, n, r, q) = vdot(n, aux + r, n, Sinv + (long) q * n, 1);
            }
            return tr;
        case T_DENSE:    /* :1184-1223: aux = A S^-1 by n packed matvecs, then n(n+1)/2 dots */
            for (int j = 0; j < n; ++j) packed_symv(n, c->pk, Sinv + (long) j * n, aux + (long) j * n);
            for (int j = 0; j < n; ++j) {
                tr += FE(aux, n, j, j);
                for (int r = 0; r <= j; ++r) FE(B, n, j, r) = vdot(n, aux + (long) j * n, 1, Sinv + (long) r * n, 1);
            }
            return tr;
        case T_SPR1: case T_DSR1:  /* :1225-1270: B = sign v v' */
            memset(B, 0, sizeof(double) * (size_t) n * n);
            r1_solve(c, Sinv, aux);
            tr = r1_trace_asinv(c, aux);
            for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i) FE(B, n, i, j) += c->sign * aux[i] * aux[j];
            return tr;
        default: return 0.0;
    }
}
static double a_dot_b(const coef *c, const double *B, double *aux) {   /* :1275-1359 */
    const int n = c->n;
    double s = 0.0;
    switch (c->type) {
        case T_SPARSE:
            for (int k = 0; k < c->nt; ++k) s += (c->ri[k] == c->ci[k] ? 0.5 : 1.0) * c->x[k] * FE(B, n, c->ri[k], c->ci[k]);
            return 2.0 * s;
        case T_DENSE: { const double *ac = c->pk, *bc = B;
            for (int j = 0; j < n; ++j) {
                s += 0.5 * ac[0] * bc[0];
                for (int i = 1; i < n - j; ++i) s += ac[i] * bc[i];
                ac += n - j; bc += n + 1;
            }
            return 2.0 * s; }
        case T_SPR1:
            for (int q = 0; q < c->k; ++q) {
                s += 0.5 * c->kval[q] * c->kval[q] * FE(B, n, c->kidx[q], c->kidx[q]);
                for (int p = q + 1; p < c->k; ++p) s += c->kval[p] * c->kval[q] * FE(B, n, c->kidx[p], c->kidx[q]);
            }
            return 2.0 * c->sign * s;
        case T_DSR1: {  /* symv with the LOWER triangle of B, then a dot */
            for (int i = 0; i < n; ++i) aux[i] = 0.0;
            for (int j = 0; j < n; ++j) {
                aux[j] += FE(B, n, j, j) * c->fac[j];
                for (int i = j + 1; i < n; ++i) { aux[i] += FE(B, n, i, j) * c->fac[j]; aux[j] += FE(B, n, i, j) * c->fac[i]; }
            }
            return c->sign * vdot(n, c->fac, 1, aux, 1); }
        default: return 0.0;
    }
}

/* ---------------------------------------------------------------- M4 kernels (hdsdp_sdpdata.c:1364-1690) */
static double a_sinv(const coef *c, const double *Sinv, double *aux, double Rd, double *B) {
    const int n = c->n;
    double t = 0.0;
    memset(B, 0, sizeof(double) * (size_t) n * n);
    switch (c->type) {
        case T_SPARSE:   /* :1369-1436 */
            for (int k = 0; k < c->nt; ++k) {
                int r = c->ri[k], q = c->ci[k];
                vaxpy(n, c->x[k], Sinv + (long) r * n, 1, B + q, n);
                if (r != q) vaxpy(n, c->x[k], Sinv + (long) q * n, 1, B + r, n);
            }
            if (Rd == 0.0) return 0.0;
            if (c->nt > 0.1 * n) return vdot(n * n, Sinv, 1, B, 1);
            for (int k = 0; k < c->nt; ++k) {
                int r = c->ri[k], q = c->ci[k];
                double d = vdot(n, Sinv + (long) q * n, 1, Sinv + (long) r * n, 1);
                t += c->x[k] * d;
                if (r != q) t += c->x[k] * d;
            }
            return t;
        case T_DENSE:    /* :1438-1465 */
            for (int j = 0; j < n; ++j) packed_symv(n, c->pk, Sinv + (long) j * n, B + (long) j * n);
            if (Rd == 0.0) return 0.0;
            return vdot(n * n, Sinv, 1, B, 1);
        case T_SPR1:     /* :1467-1523 */
            if (c->k >= 0.5 * sqrt((double) n)) {
                r1_solve(c, Sinv, aux);
                for (int j = 0; j < n; ++j) vaxpy(n, c->sign * aux[j], c->fac, 1, B + (long) j * n, 1);
                if (Rd == 0.0) return 0.0;
                return c->sign * vdot(n, aux, 1, aux, 1);
            }
            for (int p = 0; p < c->k; ++p) for (int q = 0; q < c->k; ++q) {
                double e = c->sign * c->kval[p] * c->kval[q];
                const double *sc = Sinv + (long) c->kidx[q] * n;
                vaxpy(n, e, sc, 1, B + c->kidx[p], n);
                if (p <= q && Rd != 0.0) {
                    if (p == q) t += 0.5 * e * vdot(n, sc, 1, sc, 1);
                    else t += e * vdot(n, Sinv + (long) c->kidx[p] * n, 1, sc, 1);
                }
            }
            return 2.0 * t;
        case T_DSR1:     /* :1525-1549 */
            r1_solve(c, Sinv, aux);
            for (int j = 0; j < n; ++j) vaxpy(n, c->sign * aux[j], c->fac, 1, B + (long) j * n, 1);
            if (Rd == 0.0) return 0.0;
            return c->sign * vdot(n, aux, 1, aux, 1);
        default: return 0.0;
    }
}
static double a_dot_sinv_b(const coef *c, const double *Sinv, const double *ASinv, double *aux) {  /* :1554-1690 */
    const int n = c->n;
    double s = 0.0;
    switch (c->type) {
        case T_SPARSE:
            for (int k = 0; k < c->nt; ++k) {
                int r = c->ri[k], q = c->ci[k];
                s += c->x[k] * vdot(n, Sinv + (long) r * n, 1, ASinv + (long) q * n, 1);
                if (r != q) s += c->x[k] * vdot(n, Sinv + (long) q * n, 1, ASinv + (long) r * n, 1);
            }
            return s;
        case T_DENSE: { const double *ac = c->pk;
            for (int j = 0; j < n; ++j) {
                s += 0.5 * ac[0] * vdot(n, Sinv + (long) j * n, 1, ASinv + (long) j * n, 1);
                for (int i = j + 1; i < n; ++i) {
                    double a = ac[i - j];
                    if (fabs(a) >= 1e-15) s += a * vdot(n, Sinv + (long) i * n, 1, ASinv + (long) j * n, 1);  /* :1615 */
                }
                ac += n - j;
            }
            return 2.0 * s; }
        case T_SPR1:
            if (c->k >= sqrt((double) n)) {
                double *z = aux + n;
                r1_solve(c, Sinv, aux);
                memset(z, 0, sizeof(double) * n);
                for (int q = 0; q < c->k; ++q) vaxpy(n, c->kval[q], ASinv + (long) c->kidx[q] * n, 1, z, 1);
                return c->sign * vdot(n, aux, 1, z, 1);
            }
            for (int p = 0; p < c->k; ++p) {
                for (int q = 0; q < p; ++q)
                    s += c->kval[p] * c->kval[q] * vdot(n, Sinv + (long) c->kidx[p] * n, 1, ASinv + (long) c->kidx[q] * n, 1);
                s += 0.5 * c->kval[p] * c->kval[p] * vdot(n, Sinv + (long) c->kidx[p] * n, 1, ASinv + (long) c->kidx[p] * n, 1);
            }
            return 2.0 * c->sign * s;
        case T_DSR1: { double *z = aux + n;
            r1_solve(c, Sinv, aux);
            for (int i = 0; i < n; ++i) z[i] = 0.0;
            for (int j = 0; j < n; ++j) vaxpy(n, c->fac[j], ASinv + (long) j * n, 1, z, 1);   /* fds_gemv dense_opts.c:78 */
            return c->sign * vdot(n, aux, 1, z, 1); }
        default: return 0.0;
    }
}

/* ---------------------------------------------------------------- M5 kernels (hdsdp_sdpdata.c:1711-2165) */
static double sparse_entry_sum(const coef *B, const double *sr, const double *sc) {
    /* sum over the symmetric entries of B of  b_pq * sr[p] * sc[q]  (inner loops of the KKT5Pair_Sparse_* kernels) */
    const int n = B->n;
    double t = 0.0;
    switch (B->type) {
        case T_SPARSE:
            for (int k = 0; k < B->nt; ++k) {
                int p = B->ri[k], q = B->ci[k];
                t += B->x[k] * sr[p] * sc[q];
                if (p != q) t += B->x[k] * sr[q] * sc[p];
            }
            return t;
        case T_DENSE: { const double *bc = B->pk;
            for (int j = 0; j < n; ++j) {
                t += bc[0] * sr[j] * sc[j];
                for (int i = j + 1; i < n; ++i) { t += bc[i - j] * sr[i] * sc[j]; t += bc[i - j] * sr[j] * sc[i]; }
                bc += n - j;
            }
            return t; }
        case T_SPR1:
            for (int p = 0; p < B->k; ++p) {
                for (int q = 0; q < p; ++q) {
                    double e = B->kval[p] * B->kval[q];
                    t += e * sr[B->kidx[p]] * sc[B->kidx[q]];
                    t += e * sr[B->kidx[q]] * sc[B->kidx[p]];
                }
                t += B->kval[p] * B->kval[p] * sr[B->kidx[p]] * sc[B->kidx[p]];
            }
            return B->sign * t;
        default: return 0.0;
    }
}
static double pair_trace(const coef *A, const coef *B, const double *Sinv, double *aux, int *ok) {
    /* tr(A S^-1 B S^-1) without intermediates; dispatch table of hdsdp_sdpdata.c:1965-2058 */
    const int n = A->n;
    *ok = 1;
    if (A->type == T_ZERO || B->type == T_ZERO) return 0.0;
    if (A->type == T_DENSE) { *ok = 0; return 0.0; }                     /* :1997-2001 assert(0) */
    if (A->type == T_SPARSE) {
        if (B->type == T_DSR1) { full_symv(n, Sinv, B->fac, aux); return B->sign * quad_form(A, aux, aux + n); }  /* :1872-1885 */
        double s = 0.0;                                                   /* :1711-1870 */
        for (int k = 0; k < A->nt; ++k) {
            double t = sparse_entry_sum(B, Sinv + (long) A->ri[k] * n, Sinv + (long) A->ci[k] * n);
            s += (A->ri[k] == A->ci[k] ? 0.5 : 1.0) * A->x[k] * t;
        }
        return 2.0 * s;
    }
    if (A->type == T_SPR1) {
        if (B->type == T_SPARSE) return pair_trace(B, A, Sinv, aux, ok);  /* :2017 */
        if (B->type == T_DENSE) {                                         /* :1887-1901 */
            memset(aux, 0, sizeof(double) * n);
            for (int q = 0; q < A->k; ++q) vaxpy(n, A->kval[q], Sinv + (long) A->kidx[q] * n, 1, aux, 1);
            return A->sign * quad_form(B, aux, aux + n);
        }
        if (B->type == T_SPR1) {                                          /* :1903-1922 */
            double s = 0.0;
            for (int p = 0; p < A->k; ++p) for (int q = 0; q < B->k; ++q) s += A->kval[p] * B->kval[q] * FE(Sinv, n, A->kidx[p], B->kidx[q]);
            return s * s * A->sign * B->sign;
        }
        memset(aux, 0, sizeof(double) * n);                               /* SPR1 x DSR1 :1924-1944 */
        for (int q = 0; q < A->k; ++q) vaxpy(n, A->kval[q], Sinv + (long) A->kidx[q] * n, 1, aux, 1);
        double d = vdot(n, aux, 1, B->fac, 1);
        return A->sign * B->sign * d * d;
    }
    /* A is DSR1 (:1946-1963, :2029-2058) */
    if (B->type == T_SPR1) return pair_trace(B, A, Sinv, aux, ok);
    full_symv(n, Sinv, A->fac, aux);
    if (B->type == T_DSR1) { double d = vdot(n, B->fac, 1, aux, 1); return A->sign * B->sign * d * d; }
    return A->sign * quad_form(B, aux, aux + n);
}
static double sinv_a_dot_sinv(const coef *c, const double *Sinv, double *aux) {   /* :2063-2165 */
    const int n = c->n;
    double s = 0.0;
    switch (c->type) {
        case T_SPARSE:
            for (int d = 0; d < n; ++d) { const double *col = Sinv + (long) d * n;
                for (int k = 0; k < c->nt; ++k) s += (c->ri[k] == c->ci[k] ? 0.5 : 1.0) * c->x[k] * col[c->ri[k]] * col[c->ci[k]]; }
            return 2.0 * s;
        case T_DENSE:
            for (int r = 0; r < n; ++r) {
                for (int q = 0; q < r; ++q) s += c->pk[PK(n, r, q)] * vdot(n, Sinv + (long) r * n, 1, Sinv + (long) q * n, 1);
                s += 0.5 * c->pk[PK(n, r, r)] * vdot(n, Sinv + (long) r * n, 1, Sinv + (long) r * n, 1);
            }
            return 2.0 * s;
        case T_SPR1:
            for (int d = 0; d < n; ++d) { const double *col = Sinv + (long) d * n;
                for (int p = 0; p < c->k; ++p) {
                    for (int q = 0; q < p; ++q) s += c->kval[p] * c->kval[q] * col[c->kidx[p]] * col[c->kidx[q]];
                    s += 0.5 * c->kval[p] * c->kval[p] * col[c->kidx[p]] * col[c->kidx[p]];
                } }
            return 2.0 * s * c->sign;
        case T_DSR1: r1_solve(c, Sinv, aux); return c->sign * vdot(n, aux, 1, aux, 1);
        default: return 0.0;
    }
}

/* ---------------------------------------------------------------- column builders (hdsdp_conic_sdp.c:687-985) */
typedef struct { const orc_block *b; const double *Sinv; double Rd; int hsd; double *M, *asinv, *asinvrd, *asinvc;
                 double *buf, *buf2; } kctx;

static void put_M(kctx *k, int prow, int pcol, double v) {   /* writes M[max, min] (:764-775) */
    const int m = k->b->m;
    if (prow >= pcol) FE(k->M, m, prow, pcol) += v; else FE(k->M, m, pcol, prow) += v;
}

static int column_m2(kctx *k, int pos) {
    const orc_block *b = k->b; const int n = b->n, i = b->perm[pos];
    const coef *A = &b->rows[i];
    if (A->rank != 1) return 1;
    double *v = k->buf, *aux = k->buf + n;
    r1_solve(A, k->Sinv, v);
    k->asinv[i] += r1_trace_asinv(A, v);
    if (k->Rd != 0.0) k->asinvrd[i] += A->sign * k->Rd * vdot(n, v, 1, v, 1);
    if (k->hsd) k->asinvc[i] += A->sign * quad_form(&b->obj, v, aux);
    for (int r = pos; r < b->m; ++r) { int j = b->perm[r]; put_M(k, j, i, A->sign * quad_form(&b->rows[j], v, aux)); }
    return 0;
}
static int column_m3(kctx *k, int pos) {
    const orc_block *b = k->b; const int n = b->n, i = b->perm[pos];
    double *B = k->buf, *aux = k->buf2;
    k->asinv[i] += sinv_a_sinv(&b->rows[i], k->Sinv, aux, B);
    if (k->Rd != 0.0) { double t = 0.0; for (int d = 0; d < n; ++d) t += FE(B, n, d, d); k->asinvrd[i] += t * k->Rd; }
    if (k->hsd) k->asinvc[i] += a_dot_b(&b->obj, B, aux);
    for (int r = pos; r < b->m; ++r) { int j = b->perm[r]; put_M(k, j, i, a_dot_b(&b->rows[j], B, aux)); }
    return 0;
}
static int column_m4(kctx *k, int pos) {
    const orc_block *b = k->b; const int n = b->n, i = b->perm[pos];
    double *B = k->buf, *aux = k->buf2;
    k->asinvrd[i] += k->Rd * a_sinv(&b->rows[i], k->Sinv, aux, k->Rd, B);
    double t = 0.0; for (int d = 0; d < n; ++d) t += FE(B, n, d, d);
    k->asinv[i] += t;
    if (k->hsd) k->asinvc[i] += a_dot_sinv_b(&b->obj, k->Sinv, B, aux);
    for (int r = pos; r < b->m; ++r) { int j = b->perm[r]; put_M(k, j, i, a_dot_sinv_b(&b->rows[j], k->Sinv, B, aux)); }
    return 0;
}
static int column_m5(kctx *k, int pos) {
    const orc_block *b = k->b; const int i = b->perm[pos];
    const coef *A = &b->rows[i];
    double *aux = k->buf2; int ok = 1;
    k->asinv[i] += a_dot_b(A, k->Sinv, aux);                       /* :958 reuses the M3 dot */
    if (k->Rd != 0.0) k->asinvrd[i] += sinv_a_dot_sinv(A, k->Sinv, aux) * k->Rd;
    if (k->hsd) { k->asinvc[i] += pair_trace(A, &b->obj, k->Sinv, aux, &ok); if (!ok) return 1; }
    for (int r = pos; r < b->m; ++r) { int j = b->perm[r];
        double v = pair_trace(A, &b->rows[j], k->Sinv, aux, &ok); if (!ok) return 1; put_M(k, j, i, v); }
    return 0;
}

int orc_kkt_build(const orc_block *b, const double *Sinv, double Rd, int typeKKT, int fixedStrategy, double *M,
                  double *asinv, double *asinvrd, double *asinvc, double *scal) {
    const int n = b->n, m = b->m;
    kctx k = { b, Sinv, Rd, typeKKT == 2, M, asinv, asinvrd, asinvc, NULL, NULL };
    int rc = 0;
    k.buf = malloc(sizeof(double) * ((size_t) n * n + 2 * n)); k.buf2 = malloc(sizeof(double) * ((size_t) n * n + 2 * n));
    /* HKKTClean (hdsdp_schur.c:141-165) */
    memset(asinv, 0, sizeof(double) * m); memset(asinvrd, 0, sizeof(double) * m);
    if (typeKKT == 2) { memset(asinvc, 0, sizeof(double) * m); scal[0] = scal[1] = scal[2] = 0.0; }
    scal[3] = 0.0;
    if (typeKKT != 1) memset(M, 0, sizeof(double) * (size_t) m * m);
    if (typeKKT == 1) {  /* corrector: hdsdp_conic_sdp.c:1035-1056 */
        for (int i = 0; i < m; ++i) asinv[i] += a_dot_b(&b->rows[i], Sinv, k.buf);
        if (Rd != 0.0) for (int i = 0; i < m; ++i) asinvrd[i] += Rd * sinv_a_dot_sinv(&b->rows[i], Sinv, k.buf);
        goto done;
    }
    if (Rd != 0.0) for (int d = 0; d < n; ++d) scal[3] += FE(Sinv, n, d, d);      /* :1764-1768 */
    for (int pos = 0; pos < m && !rc; ++pos) {                                     /* :1770-1804 */
        if (b->rows[b->perm[pos]].type == T_ZERO) continue;
        int s = fixedStrategy >= 0 ? fixedStrategy : b->strategy[pos];
        if (typeKKT == 3 && fixedStrategy < 0 && s == S_M2) s = S_M5;   /* "primal method cannot use KKT 2", :1782-1788 */
        switch (s) {
            case S_M2: rc = column_m2(&k, pos); break;
            case S_M3: rc = column_m3(&k, pos); break;
            case S_M4: rc = column_m4(&k, pos); break;
            case S_M5: rc = column_m5(&k, pos); break;
            default: rc = 1;
        }
    }
    if (!rc && typeKKT == 2 && b->obj.type != T_ZERO) {   /* HSD scalars: hdsdp_conic_sdp.c:987-1033 */
        const coef *C = &b->obj; int ok = 1;
        if (C->type == T_SPR1) {                          /* (:1005 tests SPR1 twice: DSR1/sparse C take the M3 branch) */
            scal[1] += pair_trace(C, C, Sinv, k.buf, &ok);
            scal[0] += a_dot_b(C, Sinv, k.buf);
            if (Rd != 0.0) scal[2] += Rd * sinv_a_dot_sinv(C, Sinv, k.buf);
        } else {
            scal[0] += sinv_a_sinv(C, Sinv, k.buf2, k.buf);
            scal[1] += a_dot_b(C, k.buf, k.buf2);
            if (Rd != 0.0) { double t = 0.0; for (int d = 0; d < n; ++d) t += FE(k.buf, n, d, d); scal[2] += t * Rd; }
        }
    }
done:
    free(k.buf); free(k.buf2);
    return rc;
}

/* ---------------------------------------------------------------- PCG (hdsdp_linsolver.c:1405-1588) */
static void lower_symv(int m, const double *M, const double *x, double *y) {
    for (int i = 0; i < m; ++i) y[i] = 0.0;
    for (int j = 0; j < m; ++j) {
        y[j] += FE(M, m, j, j) * x[j];
        for (int i = j + 1; i < m; ++i) { double a = FE(M, m, i, j); y[i] += a * x[j]; y[j] += a * x[i]; }
    }
}
static void chol_solve(int m, const double *L, double *x) {
    for (int i = 0; i < m; ++i) { x[i] = (x[i] - vdot(i, L + i, m, x, 1)) / FE(L, m, i, i); }
    for (int i = m - 1; i >= 0; --i) { x[i] = (x[i] - vdot(m - 1 - i, L + (long) i * m + i + 1, 1, x + i + 1, 1)) / FE(L, m, i, i); }
}
static int pcg(int m, const double *M, const double *rhs, double *x, double relTol, double absTol, int maxIter,
               int jacobi, const double *L, int *status) {
    double *r = malloc(sizeof(double) * 5 * (size_t) m), *rn = r + m, *d = rn + m, *Md = d + m, *z = Md + m;
    int freq = 20, it = 0;
    *status = 0;
    memset(x, 0, sizeof(double) * m);
    memcpy(r, rhs, sizeof(double) * m);
    double rhsn = sqrt(vdot(m, rhs, 1, rhs, 1)), resn = rhsn;
    double tol = fmin(absTol, rhsn * relTol);
    tol = fmax(tol, 0.1 * absTol);
    if (resn < tol) { free(r); return 0; }
#define PRECOND(v) do { if (jacobi) { for (int q_ = 0; q_ < m; ++q_) (v)[q_] /= FE(M, m, q_, q_); } else chol_solve(m, L, (v)); } while (0)
    memcpy(d, r, sizeof(double) * m); PRECOND(d);
    memcpy(z, d, sizeof(double) * m);
    lower_symv(m, M, d, Md);
    for (it = 0; it < maxIter; ++it) {
        double rz = vdot(m, z, 1, r, 1), dMd = vdot(m, d, 1, Md, 1), alpha = rz / dMd;
        vaxpy(m, alpha, d, 1, x, 1);
        if (it % freq == 5 && jacobi) {                     /* restart (:1509-1523) */
            memcpy(r, rhs, sizeof(double) * m);
            lower_symv(m, M, x, d);
            vaxpy(m, -1.0, d, 1, r, 1);
            memcpy(d, r, sizeof(double) * m); PRECOND(d);
            lower_symv(m, M, d, Md);
            memcpy(z, r, sizeof(double) * m); PRECOND(z);
            continue;
        }
        memcpy(rn, r, sizeof(double) * m);
        vaxpy(m, -alpha, Md, 1, rn, 1);
        memcpy(z, rn, sizeof(double) * m); PRECOND(z);
        double beta = vdot(m, rn, 1, z, 1) / rz;
        for (int q = 0; q < m; ++q) d[q] = z[q] + beta * d[q];
        lower_symv(m, M, d, Md);
        memcpy(r, rn, sizeof(double) * m);
        resn = sqrt(vdot(m, r, 1, r, 1));
        if (resn != resn) { *status = 2; break; }
        if (it > 20 && resn > 0.01 * rhsn) { *status = 1; break; }
        if (resn < tol) break;
    }
    if (it >= maxIter) *status = 1;
#undef PRECOND
    free(r);
    return it;
}
int orc_pcg_solve(int m, const double *M, const double *rhs, double *x, double relTol, double absTol, int maxIter) {
    int status = 0;
    if (maxIter <= 0) maxIter = (m > 50 ? m : 50);        /* hdsdp_linsolver.c:1340-1345 default budget */
    int it = pcg(m, M, rhs, x, relTol, absTol, maxIter, 1, NULL, &status);
    if (status == 2) return -1;
    if (status == 1) {                                     /* escalate to the Cholesky preconditioner (:1558-1567) */
        double *L = malloc(sizeof(double) * (size_t) m * m);
        memcpy(L, M, sizeof(double) * (size_t) m * m);
        if (orc_potrf(m, L)) { free(L); return -1; }
        it = pcg(m, M, rhs, x, relTol, absTol, maxIter, 0, L, &status);
        free(L);
        if (status) return -1;
    }
    return it;
}

/* ---------------------------------------------------------------- symmetric-indefinite fallback
 * The reference switches the Schur system to LAPACK dsytrf / dsytrs (hdsdp_linsolver.c:1706-1780) when its PCG gives
 * up.  LAPACK is a third-party dependency that is not in the reference tree (any conforming build; MKL in this image);
 * below is its published algorithm, the unblocked Bunch-Kaufman LDL^T with partial pivoting of dsytf2 (lower storage,
 * alpha = (1+sqrt(17))/8) and the matching dsytrs substitution.  ipiv: >= 0 one-by-one pivot exchanged with that row;
 * < 0: two-by-two pivot in (k, k+1), row k+1 exchanged with -ipiv-1.  Returns 0, or j+1 for an exactly zero pivot. */
int orc_sytrf(int n, double *A, int *ipiv) {
    const double alpha = (1.0 + sqrt(17.0)) / 8.0;
    int info = 0, k = 0;
    while (k < n) {
        int kstep = 1, kp = k, imax = k;
        double absakk = fabs(FE(A, n, k, k)), colmax = 0.0;
        for (int i = k + 1; i < n; ++i) if (fabs(FE(A, n, i, k)) > colmax) { colmax = fabs(FE(A, n, i, k)); imax = i; }
        if (fmax(absakk, colmax) == 0.0) {
            if (!info) info = k + 1;
        } else {
            if (absakk >= alpha * colmax) kp = k;
            else {
                double rowmax = 0.0;
                for (int j = k; j < imax; ++j) rowmax = fmax(rowmax, fabs(FE(A, n, imax, j)));
                for (int i = imax + 1; i < n; ++i) rowmax = fmax(rowmax, fabs(FE(A, n, i, imax)));
                if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
                else if (fabs(FE(A, n, imax, imax)) >= alpha * rowmax) kp = imax;
                else { kp = imax; kstep = 2; }
            }
            const int kk = k + kstep - 1;
            if (kp != kk) {      /* symmetric exchange of rows/columns kk and kp inside the trailing block */
                double t;
                for (int i = kp + 1; i < n; ++i) { t = FE(A, n, i, kk); FE(A, n, i, kk) = FE(A, n, i, kp); FE(A, n, i, kp) = t; }
                for (int i = kk + 1; i < kp; ++i) { t = FE(A, n, i, kk); FE(A, n, i, kk) = FE(A, n, kp, i); FE(A, n, kp, i) = t; }
                t = FE(A, n, kk, kk); FE(A, n, kk, kk) = FE(A, n, kp, kp); FE(A, n, kp, kp) = t;
                if (kstep == 2) { t = FE(A, n, k + 1, k); FE(A, n, k + 1, k) = FE(A, n, kp, k); FE(A, n, kp, k) = t; }
            }
            if (kstep == 1) {
                const double d = 1.0 / FE(A, n, k, k);
                for (int j = k + 1; j < n; ++j) {
                    const double w = -d * FE(A, n, j, k);
                    if (w != 0.0) for (int i = j; i < n; ++i) FE(A, n, i, j) += FE(A, n, i, k) * w;
                }
                for (int i = k + 1; i < n; ++i) FE(A, n, i, k) *= d;
            } else if (k < n - 2) {
                double d21 = FE(A, n, k + 1, k);
                const double d11 = FE(A, n, k + 1, k + 1) / d21, d22 = FE(A, n, k, k) / d21;
                const double t = 1.0 / (d11 * d22 - 1.0);
                d21 = t / d21;
                for (int j = k + 2; j < n; ++j) {
                    const double wk = d21 * (d11 * FE(A, n, j, k) - FE(A, n, j, k + 1));
                    const double wk1 = d21 * (d22 * FE(A, n, j, k + 1) - FE(A, n, j, k));
                    for (int i = j; i < n; ++i) FE(A, n, i, j) -= FE(A, n, i, k) * wk + FE(A, n, i, k + 1) * wk1;
                    FE(A, n, j, k) = wk;
                    FE(A, n, j, k + 1) = wk1;
                }
            }
        }
        if (kstep == 1) ipiv[k] = kp;
        else ipiv[k] = ipiv[k + 1] = -kp - 1;
        k += kstep;
    }
    return info;
}
void orc_sytrs(int n, const double *A, const int *ipiv, double *b) {
    double t;
    int k = 0;
    while (k < n) {                       /* b <- D^-1 L^-1 P^T b */
        if (ipiv[k] >= 0) {
            const int kp = ipiv[k];
            if (kp != k) { t = b[k]; b[k] = b[kp]; b[kp] = t; }
            for (int i = k + 1; i < n; ++i) b[i] -= FE(A, n, i, k) * b[k];
            b[k] /= FE(A, n, k, k);
            k += 1;
        } else {
            const int kp = -ipiv[k] - 1;
            if (kp != k + 1) { t = b[k + 1]; b[k + 1] = b[kp]; b[kp] = t; }
            for (int i = k + 2; i < n; ++i) b[i] -= FE(A, n, i, k) * b[k] + FE(A, n, i, k + 1) * b[k + 1];
            const double akm1k = FE(A, n, k + 1, k), akm1 = FE(A, n, k, k) / akm1k, ak = FE(A, n, k + 1, k + 1) / akm1k;
            const double denom = akm1 * ak - 1.0, bkm1 = b[k] / akm1k, bk = b[k + 1] / akm1k;
            b[k] = (ak * bkm1 - bk) / denom;
            b[k + 1] = (akm1 * bk - bkm1) / denom;
            k += 2;
        }
    }
    k = n - 1;
    while (k >= 0) {                      /* b <- P L^-T b */
        if (ipiv[k] >= 0) {
            for (int i = k + 1; i < n; ++i) b[k] -= FE(A, n, i, k) * b[i];
            const int kp = ipiv[k];
            if (kp != k) { t = b[k]; b[k] = b[kp]; b[kp] = t; }
            k -= 1;
        } else {
            for (int i = k + 1; i < n; ++i) { b[k] -= FE(A, n, i, k) * b[i]; b[k - 1] -= FE(A, n, i, k - 1) * b[i]; }
            const int kp = -ipiv[k] - 1;
            if (kp != k) { t = b[k]; b[k] = b[kp]; b[kp] = t; }
            k -= 2;
        }
    }
}
/* HFpLinsysNumeric + HFpLinsysSolve on the Schur system object (hdsdp_linsolver.c:2029-2044, 2085-2110): *linType is
 * the object's solver class, 5 = DENSE_ITERATIVE (PCG) or 6 = DENSE_INDEFINITE; a failed PCG (or a NaN result)
 * switches the object to 6 for good and solves again with LDL^T.  Returns 0, or 1 for a failed solve. */
int orc_schur_solve(int m, const double *M, const double *rhs, double *x, double relTol, double absTol, int maxIter,
                    int *linType) {
    if (*linType == 5) {
        int it = orc_pcg_solve(m, M, rhs, x, relTol, absTol, maxIter);
        if (it >= 0 && x[0] == x[0] && rhs[0] == rhs[0]) return 0;
        *linType = 6;
    }
    double *F = malloc(sizeof(double) * (size_t) m * m);
    int *ipiv = malloc(sizeof(int) * (size_t) m);
    memcpy(F, M, sizeof(double) * (size_t) m * m);
    int info = orc_sytrf(m, F, ipiv);
    if (!info) { memcpy(x, rhs, sizeof(double) * m); orc_sytrs(m, F, ipiv, x); }
    free(F); free(ipiv);
    return (info || x[0] != x[0]) ? 1 : 0;
}

/* ---------------------------------------------------------------- SURVEY.md 8(d) generator */
static uint64_t g_state;
static double draw(void) {
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return 2.0 * ((double) (z >> 11) / 9007199254740992.0) - 1.0;
}
void orc_synth_csc(int n, int m, int **pbeg, int **pidx, double **pval, double **pb) {
    const long P = (long) n * (n + 1) / 2;
    g_state = 0x9E3779B97F4A7C15ULL;
    double *A = calloc((size_t) P * m, sizeof(double)), *C = calloc(P, sizeof(double)), *b = calloc(m, sizeof(double));
    char *keep = calloc((size_t) P * m, 1);
    for (int c = 0; c < m; ++c) { long k = 0;
        for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i, ++k) {
            double v = draw(); int kp = draw() >= 0.2;
            if (i == j || kp) { A[(size_t) c * P + k] = v; keep[(size_t) c * P + k] = 1; }
        } }
    for (int j = 0; j < n; ++j) C[PK(n, j, j)] = 1.0;
    for (int c = 0; c < m; ++c) { double y0 = draw(); for (long k = 0; k < P; ++k) C[k] += y0 * A[(size_t) c * P + k]; }
    long tot = P; for (size_t q = 0; q < (size_t) P * m; ++q) tot += keep[q];
    int *beg = calloc(m + 2, sizeof(int)), *idx = malloc(sizeof(int) * tot); double *val = malloc(sizeof(double) * tot);
    long pos = 0;
    for (long k = 0; k < P; ++k) { idx[pos] = (int) k; val[pos++] = C[k]; }
    beg[1] = (int) pos;
    for (int c = 0; c < m; ++c) {
        for (long k = 0; k < P; ++k) if (keep[(size_t) c * P + k]) { idx[pos] = (int) k; val[pos++] = A[(size_t) c * P + k]; }
        beg[c + 2] = (int) pos;
        for (int j = 0; j < n; ++j) b[c] += A[(size_t) c * P + PK(n, j, j)];
    }
    free(A); free(C); free(keep);
    *pbeg = beg; *pidx = idx; *pval = val; *pb = b;
}
void orc_free_csc(int *beg, int *idx, double *val, double *b) { free(beg); free(idx); free(val); free(b); }

/* The same family WITHOUT the CSC (the int32 CSC cannot hold n = m = 2000): draw number t (0-based) of the one splitmix64
 * stream is a pure function of t, so single matrices and the objective can be generated on their own.  Used by the
 * full-size parity tests (tests/test_gpu_parity.py) to form host-side fp64 answers at the BASELINE sizes. */
static double draw_at(uint64_t t) {
    uint64_t z = 0x9E3779B97F4A7C15ULL * (t + 2);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return 2.0 * ((double) (z >> 11) / 9007199254740992.0) - 1.0;
}
/* constraint matrix c (0-based) as a full symmetric n x n matrix */
void orc_synth_matrix(int n, int c, double *A) {
    const uint64_t P = (uint64_t) n * (n + 1) / 2;
    uint64_t t = 2 * (uint64_t) c * P;
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i, t += 2) {
            const double v = draw_at(t);
            const int kp = draw_at(t + 1) >= 0.2;
            const double a = (i == j || kp) ? v : 0.0;
            A[(size_t) j * n + i] = a; A[(size_t) i * n + j] = a;
        }
}
/* y0 (m multipliers) and the objective C = I + sum_c y0_c A_c as a full symmetric matrix; `threads` host threads */
void orc_synth_objective(int n, int m, double *y0, double *C, int threads) {
    const uint64_t P = (uint64_t) n * (n + 1) / 2;
    for (int c = 0; c < m; ++c) y0[c] = draw_at(2 * (uint64_t) m * P + (uint64_t) c);
    if (threads < 1) threads = 1;
    /* every entry sums its m terms in constraint order starting from the identity's, exactly like orc_synth_csc, whatever
     * the thread count: bit-reproducible */
    #pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
    for (int j = 0; j < n; ++j) {
        long k = PK(n, j, j);
        for (int i = j; i < n; ++i, ++k) {
            double acc = (i == j) ? 1.0 : 0.0;
            for (int c = 0; c < m; ++c) {
                const uint64_t t = 2 * ((uint64_t) c * P + (uint64_t) k);
                const double v = draw_at(t);
                if (i == j || draw_at(t + 1) >= 0.2) acc += y0[c] * v;
            }
            C[(size_t) j * n + i] = acc; C[(size_t) i * n + j] = acc;
        }
    }
}
/* ---------------------------------------------------------------------------------------------------------------
 * Ratio test: largest alpha with S + alpha dS >= 0 (sdpDenseConeRatioTestImpl hdsdp_conic_sdp.c:1640-1686,
 * operator sdpDenseConeILanczosMultiply :462-505, solver HLanczosSolve linalg/hdsdp_lanczos.c:161-292).
 * Same recurrence, same libc-seeded start vector (:33-53), same Ritz check cadence and acceptance rule; the small
 * dsyevr call (:230) is a cyclic Jacobi here.  `state` carries the warm start between calls like hdsdp_lanczos.
 * ------------------------------------------------------------------------------------------------------------- */
struct orc_lanczos_s { int n, nComputed; double *warm; };

orc_lanczos *orc_lanczos_create(int n) {
    orc_lanczos *st = calloc(1, sizeof(*st));
    st->n = n; st->warm = calloc(n, sizeof(double));
    return st;
}
void orc_lanczos_free(orc_lanczos *st) { if (st) { free(st->warm); free(st); } }

static void lz_rand_vec(int n, double *v, double scale, int add) {   /* HLanczosIPrepare / HLanczosIPerturb */
    srand((unsigned int) n);
    for (int i = 0; i < n; ++i) {
        srand((unsigned int) rand());
        double a = sqrt(sqrt((double) (rand() % 1627)));
        double p = a * (rand() % 2 - 0.5);
        v[i] = add ? v[i] + scale * p : scale * p;
    }
}

static void lz_apply(int n, const double *L, const double *dS, const double *in, double *out, double *t) {
    /* out = L^-1 ( -dS ( L^-T in ) ), L lower column-major, dS lower-valid column-major */
    for (int i = n - 1; i >= 0; --i) {               /* backward solve L^T x = in */
        double s = in[i];
        for (int k = i + 1; k < n; ++k) s -= L[(size_t) i * n + k] * out[k];
        out[i] = s / L[(size_t) i * n + i];
    }
    for (int i = 0; i < n; ++i) t[i] = 0.0;
    for (int j = 0; j < n; ++j) {                     /* t = -dS x with the lower triangle only */
        t[j] -= dS[(size_t) j * n + j] * out[j];
        for (int i = j + 1; i < n; ++i) {
            const double a = dS[(size_t) j * n + i];
            t[i] -= a * out[j];
            t[j] -= a * out[i];
        }
    }
    for (int i = 0; i < n; ++i) {                     /* forward solve L z = t */
        double s = t[i];
        for (int k = 0; k < i; ++k) s -= L[(size_t) k * n + i] * out[k];
        out[i] = s / L[(size_t) i * n + i];
    }
}

static void lz_jacobi(int k, double *A, double *d, double *Y) {  /* A k x k column-major symmetric; ascending d */
    for (int i = 0; i < k * k; ++i) Y[i] = 0.0;
    for (int i = 0; i < k; ++i) Y[i * k + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < k; ++p) for (int q = p + 1; q < k; ++q) off += A[q * k + p] * A[q * k + p];
        if (off < 1e-300) break;
        for (int p = 0; p < k; ++p) for (int q = p + 1; q < k; ++q) {
            const double apq = A[q * k + p];
            if (fabs(apq) < 1e-300) continue;
            const double th = (A[q * k + q] - A[p * k + p]) / (2.0 * apq);
            const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s_ = t * c;
            for (int r = 0; r < k; ++r) { double x = A[p * k + r], y = A[q * k + r]; A[p * k + r] = c * x - s_ * y; A[q * k + r] = s_ * x + c * y; }
            for (int r = 0; r < k; ++r) { double x = A[r * k + p], y = A[r * k + q]; A[r * k + p] = c * x - s_ * y; A[r * k + q] = s_ * x + c * y; }
            for (int r = 0; r < k; ++r) { double x = Y[p * k + r], y = Y[q * k + r]; Y[p * k + r] = c * x - s_ * y; Y[q * k + r] = s_ * x + c * y; }
        }
    }
    for (int i = 0; i < k; ++i) d[i] = A[i * k + i];
    for (int i = 0; i < k; ++i) {
        int mn = i;
        for (int j = i + 1; j < k; ++j) if (d[j] < d[mn]) mn = j;
        if (mn != i) {
            double td = d[i]; d[i] = d[mn]; d[mn] = td;
            for (int r = 0; r < k; ++r) { double ty = Y[i * k + r]; Y[i * k + r] = Y[mn * k + r]; Y[mn * k + r] = ty; }
        }
    }
}

/* returns 0 and the step (HUGE_VAL if unbounded), or 1 on the reference's failure exit */
int orc_ratio_test(const orc_block *b, const double *L, double dTauStep, const double *dy, double dEyeCoef,
                   orc_lanczos *st, double *maxStep) {
    const int n = b->n, md = 30, nh = md + 1;
    double *dS = calloc((size_t) n * n, sizeof(double));
    orc_assemble_S(b, dTauStep, dy, -dEyeCoef, dS);       /* dS = dTau*C - sum dy_i A_i + dEyeCoef*I */
    double *V = calloc((size_t) n * (md + 1), sizeof(double)), *H = calloc((size_t) nh * nh, sizeof(double));
    double *v = calloc(n, sizeof(double)), *w = calloc(n, sizeof(double)), *t = calloc(n, sizeof(double));
    double *z1 = calloc(n, sizeof(double)), *z2 = calloc(n, sizeof(double));
    double *U = calloc((size_t) md * md, sizeof(double)), *Y = calloc((size_t) md * md, sizeof(double)), *d = calloc(md, sizeof(double));
    int rc = 0;
    if (st->nComputed == 0) lz_rand_vec(n, v, 1.0, 0);
    else { memcpy(v, st->warm, sizeof(double) * n); lz_rand_vec(n, v, 1e-03, 1); }
    double nv = 0.0; for (int i = 0; i < n; ++i) nv += v[i] * v[i];
    nv = sqrt(nv); for (int i = 0; i < n; ++i) v[i] /= nv;
    memcpy(V, v, sizeof(double) * n);
    int freq = md / 5; if (freq > 3) freq = 3;
    double step = 0.0;
#define HH(i, j) H[(size_t) (j) * nh + (i)]
    for (int k = 0; k < md; ++k) {
        lz_apply(n, L, dS, v, w, t);
        if (k > 0) { const double h = HH(k, k - 1); for (int i = 0; i < n; ++i) w[i] -= h * V[(size_t) (k - 1) * n + i]; }
        double dt = 0.0; for (int i = 0; i < n; ++i) dt += w[i] * V[(size_t) k * n + i];
        const double alp = -dt;
        for (int i = 0; i < n; ++i) w[i] += alp * V[(size_t) k * n + i];
        double nrm = 0.0; for (int i = 0; i < n; ++i) nrm += w[i] * w[i];
        nrm = sqrt(nrm);
        HH(k, k) = -alp;
        if (nrm > 0.0) {
            for (int i = 0; i < n; ++i) v[i] = w[i] / nrm;
            memcpy(V + (size_t) (k + 1) * n, v, sizeof(double) * n);
            HH(k + 1, k) = HH(k, k + 1) = nrm;
        }
        if ((k + 1) % freq == 0 || k > md - 1 || nrm == 0.0) {
            const int kp = k + 1;
            if (kp < 2) { rc = 1; break; }                 /* the reference's dsyevr(il = 0) abort */
            for (int j = 0; j < kp; ++j) for (int i = 0; i < kp; ++i) U[j * kp + i] = 0.5 * (HH(i, j) + HH(j, i));
            lz_jacobi(kp, U, d, Y);
            const double e1 = d[kp - 1], e2 = d[kp - 2];
            const double *y1 = Y + (size_t) (kp - 1) * kp, *y2 = Y + (size_t) (kp - 2) * kp;
            const double resi = fabs(HH(kp, k) * y1[k]);
            if (resi < 1e-04 || k >= md - 1) {
                for (int i = 0; i < n; ++i) { double s_ = 0.0; for (int c = 0; c < kp; ++c) s_ += V[(size_t) c * n + i] * y1[c]; z1[i] = s_; }
                lz_apply(n, L, dS, z1, z2, t);
                memcpy(st->warm, z2, sizeof(double) * n);
                double r1 = 0.0; for (int i = 0; i < n; ++i) { double x = z2[i] - e1 * z1[i]; r1 += x * x; }
                r1 = sqrt(r1);
                for (int i = 0; i < n; ++i) { double s_ = 0.0; for (int c = 0; c < kp; ++c) s_ += V[(size_t) c * n + i] * y2[c]; z2[i] = s_; }
                lz_apply(n, L, dS, z2, z1, t);
                double r2 = 0.0; for (int i = 0; i < n; ++i) { double x = z1[i] - e1 * z2[i]; r2 += x * x; }
                r2 = sqrt(r2);
                const double diff = e1 - e2 - r2;
                double gam = diff > 0 ? diff : 1e-16;
                const double sq = r1 * r1 / gam;
                gam = r1 < sq ? r1 : sq;
                if (gam < 1e-03 || gam + e1 <= 0.5) { step = (gam + e1 <= 0.0) ? HUGE_VAL : 1.0 / (gam + e1); break; }
                if (nrm == 0.0) { rc = 1; break; }
                step = 1.0 / (gam + e1);
            }
        }
    }
#undef HH
    if (!rc) { st->nComputed += 1; *maxStep = step; }
    free(dS); free(V); free(H); free(v); free(w); free(t); free(z1); free(z2); free(U); free(Y); free(d);
    return rc;
}


/* ---------------------------------------------------------------------------------------------------------------
 * Primal recovery (sdpDenseConeGetPrimal, hdsdp_conic_sdp.c:2393-2446): S = C - sum y_i A_i (no residual term) = L L^T,
 * dS = sum dy_i A_i;  two forward solves, "+ I" and symmetrise, two backward solves, symmetrise and scale by mu.
 * Returns 1 (X untouched) when S is not positive definite -- the reference prints "Recovery step is infeasible".
 * ------------------------------------------------------------------------------------------------------------- */
static void pr_fsolve(int n, const double *L, double *B) {          /* B <- L^-1 B, all n columns */
    for (int c = 0; c < n; ++c) {
        double *x = B + (size_t) c * n;
        for (int i = 0; i < n; ++i) {
            double s_ = x[i];
            for (int k = 0; k < i; ++k) s_ -= L[(size_t) k * n + i] * x[k];
            x[i] = s_ / L[(size_t) i * n + i];
        }
    }
}
static void pr_bsolve(int n, const double *L, double *B) {          /* B <- L^-T B */
    for (int c = 0; c < n; ++c) {
        double *x = B + (size_t) c * n;
        for (int i = n - 1; i >= 0; --i) {
            double s_ = x[i];
            for (int k = i + 1; k < n; ++k) s_ -= L[(size_t) i * n + k] * x[k];
            x[i] = s_ / L[(size_t) i * n + i];
        }
    }
}
static void pr_transpose(int n, double *A) {
    for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) { double t = A[(size_t) j * n + i]; A[(size_t) j * n + i] = A[(size_t) i * n + j]; A[(size_t) i * n + j] = t; }
}
int orc_get_primal(const orc_block *b, double mu, const double *y, const double *dy, double *X) {
    const int n = b->n, m = b->m;
    double *L = calloc((size_t) n * n, sizeof(double)), *W = calloc((size_t) n * n, sizeof(double));
    double *ndy = malloc(sizeof(double) * (m > 0 ? m : 1));
    int rc = 0;
    orc_assemble_S(b, 1.0, y, 0.0, L);
    if (orc_potrf(n, L) != 0) { rc = 1; goto done; }
    for (int i = 0; i < m; ++i) ndy[i] = -dy[i];
    orc_assemble_S(b, 0.0, ndy, 0.0, W);                              /* lower triangle of dS */
    for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) W[(size_t) i * n + j] = W[(size_t) j * n + i];
    pr_fsolve(n, L, W); pr_transpose(n, W); pr_fsolve(n, L, W);      /* L^-1 dS L^-T */
    for (int j = 0; j < n; ++j) {
        W[(size_t) j * n + j] += 1.0;
        for (int i = j + 1; i < n; ++i) { double t = 0.5 * (W[(size_t) j * n + i] + W[(size_t) i * n + j]); W[(size_t) j * n + i] = W[(size_t) i * n + j] = t; }
    }
    pr_bsolve(n, L, W); pr_transpose(n, W); pr_bsolve(n, L, W);
    for (int j = 0; j < n; ++j) {
        X[(size_t) j * n + j] = mu * W[(size_t) j * n + j];
        for (int i = j + 1; i < n; ++i) { double t = 0.5 * (W[(size_t) j * n + i] + W[(size_t) i * n + j]); X[(size_t) j * n + i] = X[(size_t) i * n + j] = mu * t; }
    }
done:
    free(L); free(W); free(ndy);
    return rc;
}
