// coeff.cpp -- presolve of one SDP block on the host (see coeff.h for the reference map).
#include "coeff.h"
#include <algorithm>
#include <cmath>
#include <cstdio>

namespace {

inline long pack_nnz(long n) { return n * (n + 1) / 2; }

// packed index -> (row, col), row >= col   (sparse_opts.c:427-441 walks the columns the same way)
inline void unpack_index(int n, long p, int &i, int &j) {
    j = 0;
    long start = 0;
    while (p >= start + (n - j)) { start += n - j; ++j; }
    i = j + (int) (p - start);
}

// Rank-one test for a triplet matrix sorted by packed index: first entry must be a diagonal (i,i);
// the entries of column i give a = A[:,i]/sqrt|A_ii|; the support must be a full anz x anz lower
// block and reproduce every entry to 1e-10 in the 1-norm (sparse_opts.c:453-516).
bool sparse_rank_one(int n, const std::vector<int> &ri, const std::vector<int> &ci, const std::vector<double> &x,
                     double &sgn, std::vector<double> &a) {
    const int nnz = (int) x.size();
    int i = ri[0], j = ci[0];
    double v = x[0];
    if (i != j) return false;
    if (nnz == 1) { sgn = x[0]; a[i] = 1.0; return true; }
    double s = (v > 0) ? 1.0 : -1.0;
    v = std::sqrt(std::fabs(v));
    int k = 0, anz = 0;
    for (k = 0; k < nnz; ++k) {
        if (ci[k] > i) break;
        a[ri[k]] = x[k] / v;
        anz += 1;
    }
    if (nnz != (int) (anz * (anz + 1) / 2)) return false;
    if (k == n) return false;
    double eps = 0.0;
    for (k = 0; k < nnz; ++k) eps += std::fabs(x[k] - s * a[ri[k]] * a[ci[k]]);
    if (eps > 1e-10) return false;
    sgn = s;
    return true;
}

// Same test on a packed dense matrix (dense_opts.c:233-285)
bool dense_rank_one(int n, const std::vector<double> &A, double &sgn, std::vector<double> &a) {
    int i = 0;
    long k = 0;
    for (i = 0; i < n; ++i) {
        if (A[k] != 0) break;
        k += n - i;
    }
    if (i == n) return false;
    double s = (A[k] > 0) ? 1.0 : -1.0;
    double v = std::sqrt(std::fabs(A[k]));
    const long coli = (long) (2 * n - i - 1) * i / 2;
    for (int r = 0; r < n; ++r) a[r] = A[coli + r] / v;
    double eps = 0.0;
    long id = 0;
    for (int c = 0; c < n; ++c) {
        for (int r = 0; r < n - c; ++r) eps += std::fabs(A[id + r] - s * a[c] * a[c + r]);
        id += n - c;
        if (eps > 1e-10) return false;
    }
    sgn = s;
    return true;
}

void build_coeff(MiCoeff &c, int n, int nnz, const int *idx, const double *val) {
    c.idx.assign(idx, idx + nnz);
    c.val.assign(val, val + nnz);
    if (!std::is_sorted(c.idx.begin(), c.idx.end())) {
        std::vector<int> o(nnz);
        for (int k = 0; k < nnz; ++k) o[k] = k;
        std::stable_sort(o.begin(), o.end(), [&](int x, int y) { return idx[x] < idx[y]; });
        for (int k = 0; k < nnz; ++k) { c.idx[k] = idx[o[k]]; c.val[k] = val[o[k]]; }
    }
    const long P = pack_nnz(n);
    if (nnz == 0) { c.type = MI_COEFF_ZERO; c.nnz = 0; c.rank = 0; return; }
    std::vector<double> a(n, 0.0);
    double sgn = 0.0;
    bool r1 = false;
    if (nnz > 0.3 * P) {
        c.type = MI_COEFF_DENSE;
        std::vector<double> pk(P, 0.0);
        for (int k = 0; k < nnz; ++k) pk[c.idx[k]] = c.val[k];
        r1 = dense_rank_one(n, pk, sgn, a);
        c.nnz = (int) P;
    } else {
        c.type = MI_COEFF_SPARSE;
        std::vector<int> ri(nnz), ci(nnz);
        for (int k = 0; k < nnz; ++k) unpack_index(n, c.idx[k], ri[k], ci[k]);
        r1 = sparse_rank_one(n, ri, ci, c.val, sgn, a);
        c.nnz = nnz;
    }
    c.rank = n;
    if (!r1) return;
    int fn = 0;
    for (int r = 0; r < n; ++r) fn += (std::fabs(a[r]) > 1e-10);
    double nrm = 0.0;
    if (fn > 0.5 * n) {
        c.type = MI_COEFF_DSR1;
        for (int r = 0; r < n; ++r) nrm += a[r] * a[r];
        c.nnz = (int) P;
    } else {
        c.type = MI_COEFF_SPR1;
        for (int r = 0; r < n; ++r) {
            if (std::fabs(a[r]) > 1e-10) nrm += a[r] * a[r]; else a[r] = 0.0;
        }
        c.nnz = (int) pack_nnz(fn);
    }
    nrm = std::sqrt(nrm);
    c.sign = sgn * nrm * nrm;  // scale goes into the sign (hdsdp_sdpdata.c:880-899)
    for (int r = 0; r < n; ++r) a[r] /= nrm;
    c.factor = a;
    c.factor_nnz = fn;
    c.rank = 1;
}

// quicksort partition used by the reference for the descending nnz ordering; ties end up where
// this exact scheme leaves them (hdsdp_utils.c:93-112)
int partition_desc(std::vector<int> &ind, std::vector<int> &val, int l, int h) {
    int tmp = l, p = val[l];
    while (l < h) {
        while (l < h && val[h] <= p) --h;
        while (l < h && val[l] >= p) ++l;
        if (l < h) { std::swap(val[l], val[h]); std::swap(ind[l], ind[h]); }
    }
    std::swap(val[l], val[tmp]);
    std::swap(ind[l], ind[tmp]);
    return l;
}
void sort_desc(std::vector<int> &ind, std::vector<int> &val, int low, int up) {
    if (low < up) {
        int p = partition_desc(ind, val, low, up);
        sort_desc(ind, val, low, p - 1);
        sort_desc(ind, val, p + 1, up);
    }
}

int choose_strategy(const std::vector<int> &ranks, const std::vector<int> &sparsity, const std::vector<int> &perm,
                    int nRow, int nCol, int iPerm) {
    const double kappa = 1.5;  // SPARSE_EFFICIENCY
    int best = 0;
    double bestScore = INFINITY;
    const int rowRank = ranks[perm[iPerm]];
    const double n3 = (double) nCol * nCol * nCol;
    double after = 0.0;
    for (int i = iPerm; i < nRow; ++i) after += sparsity[i];
    const double f = sparsity[iPerm];
    const double s2 = rowRank * (f * nCol + 3 * kappa * after);
    const double s3 = (double) nCol * kappa * f + n3 + kappa * after + n3 / nRow;
    const double s4 = (double) nCol * kappa * f + kappa * (nCol + 1) * after + n3 / nRow;
    const double s5 = kappa * (2.0 * kappa * f + 1) * after + n3 / nRow;
    if (s2 <= bestScore) { best = 1; bestScore = s2; }
    if (s3 < bestScore) { best = 2; bestScore = s3; }
    if (s4 < bestScore) { best = 3; bestScore = s4; }
    if (s5 < bestScore) { best = 4; bestScore = s5; }
    return best;
}

}  // namespace

int mi_block_from_csc(MiBlockData &blk, int m, int n, const int *beg, const int *idx, const double *val) {
    blk.n = n;
    blk.m = m;
    blk.rows.assign(m, MiCoeff());
    for (int t = 0; t < 5; ++t) blk.counts[t] = 0;
    build_coeff(blk.obj, n, beg[1] - beg[0], idx + beg[0], val + beg[0]);
    for (int i = 0; i < m; ++i) {
        build_coeff(blk.rows[i], n, beg[i + 2] - beg[i + 1], idx + beg[i + 1], val + beg[i + 1]);
        blk.counts[blk.rows[i].type] += 1;
    }
    std::vector<int> ranks(m), sparsity(m);
    blk.perm.resize(m);
    blk.strategy.resize(m);
    for (int i = 0; i < m; ++i) { blk.perm[i] = i; ranks[i] = blk.rows[i].rank; sparsity[i] = blk.rows[i].nnz; }
    sort_desc(blk.perm, sparsity, 0, m - 1);
    for (int p = 0; p < m; ++p) blk.strategy[p] = choose_strategy(ranks, sparsity, blk.perm, m, n, p);
    return 0;
}
