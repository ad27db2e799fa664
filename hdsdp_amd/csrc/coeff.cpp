// coeff.cpp -- presolve of one SDP block on the host: classes, rank-one forms, order and plan (specification and reference
// map: coeff.h).  Written against that specification; where the reference's behaviour has corners the goldens pin (which
// storage positions the dense rank-one probe reads, where its quicksort leaves ties, which comparisons are strict) the
// corner is stated where it is implemented.
#include "coeff.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <utility>

// host threads for the passes over a block's entries (presolve here, staging of the upload in engine_create.h):
// HDSDP_MI355X_HOST_THREADS, default min(16, hardware threads)
int mi_host_threads() {
    static const int n = [] {
        if (const char *e = getenv("HDSDP_MI355X_HOST_THREADS")) return std::max(1, atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return (int) std::max(1u, std::min(16u, hw ? hw : 1u));
    }();
    return n;
}
void mi_parallel(int nthreads, const std::function<void(int)> &fn) {
    if (nthreads <= 0) nthreads = mi_host_threads();
    if (nthreads == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(fn, t);
    fn(0);
    for (auto &t : th) t.join();
}

namespace {

constexpr double kRankOneTol = 1e-10;      // 1-norm residual of the rank-one probe, and the cut below which a factor entry is dropped
constexpr double kDenseShare = 0.3;        // more than this share of the P packed positions: DENSE
constexpr double kDenseFactorShare = 0.5;  // more than this share of the n factor entries: DSR1

// the packed lower triangle, column by column: column j holds rows j .. n-1
struct PackedIndex {
    long n;
    long columns_before(long j) const { return j * (2 * n - j + 1) / 2; }   // first position of column j
    long size() const { return n * (n + 1) / 2; }
};

// (row, col) of packed positions given in ascending order: one walk over the columns for the whole list
void rows_and_cols(int n, const std::vector<int> &pos, std::vector<int> &row, std::vector<int> &col) {
    const PackedIndex px{n};
    row.resize(pos.size());
    col.resize(pos.size());
    long j = 0, first = 0, next = n;                 // column j occupies [first, next)
    for (size_t e = 0; e < pos.size(); ++e) {
        while (pos[e] >= next) { ++j; first = next; next += n - j; }
        col[e] = (int) j;
        row[e] = (int) (j + (pos[e] - first));
    }
    (void) px;
}

// Rank-one probe of a SPARSE-class matrix (entries sorted by packed position).  The candidate factor is the column of the
// FIRST stored entry, which therefore has to be a diagonal one; the support must be exactly the k(k+1)/2 lower positions of
// a k x k principal block, k = entries of that column; the factor has to reproduce every stored entry, 1e-10 in the 1-norm.
// Corners kept as the reference has them: a single (diagonal) entry is rank one with an unnormalised factor e_i and the
// whole value as its sign; a candidate column with n entries is refused (such a matrix is not SPARSE-class anyway unless
// n is tiny); the sign of a zero pivot counts as negative.
bool probe_rank_one_sparse(int n, const std::vector<int> &row, const std::vector<int> &col, const std::vector<double> &v,
                           double &sign, std::vector<double> &a) {
    const size_t cnt = v.size();
    if (row[0] != col[0]) return false;
    const int pc = col[0];
    if (cnt == 1) { sign = v[0]; a[pc] = 1.0; return true; }
    const double s = (v[0] > 0) ? 1.0 : -1.0;
    const double scale = std::sqrt(std::fabs(v[0]));
    size_t k = 0;
    while (k < cnt && col[k] <= pc) { a[row[k]] = v[k] / scale; ++k; }
    if (cnt != k * (k + 1) / 2 || k == (size_t) n) return false;
    double resid = 0.0;
    for (size_t e = 0; e < cnt; ++e) resid += std::fabs(v[e] - s * a[row[e]] * a[col[e]]);
    if (resid > kRankOneTol) return false;
    sign = s;
    return true;
}

// Rank-one probe of a DENSE-class matrix in packed storage: the candidate is the column of the first diagonal entry that is
// not exactly zero.  The reference reads the n candidate entries from the n storage positions that END with that column's
// (columns_before(i) - i + r for r = 0 .. n-1): for r >= i that is A(r, i); for r < i it is whatever the tail of the
// columns before holds -- zero for a matrix that really is a a' with a_r = 0 there, and for anything else the residual test
// decides.  The same positions are read here, so that accept / reject agree in every case.  The residual is summed column
// by column and the probe gives up after the first column that takes it past the tolerance.
bool probe_rank_one_dense(int n, const std::vector<double> &A, double &sign, std::vector<double> &a) {
    const PackedIndex px{n};
    int pc = 0;
    while (pc < n && A[px.columns_before(pc)] == 0) ++pc;
    if (pc == n) return false;
    const double d = A[px.columns_before(pc)];
    const double s = (d > 0) ? 1.0 : -1.0, scale = std::sqrt(std::fabs(d));
    const long origin = px.columns_before(pc) - pc;
    for (int r = 0; r < n; ++r) a[r] = A[origin + r] / scale;
    double resid = 0.0;
    long at = 0;
    for (int c = 0; c < n; ++c) {
        const long h = n - c;
        for (long r = 0; r < h; ++r) resid += std::fabs(A[at + r] - s * a[c] * a[c + r]);
        at += h;
        if (resid > kRankOneTol) return false;
    }
    sign = s;
    return true;
}

// The first column of probe_rank_one_dense on the sorted entry list (same positions, same arithmetic, same order of the sum):
// false iff the probe would give up after column 0 -- or finds no non-zero diagonal entry at all.
bool dense_probe_may_pass(int n, const std::vector<int> &idx, const std::vector<double> &val) {
    const PackedIndex px{n};
    auto at = [&](long pos) {   // value at a packed position (0 where nothing is stored)
        auto it = std::lower_bound(idx.begin(), idx.end(), (int) pos);
        return (it != idx.end() && *it == (int) pos) ? val[(size_t) (it - idx.begin())] : 0.0;
    };
    int pc = 0;
    while (pc < n && at(px.columns_before(pc)) == 0) ++pc;
    if (pc == n) return false;
    const double d = at(px.columns_before(pc));
    const double s = (d > 0) ? 1.0 : -1.0, scale = std::sqrt(std::fabs(d));
    const long origin = px.columns_before(pc) - pc;
    // a[0 .. n-1] = A[origin + r] / scale: one contiguous range of packed positions
    std::vector<double> a((size_t) n, 0.0);
    {
        size_t k = (size_t) (std::lower_bound(idx.begin(), idx.end(), (int) origin) - idx.begin());   // (origin >= 0)
        for (; k < idx.size() && idx[k] < origin + n; ++k) a[(size_t) (idx[k] - origin)] = val[k];
        for (int r = 0; r < n; ++r) a[r] /= scale;
    }
    double resid = 0.0;
    size_t k = 0;
    for (long r = 0; r < n; ++r) {          // column 0 occupies packed positions 0 .. n-1
        double v = 0.0;
        if (k < idx.size() && idx[k] == r) v = val[k++];
        resid += std::fabs(v - s * a[0] * a[r]);
    }
    return !(resid > kRankOneTol);
}

// Descending order by key with the reference's tie behaviour: its quicksort takes the first element of a range as the pivot,
// closes in from both ends (from the right past keys <= pivot, from the left past keys >= pivot), exchanges, and finally
// puts the pivot where the ends met; keys and indices move together.  Ties therefore end up in an order of their own, which
// the plan -- and the goldens -- depend on.  Ranges are kept on an explicit stack (a block with 1e5 equal rows must not
// recurse 1e5 deep), and a range whose keys are all equal is left alone: every level of the scheme would return it unchanged.
void order_descending(std::vector<int> &ind, std::vector<int> &key) {
    std::vector<std::pair<int, int>> todo;
    if (!key.empty()) todo.push_back({0, (int) key.size() - 1});
    while (!todo.empty()) {
        const int lo = todo.back().first, hi = todo.back().second;
        todo.pop_back();
        if (lo >= hi) continue;
        bool flat = true;
        for (int t = lo + 1; t <= hi && flat; ++t) flat = (key[t] == key[lo]);
        if (flat) continue;
        const int pivot = key[lo];
        int l = lo, h = hi;
        while (l < h) {
            while (l < h && key[h] <= pivot) --h;
            while (l < h && key[l] >= pivot) ++l;
            if (l < h) { std::swap(key[l], key[h]); std::swap(ind[l], ind[h]); }
        }
        std::swap(key[l], key[lo]);
        std::swap(ind[l], ind[lo]);
        // (the reference finishes the left part first; the two parts are disjoint, so the order of work does not matter)
        todo.push_back({l + 1, hi});
        todo.push_back({lo, l - 1});
    }
}

}  // namespace

int mi_coeff_build(MiCoeff &c, int n, long nnz, const int *idx, const double *val) {
    const PackedIndex px{n};
    const long P = px.size();
    c = MiCoeff();
    c.stored = nnz;
    if (nnz < 0) return 1;                                   // column pointers that run backwards
    if (nnz == 0) return 0;                                  // ZERO
    if (nnz > P) return 1;
    c.idx.assign(idx, idx + nnz);
    c.val.assign(val, val + nnz);
    if (!std::is_sorted(c.idx.begin(), c.idx.end())) {       // entries in file order: bring them to packed order, stably
        std::vector<long> o((size_t) nnz);
        for (long k = 0; k < nnz; ++k) o[k] = k;
        std::stable_sort(o.begin(), o.end(), [&](long x, long y) { return idx[x] < idx[y]; });
        for (long k = 0; k < nnz; ++k) { c.idx[k] = idx[o[k]]; c.val[k] = val[o[k]]; }
    }
    if (c.idx.front() < 0 || c.idx.back() >= P) return 1;
    for (long k = 1; k < nnz; ++k)
        if (c.idx[k] == c.idx[k - 1]) return 1;              // the same packed position twice: not a matrix (the classes would disagree on its value)
    {   // trace: the diagonal positions are the column starts
        long j = 0, first = 0;
        for (long k = 0; k < nnz; ++k) {
            while (j < n && first < c.idx[k]) { first += n - j; ++j; }
            if (j < n && first == c.idx[k]) c.trace += c.val[k];
        }
    }
    std::vector<double> a((size_t) n, 0.0);
    double sgn = 0.0;
    bool one = false;
    if ((double) nnz > kDenseShare * (double) P) {
        c.type = MI_COEFF_DENSE;
        c.nnz = (int) P;
        // The probe reads the whole packed matrix only if it IS rank one; for anything else its residual passes the tolerance in
        // the first column.  That first column is decided on the sorted entries (n positions of the candidate factor + the
        // entries of column 0: two short walks), and the 8 P-byte dense image -- 16 MB at n = 2000, per constraint -- is made
        // only for a matrix that survives it.
        if (dense_probe_may_pass(n, c.idx, c.val)) {
            std::vector<double> full((size_t) P, 0.0);
            for (long k = 0; k < nnz; ++k) full[c.idx[k]] = c.val[k];
            one = probe_rank_one_dense(n, full, sgn, a);
        }
    } else {
        c.type = MI_COEFF_SPARSE;
        c.nnz = (int) nnz;
        std::vector<int> row, col;
        rows_and_cols(n, c.idx, row, col);
        one = probe_rank_one_sparse(n, row, col, c.val, sgn, a);
        if (!one && nnz == n) {
            bool eye = true;
            for (long k = 0; k < nnz && eye; ++k) eye = (row[k] == col[k]) && (c.val[k] == c.val[0]);
            c.is_eye = eye;
            c.eye_val = c.val[0];
        }
    }
    c.rank = n;
    if (!one) return 0;
    // rank one: dense or sparse factor, normalised, the scale folded into the sign
    int kept = 0;
    for (int r = 0; r < n; ++r) kept += (std::fabs(a[r]) > kRankOneTol);
    double norm2 = 0.0;
    if ((double) kept > kDenseFactorShare * (double) n) {
        c.type = MI_COEFF_DSR1;
        c.nnz = (int) P;
        for (int r = 0; r < n; ++r) norm2 += a[r] * a[r];
    } else {
        c.type = MI_COEFF_SPR1;
        c.nnz = (int) ((long) kept * (kept + 1) / 2);
        for (int r = 0; r < n; ++r) {
            if (std::fabs(a[r]) > kRankOneTol) norm2 += a[r] * a[r];
            else a[r] = 0.0;
        }
    }
    const double norm = std::sqrt(norm2);
    c.sign = sgn * norm * norm;
    for (int r = 0; r < n; ++r) a[r] /= norm;
    c.factor.swap(a);
    c.factor_nnz = kept;
    c.rank = 1;
    if (c.type == MI_COEFF_SPR1 && kept == 1 && c.sign == 1.0)
        for (int r = 0; r < n; ++r)
            if (c.factor[r] != 0.0) { if (c.factor[r] == 1.0) c.unit_col = r; break; }
    return 0;
}

void mi_block_plan(MiBlockData &blk) {
    const int m = blk.m, n = blk.n;
    for (int t = 0; t < 5; ++t) blk.counts[t] = 0;
    blk.stored = blk.obj.stored;
    for (int i = 0; i < m; ++i) { blk.counts[blk.rows[i].type] += 1; blk.stored += blk.rows[i].stored; }
    std::vector<int> key((size_t) m);
    blk.perm.resize((size_t) m);
    for (int i = 0; i < m; ++i) { blk.perm[i] = i; key[i] = blk.rows[i].nnz; }
    order_descending(blk.perm, key);
    // Cost model per position p of the order (r = rank of the row there, f = its nnz, z = nnz of this and all later rows,
    // kappa = 1.5): M2 r (f n + 3 kappa z); M3 n kappa f + n^3 + kappa z + n^3/m; M4 n kappa f + kappa (n + 1) z + n^3/m;
    // M5 kappa (2 kappa f + 1) z + n^3/m.  M2 is taken on a tie with "nothing yet", every later candidate has to be strictly
    // cheaper.  z is an exact integer sum (below 2^53), whichever end it is accumulated from.
    std::vector<double> tail((size_t) m + 1, 0.0);
    for (int p = m - 1; p >= 0; --p) tail[p] = tail[p + 1] + (double) key[p];
    blk.strategy.resize((size_t) m);
    const double kappa = 1.5, n3 = (double) n * n * n;
    for (int p = 0; p < m; ++p) {
        const double r = blk.rows[blk.perm[p]].rank, f = key[p], z = tail[p];
        const double cost[4] = {r * (f * n + 3 * kappa * z),
                                (double) n * kappa * f + n3 + kappa * z + n3 / m,
                                (double) n * kappa * f + kappa * (n + 1) * z + n3 / m,
                                kappa * (2.0 * kappa * f + 1) * z + n3 / m};
        int pick = 0;
        double best = INFINITY;
        for (int q = 0; q < 4; ++q)
            if (q == 0 ? cost[q] <= best : cost[q] < best) { best = cost[q]; pick = q + 1; }
        blk.strategy[p] = pick;
    }
}

template <class Off> static int block_from_csc(MiBlockData &blk, int m, int n, const Off *beg, const int *idx, const double *val) {
    blk = MiBlockData();
    blk.n = n;
    blk.m = m;
    blk.rows.assign((size_t) m, MiCoeff());
    if (mi_coeff_build(blk.obj, n, (long) (beg[1] - beg[0]), idx + beg[0], val + beg[0])) return 1;
    // the columns are independent: host threads take them one by one (19 GB of entries at n = m = 2000: copy, order check,
    // trace, class and rank-one probe are all passes over memory)
    std::atomic<int> next{0}, bad{0};
    mi_parallel((long) (beg[m + 1] - beg[1]) >= (1L << 22) ? 0 : 1, [&](int) {
        for (int i = next.fetch_add(1); i < m; i = next.fetch_add(1))
            if (mi_coeff_build(blk.rows[i], n, (long) (beg[i + 2] - beg[i + 1]), idx + beg[i + 1], val + beg[i + 1])) bad.store(1);
    });
    if (bad.load()) return 1;
    mi_block_plan(blk);
    return 0;
}
int mi_block_from_csc(MiBlockData &blk, int m, int n, const int *beg, const int *idx, const double *val) {
    return block_from_csc(blk, m, n, beg, idx, val);
}
int mi_block_from_csc(MiBlockData &blk, int m, int n, const int64_t *beg, const int *idx, const double *val) {
    return block_from_csc(blk, m, n, beg, idx, val);
}
