// sdpa.cpp -- SDPA sparse-format (.dat-s) reader producing the reference's per-block user-data layout:
// for every SDP block a CSC matrix of shape n(n+1)/2 x (m+1), column 0 = objective, rows = packed lower
// index (interface/def_hdsdp_user_data.h:16-32).  Semantics follow interface/hdsdp_file_io.c:34-381:
// C = -F0 (:248-250), b = the c vector, 1-based indices, entries given in either triangle are mirrored to
// the lower one, |entry| < 1e-12 dropped (:224-230), a trailing negative block size is an LP block,
// `*` / `"` lines are comments, `{}(),'` are separators.  Host only: no device is touched.
#include "../../include/hdsdp_mi355x.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>

struct HMiSDPA_s {
    int m = 0, nblk = 0, nlp = 0;
    std::vector<int> dims;
    std::vector<double> rhs;
    // column pointers are 64-bit: a block's columns may hold more than 2^31 - 1 entries together (the reference's reader
    // and user data cannot: hdsdp_file_io.c / def_hdsdp_user_data.h:22-32 use int); beg32 is made on demand for callers of
    // the reference-shaped accessor and only where the block fits it
    std::vector<std::vector<int64_t>> beg;
    std::vector<std::vector<int>> beg32, idx;
    std::vector<std::vector<double>> val;
    std::vector<int64_t> lpBeg;
    std::vector<int> lpIdx;
    std::vector<double> lpVal;
};

namespace {
struct Trip { int col; int row; double v; };

bool next_data_line(FILE *f, std::string &line) {
    char buf[1 << 16];
    line.clear();
    while (fgets(buf, sizeof(buf), f)) {
        line += buf;
        if (!line.empty() && line.back() != '\n' && !feof(f)) continue;  // long line: keep reading
        return true;
    }
    return !line.empty();
}
void soften(std::string &s) {
    for (char &c : s)
        if (c == '{' || c == '}' || c == '(' || c == ')' || c == ',' || c == '\'') c = ' ';
}
void bucket(int ncol, std::vector<Trip> &t, std::vector<int64_t> &beg, std::vector<int> &idx, std::vector<double> &val) {
    beg.assign((size_t) ncol + 1, 0);
    for (auto &e : t) beg[e.col + 1]++;
    for (int c = 0; c < ncol; ++c) beg[c + 1] += beg[c];
    idx.resize(t.size());
    val.resize(t.size());
    std::vector<int64_t> pos(beg.begin(), beg.end() - 1);
    for (auto &e : t) { idx[pos[e.col]] = e.row; val[pos[e.col]++] = e.v; }  // stable: file order inside a column
}
}  // namespace

extern "C" {

hdsdp_retcode HMiReadSDPA(const char *fname, HMiSDPA **out) {
    if (!fname || !out) return HDSDP_RETCODE_FAILED;
    FILE *f = fopen(fname, "r");
    if (!f) { fprintf(stderr, "[hdsdp_mi355x] cannot open %s\n", fname); return HDSDP_RETCODE_FAILED; }
    HMiSDPA_s *p = new HMiSDPA_s();
    std::string line;
    auto fail = [&](const char *why) { fprintf(stderr, "[hdsdp_mi355x] %s: %s\n", fname, why); fclose(f); delete p; return HDSDP_RETCODE_FAILED; };
    do { if (!next_data_line(f, line)) return fail("empty file"); } while (line[0] == '*' || line[0] == '"');
    if (sscanf(line.c_str(), "%d", &p->m) != 1 || p->m < 1) return fail("bad constraint count");
    if (!next_data_line(f, line)) return fail("missing block count");
    int nb = 0;
    if (sscanf(line.c_str(), "%d", &nb) != 1 || nb < 1) return fail("bad block count");
    // block sizes: may span lines
    std::vector<int> sizes;
    while ((int) sizes.size() < nb) {
        if (!next_data_line(f, line)) return fail("missing block sizes");
        soften(line);
        const char *s = line.c_str(); char *e;
        for (;;) { long v = strtol(s, &e, 10); if (e == s) break; sizes.push_back((int) v); s = e; }
    }
    for (int i = 0; i < nb; ++i) {
        if (sizes[i] > 0) p->dims.push_back(sizes[i]);
        else if (sizes[i] < 0 && i == nb - 1) p->nlp = -sizes[i];
        else return fail("only one diagonal (LP) block is supported and it must come last");  // hdsdp_file_io.c:108-114
    }
    p->nblk = (int) p->dims.size();
    while ((int) p->rhs.size() < p->m) {
        if (!next_data_line(f, line)) return fail("missing objective vector");
        soften(line);
        const char *s = line.c_str(); char *e;
        for (;;) { double v = strtod(s, &e); if (e == s) break; p->rhs.push_back(v); s = e; }
    }
    p->rhs.resize(p->m);
    std::vector<std::vector<Trip>> trips(p->nblk);
    std::vector<Trip> lpt;
    while (next_data_line(f, line)) {
        int mat, blk, i, j; double v;
        if (sscanf(line.c_str(), "%d %d %d %d %lg", &mat, &blk, &i, &j, &v) != 5) {
            bool blank = true;
            for (char c : line) if (!isspace((unsigned char) c)) { blank = false; break; }
            if (blank || line.rfind("BEGIN.COMMENT", 0) == 0 || line[0] == '*') continue;
            return fail("malformed entry line");
        }
        blk -= 1; i -= 1; j -= 1;
        if (std::fabs(v) < 1e-12) continue;
        if (mat < 0 || mat > p->m) return fail("matrix index out of range");
        if (mat == 0) v = -v;
        if (blk == p->nblk && p->nlp > 0) {
            if (i < 0 || i >= p->nlp) return fail("LP index out of range");
            lpt.push_back({mat, i, v});
            continue;
        }
        if (blk < 0 || blk >= p->nblk) return fail("block index out of range");
        const int n = p->dims[blk];
        if (n > 65535) return fail("block dimension above 65535: the packed index would not fit 32 bits");
        if (i < 0 || j < 0 || i >= n || j >= n) return fail("entry index out of range");
        if (i > j) { int t = i; i = j; j = t; }               // (i <= j) -> packed lower (row j, col i)
        const long pk = (long) (2 * n - i - 1) * i / 2 + j;
        trips[blk].push_back({mat, (int) pk, v});
    }
    fclose(f);
    p->beg.resize(p->nblk); p->beg32.resize(p->nblk); p->idx.resize(p->nblk); p->val.resize(p->nblk);
    for (int b = 0; b < p->nblk; ++b) {
        bucket(p->m + 1, trips[b], p->beg[b], p->idx[b], p->val[b]);
        // the reference-style int column pointers of a block that fits them, made here so that the accessors only read
        if (p->beg[b].back() <= 2147483647LL) p->beg32[b].assign(p->beg[b].begin(), p->beg[b].end());
    }
    if (p->nlp > 0) bucket(p->m + 1, lpt, p->lpBeg, p->lpIdx, p->lpVal);
    *out = p;
    return HDSDP_RETCODE_OK;
}

void HMiSDPAGetDims(const HMiSDPA *p, int *nConstrs, int *nBlks, int *nLpCols) {
    if (nConstrs) *nConstrs = p->m;
    if (nBlks) *nBlks = p->nblk;
    if (nLpCols) *nLpCols = p->nlp;
}
hdsdp_retcode HMiSDPAGetBlock64(const HMiSDPA *p, int iBlk, int *dim, const int64_t **beg, const int **idx, const double **val) {
    if (iBlk < 0 || iBlk >= p->nblk) return HDSDP_RETCODE_FAILED;
    if (dim) *dim = p->dims[iBlk];
    if (beg) *beg = p->beg[iBlk].data();
    if (idx) *idx = p->idx[iBlk].data();
    if (val) *val = p->val[iBlk].data();
    return HDSDP_RETCODE_OK;
}
hdsdp_retcode HMiSDPAGetBlock(const HMiSDPA *p, int iBlk, int *dim, const int **beg, const int **idx, const double **val) {
    if (iBlk < 0 || iBlk >= p->nblk) return HDSDP_RETCODE_FAILED;
    if (p->beg[iBlk].back() > 2147483647LL) {
        fprintf(stderr, "[hdsdp_mi355x] block %d holds %lld entries: beyond the reference's int column pointers, use HMiSDPAGetBlock64\n",
                iBlk, (long long) p->beg[iBlk].back());
        return HDSDP_RETCODE_FAILED;
    }
    if (dim) *dim = p->dims[iBlk];
    if (beg) *beg = p->beg32[iBlk].data();
    if (idx) *idx = p->idx[iBlk].data();
    if (val) *val = p->val[iBlk].data();
    return HDSDP_RETCODE_OK;
}
const double *HMiSDPAGetRHS(const HMiSDPA *p) { return p->rhs.data(); }
void HMiSDPAFree(HMiSDPA **pp) {
    if (pp && *pp) { delete *pp; *pp = nullptr; }
}

}  // extern "C"
