// bsparse.hip -- block-sparse (128 x 128 tiles) storage and left-looking level-scheduled LDL' (signed Cholesky) of a sparse Schur matrix.
// See bsparse.h for the design.  Reference counterpart: the aggregated-pattern CSC operator with its sparse direct solver,
// interface/hdsdp_schur.c:46-139 and linalg/hdsdp_linsolver.c:509-809 (QDLDL: elimination tree, symbolic and numeric LDL').
#include "bsparse.h"
#include "chol.h"
#include <algorithm>
#include <cstring>
#include <numeric>

#define BT 128                 // tile edge
#define BTT 16384              // doubles per tile
#define BS_LD 144              // LDS row stride of a 16-deep slab (as the GEMM family's M-major image)

// ---------------------------------------------------------------------------------------------------------------------
// reverse Cuthill-McKee (moved here from the operator's set-up code: the tile store orders its non-dense rows with it too)
// ---------------------------------------------------------------------------------------------------------------------
std::vector<int> hdm_rcm_order(int m, const std::vector<int> &beg, const std::vector<int> &idx) {
    std::vector<int> deg(m, 0);
    for (int c = 0; c < m; ++c)
        for (int q = beg[c]; q < beg[c + 1]; ++q) if (idx[q] != c) { deg[c] += 1; deg[idx[q]] += 1; }
    std::vector<int> ap(m + 1, 0);
    for (int v = 0; v < m; ++v) ap[v + 1] = ap[v] + deg[v];
    std::vector<int> adj((size_t) ap[m]), fill(ap.begin(), ap.end() - 1);
    for (int c = 0; c < m; ++c)
        for (int q = beg[c]; q < beg[c + 1]; ++q) if (idx[q] != c) { adj[fill[c]++] = idx[q]; adj[fill[idx[q]]++] = c; }
    std::vector<int> order; order.reserve(m);
    std::vector<char> seen(m, 0);
    std::vector<int> level(m, -1), queue;
    auto bfs = [&](int root, std::vector<int> &out) {          // breadth-first from root over the unseen part; returns the last level's vertex of minimal degree
        out.clear(); out.push_back(root);
        std::vector<int> touched{root};
        level[root] = 0;
        for (size_t h = 0; h < out.size(); ++h) {
            const int v = out[h];
            const size_t first_child = out.size();
            for (int q = ap[v]; q < ap[v + 1]; ++q) {
                const int w = adj[q];
                if (seen[w] || level[w] >= 0) continue;
                level[w] = level[v] + 1; out.push_back(w); touched.push_back(w);
            }
            std::sort(out.begin() + first_child, out.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
        }
        const int last_level = level[out.back()];
        int best = out.back();
        for (int v : out) if (level[v] == last_level && (deg[v] < deg[best] || (deg[v] == deg[best] && v < best))) best = v;
        for (int v : touched) level[v] = -1;
        return best;
    };
    for (int s0 = 0; s0 < m; ++s0) {
        if (seen[s0]) continue;
        int root = s0;
        for (int sweep = 0; sweep < 3; ++sweep) root = bfs(root, queue);
        bfs(root, queue);
        for (int v : queue) { seen[v] = 1; order.push_back(v); }
    }
    std::vector<int> perm(m);
    for (int k = 0; k < m; ++k) perm[order[k]] = m - 1 - k;     // reversed
    return perm;
}

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
typedef double bs_d4 __attribute__((ext_vector_type(4)));

// One workgroup (4 waves, 64 x 64 quadrants, sixteen 16 x 16 fp64 MFMA accumulators each) per tile operation.
//   MODE 0 (update):  C <- C - sum_s A_s S_j B_s^T   over the target's source list (j = the sources' block column)
//   MODE 1 (panel):   C <- C W_k^T S_k               in place (every load of C has passed the last barrier before the first store)
// S = diag(sgn) are the pivot signs of the LDL' form M = L~ S L~' (bsparse.h); all +1 for a positive definite matrix, where
// this is the Cholesky factorisation, bit for bit.
// Tiles are 128 x 128 column-major: a 16-deep k slab of an operand is 2048 consecutive doubles.
template <int MODE>
__global__ __launch_bounds__(256) void bs_tile_kernel(double *__restrict__ L, const double *__restrict__ Winv, const int *__restrict__ tgt_tile,
                                                      const int *__restrict__ tgt_src_ptr, const int2 *__restrict__ src,
                                                      const int2 *__restrict__ pan, int first, const double *__restrict__ sgn,
                                                      const int *__restrict__ src_col) {
    __shared__ __attribute__((aligned(16))) double sA[16 * BS_LD], sB[16 * BS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int op = first + blockIdx.x;
    int s0 = 0, s1 = 1;
    double *C;
    const double *Wk = nullptr;
    if (MODE == 0) { s0 = tgt_src_ptr[op]; s1 = tgt_src_ptr[op + 1]; C = L + ((long) tgt_tile[op] << 14); }
    else { const int2 p = pan[op]; C = L + ((long) p.x << 14); Wk = Winv + ((long) p.y << 14); }
    bs_d4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (bs_d4){0.0, 0.0, 0.0, 0.0};
    const int sk = tid >> 4, si = (tid & 15) * 8;          // staging: k row of the slab, first of 8 consecutive elements
    for (int s = s0; s < s1; ++s) {
        const double *A, *B, *sg = nullptr;
        if (MODE == 0) { const int2 ab = src[s]; A = L + ((long) ab.x << 14); B = L + ((long) ab.y << 14); sg = sgn + (long) src_col[s] * BT; }
        else { A = C; B = Wk; }
        for (int k0 = 0; k0 < BT; k0 += 16) {
            const double2 *pa = reinterpret_cast<const double2 *>(A + (long) (k0 + sk) * BT + si);
            const double2 *pb = reinterpret_cast<const double2 *>(B + (long) (k0 + sk) * BT + si);
            const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3];
            double2 b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3];
            if (MODE == 0) {                                 // column k of the source tiles carries the sign of pivot k of their block column
                const double g = sg[k0 + sk];
                b0.x *= g; b0.y *= g; b1.x *= g; b1.y *= g; b2.x *= g; b2.y *= g; b3.x *= g; b3.y *= g;
            }
            __syncthreads();                                 // the previous slab has been consumed
            double2 *da = reinterpret_cast<double2 *>(sA + sk * BS_LD + si), *db = reinterpret_cast<double2 *>(sB + sk * BS_LD + si);
            da[0] = a0; da[1] = a1; da[2] = a2; da[3] = a3; db[0] = b0; db[1] = b1; db[2] = b2; db[3] = b3;
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; kk += 4) {
                double fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fb[i] = sA[(kk + lq) * BS_LD + wm * 64 + i * 16 + l15];
#pragma unroll
                for (int j = 0; j < 4; ++j) fa[j] = sB[(kk + lq) * BS_LD + wn * 64 + j * 16 + l15];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[j], fb[i], acc[j][i], 0, 0, 0);
            }
        }
    }
    // lane l, register r of acc[j][i]: C[wm * 64 + i * 16 + l15][wn * 64 + j * 16 + lq + 4 r]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cc = wn * 64 + j * 16 + lq + 4 * r;
                double *q = C + (wm * 64 + i * 16 + l15) + (long) cc * BT;
                if (MODE == 0) *q -= acc[j][i][r];
                else *q = acc[j][i][r] * sgn[(long) pan[op].y * BT + cc];
            }
}

// forward substitution, one level: y_k = W_k (b_k - sum_{j < k} L(k, j) y_j) for the block columns of the level
__global__ __launch_bounds__(256) void bs_fwd_kernel(const double *__restrict__ L, const double *__restrict__ Winv, const int *__restrict__ cols,
                                                     const int *__restrict__ row_ptr, const int *__restrict__ row_col,
                                                     const int *__restrict__ row_tile, double *__restrict__ v, int first) {
    __shared__ double part[2][BT], t[BT];
    const int k = cols[first + blockIdx.x], tid = threadIdx.x, r = tid & 127, h = tid >> 7;
    double acc = 0.0;
    for (int q = row_ptr[k]; q < row_ptr[k + 1]; ++q) {
        const double *T = L + ((long) row_tile[q] << 14) + r, *y = v + (long) row_col[q] * BT;
        for (int c = 64 * h; c < 64 * h + 64; ++c) acc += T[(long) c * BT] * y[c];
    }
    part[h][r] = acc;
    __syncthreads();
    if (tid < BT) t[tid] = v[(long) k * BT + tid] - (part[0][tid] + part[1][tid]);
    __syncthreads();
    const double *W = Winv + ((long) k << 14) + r;
    acc = 0.0;
    for (int c = 64 * h; c < 64 * h + 64; ++c) if (c <= r) acc += W[(long) c * BT] * t[c];     // W lower triangular
    part[h][r] = acc;
    __syncthreads();
    if (tid < BT) v[(long) k * BT + tid] = part[0][tid] + part[1][tid];
}

// backward substitution, one level (levels walked downwards): x_k = W_k^T (y_k - sum_{i > k} L(i, k)^T x_i)
// (v_k still holds the forward result y_k when its level is reached: the signs S_k of L~ S L~' x = b are applied here)
__global__ __launch_bounds__(256) void bs_bwd_kernel(const double *__restrict__ L, const double *__restrict__ Winv, const int *__restrict__ cols,
                                                     const int *__restrict__ col_ptr, const int *__restrict__ col_row,
                                                     const int *__restrict__ col_tile, double *__restrict__ v, int first,
                                                     const double *__restrict__ sgn) {
    __shared__ double part[2][BT], t[BT];
    const int k = cols[first + blockIdx.x], tid = threadIdx.x, c = tid & 127, h = tid >> 7;
    double acc = 0.0;
    for (int q = col_ptr[k]; q < col_ptr[k + 1]; ++q) {
        const double *T = L + ((long) col_tile[q] << 14) + (long) c * BT, *x = v + (long) col_row[q] * BT;
        for (int r = 64 * h; r < 64 * h + 64; ++r) acc += T[r] * x[r];
    }
    part[h][c] = acc;
    __syncthreads();
    if (tid < BT) t[tid] = sgn[(long) k * BT + tid] * v[(long) k * BT + tid] - (part[0][tid] + part[1][tid]);
    __syncthreads();
    const double *W = Winv + ((long) k << 14) + (long) c * BT;
    acc = 0.0;
    for (int r = 64 * h; r < 64 * h + 64; ++r) if (r >= c) acc += W[r] * t[r];
    part[h][c] = acc;
    __syncthreads();
    if (tid < BT) v[(long) k * BT + tid] = part[0][tid] + part[1][tid];
}

// identity on the diagonal of the rows past m in the last diagonal tile (they are never pivots, but the tile must be a factor)
__global__ void bs_pad_diag_kernel(double *L, const int *diag_tile, int nb, int m) {
    const int r = (nb - 1) * BT + threadIdx.x;
    if (threadIdx.x < BT && r >= m) L[((long) diag_tile[nb - 1] << 14) + threadIdx.x * (BT + 1)] = 1.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// host: symbolic phase
// ---------------------------------------------------------------------------------------------------------------------
template <class T> static int bs_upload(T **dev, const std::vector<T> &h) {
    HDM_HIP_CHECK(hipMalloc((void **) dev, sizeof(T) * std::max<size_t>(1, h.size())));
    if (!h.empty()) HDM_HIP_CHECK(hdm_memcpy_h2d_sync(*dev, h.data(), sizeof(T) * h.size()));
    return 0;
}

int HdmBsp::init(int m_, const int *beg, const int *idx, double max_fraction) {
    m = m_;
    nb = (m + BT - 1) / BT;
    if (nb < 2 || (long) nb * nb > (1L << 26)) return 1;
    // ---- row order: rows that reach a large part of the matrix go last (an arrow's shaft: eliminated last they cause no fill),
    // reverse Cuthill-McKee for the others (neighbours end up in the same or in neighbouring tiles)
    std::vector<int> deg(m, 0);
    for (int c = 0; c < m; ++c)
        for (int q = beg[c]; q < beg[c + 1]; ++q) if (idx[q] != c) { deg[c] += 1; deg[idx[q]] += 1; }
    std::vector<int> sorted(deg);
    std::nth_element(sorted.begin(), sorted.begin() + m / 2, sorted.end());
    const int dense_deg = std::max(BT, 10 * std::max(1, sorted[m / 2]));
    std::vector<int> sub(m, -1);
    int ms = 0;
    for (int v = 0; v < m; ++v) if (deg[v] <= dense_deg) sub[v] = ms++;
    std::vector<int> sbeg(ms + 1, 0), sidx;
    {
        int sc = 0;
        for (int c = 0; c < m; ++c) {
            if (sub[c] < 0) continue;
            for (int q = beg[c]; q < beg[c + 1]; ++q) if (sub[idx[q]] >= 0) sidx.push_back(sub[idx[q]]);
            sbeg[++sc] = (int) sidx.size();
        }
    }
    // (the reduced pattern is still lower triangular in its own numbering: `sub` is increasing)
    std::vector<int> sperm = ms > 0 ? hdm_rcm_order(ms, sbeg, sidx) : std::vector<int>();
    perm.assign(m, 0);
    {
        int tail = ms;
        for (int v = 0; v < m; ++v) perm[v] = (sub[v] >= 0) ? sperm[sub[v]] : tail++;
    }
    // ---- block pattern of the reordered matrix and its symbolic Cholesky factorisation (block quotient graph)
    std::vector<std::vector<int>> col(nb);
    for (int k = 0; k < nb; ++k) col[k].push_back(k);
    for (int c = 0; c < m; ++c)
        for (int q = beg[c]; q < beg[c + 1]; ++q) {
            int bi = perm[idx[q]] / BT, bj = perm[c] / BT;
            if (bi < bj) std::swap(bi, bj);
            col[bj].push_back(bi);
        }
    std::vector<int> parent(nb, -1);
    for (int k = 0; k < nb; ++k) {
        std::sort(col[k].begin(), col[k].end());
        col[k].erase(std::unique(col[k].begin(), col[k].end()), col[k].end());
        if (col[k].size() > 1) {
            const int p = col[k][1];
            parent[k] = p;
            col[p].insert(col[p].end(), col[k].begin() + 2, col[k].end());
        }
    }
    bptr.assign(nb + 1, 0);
    brow.clear();
    for (int k = 0; k < nb; ++k) { brow.insert(brow.end(), col[k].begin(), col[k].end()); bptr[k + 1] = (int) brow.size(); }
    ntiles = (int) brow.size();
    if ((double) ntiles > max_fraction * (double) dense_tiles()) return 1;
    std::vector<int> tmap((size_t) nb * nb, -1), diag(nb);
    for (int k = 0; k < nb; ++k)
        for (int q = bptr[k]; q < bptr[k + 1]; ++q) { tmap[(size_t) brow[q] + (size_t) k * nb] = q; if (brow[q] == k) diag[k] = q; }
    // ---- levels of the block elimination tree
    std::vector<int> level(nb, 0);
    for (int k = 0; k < nb; ++k) if (parent[k] >= 0) level[parent[k]] = std::max(level[parent[k]], level[k] + 1);
    nlevels = 1 + *std::max_element(level.begin(), level.end());
    std::vector<int> cols_by_level(nb);
    std::iota(cols_by_level.begin(), cols_by_level.end(), 0);
    std::stable_sort(cols_by_level.begin(), cols_by_level.end(), [&](int a, int b) { return level[a] < level[b]; });
    lvl_ptr.assign(nlevels + 1, 0);
    for (int k = 0; k < nb; ++k) lvl_ptr[level[k] + 1] += 1;
    for (int l = 0; l < nlevels; ++l) lvl_ptr[l + 1] += lvl_ptr[l];
    // ---- left-looking update lists: target tile (a, b) collects L(a, j) L(b, j)^T from every column j with both tiles
    std::vector<std::vector<int2>> srcs(ntiles);
    for (int j = 0; j < nb; ++j)
        for (int qa = bptr[j] + 1; qa < bptr[j + 1]; ++qa)
            for (int qb = bptr[j] + 1; qb <= qa; ++qb) {
                const int t = tmap[(size_t) brow[qa] + (size_t) brow[qb] * nb];
                if (t < 0) return 1;                          // cannot happen: the symbolic factorisation closed the pattern
                srcs[t].push_back(make_int2(qa, qb));
            }
    std::vector<int> h_tgt, h_tsp{0}, h_scol;
    std::vector<int2> h_src, h_pan;
    std::vector<int> tile_col((size_t) ntiles);
    for (int k = 0; k < nb; ++k) for (int q = bptr[k]; q < bptr[k + 1]; ++q) tile_col[q] = k;
    lvl_tgt_ptr.assign(nlevels + 1, 0);
    lvl_pan_ptr.assign(nlevels + 1, 0);
    for (int l = 0; l < nlevels; ++l) {
        for (int c = lvl_ptr[l]; c < lvl_ptr[l + 1]; ++c) {
            const int k = cols_by_level[c];
            for (int q = bptr[k]; q < bptr[k + 1]; ++q) {
                if (!srcs[q].empty()) {
                    h_tgt.push_back(q);
                    h_src.insert(h_src.end(), srcs[q].begin(), srcs[q].end());
                    for (const int2 &ab : srcs[q]) h_scol.push_back(tile_col[ab.x]);
                    h_tsp.push_back((int) h_src.size());
                }
                if (q > bptr[k]) h_pan.push_back(make_int2(q, k));
            }
        }
        lvl_tgt_ptr[l + 1] = (int) h_tgt.size();
        lvl_pan_ptr[l + 1] = (int) h_pan.size();
    }
    // ---- strictly lower tiles by block row and by block column (substitutions)
    std::vector<int> rp(nb + 1, 0), rc, rt, cp(nb + 1, 0), cr, ct;
    for (int k = 0; k < nb; ++k)
        for (int q = bptr[k] + 1; q < bptr[k + 1]; ++q) rp[brow[q] + 1] += 1;
    for (int k = 0; k < nb; ++k) rp[k + 1] += rp[k];
    rc.resize(rp[nb]); rt.resize(rp[nb]);
    {
        std::vector<int> fill(rp.begin(), rp.end() - 1);
        for (int k = 0; k < nb; ++k)
            for (int q = bptr[k] + 1; q < bptr[k + 1]; ++q) { rc[fill[brow[q]]] = k; rt[fill[brow[q]]++] = q; }
    }
    for (int k = 0; k < nb; ++k) {
        for (int q = bptr[k] + 1; q < bptr[k + 1]; ++q) { cr.push_back(brow[q]); ct.push_back(q); }
        cp[k + 1] = (int) cr.size();
    }
    // ---- device side
    if (bs_upload(&perm_dev, perm) || bs_upload(&tilemap, tmap) || bs_upload(&lvl_cols, cols_by_level) || bs_upload(&tgt_tile, h_tgt) ||
        bs_upload(&tgt_src_ptr, h_tsp) || bs_upload(&src, h_src) || bs_upload(&src_col, h_scol) || bs_upload(&pan, h_pan) || bs_upload(&row_ptr, rp) ||
        bs_upload(&row_col, rc) || bs_upload(&row_tile, rt) || bs_upload(&col_ptr, cp) || bs_upload(&col_row, cr) ||
        bs_upload(&col_tile, ct) || bs_upload(&diag_tile, diag))
        return 1;
    const size_t tb = sizeof(double) * BTT * ((size_t) ntiles + 1);
    if (hipMalloc((void **) &Mval, tb) != hipSuccess || hipMalloc((void **) &Lval, tb) != hipSuccess ||
        hipMalloc((void **) &Winv, sizeof(double) * BTT * (size_t) nb) != hipSuccess ||
        hipMalloc((void **) &vec, sizeof(double) * BT * (size_t) nb) != hipSuccess ||
        hipHostMalloc((void **) &hvec, sizeof(double) * BT * (size_t) nb, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **) &sgn, sizeof(double) * BT * (size_t) nb) != hipSuccess ||
        hipMalloc((void **) &info_dev, 2 * sizeof(int)) != hipSuccess) {
        (void) hipGetLastError();
        fprintf(stderr, "[hdsdp_mi355x] block-sparse Schur matrix: out of device memory (%.1f GiB of tiles)\n", (double) bytes() / (1 << 30));
        return 1;
    }
    HDM_HIP_CHECK(hdm_memset_sync(Mval, 0, tb));
    HDM_HIP_CHECK(hdm_memset_sync(Lval, 0, tb));
    if (hdm_potrf_sweep_configure()) return 1;
    factored = false;
    return 0;
}

void HdmBsp::destroy() {
    for (void *p : {(void *) perm_dev, (void *) tilemap, (void *) Mval, (void *) Lval, (void *) Winv, (void *) lvl_cols, (void *) tgt_tile,
                    (void *) tgt_src_ptr, (void *) src, (void *) pan, (void *) row_ptr, (void *) row_col, (void *) row_tile, (void *) col_ptr,
                    (void *) col_row, (void *) col_tile, (void *) diag_tile, (void *) vec, (void *) info_dev, (void *) sgn, (void *) src_col})
        if (p) (void) hipFree(p);
    if (hvec) (void) hipHostFree(hvec);
    perm_dev = tilemap = lvl_cols = tgt_tile = tgt_src_ptr = row_ptr = row_col = row_tile = col_ptr = col_row = col_tile = diag_tile = info_dev = nullptr;
    Mval = Lval = Winv = vec = hvec = sgn = nullptr;
    src_col = nullptr;
    src = pan = nullptr;
}

HdmMatView HdmBsp::view_M() const { HdmMatView v; v.base = Mval; v.ld = 0; v.tilemap = tilemap; v.perm = perm_dev; v.nbt = nb; v.trash = ntiles; return v; }
HdmMatView HdmBsp::view_L() const { HdmMatView v = view_M(); v.base = Lval; return v; }
int HdmBsp::zero_M(hipStream_t s) { HDM_HIP_CHECK(hipMemsetAsync(Mval, 0, sizeof(double) * BTT * (size_t) ntiles, s)); return 0; }
int HdmBsp::zero_L(hipStream_t s) { HDM_HIP_CHECK(hipMemsetAsync(Lval, 0, sizeof(double) * BTT * (size_t) ntiles, s)); factored = false; return 0; }
int HdmBsp::load_M(hipStream_t s) {
    HDM_HIP_CHECK(hipMemcpyAsync(Lval, Mval, sizeof(double) * BTT * (size_t) ntiles, hipMemcpyDeviceToDevice, s));
    factored = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// numeric factorisation and substitutions
// ---------------------------------------------------------------------------------------------------------------------
int HdmBsp::factor(hipStream_t s, int *info_host, int *nneg_host) {
    HDM_HIP_CHECK(hipMemsetAsync(info_dev, 0, 2 * sizeof(int), s));
    if (m % BT) hipLaunchKernelGGL(bs_pad_diag_kernel, dim3(1), dim3(BT), 0, s, Lval, diag_tile, nb, m);
    for (int l = 0; l < nlevels; ++l) {
        const int nt = lvl_tgt_ptr[l + 1] - lvl_tgt_ptr[l], nc = lvl_ptr[l + 1] - lvl_ptr[l], np = lvl_pan_ptr[l + 1] - lvl_pan_ptr[l];
        if (nt > 0)
            hipLaunchKernelGGL(bs_tile_kernel<0>, dim3(nt), dim3(256), 0, s, Lval, Winv, tgt_tile, tgt_src_ptr, src, pan, lvl_tgt_ptr[l], sgn, src_col);
        if (hdm_potrf_sweep_batched(Lval, diag_tile, lvl_cols + lvl_ptr[l], nc, Winv, info_dev, m, s, sgn)) return 1;
        if (np > 0)
            hipLaunchKernelGGL(bs_tile_kernel<1>, dim3(np), dim3(256), 0, s, Lval, Winv, tgt_tile, tgt_src_ptr, src, pan, lvl_pan_ptr[l], sgn, src_col);
    }
    HDM_HIP_CHECK(hipGetLastError());
    int info[2] = {0, 0};
    HDM_HIP_CHECK(hipMemcpyAsync(info, info_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    HDM_HIP_CHECK(hipStreamSynchronize(s));
    if (info[0] > m) info[0] = 0;
    if (info_host) *info_host = info[0];
    if (nneg_host) *nneg_host = info[1];
    negative = info[1];
    factored = (info[0] == 0);
    return 0;
}

int HdmBsp::solve_host(const double *rhs, double *sol, hipStream_t s) {
    if (!factored) return 1;
    HDM_HIP_CHECK(hipStreamSynchronize(s));                     // the staging buffer is free
    memset(hvec, 0, sizeof(double) * BT * (size_t) nb);
    for (int i = 0; i < m; ++i) hvec[perm[i]] = rhs[i];
    HDM_HIP_CHECK(hipMemcpyAsync(vec, hvec, sizeof(double) * BT * (size_t) nb, hipMemcpyHostToDevice, s));
    for (int l = 0; l < nlevels; ++l)
        hipLaunchKernelGGL(bs_fwd_kernel, dim3(lvl_ptr[l + 1] - lvl_ptr[l]), dim3(256), 0, s, Lval, Winv, lvl_cols, row_ptr, row_col, row_tile,
                           vec, lvl_ptr[l]);
    for (int l = nlevels - 1; l >= 0; --l)
        hipLaunchKernelGGL(bs_bwd_kernel, dim3(lvl_ptr[l + 1] - lvl_ptr[l]), dim3(256), 0, s, Lval, Winv, lvl_cols, col_ptr, col_row, col_tile,
                           vec, lvl_ptr[l], sgn);
    HDM_HIP_CHECK(hipGetLastError());
    HDM_HIP_CHECK(hipMemcpyAsync(hvec, vec, sizeof(double) * BT * (size_t) nb, hipMemcpyDeviceToHost, s));
    HDM_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < m; ++i) sol[i] = hvec[perm[i]];
    return 0;
}

// one kernel of this translation unit (= one code object): what the preload thread asks the runtime about (engine.hip: preload_modules)
const void *hdm_module_handle_bsparse() { return (const void *) bs_pad_diag_kernel; }
