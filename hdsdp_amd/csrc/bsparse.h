// bsparse.h -- block-sparse storage and Cholesky factorisation of a sparse Schur matrix (bsparse.hip)
//
// The reference keeps a sparse Schur matrix as the aggregated-pattern CSC and factors it with a sparse direct solver
// (interface/hdsdp_schur.c:46-139, linalg/hdsdp_linsolver.c:509-809: QDLDL).  The device counterpart stores and factors
// 128 x 128 TILES: the rows are reordered (rows that touch a large share of the matrix last, reverse Cuthill-McKee for the
// rest), cut into blocks of 128, and only the tiles inside the block pattern of the Cholesky factor (a symbolic
// factorisation of the block quotient graph) exist -- memory O(tiles of L), nothing of the m x m square.  The numeric
// factorisation -- an LDL' in signed-Cholesky form, see HdmBsp::factor -- is left-looking by levels of the block elimination tree: all block columns of a level are independent, so a
// level is three launches whatever its width (tile updates on the fp64 MFMA, diagonal-block sweeps, panel products), and a
// block-diagonal matrix with a few linking rows -- many small SDP blocks that share a handful of constraints -- is three
// levels, not m / 128 dependent steps.
#pragma once
#include "hdm_common.h"
#include <vector>

// where the Schur builders put an entry of M: a dense column-major matrix (tilemap == nullptr) or the tile store.
// (r, c) are the DRIVER's row / column numbers, r >= c.
struct HdmMatView {
    double *base = nullptr;
    long ld = 0;                 // dense: leading dimension
    const int *tilemap = nullptr;   // tiles: nbt x nbt, block row + block column * nbt -> tile number (lower triangle), -1 = not stored
    const int *perm = nullptr;      // tiles: row renumbering, old -> new
    int nbt = 0, trash = 0;         // tiles: block count; number of the tile that absorbs writes outside the pattern (there are none)
};
__device__ __forceinline__ double *hdm_mat_at(const HdmMatView &v, int r, int c) {
    if (!v.tilemap) return v.base + r + (long) c * v.ld;
    int pr = v.perm[r], pc = v.perm[c];
    if (pr < pc) { const int t = pr; pr = pc; pc = t; }
    int t = v.tilemap[(pr >> 7) + (long) (pc >> 7) * v.nbt];
    if (t < 0) t = v.trash;
    return v.base + ((long) t << 14) + (pr & 127) + ((pc & 127) << 7);
}

struct HdmBsp {
    int m = 0, nb = 0, ntiles = 0, nlevels = 0;
    std::vector<int> perm;            // old -> new
    std::vector<int> bptr, brow;      // block pattern of L, lower incl. diagonal, by block column (host); tile number = position
    std::vector<int> lvl_ptr, lvl_tgt_ptr, lvl_pan_ptr;   // per level: block columns, update targets, panel tiles
    // device
    int *perm_dev = nullptr, *tilemap = nullptr;
    double *Mval = nullptr, *Lval = nullptr, *Winv = nullptr;   // (ntiles + 1) tiles each; nb inverted diagonal tiles
    int *lvl_cols = nullptr;          // block columns ordered by level
    int *tgt_tile = nullptr, *tgt_src_ptr = nullptr;   // update targets (ordered by level) and their source lists
    int2 *src = nullptr;              // (tile (i, j), tile (k, j)) pairs: target (i, k) -= L(i, j) L(k, j)^T
    int2 *pan = nullptr;              // (tile (i, k), k): L(i, k) <- A(i, k) W_k^T
    int *row_ptr = nullptr, *row_col = nullptr, *row_tile = nullptr;   // strictly lower tiles by block row (forward substitution)
    int *col_ptr = nullptr, *col_row = nullptr, *col_tile = nullptr;   // ... by block column (backward substitution)
    int *diag_tile = nullptr;
    double *vec = nullptr;            // nb * 128 solve vector
    double *hvec = nullptr;           // pinned host staging of the same size
    int *info_dev = nullptr;          // two words: first zero pivot + 1, negative pivots
    double *sgn = nullptr;            // nb * 128 pivot signs of the last factorisation (+1 in the padding)
    int *src_col = nullptr;           // block column of every source pair (whose signs the update applies)
    bool factored = false;
    int negative = 0;                 // negative pivots of the last factorisation (0: the matrix is positive definite)
    long dense_tiles() const { return (long) nb * (nb + 1) / 2; }
    size_t bytes() const { return sizeof(double) * 16384 * ((size_t) 2 * (ntiles + 1) + nb); }

    // symbolic phase from the lower-triangular CSC pattern (host).  Returns 0, or 1 when the tile form would not pay
    // (`max_fraction` of the dense lower triangle's tiles) or cannot be built.
    int init(int m, const int *beg, const int *idx, double max_fraction);
    void destroy();
    HdmMatView view_M() const;        // the accumulation store the builders write
    HdmMatView view_L() const;        // the factor store (values scattered from the host CSC land here)
    int zero_M(hipStream_t s);
    int zero_L(hipStream_t s);
    int load_M(hipStream_t s);        // L store <- M store
    // In the L store, as M = L~ S L~' with S = diag(+-1) the pivots' signs and L~ = L |D|^1/2 -- the reference's sparse direct
    // solver is an LDL' without pivoting (external/qdldl.c: an indefinite matrix factors, only a pivot that is exactly zero
    // fails, linalg/hdsdp_linsolver.c:596-626), and so is this; for a positive definite matrix it IS the Cholesky factorisation,
    // bit for bit.  info = 0, or first zero / non-finite pivot + 1 (renumbered order); nneg = negative pivots.
    int factor(hipStream_t s, int *info, int *nneg = nullptr);
    int solve_host(const double *rhs, double *sol, hipStream_t s);   // driver's numbering in and out
};

// reverse Cuthill-McKee order of a symmetric pattern given as its lower triangle in CSC form: perm[old] = new
std::vector<int> hdm_rcm_order(int m, const std::vector<int> &beg, const std::vector<int> &idx);
