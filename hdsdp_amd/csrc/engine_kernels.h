// engine_kernels.h -- small kernels of the engine at global scope (row put, dots, data norms, scaling, CSC gather / scatter, the shards' sum)
// Implementation header of engine.hip: included exactly once, there, in this order (global scope, between the anonymous-namespace pieces); split out of a 3 300-line file in round 3, nothing else changed.
// kernels local to this file -------------------------------------------------------------------
// row i of the lower triangle of M (column-major, ld): M[i, j] += v[j] for j <= i
__global__ void mi_put_row_kernel(HdmMatView Mv, int i, const double *__restrict__ v, int m) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < m && j <= i) *hdm_mat_at(Mv, i, j) += v[j];
}
// out[j] = M(i, j) of the symmetric matrix behind the view (lower triangle stored), 0 where the tile store has no tile
__global__ void mi_get_row_kernel(HdmMatView Mv, int i, int m, double *__restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int r = i > j ? i : j, c = i > j ? j : i;
    if (Mv.tilemap) {
        int pr = Mv.perm[r], pc = Mv.perm[c];
        if (pr < pc) { const int t = pr; pr = pc; pc = t; }
        if (Mv.tilemap[(pr >> 7) + (long) (pc >> 7) * Mv.nbt] < 0) { out[j] = 0.0; return; }
    }
    out[j] = *hdm_mat_at(Mv, r, c);
}

__global__ void mi_col_dot_kernel(const double *__restrict__ X, const double *__restrict__ Y, long ld, int n,
                                  const double *__restrict__ sgn, const int *__restrict__ rows, int count,
                                  double *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cidx = blockIdx.x * 4 + wave;
    if (cidx >= count) return;
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += X[i + (long) cidx * ld] * Y[i + (long) cidx * ld];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[rows[cidx]] += sgn[cidx] * s;
}

__global__ void mi_mat_dot_kernel(const double *__restrict__ X, long ldx, const double *__restrict__ Y, long ldy, int n,
                                  int diag_only, double scale, double *__restrict__ out) {
    // single workgroup: out += scale * <X, Y> over n x n (or trace(X) if Y == nullptr and diag_only)
    __shared__ double red[4];
    double s = 0.0;
    if (diag_only) {
        for (int i = threadIdx.x; i < n; i += 256) s += X[i + (long) i * ldx];
    } else {
        for (long e = threadIdx.x; e < (long) n * n; e += 256) {
            int i = (int) (e % n), j = (int) (e / n);
            s += X[i + (long) j * ldx] * Y[i + (long) j * ldy];
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out += scale * (red[0] + red[1] + red[2] + red[3]);
}

// out[0] += sum |A_ij|, out[1] += sum A_ij^2 over the full symmetric matrices given by their lower triangles; a_l_form:
// the diagonal is stored halved (engine layout of the constraint matrices).  One workgroup per matrix.
__global__ void mi_low_norms_kernel(const double *__restrict__ A, long astride, int n, long ld, int count, int a_l_form,
                                    double *__restrict__ out) {
    __shared__ double ra[4], rf[4];
    const double *M = A + (long) blockIdx.x * astride;
    double sa = 0.0, sf = 0.0;
    if (a_l_form) {
        // A_L form in skyline storage (hdm_common.h): everything that is stored and not zero is an entry on or below the
        // diagonal, so one linear pass over the matrix's storage does it (this loop once walked the n x n index space with
        // a division and the skyline offset per element: 1.0 s for 2000 matrices at n = 2000, now HBM-bound).  Off-diagonal
        // entries count twice; the diagonal is stored halved: |2v| = 2|v| as well, and (2v)^2 = 2v^2 + 2v^2 -- the second
        // half comes from the short loop over the diagonal.
        const long cnt = hdm_sky_size((int) ld);
        for (long e = threadIdx.x; e < cnt; e += 256) { const double v = M[e]; sa += 2.0 * fabs(v); sf += 2.0 * v * v; }
        for (int i = threadIdx.x; i < n; i += 256) { const double v = M[hdm_sky_off(i, i, (int) ld)]; sf += 2.0 * v * v; }
    } else {
        for (int j = 0; j < n; ++j)
            for (int i = j + threadIdx.x; i < n; i += 256) {
                const double v = M[i + (long) j * ld];
                if (i == j) { sa += fabs(v); sf += v * v; }
                else { sa += 2.0 * fabs(v); sf += 2.0 * v * v; }
            }
    }
    for (int off = 32; off > 0; off >>= 1) { sa += __shfl_down(sa, off, 64); sf += __shfl_down(sf, off, 64); }
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = sa; rf[threadIdx.x >> 6] = sf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out, ra[0] + ra[1] + ra[2] + ra[3]);
        atomicAdd(out + 1, rf[0] + rf[1] + rf[2] + rf[3]);
    }
    (void) count;
}

__global__ void mi_scale_kernel(double *__restrict__ A, long count, double s) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) A[e] *= s;
}

// out += <S, X> with S given by its lower triangle (fds_dot_fds, dense_opts.c:134-156): 2 * (sum_{i>j} + half the diagonal)
__global__ void mi_lower_dot_kernel(const double *__restrict__ S, long lds_, const double *__restrict__ X, long ldx, int n,
                                    double *__restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (long e = threadIdx.x; e < (long) n * n; e += 256) {
        const int i = (int) (e % n), j = (int) (e / n);
        if (i < j) continue;
        const double v = S[i + (long) j * lds_] * X[i + (long) j * ldx];
        s += (i == j) ? 0.5 * v : v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out += 2.0 * (red[0] + red[1] + red[2] + red[3]);
}



// sparse Schur operator: entry p of the aggregated CSC pattern <-> element (rows[p], cols[p]) of the dense device matrix
__global__ void mi_csc_gather_kernel(HdmMatView Mv, const int *__restrict__ rows, const int *__restrict__ cols, long nnz,
                                     double *__restrict__ vals) {
    const long p = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) vals[p] = *hdm_mat_at(Mv, rows[p], cols[p]);
}
__global__ void mi_csc_scatter_kernel(HdmMatView Mv, const int *__restrict__ rows, const int *__restrict__ cols, long nnz,
                                      const double *__restrict__ vals) {
    const long p = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) *hdm_mat_at(Mv, rows[p], cols[p]) = vals[p];
}

// the copy transport's all-reduce: out[i] = sum over the shards' buffers, in shard order
struct MiGrpPtrs { const double *p[16]; };
__global__ void mi_grp_sum_kernel(MiGrpPtrs pl, int W, long lo, long cnt, double *__restrict__ out) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    double s = 0.0;
    for (int q = 0; q < W; ++q) s += pl.p[q][lo + i];
    out[i] = s;
}
