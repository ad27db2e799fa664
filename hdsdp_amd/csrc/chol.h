// chol.h -- device-resident blocked Cholesky object (see chol.hip)
#pragma once
#include "hdm_common.h"
#include <vector>

struct HdmChol {
    int n = 0, npad = 0, nblk = 0;
    double *L = nullptr;     // npad x npad, column-major; lower triangle = Cholesky factor after factor()
    double *Linv = nullptr;  // npad x npad, lower triangular inverse (explicit zeros above the diagonal)
    double *Dinv = nullptr;  // nblk x (128 x 128) inverted diagonal blocks
    double *Z = nullptr;     // npad x 128 scratch for invert_factor
    double *Zd = nullptr;    // npad x npad scratch for the recursive-doubling variant of invert_factor
    double *vec = nullptr;   // 4 * npad scratch vectors
    int *info_dev = nullptr;
    bool factored = false, have_inv = false;
    bool logdet_ok = false;  // log det of the factored matrix is known (single-launch small-block check): 2 sum log L_kk
    double logdet_val = 0.0;
    // The factorisation (3 launches per 128-block) and the block substitutions (2 launches per block) are chains of
    // short dependent launches with fixed arguments: after one eager run they can be captured into a hipGraph and
    // replayed (policy and measurements: hdm_graph_level in chol.hip).  A failed capture falls back to eager launches.
    struct Replay { const void *b = nullptr, *x = nullptr; int nrhs = 0, which = 0; long ldv = 0; hipGraphExec_t exec = nullptr; };
    hipGraphExec_t factor_graph = nullptr;
    Replay solves[4];
    int nsolves = 0, factor_runs = 0, solve_runs = 0;
    bool graphs_ok = true;
    // single-launch block substitution (hdm_trsv_flow_kernel): per (rhs, direction, block) publication flags + error word
    int *flow_flags = nullptr;
    int *flow_err = nullptr;     // mapped host word: a workgroup gave up waiting
    int flow_epoch = 0;
    bool flow_ok = true, flow_pending = false;
    int flow_cap = -1;           // co-residency bound on this object's device (workgroups), computed at the first solve
    std::vector<double> host_stage;   // solve_host: solutions land here first (in-place solves, retry after a give-up)
    // Block envelope of a matrix with structural zeros (the sparse Schur operator's M): env_first[i] = first 128-block column
    // with an entry in block row i, env_colh[k] = last block row whose envelope reaches block column k (non-decreasing).
    // A Cholesky factor fills inside the row envelope only, so the factorisation's panel and trailing update of block column k
    // stop at row env_colh[k] and the substitutions skip the blocks outside.  Empty = dense.
    std::vector<int> env_first, env_colh;
    int *env_dev = nullptr;      // device copy: first[nblk], then colh[nblk]

    int init(int n);
    int set_envelope(const int *first_blockcol_of_blockrow);   // nblk entries; nullptr = dense.  Before the first factor().
    void destroy();
    int load_host(const double *A, long lda, hipStream_t s);
    int load_device(const double *A, long lda, hipStream_t s);
    int finish_load(hipStream_t s);
    int factor(hipStream_t s, int *info_host);          // info = 0 ok, j+1 = first non-positive pivot
    int invert_factor(hipStream_t s);                   // builds Linv (idempotent until the next load)
    int set_reverse_inverse(hipStream_t s);             // primal builds: Linv <- W, W^T W = X, from the factor of J X J
    int get_diag(double *diag_host, hipStream_t s);
    int solve_device(double *b_dev, double *x_dev, int nrhs, long ldv, int which, hipStream_t s);
    int enqueue_factor(hipStream_t s);
    int enqueue_solve(double *b_dev, double *x_dev, int nrhs, long ldv, int which, hipStream_t s);
    int solve_host(const double *rhs, double *sol, int nrhs, int which, hipStream_t s);
    int inverse_full(double *out_dev, long ldo, hipStream_t s);
};

// diagonal-tile factorisations of a block-sparse matrix (bsparse.hip): 128 x 128 tiles, 16384 doubles apart; tile
// diag_tile[cols[b]] is factored in place (lower), its inverse goes to Winv + cols[b] * 16384; rows past m are padding
int hdm_potrf_sweep_configure();
// sgn == nullptr: Cholesky sweeps (info[0] = first non-positive pivot + 1).  sgn != nullptr: the signed LDL' sweep of sweep128.h --
// sgn[128 k ..] <- pivot signs of block column k, info[0] = first zero / non-finite pivot + 1, info[1] += negative pivots
int hdm_potrf_sweep_batched(double *tiles, const int *diag_tile, const int *cols, int ncols, double *Winv, int *info, int m, hipStream_t s,
                            double *sgn = nullptr);

// several engine shards share this device: the single-launch substitution (which needs co-resident workgroups) is off
void hdm_flow_set_shared_device(int on);
bool hdm_flow_shared_device();   // several engine instances share one device: no kernel may count on its grid being co-resident
double hdm_diag_block_probe(int variant, int reps, hipStream_t s);   // diagnostic: us per diagonal-block kernel (0: LDS panels, 1: register sweep)
