// coeff.h -- host-side constraint data model of one SDP block (presolve-time, no GPU).
//
// What the engine needs to know about a block's coefficient matrices before it lays them out in HBM, stated as the
// SPECIFICATION the reference's presolve implements (the behaviour is pinned by
// tests/test_abi_cpu.py::test_host_presolve_matches_reference and tests/test_gpu_parity.py::test_presolve_plan_matches_reference
// against dumps of the compiled reference):
//   class of a matrix    no entry: ZERO; more than 0.3 P entries (P = n(n+1)/2): DENSE; else SPARSE
//                        (linalg/hdsdp_sdpdata.c:2321-2345)
//   rank-one form        A = sign * a a' iff the stored lower entries are reproduced to 1e-10 in the 1-norm by the factor
//                        read off the column of the first stored (resp. first non-zero) diagonal entry; a factor with more
//                        than n/2 entries above 1e-10 makes the matrix DSR1, else SPR1; |a| = 1, the scale goes into `sign`
//                        (hdsdp_sdpdata.c:2373-2458, :880-899; sparse_opts.c:453-516; dense_opts.c:233-285)
//   reported nnz / rank  DENSE, DSR1: P; SPARSE: entries; SPR1: k(k+1)/2 for k factor entries; rank 1 or n
//                        (hdsdp_sdpdata.c:388-435, :2347-2358)
//   order                rows by reported nnz, descending, ties where the reference's quicksort leaves them
//                        (interface/hdsdp_utils.c:93-112, :468-477)
//   plan                 per position the cheapest of M2..M5 by the cost model of hdsdp_conic_sdp.c:539-600
// All strategies give the same numbers on the device; the plan is reported, the classes choose the
// device path.
//
// Ingest is column by column (mi_coeff_build), with 64-bit totals: a block's columns may hold more than 2^31 entries
// together (fully dense n = m = 2000: 4.0e9), which the reference's `int` column pointers cannot express
// (interface/def_hdsdp_user_data.h:22-32).  One column holds at most P < 2^31 entries (n <= 65535).
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

// fn(t) on `nthreads` host threads (0: HDSDP_MI355X_HOST_THREADS, default min(16, hardware threads)); returns when all are done
int mi_host_threads();
void mi_parallel(int nthreads, const std::function<void(int)> &fn);

enum MiCoeffType { MI_COEFF_ZERO = 0, MI_COEFF_SPARSE = 1, MI_COEFF_DENSE = 2, MI_COEFF_SPR1 = 3, MI_COEFF_DSR1 = 4 };

struct MiCoeff {
    int type = MI_COEFF_ZERO;
    int nnz = 0;        // what the reference's getnnz reports for the final type
    int rank = 0;       // sdpDataMatGetRank
    long stored = 0;    // entries handed in (stays valid after release())
    double trace = 0.0; // sum of the stored diagonal entries
    // lower-triangular entries as given (packed index, value), sorted by packed index; empty after release()
    std::vector<int> idx;
    std::vector<double> val;
    // rank-one form  A = sign * a a',  |a|_2 = 1 (types SPR1 / DSR1)
    double sign = 0.0;
    std::vector<double> factor;  // dense length-n factor
    int factor_nnz = 0;
    // a multiple of the identity as the reference's feature detection sees it (SPARSE class, exactly n stored entries, all on
    // the diagonal, all equal: linalg/hdsdp_sdpdata.c:906-931); decided while the entries are at hand
    bool is_eye = false;
    double eye_val = 0.0;
    int unit_col = -1;  // i if the matrix is e_i e_i' exactly (SPR1, one factor entry 1.0, sign 1.0: hdsdp_sdpdata.c:963-975), else -1
    // the entries are on the device (or not needed there): give the host copy back; class, counts, trace and factor stay
    void release() { std::vector<int>().swap(idx); std::vector<double>().swap(val); }
};

struct MiBlockData {
    int n = 0, m = 0;
    MiCoeff obj;
    std::vector<MiCoeff> rows;
    std::vector<int> perm;      // sdpConePerm
    std::vector<int> strategy;  // KKTStrategies[position]
    int counts[5] = {0, 0, 0, 0, 0};
    int64_t stored = 0;         // entries of all columns together
};

// one column: `nnz` lower-triangular entries (packed index, value; any order, no duplicates) of an n x n matrix -> class,
// rank-one form, counts.  Returns 1 on an index outside [0, P).
int mi_coeff_build(MiCoeff &c, int n, long nnz, const int *idx, const double *val);
// after all m rows are in place: class counts, order and plan
void mi_block_plan(MiBlockData &blk);
// CSC of shape n(n+1)/2 x (m+1): column 0 = objective C, column i = A_i (def_hdsdp_user_data.h:16-32); column pointers
// 32-bit (the reference's layout) or 64-bit
int mi_block_from_csc(MiBlockData &blk, int m, int n, const int *beg, const int *idx, const double *val);
int mi_block_from_csc(MiBlockData &blk, int m, int n, const int64_t *beg, const int *idx, const double *val);
