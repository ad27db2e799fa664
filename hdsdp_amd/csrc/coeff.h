// coeff.h -- host-side constraint data model of one SDP block (presolve-time, no GPU).
//
// Mirrors what the reference derives in its presolve so that (a) the classification / ordering /
// strategy plan can be compared 1:1 with the reference's (tests/golden: coef_type, kkt_perm,
// kkt_strategy) and (b) the device path can pick its rank-one fast path:
//   classification     linalg/hdsdp_sdpdata.c:2321-2345 (sdpDataMatSetData)
//   rank-one detection linalg/hdsdp_sdpdata.c:2373-2458, sparse_opts.c:453-516, dense_opts.c:233-285
//   nnz / rank         linalg/hdsdp_sdpdata.c:388-435, :2347-2358
//   ordering + plan    interface/hdsdp_conic_sdp.c:539-676, interface/hdsdp_utils.c:93-112,468-477
#pragma once
#include <vector>

enum MiCoeffType { MI_COEFF_ZERO = 0, MI_COEFF_SPARSE = 1, MI_COEFF_DENSE = 2, MI_COEFF_SPR1 = 3, MI_COEFF_DSR1 = 4 };

struct MiCoeff {
    int type = MI_COEFF_ZERO;
    int nnz = 0;   // what the reference's getnnz reports for the final type
    int rank = 0;  // sdpDataMatGetRank
    // raw lower-triangular entries as given (packed index, value), sorted by packed index
    std::vector<int> idx;
    std::vector<double> val;
    // rank-one form  A = sign * a a',  |a|_2 = 1 (types SPR1/DSR1)
    double sign = 0.0;
    std::vector<double> factor;  // dense length-n factor
    int factor_nnz = 0;
};

struct MiBlockData {
    int n = 0, m = 0;
    MiCoeff obj;
    std::vector<MiCoeff> rows;
    std::vector<int> perm;      // sdpConePerm
    std::vector<int> strategy;  // KKTStrategies[position]
    int counts[5] = {0, 0, 0, 0, 0};
};

// CSC of shape n(n+1)/2 x (m+1): column 0 = objective C, column i = A_i (def_hdsdp_user_data.h:16-32)
int mi_block_from_csc(MiBlockData &blk, int m, int n, const int *beg, const int *idx, const double *val);
