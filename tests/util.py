"""shared helpers for the parity tests"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# the reference's own acceptance bar for Schur quantities (interface/hdsdp_utils.c:621-641):
#   |a - b| / (|a| + 1e-4) < 1e-8
KKT_TOL = 1e-8


def kkt_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(got - ref) / (np.abs(ref) + 1e-4))) if ref.size else 0.0


# The reference's bar has an absolute floor of 1e-4, which is vacuous when S ~ 1e3 I makes every entry
# of M ~ 1e-6; so every comparison ALSO has to pass a norm-wise relative bound.
REL_TOL = 1e-10
# ratio test (Lanczos estimate of the largest feasible step): the reference's own estimate carries a safety term of up
# to 1e-3 relative (hdsdp_lanczos.c:268-276); two runs of the same recurrence agree far better than that
RATIO_TOL = 1e-6


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = float(np.max(np.abs(ref))) if ref.size else 0.0
    if den == 0.0:
        return float(np.max(np.abs(got))) if got.size else 0.0
    return float(np.max(np.abs(got - ref))) / den


def check_close(got, ref, what=""):
    e1, e2 = kkt_err(got, ref), rel_err(got, ref)
    assert e1 < KKT_TOL, f"{what}: reference-bar error {e1:.3e}"
    assert e2 < REL_TOL, f"{what}: norm-wise relative error {e2:.3e}"
    return e2


def lower_mask(m):
    """golden / KKT.M arrays are column-major m x m seen in C order: element (row i, col j) sits at [j, i];
    the reference fills row >= col only."""
    return np.triu(np.ones((m, m), dtype=bool))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def y_of(g):
    return np.asarray(g["y"], dtype=np.float64)


def primal_X(n):
    """the fixed symmetric positive definite matrix oracle/ref_dump.c registers with HKKTRegisterPSDP for the
    KKT_TYPE_PRIMAL goldens (closed form, diagonally dominant)"""
    i = np.arange(n, dtype=np.float64)
    I, J = np.meshgrid(i, i, indexing="ij")
    X = 0.5 / n * np.cos(0.37 * (I + J) + 0.11 * I * J)
    X[np.arange(n), np.arange(n)] = 2.0 + 0.01 * (np.arange(n) % 7)
    return np.ascontiguousarray(X)


def golden_schur_dense(g, key):
    """the Schur matrix of a multi-block golden as the dense m x m array the tests compare (C-order view of the
    column-major matrix, lower triangle at [col, row]); a golden dumped from the reference's SPARSE operator
    (kkt_sparse = 1: aggregated CSC pattern kkt_beg / kkt_idx + nnz values) is expanded"""
    if "kkt_sparse" not in g or int(g["kkt_sparse"][0]) == 0:
        return g[key]
    m = int(g["mb_dims"][1])
    beg, idx, val = g["kkt_beg"], g["kkt_idx"], g[key]
    D = np.zeros((m, m))
    for j in range(m):
        D[j, idx[beg[j]:beg[j + 1]]] = val[beg[j]:beg[j + 1]]
    return D
