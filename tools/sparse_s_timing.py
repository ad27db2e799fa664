"""Sparse dual matrix (hdsdp_linsolver.c:509-809): what the engine's choice -- accept the CSC, scatter it into a
dense device matrix, blocked MFMA Cholesky, dense inverse -- costs at sparse-cone sizes, beside a sparse direct solver on the
host (scipy's SuperLU as a stand-in for the reference's QDLDL: factor once, then n solves for the dense inverse the
reference forms as well, hdsdp_conic_sdp.c `dualMatInv`).  Patterns: a band of half-width 5 plus an arrow row (maxG-like),
and 5 random entries per column."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hdsdp_amd import api


def pattern(n, kind, rng):
    rows, cols, vals = [], [], []
    if kind == "band+arrow":
        for d in range(1, 6):
            i = np.arange(d, n); rows += i.tolist(); cols += (i - d).tolist(); vals += rng.uniform(-1, 1, n - d).tolist()
        i = np.arange(0, n - 1); rows += [n - 1] * (n - 1); cols += i.tolist(); vals += rng.uniform(-1, 1, n - 1).tolist()
    else:
        i = rng.integers(0, n, 5 * n); j = rng.integers(0, n, 5 * n)
        lo, hi = np.minimum(i, j), np.maximum(i, j)
        k = lo != hi
        rows, cols, vals = hi[k].tolist(), lo[k].tolist(), rng.uniform(-1, 1, k.sum()).tolist()
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsc()
    A.sum_duplicates()
    S = A + A.T
    d = np.asarray(abs(S).sum(axis=1)).ravel() + 4.0
    return (S + sp.diags(d)).tocsc()


def main():
    rng = np.random.default_rng(5)
    for kind in ("band+arrow", "random 5 per column"):
        for n in (1000, 4000, 8000):
            S = pattern(n, kind, rng)
            L = sp.tril(S).tocsc()
            L.sort_indices()
            ls = api.LinSys(n, api.HDSDP_LINSYS_SPARSE_DIRECT)
            ls.symbolic(L.indptr, L.indices)
            assert ls.psd_check_csc(L.data)
            t0 = time.perf_counter()
            for _ in range(3):
                assert ls.psd_check_csc(L.data)          # scatter + blocked Cholesky on the device (HFpLinsysPsdCheck = numeric factorisation)
            t_fac = (time.perf_counter() - t0) / 3
            t0 = time.perf_counter()
            X = ls.invert()                              # dense inverse to the host (n^2 doubles over PCIe included)
            t_inv = time.perf_counter() - t0
            b = rng.uniform(-1, 1, n)
            res = np.linalg.norm(S @ ls.solve(b) - b) / np.linalg.norm(b)
            ls.destroy()
            t0 = time.perf_counter()
            lu = spl.splu(S, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
            t_hfac = time.perf_counter() - t0
            cols = min(n, 256)
            t0 = time.perf_counter()
            lu.solve(np.eye(n, cols))
            t_hinv = (time.perf_counter() - t0) * n / cols
            fill = (lu.L.nnz + lu.U.nnz) / 2
            err = np.max(np.abs(X[:, :cols] - lu.solve(np.eye(n, cols))))
            print(f"{kind:22s} n={n:5d} nnz(tril S)={L.nnz:7d} host factor nnz={int(fill):9d} | device: factor {t_fac*1e3:7.1f} ms, "
                  f"dense inverse to host {t_inv*1e3:7.1f} ms (residual {res:.1e}) | host SuperLU 1 core: factor {t_hfac*1e3:7.1f} ms, "
                  f"n solves {t_hinv*1e3:8.1f} ms (from {cols} columns) | max |X - X_host| {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
