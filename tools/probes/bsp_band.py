"""tile-form Cholesky (csrc/bsparse.hip) on big sparse patterns: a band at m = 200 000 (a dense m x m device matrix would be 320 GB)
and an arrow at m = 100 000; factorisation time, tiles, levels, residual of the solve"""
import ctypes as C
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdsdp_amd import api
lib = api.load_library()
ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)


def run(name, A):
    m = A.shape[0]
    L = sp.tril(A, format="csc")
    L.sort_indices()
    beg, idx, val = L.indptr.astype(np.int32), L.indices.astype(np.int32), L.data.astype(np.float64)
    assert np.all(idx[beg[:-1]] == np.arange(m))          # diagonal first
    b = np.cos(0.01 * np.arange(m))
    x = np.zeros(m)
    info, stats, ms = C.c_int(-1), (C.c_int * 4)(), C.c_double(0.0)
    t0 = time.time()
    rc = lib.HMiBspSolve(m, beg.ctypes.data_as(ip), idx.ctypes.data_as(ip), val.ctypes.data_as(dp), b.ctypes.data_as(dp), x.ctypes.data_as(dp),
                         C.byref(info), stats, C.byref(ms))
    t1 = time.time()
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b) if rc == 0 and info.value == 0 else float("nan")
    nb, nt, nl = stats[0], stats[1], stats[2]
    print("%-28s m %7d nnz(lower) %9d  rc %d info %d  tiles %6d of %9d (%.2f GiB of tile stores vs %.0f GiB dense)  levels %5d  "
          "factor %.1f ms  whole call %.1f s  residual %.1e" % (name, m, len(val), rc, info.value, nt, nb * (nb + 1) // 2,
          8.0 * 16384 * (2 * (nt + 1) + nb) / 2 ** 30, 2 * 8.0 * (nb * 128.0) ** 2 / 2 ** 30, nl, ms.value, t1 - t0, res), flush=True)


rng = np.random.RandomState(3)
for m, band in ((20000, 40), (200000, 40)):
    diags = [rng.uniform(-1, 1, m - k) for k in range(1, band + 1)]
    off = sp.diags(diags, [-k for k in range(1, band + 1)], shape=(m, m), format="csr")
    A = off + off.T
    A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)
    run("band %d" % band, A.tocsc())
m, link = 100048, 48
blocks = sp.block_diag([sp.csr_matrix(np.tril(rng.uniform(-1, 1, (10, 10)), -1)) for _ in range((m - link) // 10)], format="csr")
low = sp.vstack([sp.hstack([blocks, sp.csr_matrix((m - link, link))]), sp.csr_matrix(rng.uniform(-0.01, 0.01, (link, m)))]).tocsr()
low = sp.tril(low, -1)
A = low + low.T
A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)
run("arrow (10-blocks + 48 rows)", A.tocsc())
