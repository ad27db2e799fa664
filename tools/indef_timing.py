"""Time the Schur system's symmetric-indefinite fallback (lu.hip) next to the regular Cholesky path, through the C ABI
(host matrix in, host solution out, like HFpLinsysNumeric / HFpLinsysSolve)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hdsdp_amd import api

for m in (500, 2000, 8000):
    rng = np.random.default_rng(m)
    G = rng.uniform(-1, 1, (m, m))
    A = 0.5 * (G + G.T)
    P = np.triu(A @ A.T / m + np.eye(m))
    A = np.triu(A)
    b = rng.uniform(-1, 1, m)
    row = []
    for tag, Mx in (("cholesky", P), ("pivoted", A)):
        ls = api.LinSys(m, api.HDSDP_LINSYS_DENSE_ITERATIVE)
        ls.numeric(Mx); ls.solve(b)
        t0 = time.perf_counter(); ls.numeric(Mx); t1 = time.perf_counter(); ls.solve(b); t2 = time.perf_counter()
        row.append("%s: factor %.2f ms (incl. %.0f MB upload), solve %.2f ms" % (tag, (t1 - t0) * 1e3, m * m * 8 / 1e6, (t2 - t1) * 1e3))
        ls.destroy()
    print("m=%d  " % m + "   ".join(row), flush=True)
