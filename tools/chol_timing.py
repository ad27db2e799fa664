"""Time HFpLinsysPsdCheck (upload + blocked Cholesky) and one solve at a few sizes through the C ABI."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hdsdp_amd import api
for n in (100, 500, 2000, 4000):
    rng = np.random.default_rng(n)
    G = rng.uniform(-1, 1, (n, n))
    S = np.triu(G @ G.T / n + np.eye(n))
    b = rng.uniform(-1, 1, n)
    ls = api.LinSys(n)
    for _ in range(3):
        assert ls.psd_check(S)
        ls.solve(b)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ls.psd_check(S)
    t1 = time.perf_counter()
    for _ in range(reps):
        ls.solve(b)
    t2 = time.perf_counter()
    print("n=%d  factor (incl. %.1f MB upload) %.3f ms   solve %.3f ms" % (n, n * n * 8 / 1e6, (t1 - t0) / reps * 1e3, (t2 - t1) / reps * 1e3), flush=True)
    ls.destroy()
