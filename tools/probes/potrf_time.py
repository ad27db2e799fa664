"""wall time of HMiPotrf (device matrix in, factor out, one synchronisation) at a few sizes; HDM_DIAG_SWEEP=0/1 for A/B"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdsdp_amd import api
import ctypes as C
lib = api.load_library()
for n in (100, 128, 256, 1024, 2000):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    d = torch.from_numpy(np.asfortranarray(A).T.copy()).cuda()   # column-major image of the symmetric matrix
    info = C.c_int(0)
    for _ in range(3):
        d2 = d.clone(); lib.HMiPotrf(C.c_void_p(d2.data_ptr()), n, n, C.byref(info))
    L = np.tril(d2.cpu().numpy().T)
    err = np.linalg.norm(L @ L.T - A) / np.linalg.norm(A)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.HMiPotrf(C.c_void_p(d2.data_ptr()), n, n, C.byref(info))   # (re-factoring a factor: same cost, may fail: irrelevant here)
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n}: {dt*1e6:.0f} us per HMiPotrf (alloc + load + factor + copy + sync), residual {err:.1e}, HDM_DIAG_SWEEP={os.environ.get('HDM_DIAG_SWEEP','1')}", flush=True)
for v in (0, 1):
    print(f"diagonal-block kernel variant {v}: {lib.HMiDiagBlockProbe(v, 50):.1f} us (HIP events, 50 launches)")
