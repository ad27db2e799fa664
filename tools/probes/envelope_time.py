"""factorisation of a block-banded Schur matrix: dense vs on its block envelope (HdmChol::set_envelope)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hdsdp_amd import api
import ctypes as C
lib = api.load_library()
print("# m, band (128-blocks below the diagonal block): Cholesky of the dense device matrix vs on the pattern's block envelope, ms (HIP events)")
for n, band in ((4096, 2), (16384, 2), (16384, 8), (32768, 2), (49152, 2)):
    a, b = np.zeros(1), np.zeros(1)
    rc = lib.HMiCholEnvelopeProbe(n, band, 3 if n <= 16384 else 1, a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)))
    print(f"m={n:6d} band={band}: dense {a[0]:9.1f} ms   envelope {b[0]:8.1f} ms   (rc {rc})", flush=True)
