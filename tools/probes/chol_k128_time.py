"""factorisation time (HIP events, mean of 20) of dense n x n matrices: the blocked Cholesky with its panel / update products as
64-row tiles straight from global memory (default) or through the general GEMM kernel (HDM_CHOL_K128=0); run once per setting"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hdsdp_amd import api
lib = api.load_library()
lib.HMiCholEnvelopeProbe.restype = C.c_int
for n in (512, 1024, 2000, 4000, 8000):
    a, b = C.c_double(0), C.c_double(0)
    rc = lib.HMiCholEnvelopeProbe(n, n // 128 + 1, 20, C.byref(a), C.byref(b))
    print("HDM_CHOL_K128=%s  n = %5d: %.3f ms per factorisation (rc %d)" % (os.environ.get("HDM_CHOL_K128", "1"), n, a.value, rc), flush=True)
