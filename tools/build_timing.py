"""per-role kernel time of ONE HKKTBuildUp at n=m=2000 (no factorisation: usable with the timing-only kernel
ablations selected by HDM_VAR, whose results are wrong)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdsdp_amd import api
n = m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
lib = api.load_library()
cone = api.SDPCone.synthetic(n, m)
kkt = api.KKT(m, [cone], host_mirror=False)
cone.set_start(-10.0 * n)
assert cone.check_is_interior(1.0, np.zeros(m))
kkt.build_up(0)
lib.HMiSetKernelTiming(1)
for _ in range(2):
    kkt.build_up(0)
lib.HMiSetKernelTiming(0)
kms, kfl = np.zeros(5), np.zeros(5)
kln = np.zeros(5, dtype=np.int64)
lib.HMiGetKernelTiming(kms.ctypes.data_as(C.POINTER(C.c_double)), kfl.ctypes.data_as(C.POINTER(C.c_double)),
                       kln.ctypes.data_as(C.POINTER(C.c_int64)))
print("HDM_VAR=%s  K1 %.2f  K2 %.2f (+ diagonal tiles %.2f)  gram %.2f ms per build" % (os.environ.get("HDM_VAR", "-"), kms[1] / 2, kms[2] / 2, kms[4] / 2, kms[3] / 2))
