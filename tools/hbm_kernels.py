#!/usr/bin/env python3
"""achieved HBM rate of the two memory-bound passes over the resident constraint data at n=m=2000:
S assembly with y != 0 (hdm_sym_combine_kernel) and the corrector's <A_i, S^-1>, <A_i, S^-2> (hdm_sym_dot2_kernel)"""
import os, sys, time
os.environ["HDSDP_MI355X_AFFINE_S"] = "0"   # every call assembles (the engine would otherwise recognise the repeated point)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdsdp_amd import api
n = m = 2000
cone = api.SDPCone.synthetic(n, m)
kkt = api.KKT(m, [cone], host_mirror=False)
cone.set_start(-10.0 * n)
y0, y1 = np.zeros(m), 1e-3 * np.cos(np.arange(m))
def t(f, reps=3):
    f(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps
t_zero = t(lambda: cone.check_is_interior(1.0, y0))
t_full = t(lambda: cone.check_is_interior(1.0, y1))
kkt.build_up(0)
t_cor = t(lambda: kkt.build_up(api.KKT_TYPE_CORRECTOR))
low = m * n * (n + 1) / 2 * 8          # lower triangles actually read
full = m * n * n * 8                  # whole stored squares
print("S assembly: %.2f ms over the lower triangles of A (%.1f GB) -> %.2f TB/s" % ((t_full - t_zero) * 1e3, low / 1e9, low / (t_full - t_zero) / 1e12))
print("corrector build: %.2f ms (S^-1, S^-2 by GEMM + one pass over A: %.1f GB lower / %.1f GB stored) -> >= %.2f TB/s" % (
    t_cor * 1e3, low / 1e9, full / 1e9, low / t_cor / 1e12))
