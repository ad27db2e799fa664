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
cone.use_sweep_copy(0)
t_full = t(lambda: cone.check_is_interior(1.0, y1))
t0 = time.perf_counter(); cone.use_sweep_copy(1); t_build = time.perf_counter() - t0
used, vals, pos = cone.sweep_info()
t_zs = t(lambda: cone.check_is_interior(1.0, y1))
dy = np.cos(0.7 * np.arange(m) + 0.2)
t_ratio_zs = t(lambda: cone.ratio_test(0.0, dy, 0.0))
cone.use_sweep_copy(0)
t_ratio = t(lambda: cone.ratio_test(0.0, dy, 0.0))
cone.use_sweep_copy(1)
kkt.build_up(0)
t_cor = t(lambda: kkt.build_up(api.KKT_TYPE_CORRECTOR))
low = m * n * (n + 1) / 2 * 8          # lower triangles actually read
full = m * n * n * 8                  # whole stored squares
print("S assembly: %.2f ms over the lower triangles of A (%.1f GB) -> %.2f TB/s" % ((t_full - t_zero) * 1e3, low / 1e9, low / (t_full - t_zero) / 1e12))
print("corrector build: %.2f ms (S^-1, S^-2 by GEMM + one pass over A: %.1f GB lower / %.1f GB stored) -> >= %.2f TB/s" % (
    t_cor * 1e3, low / 1e9, full / 1e9, low / t_cor / 1e12))
zb = vals * 8 + pos / 1024 * 192
print("S assembly from the zero-suppressed copy (%.1f %% of the positions are non-zero, %.1f GB, built in %.2f s): %.2f ms -> %.2f TB/s of its own bytes, %.2fx the dense sweep" % (
    100.0 * vals / max(pos, 1), zb / 1e9, t_build, (t_zs - t_zero) * 1e3, zb / (t_zs - t_zero) / 1e12, (t_full - t_zero) / (t_zs - t_zero)))
print("ratio test (dS sweep + Lanczos): %.2f ms dense, %.2f ms zero-suppressed" % (t_ratio * 1e3, t_ratio_zs * 1e3))
