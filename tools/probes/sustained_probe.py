"""does the fp64 MFMA rate hold when the register-only loop runs as long as the real kernels (80-120 ms per launch)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdsdp_amd import api
lib = api.load_library()
lib.HMiDeviceInit(0)
for mode in (300, 0):
    for iters in (4000, 40000, 400000, 1600000):
        t = time.time()
        v = lib.HMiMfmaIssueProbe(mode, 2, iters)
        dt = time.time() - t
        print(f"mode {mode} (2 waves/SIMD), {iters:8d} iterations: {v:6.2f} TFLOP/s = {v / 78.6:.3f} of 78.6   (two launches, {dt * 1e3:.0f} ms wall)", flush=True)
# back to back: ten long launches in a row
for rep in range(6):
    v = lib.HMiMfmaIssueProbe(300, 2, 800000)
    print(f"rep {rep}: {v:6.2f} TFLOP/s", flush=True)
