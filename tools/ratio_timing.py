#!/usr/bin/env python3
"""wall time of the device ratio test (Lanczos) and of one interior check at n = 2000"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdsdp_amd import api
n, m = 2000, 64
cone = api.SDPCone.synthetic(n, m)
cone.set_start(-10.0 * n)
y = np.zeros(m)
assert cone.check_is_interior(1.0, y)
dy = 40.0 * np.cos(0.7 * np.arange(m) + 0.2)
for rep in range(4):
    t0 = time.perf_counter(); s = cone.ratio_test(0.0, dy, 0.0); t1 = time.perf_counter()
    ok = cone.check_is_interior(1.0, y); t2 = time.perf_counter()
    print("ratio test %.2f ms (step %.6f)   interior check %.2f ms" % ((t1 - t0) * 1e3, s, (t2 - t1) * 1e3))
