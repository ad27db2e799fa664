"""wall time of the device ratio test (Lanczos) and of one interior check at n = 2000 (set HDSDP_MI355X_RATIO_DEBUG=1 for the step counts)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdsdp_amd import api
n, m = 2000, 64
cone = api.SDPCone.synthetic(n, m)
cone.set_start(-10.0 * n)
y = np.zeros(m)
assert cone.check_is_interior(1.0, y)
for scale in (40.0, 4000.0, 40000.0):
    dy = scale * np.cos(0.7 * np.arange(m) + 0.2)
    for rep in range(3):
        t0 = time.perf_counter(); s = cone.ratio_test(0.0, dy, 0.0); t1 = time.perf_counter()
        ok = cone.check_is_interior(1.0, y); t2 = time.perf_counter()
        print("dy scale %g: ratio test %.2f ms (step %.6e)   interior check %.2f ms" % (scale, (t1 - t0) * 1e3, s, (t2 - t1) * 1e3))
