"""diagnostic: what the FIRST HKKTBuildUp of a headline-size block costs beyond a later one (work buffers are allocated there)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hdsdp_amd import api
n = m = 2000
t0 = time.perf_counter(); cone = api.SDPCone.synthetic(n, m); t1 = time.perf_counter()
kkt = api.KKT(m, [cone], host_mirror=True)
cone.set_start(-10.0 * n)
assert cone.check_is_interior(1.0, np.zeros(m))
t2 = time.perf_counter()
ts = []
for k in range(3):
    t = time.perf_counter(); kkt.build_up(api.KKT_TYPE_INFEASIBLE); ts.append(time.perf_counter() - t)
print("create %.2f s, to first build %.2f s, builds: %s ms" % (t1 - t0, t2 - t1, ["%.1f" % (x * 1e3) for x in ts]), flush=True)
