"""stage split of one step at n=m=2000 (HMiGetStageTimes + wall clock of the replicated parts)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdsdp_amd import api
n = m = 2000
lib = api.load_library()
cone = api.SDPCone.synthetic(n, m)
kkt = api.KKT(m, [cone], host_mirror=False)
cone.set_start(-10.0 * n)
y = np.zeros(m)
for rep in range(3):
    t0 = time.perf_counter(); assert cone.check_is_interior(1.0, y); t1 = time.perf_counter()
    kkt.build_up(0); t2 = time.perf_counter()
    kkt.factorize(); t3 = time.perf_counter()
    b = cone.traces(); x = kkt.solve(b); t4 = time.perf_counter()
    st = kkt.stage_times_ms()
    print("rep %d: interior check %.2f  build %.2f [invert L %.2f, congruence %.2f, gram %.2f, extract %.2f]  factor M %.2f  one solve %.2f ms"
          % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, st[0], st[1], st[2], st[3], (t3 - t2) * 1e3, (t4 - t3) * 1e3))
