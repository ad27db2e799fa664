"""step time at n=m=2000 with and without the reference boundary's host round trip of M (kktMatElem mirror)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdsdp_amd import api
n = m = 2000
cone = api.SDPCone.synthetic(n, m)
cone.set_start(-10.0 * n)
y = np.zeros(m); b = cone.traces()
for mirror in (False, True):
    kkt = api.KKT(m, [cone], host_mirror=mirror)
    def step():
        assert cone.check_is_interior(1.0, y)
        kkt.build_up(0); kkt.factorize(); e = kkt.export()
        kkt.solve(b); kkt.solve(e["ASinv"]); kkt.solve(e["ASinvRdSinv"])
    step()
    t0 = time.perf_counter()
    for _ in range(3): step()
    print("host mirror %s: %.2f ms per step" % ("on " if mirror else "off", (time.perf_counter() - t0) / 3 * 1e3))
    kkt.destroy()
