"""device-resident interior check (assemble S at y = 0: one pass over C, then the blocked Cholesky) at a few dimensions: time per
check.  Run under HDM_CHOL_LOOKAHEAD=0 / 1 (and HDM_GRAPHS=0 / 1) for the A/B."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hdsdp_amd import api
out = []
for n in (512, 1000, 2000, 4000):
    m = 4
    cone = api.SDPCone.synthetic(n, m)
    cone.set_start(-10.0 * n)
    y = np.zeros(m)
    taus = (1.0, float(np.nextafter(1.0, 2.0)))
    for k in range(6):
        assert cone.check_is_interior(taus[k & 1], y)
    reps = 40
    t0 = time.perf_counter()
    for k in range(reps):
        cone.check_is_interior(taus[k & 1], y)
    out.append("n=%d %.3f ms" % (n, (time.perf_counter() - t0) / reps * 1e3))
    cone.destroy()
print({k: os.environ.get(k) for k in ("HDM_CHOL_LOOKAHEAD", "HDM_GRAPHS")}, "  ".join(out), flush=True)
