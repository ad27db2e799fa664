"""diagnostic: wall time per ratio test / interior check / barrier on the small goldens (run under rocprofv3 --kernel-trace --stats
for the kernel breakdown)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import load_golden, y_of
from hdsdp_amd import api
for name in sys.argv[1:] or ["mcp100_A", "gpp100_A", "theta1_A", "truss1_A"]:
    g = load_golden(name)
    if "csc_beg" not in g:
        continue
    n, m = int(g["dims"][0]), int(g["dims"][1])
    cone = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    cone.set_start(float(g["Rd"][0]))
    tau, y = float(g["tau"][0]), y_of(g)
    assert cone.check_is_interior(tau, y)
    dy = 0.3 * np.cos(0.7 * np.arange(m) + 0.2)
    for _ in range(3):
        cone.ratio_test(0.0, dy, 0.0)
    reps = 100
    t0 = time.perf_counter()
    for _ in range(reps):
        cone.ratio_test(0.0, dy, 0.0)
    t1 = time.perf_counter()
    for k in range(reps):
        cone.check_is_interior(tau, y * (1.0 + 1e-9 * k))
    t2 = time.perf_counter()
    for _ in range(reps):
        cone.log_barrier(tau)
    t3 = time.perf_counter()
    print("%-10s n=%d m=%d path=%d: ratio test %.1f us, interior check %.1f us, barrier %.1f us" %
          (name, n, m, cone.path, (t1 - t0) / reps * 1e6, (t2 - t1) / reps * 1e6, (t3 - t2) / reps * 1e6), flush=True)
    cone.destroy()
