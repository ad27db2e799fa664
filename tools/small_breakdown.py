"""Per-call wall time of one Phase-A pass on the small BASELINE configs (launch-latency bound)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import load_golden, y_of
from hdsdp_amd import api

for name in ("theta1_A", "mcp100_A", "gpp100_A", "syn64"):
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    if "csc_beg" in g:
        cone = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    else:
        cone = api.SDPCone.synthetic(n, m)
    for mirror in (True, False):
        kkt = api.KKT(m, [cone], host_mirror=mirror)
        cone.set_start(Rd)
        b = np.asarray(g["b"], dtype=np.float64)
        acc = np.zeros(6)
        reps = 50
        for it in range(reps + 5):
            t = [time.perf_counter()]
            cone.check_is_interior(tau, y); t.append(time.perf_counter())
            kkt.build_up(api.KKT_TYPE_INFEASIBLE); t.append(time.perf_counter())
            kkt.factorize(); t.append(time.perf_counter())
            e = kkt.export(); t.append(time.perf_counter())
            kkt.solve(b); t.append(time.perf_counter())
            kkt.solve(e["ASinv"]); kkt.solve(e["ASinvRdSinv"]); t.append(time.perf_counter())
            if it >= 5:
                acc += np.diff(t)
        acc *= 1e6 / reps
        print("%-9s n=%3d m=%3d path=%d mirror=%d  interior %.0f  build %.0f  factor %.0f  export %.0f  solve1 %.0f  solve2+3 %.0f  total %.0f us"
              % (name, n, m, cone.path, mirror, *acc, acc.sum()), flush=True)
        kkt.destroy()
    cone.destroy()
