"""diagnostic: the ingest test's sequence (theta1 three ways, then mcp100) with every result held to the golden"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import load_golden, y_of
from hdsdp_amd import api
import test_gpu_ingest as T
for name in ("theta1_A", "mcp100_A"):
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    beg, idx, val = np.asarray(g["csc_beg"]), np.asarray(g["csc_idx"]), np.asarray(g["csc_val"])
    rng = np.random.default_rng(7)
    def columns():
        for c in rng.permutation(m + 1):
            lo, hi = int(beg[c]), int(beg[c + 1])
            if hi == lo: continue
            p = rng.permutation(hi - lo)
            yield int(c), idx[lo:hi][p], val[lo:hi][p]
    for how, make in (("csc", lambda: api.SDPCone.from_csc(n, m, beg, idx, val)), ("csc64", lambda: api.SDPCone.from_csc64(n, m, beg, idx, val)),
                      ("cols", lambda: api.SDPCone.from_columns(n, m, columns()))):
        cone = make()
        st = T._state(api, cone, m, g)
        print(name, how, "path", cone.path, "err ASinvRdSinv", float(np.max(np.abs(st["ASinvRdSinv"] - g["ASinvRdSinv_inf"]))),
              "err ASinv", float(np.max(np.abs(st["ASinv"] - g["ASinv_inf"]))), "logdet", st["logdet"], float(g["logdet"][0]), flush=True)
        cone.destroy()
