"""diagnostic: the ratio tests of a golden instance (fresh, warm-started, from the checker buffer), printed to full precision;
run under HDM_LANCZOS_WHOLE=0 / 1 to compare the multi-launch and the single-launch forms"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import load_golden, y_of
from hdsdp_amd import api
for name in sys.argv[1:] or ["theta1_A", "mcp100_A", "gpp100_A", "syn64"]:
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    if "csc_beg" in g:
        cone = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    else:
        cone = api.SDPCone.synthetic(n, m)
    cone.set_start(float(g["Rd"][0]))
    assert cone.check_is_interior(float(g["tau"][0]), y_of(g))
    out = []
    for tag in ("1", "2"):
        if "rt_step" + tag in g:
            par = g["rt_par" + tag]
            st = cone.ratio_test(float(par[0]), g["rt_dy" + tag], float(par[1]))
            out.append("%s: %.17g (ref %.17g, rel %.2e)" % (tag, st, float(g["rt_step" + tag][0]), abs(st - float(g["rt_step" + tag][0])) / abs(st)))
    print(name, os.environ.get("HDM_LANCZOS_WHOLE", "1"), " ".join(out))
    cone.destroy()
