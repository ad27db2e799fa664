"""diagnostic (HDM_POISON=1): primal recovery on the syn2000x32 golden after as little as possible, under several switches"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from util import load_golden, y_of
    from hdsdp_amd import api
    g = load_golden("syn2000x32")
    n, m = int(g["dims"][0]), int(g["dims"][1])
    cone = api.SDPCone.synthetic(n, m)
    cone.set_start(float(g["Rd"][0]))
    ok = cone.check_is_interior(float(g["tau"][0]), y_of(g))
    if sys.argv[1] == "ratio":
        par = g["rt_par1"]; cone.ratio_test(float(par[0]), g["rt_dy1"], float(par[1]))
    X = cone.get_primal(float(g["pr_mu"][0]), g["pr_y"], g["pr_dy"])
    print(sys.argv[1], dict((k, os.environ.get(k)) for k in ("HDSDP_MI355X_ZS", "HDSDP_MI355X_AFFINE_S", "HDM_GRAPHS")), "interior", ok,
          "X", None if X is None else bool(np.isfinite(X).all()), "sweep copy", cone.sweep_info()[0], flush=True)
else:
    for mode in ("plain", "ratio"):
        for extra in ({}, {"HDSDP_MI355X_ZS": "0"}, {"HDSDP_MI355X_AFFINE_S": "0"}, {"HDM_GRAPHS": "0"}):
            env = dict(os.environ, HDM_POISON="1", **extra)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=env, capture_output=True, text=True)
            print((r.stdout + r.stderr).strip().splitlines()[-2:], flush=True)
