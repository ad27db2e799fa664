import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from util import load_golden, y_of
from hdsdp_amd import api
def T(label, f):
    t0 = time.perf_counter(); r = f(); print("%-34s %8.3f ms" % (label, (time.perf_counter() - t0) * 1e3)); return r
lib = T("load_library", api.load_library)
T("HMiDeviceSynchronize (ctx)", lib.HMiDeviceSynchronize)
for name in ("mcp100_A", "gpp100_A", "theta1_A"):
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    print("==", name)
    cone = T("cone create", lambda: api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"]))
    kkt = T("kkt create", lambda: api.KKT(m, [cone]))
    cone.set_start(Rd)
    T("interior check 1", lambda: cone.check_is_interior(tau, y))
    T("interior check 2", lambda: cone.check_is_interior(tau, y * 1.01))
    T("build_up 1", lambda: kkt.build_up(api.KKT_TYPE_INFEASIBLE))
    T("build_up 2", lambda: kkt.build_up(api.KKT_TYPE_INFEASIBLE))
    T("factorize 1", kkt.factorize)
    T("factorize 2", kkt.factorize)
    b = np.ones(m)
    T("solve 1", lambda: kkt.solve(b))
    T("solve 2", lambda: kkt.solve(b))
    dy = 0.3 * np.cos(np.arange(m))
    T("ratio test 1", lambda: cone.ratio_test(0.0, dy, 0.0))
    T("ratio test 2", lambda: cone.ratio_test(0.0, dy, 0.0))
    T("build corrector 1", lambda: kkt.build_up(api.KKT_TYPE_CORRECTOR))
    T("build corrector 2", lambda: kkt.build_up(api.KKT_TYPE_CORRECTOR))
    kkt.destroy(); cone.destroy()
