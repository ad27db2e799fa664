"""diagnostic: every cone slot on the blocks of tests/golden/chain16.dat-s, synchronising after each call"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hdsdp_amd import api
lib = api.load_library()
prob = api.read_sdpa(os.path.join(ROOT, "tests", "golden", sys.argv[1] if len(sys.argv) > 1 else "chain16.dat-s"))
m = prob["m"]
y = 0.1 * np.sin(1.7 * (np.arange(m) + 1))
def step(what, f):
    print(what, flush=True)
    r = f()
    assert lib.HMiDeviceSynchronize() == 0
    return r
cones = []
for k, blk in enumerate(prob["blocks"]):
    c = step(f"create {k} n={blk['n']}", lambda: api.SDPCone.from_csc(blk["n"], m, blk["beg"], blk["idx"], blk["val"], iCone=k))
    print("   path", c.path, flush=True)
    cones.append(c)
for k, c in enumerate(cones):
    step(f"norms {k}", lambda: (c.coeff_norm(1), c.coeff_norm(2), c.obj_norm(1), c.obj_norm(2)))
    step(f"scal {k}", lambda: c.scal_by_constant(1e-1))
    c.set_start(-50.0)
    assert step(f"interior {k}", lambda: c.check_is_interior(1.0, y))
    step(f"barrier {k}", lambda: c.log_barrier(1.0))
    dy = 0.3 * np.cos(0.7 * np.arange(m) + 0.2)
    step(f"ratio {k}", lambda: c.ratio_test(0.0, dy, 0.0))
    step(f"axpy {k}", lambda: c.axpy_buffer_and_check(0.1, api.BUFFER_DUALCHECK))
    X = np.eye(c.n)
    step(f"atimesx {k}", lambda: c.a_times_x(X))
    step(f"tracecx {k}", lambda: c.trace_cx(X))
    step(f"xdots {k}", lambda: c.x_dot_s(X))
    step(f"getdual {k}", lambda: c.get_dual())
    step(f"primal {k}", lambda: c.get_primal(0.5, y, 0.01 * dy))
kkt = api.KKT(m, cones)
print("sparse", kkt.is_sparse, flush=True)
for t in (api.KKT_TYPE_HOMOGENEOUS, api.KKT_TYPE_INFEASIBLE, api.KKT_TYPE_CORRECTOR):
    step(f"build {t}", lambda: kkt.build_up(t))
step("factor", kkt.factorize)
step("solve", lambda: kkt.solve(np.ones(m)))
print("ALL_OK")
