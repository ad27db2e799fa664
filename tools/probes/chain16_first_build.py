"""diagnostic: the driver's call order on the blocks of chain16 -- create, scale, interior check, FIRST build = INFEASIBLE"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hdsdp_amd import api
lib = api.load_library()
prob = api.read_sdpa(os.path.join(ROOT, "tests", "golden", "chain16.dat-s"))
m = prob["m"]
y = np.zeros(m)
which = [int(v) for v in sys.argv[1:]] or list(range(len(prob["blocks"])))
cones = []
for k in which:
    blk = prob["blocks"][k]
    c = api.SDPCone.from_csc(blk["n"], m, blk["beg"], blk["idx"], blk["val"], iCone=k)
    c.scal_by_constant(1e-5)
    c.set_start(-1e7)
    assert c.check_is_interior(1.0, y)
    cones.append(c)
    print("cone", k, "n", blk["n"], "path", c.path, flush=True)
kkt = api.KKT(m, cones)
print("build", flush=True)
kkt.build_up(api.KKT_TYPE_INFEASIBLE)
assert lib.HMiDeviceSynchronize() == 0
kkt.destroy()
print("ALL_OK")
