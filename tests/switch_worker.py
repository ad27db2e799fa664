"""child process of tests/test_gpu_switches.py: a fixed set of numbers from every device path, printed as JSON, under whatever
environment switches the parent set (the library reads its switches once per process)"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from util import load_golden, y_of  # noqa: E402
from hdsdp_amd import api  # noqa: E402

out = {}
shards = int(os.environ.get("SWITCH_WORKER_SHARDS", "1"))
if shards > 1:
    api.set_devices([0] * shards, shard_min_dim=32)      # (chain16's 5 x 5 blocks stay plain cones: its operator stays sparse)
for name in ("syn100", "gpp100_B", "theta1_A", "mix40_A", "syn640x24", "syn2304x6"):
    if name == "syn640x24":     # several 128-tiles per side: the tiled kernels, two congruence batches, many Gram splits
        n, m, Rd, tau = 640, 24, -800.0, 1.0
        y = 0.03 * np.cos(0.9 * np.arange(m))
        cone = api.SDPCone.synthetic(n, m)
    elif name == "syn2304x6":   # past 2048: the sixteen-trip form of the large-block Lanczos launch, 18 tiles per side
        n, m, Rd, tau = 2304, 6, -3000.0, 1.0
        y = 0.03 * np.cos(0.9 * np.arange(m))
        cone = api.SDPCone.synthetic(n, m)
    else:
        g = load_golden(name)
        n, m = int(g["dims"][0]), int(g["dims"][1])
        Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
        cone = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"]) if "csc_beg" in g else api.SDPCone.synthetic(n, m)
    print("switch_worker:", name, file=sys.stderr, flush=True)
    cone.set_start(Rd)
    assert cone.check_is_interior(tau, y)
    kkt = api.KKT(m, [cone])
    r = {"logdet": cone.log_barrier(tau)}
    kkt.build_up(api.KKT_TYPE_HOMOGENEOUS)
    ex = kkt.export()
    msk = np.triu(np.ones((m, m), dtype=bool))
    r["M"] = kkt.M[msk].tolist()
    for k in ("ASinv", "ASinvRdSinv", "ASinvCSinv"):
        r[k] = np.asarray(ex[k]).tolist()
    r["scal"] = [ex["CSinv"], ex["CSinvCSinv"], ex["CSinvRdSinv"], ex["TraceSinv"]]
    if name != "mix40_A":       # (the five-class block has a constraint that is zero: its Schur matrix is singular by construction)
        kkt.factorize()
        r["d2"] = kkt.solve(ex["ASinv"]).tolist()
    kkt.build_up(api.KKT_TYPE_CORRECTOR)
    r["cor"] = np.asarray(kkt.export()["ASinvRdSinv"]).tolist()
    dy = 0.3 * np.cos(0.7 * np.arange(m) + 0.2)
    r["step1"] = cone.ratio_test(0.0, dy, 0.0)
    r["step2"] = cone.ratio_test(0.1, 0.5 * dy, 0.2)       # warm-started
    ok = cone.check_is_interior(tau, y + 0.25 * min(r["step1"], 1.0) * dy)   # a point on the last-but-one direction's line family
    r["trial"] = [bool(ok), cone.log_barrier(tau) if ok else 0.0]
    r["path"] = cone.path
    out[name] = r
    kkt.destroy(); cone.destroy()
# a many-block instance whose operator comes up sparse (aggregated CSC pattern, envelope, RCM)
prob = api.read_sdpa(os.path.join(HERE, "golden", "chain16.dat-s"))
g = load_golden("chain16_A")
cones = [api.SDPCone.from_csc(b["n"], prob["m"], b["beg"], b["idx"], b["val"], iCone=k) for k, b in enumerate(prob["blocks"])]
for c in cones:
    c.set_start(float(g["Rd"][0]))
    assert c.check_is_interior(float(g["tau"][0]), y_of(g))
kkt = api.KKT(prob["m"], cones)
kkt.build_up(api.KKT_TYPE_INFEASIBLE)
kkt.factorize()
out["chain16"] = {"sparse": bool(kkt.is_sparse), "x": kkt.solve(g["b"]).tolist()}
kkt.destroy()
for c in cones:
    c.destroy()
print("SWITCH_WORKER_JSON " + json.dumps(out))
