"""Un-extrapolated anchors of the CPU baseline: FULL builds of the metric's unit of work
[chol(S) -> HKKTBuildUp(INFEASIBLE) -> HKKTFactorize -> 3 x HKKTSolve] at n = m in {200, 400} on both sides --
the compiled reference on one host core (oracle/_ref/ref_dump bench; falls back to the plain-C port) and the device
path through the C ABI -- plus a check that both end in the same step (checksum of the second solve)."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from hdsdp_amd import api

sizes = [int(a) for a in sys.argv[1:]] or [200, 400]
ref = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
for n in sizes:
    m = n
    cone = api.SDPCone.synthetic(n, m)
    kkt = api.KKT(m, [cone])           # host mirror on: the reference boundary, M crosses PCIe
    cone.set_start(-10.0 * n)
    y, b = np.zeros(m), cone.traces()

    def step():
        assert cone.check_is_interior(1.0, y)
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        kkt.factorize()
        e = kkt.export()
        return kkt.solve(b), kkt.solve(e["ASinv"]), kkt.solve(e["ASinvRdSinv"])
    for _ in range(2):
        sol = step()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        sol = step()
    gpu_s = (time.perf_counter() - t0) / reps
    path = {0: "gemm", 1: "rank-one", 2: "sparse gather"}[cone.path]
    kkt.destroy(); cone.destroy()
    kind, t = "reference", None
    if os.path.exists(ref):
        out = subprocess.run([ref, "-", "bench", str(n), str(m), str(-10.0 * n), "1.0", "0.0"], capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode == 0 and line:
            t = json.loads(line[-1])
    if t is None:
        import oracle_py
        kind, t = "port", oracle_py.bench_sample(n, m)
    cpu_s = t["chol_s"] + t["buildup_s"] + t["factor_s"] + t["solve3_s"]
    print(json.dumps({"n": n, "m": m, "device_path": path, "device_ms": round(gpu_s * 1e3, 3),
                      "device_it_per_s": round(1.0 / gpu_s, 2), "cpu_kind": kind, "cpu_cores": 1,
                      "cpu_s": round(cpu_s, 3), "cpu_it_per_s": round(1.0 / cpu_s, 4),
                      "cpu_stages_s": {k: round(v, 4) for k, v in t.items() if k.endswith("_s")},
                      "speedup": round(cpu_s / gpu_s, 1),
                      "checksum_device_sum_d2": float(np.sum(sol[1])),
                      "checksum_cpu_sum_d2": t.get("sum_d2")}), flush=True)
