#!/usr/bin/env python3
"""BASELINE configs 1-3 (theta1, mcp100, gpp100): one Phase-A pass per iteration on the device -- call by call through the
reference's operator surface, and (rank-one blocks) as the ONE fused launch of csrc/small.hip -- beside the plain-C oracle
port on one host core (same call sequence).  `measure(name)` returns a dict; bench.py appends these to its JSON line."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def measure(name, reps=50, cpu=True):
    from util import load_golden, y_of
    from hdsdp_amd import api
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    cone = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    kkt = api.KKT(m, [cone], host_mirror=False)
    cone.set_start(Rd)
    b = np.asarray(g["b"], dtype=np.float64)

    def step():
        assert cone.check_is_interior(tau, y)
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        kkt.factorize()
        e = kkt.export()
        return kkt.solve(b), kkt.solve(e["ASinv"]), kkt.solve(e["ASinvRdSinv"])
    for _ in range(3):
        ref = step()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    calls_ms = (time.perf_counter() - t0) / reps * 1e3
    out = {"config": name.split("_")[0], "n": n, "m": m, "path": {0: "gemm", 1: "rank-one", 2: "sparse gather"}[cone.path],
           "call_by_call_ms_per_pass": round(calls_ms, 4), "call_by_call_launches_per_pass": "about 40 (6 host synchronisations)"}
    # algorithmic bytes of a pass: S^-1 (8 n^2) + half of M (4 m^2) + the coefficient data
    data_bytes = 8 * n * n + 4 * m * m + 12 * int(g["csc_beg"][-1])
    out["algorithmic_bytes"] = data_bytes
    if kkt.phase_a_eligible():
        for _ in range(3):
            ok, ld, d1, d2, d3 = kkt.phase_a(tau, y, b)
        assert ok and np.linalg.norm(d2 - ref[1]) <= 1e-9 * np.linalg.norm(ref[1])
        t0 = time.perf_counter()
        for _ in range(reps * 4):
            kkt.phase_a(tau, y, b)
        fused_ms = (time.perf_counter() - t0) / (reps * 4) * 1e3
        out["fused_kernel_phases_us"] = dict(zip(("assemble_S", "factor_and_invert_S", "Sinv", "schur_build", "factor_and_invert_M", "solves"),
                                                  [round(float(v) * 1e3, 1) for v in kkt.stage_times_ms()[:6]]))
        out["fused_kernel_shader_clock_GHz"] = round(float(kkt.stage_times_ms()[6]), 3)
        out.update({"fused_ms_per_pass": round(fused_ms, 4), "fused_launches_per_pass": 1, "fused_host_syncs_per_pass": 1,
                    "fused_achieved_GBps": round(data_bytes / fused_ms / 1e6, 3)})
    if cpu:
        import oracle_py
        blk = oracle_py.Block(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
        t0 = time.perf_counter(); creps = 5
        for _ in range(creps):
            S = blk.assemble_S(tau, y, Rd); Lf, info = blk.factor(S); Sinv = blk.inverse(Lf)
            k = blk.kkt_build(Sinv, Rd, 0)
            for rhs in (b, k["ASinv"], k["ASinvRdSinv"]):
                oracle_py.pcg_solve(k["M"], rhs)
        out["cpu_port_1core_ms_per_pass"] = round((time.perf_counter() - t0) / creps * 1e3, 4)
        blk.close()
    kkt.destroy(); cone.destroy()
    return out


if __name__ == "__main__":
    for nm in ("theta1_A", "mcp100_A", "gpp100_A"):
        print(measure(nm), flush=True)
