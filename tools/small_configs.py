#!/usr/bin/env python3
"""BASELINE configs 1-3 (theta1, mcp100, gpp100): one Phase-A pass per iteration on the device, timed, beside the
plain-C oracle port on one host core (same call sequence).  These problems are launch-latency bound on a GPU."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from util import load_golden, y_of
from hdsdp_amd import api
import oracle_py

for name in ("theta1_A", "mcp100_A", "gpp100_A"):
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    cone = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    kkt = api.KKT(m, [cone])
    cone.set_start(Rd)
    b = g["b"]

    def step():
        assert cone.check_is_interior(tau, y)
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        kkt.factorize()
        e = kkt.export()
        return kkt.solve(b), kkt.solve(e["ASinv"]), kkt.solve(e["ASinvRdSinv"])
    for _ in range(3):
        step()
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        step()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    path = {0: "gemm", 1: "rank-one", 2: "sparse gather"}[cone.path]
    # oracle port, one core
    blk = oracle_py.Block(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    t0 = time.perf_counter(); creps = 5
    for _ in range(creps):
        S = blk.assemble_S(tau, y, Rd); Lf, info = blk.factor(S); Sinv = blk.inverse(Lf)
        k = blk.kkt_build(Sinv, Rd, 0)
        for rhs in (b, k["ASinv"], k["ASinvRdSinv"]):
            oracle_py.pcg_solve(k["M"], rhs)
    cpu_ms = (time.perf_counter() - t0) / creps * 1e3
    data_bytes = 8 * (n * n + m * m // 2) + 12 * int(g["csc_beg"][-1])
    print("%-9s n=%3d m=%3d path=%-13s device %.3f ms/iteration (%.2f MB of operands: %.2f GB/s)   oracle port, 1 core: %.3f ms"
          % (name, n, m, path, gpu_ms, data_bytes / 1e6, data_bytes / gpu_ms / 1e6, cpu_ms))
    kkt.destroy(); cone.destroy(); blk.close()
