"""BASELINE configs 1-3 (theta1, mcp100, gpp100): one Phase-A pass per iteration on the device -- call by call through the
reference's operator surface, and (rank-one blocks) as the ONE fused launch of csrc/small.hip -- beside the plain-C oracle
port on one host core (same call sequence).  `measure(name)` returns a dict; bench.py appends these to its JSON line.

Timing rule (round 4): a loop is warmed up BY TIME (at least 0.3 s of passes: on a fresh lease the first ~75 ms of passes run
at the idle clock and pay first-touch costs; a fixed count of three passes had left that inside the mean and the driver's
record read 0.55 ms where every other run read 0.17) and reports the MEDIAN of the per-pass wall times, with the mean and the
fastest pass beside it; the fused kernel's own in-kernel time comes from its stamps (fused_kernel_phases_us)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _timed_passes(fn, reps, warm_s=0.3):
    """per-pass wall times (ms) of `reps` calls of fn after at least `warm_s` seconds (and three calls) of warm-up"""
    t_end = time.perf_counter() + warm_s
    k = 0
    while k < 3 or time.perf_counter() < t_end:
        fn(); k += 1
    ts = np.empty(reps)
    for i in range(reps):
        t0 = time.perf_counter()
        fn()
        ts[i] = (time.perf_counter() - t0) * 1e3
    return ts


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
    ref = step()
    ts = _timed_passes(step, reps)
    out = {"config": name.split("_")[0], "n": n, "m": m, "path": {0: "gemm", 1: "rank-one", 2: "sparse gather"}[cone.path],
           "call_by_call_ms_per_pass": round(float(np.median(ts)), 4), "call_by_call_mean_ms": round(float(ts.mean()), 4),
           "call_by_call_launches_per_pass": "about 40 (6 host synchronisations)", "timing": "median of %d passes after >= 0.3 s of warm-up" % reps}
    # algorithmic bytes of a pass: S^-1 (8 n^2) + half of M (4 m^2) + the coefficient data
    data_bytes = 8 * n * n + 4 * m * m + 12 * int(g["csc_beg"][-1])
    out["algorithmic_bytes"] = data_bytes
    if kkt.phase_a_eligible():
        ok, ld, d1, d2, d3 = kkt.phase_a(tau, y, b)
        assert ok and np.linalg.norm(d2 - ref[1]) <= 1e-9 * np.linalg.norm(ref[1])
        tf = _timed_passes(lambda: kkt.phase_a(tau, y, b), reps * 4)
        fused_ms = float(np.median(tf))
        out["fused_mean_ms"], out["fused_min_ms"] = round(float(tf.mean()), 4), round(float(tf.min()), 4)
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
