#!/usr/bin/env python3
"""bench.py -- IPM iterations/sec of the Schur hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path at the fixed interior state of SURVEY.md 8(d) (y=0, tau=1,
Rd=-10n) of the synthetic dense SDP, inputs resident in HBM:
    assemble S + Cholesky(S)  ->  HKKTBuildUp(KKT_TYPE_INFEASIBLE)  ->  HKKTFactorize  ->  3 x HKKTSolve
(the Phase-A sequence interface/hdsdp_algo.c:1082-1101).

    python bench.py                         # 1 GPU, n=m=2000
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  N > 1 shards the SAME problem (strong scaling): rows cyclic over ranks,
one all-to-all + one all-reduce per step (hdsdp_amd/dist.py).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X vendor figure for dense fp64 matrix; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz


# HBM bytes per launch of the hot kernels come from the newest profiles/*_traffic.json (tools/gpu_profile.sh: separate
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command, condensed by tools/prof_summary.py --traffic;
# FETCH_SIZE x 2 for the gfx950 wide-read correction + WRITE_SIZE, KiB).  bench.py cannot run the profiler on itself, so
# the file is matched on the kernel's NAME (role and variant are template arguments): a kernel that was renamed or
# re-templated since the profile was taken yields `traffic: null` instead of a stale number.
PERSIST = os.environ.get("HDM_PERSIST", "1") != "0"   # persistent workgroups (the default) or one workgroup per tile
KERNEL_NAME = "hdm_gemm_persist_kernel" if PERSIST else "hdm_gemm_kernel"
KERNEL_SYMBOL = {1: KERNEL_NAME + "<false, true, 1, 64>", 2: KERNEL_NAME + "<false, false, 2, 64>",
                 3: KERNEL_NAME + "<true, true, 3, 64>"}


def traffic_bytes_per_launch(role):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files or os.environ.get("HDM_VAR"):
        return None, None
    try:
        with open(files[-1]) as f:
            ks = json.load(f)["kernels"]
    except Exception:
        return None, None
    hits = [v for k, v in ks.items() if KERNEL_SYMBOL[role].replace(" ", "") in k.replace(" ", "")]
    if len(hits) != 1:
        return None, os.path.basename(files[-1])
    return hits[0]["bytes_per_launch"], os.path.basename(files[-1])


def survey_flops(n, m):
    """SURVEY.md 8(d) operation count of one iteration (reference M3 count, kept for comparability)"""
    return 3.0 * m * n ** 3 + 0.5 * m * (m + 1) * n * (n + 1) + n ** 3 + m ** 3 / 3.0 + 6.0 * m * m


def executed_flops(n, m):
    """what our formulation needs on valid data: congruence n^3 per row (triangular factors), Gram over the packed index"""
    return m * 1.0 * n ** 3 + 0.5 * (m + 3) * (m + 4) * n * (n + 1) + n ** 3 + m ** 3 / 3.0 + 6.0 * m * m


def survey_equiv(n, m, world, kms, kln, steps):
    """achieved TFLOP/s if the work is priced with the reference's operation count; can exceed the
    MFMA peak because the triangular-factor congruence executes n^3, not 3 n^3, flops per constraint"""
    rows = (m + world - 1) // world
    cong_ms = (kms[1] + kms[2]) / steps
    gram_ms = kms[3] / steps
    return {"congruence": {"flops_per_unit": 3.0 * n ** 3, "unit": "constraint", "units_per_step": rows,
                           "tflops": round(3.0 * n ** 3 * rows / max(cong_ms, 1e-9) / 1e9, 2)},
            "gram": {"flops_per_unit": float(n) * (n + 1), "unit": "row pair", "units_per_step": 0.5 * m * (m + 1) / world,
                     "tflops": round(0.5 * m * (m + 1) / world * n * (n + 1) / max(gram_ms, 1e-9) / 1e9, 2)}}


def cpu_baseline(n, m, budget_cols=8):
    """Time the CPU side on this host: the real reference (oracle/_ref, kind "reference") when it was
    built, else the plain-C restatement (kind "port").  Bounded sample: the same n, `budget_cols`
    constraint matrices; extrapolated to m with the reference's own operation count."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    sample_m = budget_cols
    try:
        if os.path.exists(ref):
            out = subprocess.run([ref, "-", "bench", str(n), str(sample_m), str(-10.0 * n), "1.0", "0.0"],
                                 capture_output=True, text=True, timeout=900)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if out.returncode == 0 and line:
                t = json.loads(line[-1])
                kind = "reference"
            else:
                raise RuntimeError(out.stderr[-300:])
        else:
            raise RuntimeError("oracle/_ref not built")
    except Exception as e:  # fall back to the port
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        try:
            import oracle_py
            t = oracle_py.bench_sample(n, sample_m)
            kind = "port"
        except Exception as e2:
            return {"value": None, "unit": "it/s", "cores": 1, "kind": "unavailable",
                    "sample": f"reference: {e}; port: {e2}"}
    t_sample = t["chol_s"] + t["buildup_s"] + t["factor_s"] + t["solve3_s"]
    scale = survey_flops(n, m) / survey_flops(n, sample_m)
    t_full = t_sample * scale
    return {"value": 1.0 / t_full, "unit": "it/s", "cores": 1, "kind": kind,
            "sample": (f"n={n}, first {sample_m} of m={m} constraint matrices (same generator): "
                       f"dpotrf(S)+HKKTBuildUp+HKKTFactorize+3xHKKTSolve took {t_sample:.1f} s on 1 core; "
                       f"extrapolated x{scale:.0f} by the reference operation count 3mn^3+m(m+1)n(n+1)/2+..."),
            "sample_seconds": t_sample}


def cpu_baseline_blas3(n, m):
    """second CPU leg ("for fairness"): the same math as level-3 BLAS on the host cores this box gives us
    (oracle/blas3_baseline.py, numpy/scipy OpenBLAS), bounded sample, run in a child process"""
    script = os.path.join(ROOT, "oracle", "blas3_baseline.py")
    try:
        out = subprocess.run([sys.executable, script, str(n), str(m), "16", "192"], capture_output=True, text=True,
                             timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            raise RuntimeError(out.stderr[-300:])
        return json.loads(line[-1])
    except Exception as e:
        return {"value": None, "unit": "it/s", "cores": 0, "kind": "unavailable", "sample": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--m", type=int, default=2000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--backend", default=None, help="collective backend for N>1 (default nccl = RCCL)")
    args = ap.parse_args()

    import torch
    from hdsdp_amd import api, dist as hdist

    rank, world, local = hdist.init_process_group_from_env(args.backend)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("LOCAL_RANK", str(local))
    lib = api.load_library()
    if lib.HMiDeviceInit(local) != 0:
        raise SystemExit("no MI355X visible: bench.py has no CPU fallback")
    torch.cuda.set_device(local % torch.cuda.device_count())
    n, m = args.n, args.m

    t0 = time.time()
    cone = api.SDPCone.synthetic(n, m, rank=rank, world=world)
    ex = hdist.Exchange(cone) if world > 1 else None
    kkt = api.KKT(m, [cone], host_mirror=False)
    cone.set_start(-10.0 * n)
    y = np.zeros(m)
    b = cone.traces()
    setup_s = time.time() - t0

    def barrier():
        lib.HMiDeviceSynchronize()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    stage = {"chol_S": 0.0, "buildup": 0.0, "factor_M": 0.0, "solve3": 0.0}

    def step(timed):
        t = time.perf_counter()
        ok = cone.check_is_interior(1.0, y)
        assert ok
        t1 = time.perf_counter()
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        t2 = time.perf_counter()
        kkt.factorize()
        t3 = time.perf_counter()
        e = kkt.export()
        d1 = kkt.solve(b)
        d2 = kkt.solve(e["ASinv"])
        d3 = kkt.solve(e["ASinvRdSinv"])
        t4 = time.perf_counter()
        if timed:
            stage["chol_S"] += t1 - t; stage["buildup"] += t2 - t1; stage["factor_M"] += t3 - t2; stage["solve3"] += t4 - t3
        return d1, d2, d3

    for _ in range(args.warmup):
        step(False)
    lib.HMiSetKernelTiming(1)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        sol = step(True)
    barrier()
    elapsed = time.perf_counter() - t_start
    lib.HMiSetKernelTiming(0)
    import ctypes as C
    kms, kfl = np.zeros(4), np.zeros(4)
    kln = np.zeros(4, dtype=np.int64)
    lib.HMiGetKernelTiming(kms.ctypes.data_as(C.POINTER(C.c_double)), kfl.ctypes.data_as(C.POINTER(C.c_double)),
                           kln.ctypes.data_as(C.POINTER(C.c_int64)))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed
    # dominant kernel = the role with the largest share of the timed region (all three are fp64-MFMA bound)
    names = {1: KERNEL_NAME + "<false,true,1> (congruence step 1: U = Linv*A_L, triangular x triangular)",
             2: KERNEL_NAME + "<false,false,2> (congruence step 2: At = U*Linv^T + Linv*U^T, SYR2K form)",
             3: KERNEL_NAME + "<true,true,3> (Gram: M = Ahat*Ahat^T over the packed index)"}
    short = {1: "congruence_step1", 2: "congruence_step2", 3: "gram"}
    dom = max((1, 2, 3), key=lambda r: kms[r])
    dom_ms = kms[dom] / max(1, kln[dom])
    achieved = (kfl[dom] / max(1, kln[dom])) / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    roofline = {
        "bound": "mfma", "kernel": names[dom],
        "achieved": round(achieved, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
        "traffic": traffic_bytes_per_launch(dom)[0] if (n, m, world) == (2000, 2000, 1) else None,
        "traffic_source": traffic_bytes_per_launch(dom)[1] if (n, m, world) == (2000, 2000, 1) else None,
        "avg_launch_ms": round(float(dom_ms), 4), "launches": int(kln[dom]),
        "flops_per_launch": float(kfl[dom] / max(1, kln[dom])),
        # the same launch priced with SURVEY 8(d)'s per-unit figure (the reference's M3 count: 3 n^3 per constraint
        # for S^-1 A S^-1, n(n+1) per row pair for the trace part): congruence = steps 1+2 together
        "survey_count": survey_equiv(n, m, world, kms, kln, args.steps),
        "kernels": {short[r]: {"ms_per_step": round(float(kms[r]) / args.steps, 3),
                               "tflops": round(float(kfl[r] / max(kms[r], 1e-9) / 1e9), 2),
                               "launches_per_step": int(kln[r] // args.steps)} for r in (1, 2, 3)},
        "helper_gemms_ms_per_step": round(float(kms[0]) / args.steps, 3),
    }
    out = {
        "metric": "IPM iterations/sec (Schur build+factor+solve), n=%d m=%d dense SDP" % (n, m),
        "value": round(value, 4), "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[3]: synthetic dense SDP n=%d m=%d (SURVEY 8(d) splitmix64 family, "
                               "state y=0 tau=1 Rd=-10n), one Phase-A pass per step" % (n, m),
                   "n": n, "m": m, "parallelism": "rows%d" % world,
                   "stage_ms": {k: round(v / args.steps * 1e3, 3) for k, v in stage.items()},
                   "setup_s": round(setup_s, 1),
                   "whole_step_tflops_survey_count": round(survey_flops(n, m) / (ms_per_step * 1e-3) / 1e12, 2),
                   "whole_step_tflops_executed": round(executed_flops(n, m) / (ms_per_step * 1e-3) / 1e12, 2)},
        "roofline": roofline,
        "checksum": {"sum_d2": float(np.sum(sol[1])), "sum_d1w": float(np.dot(np.arange(1, m + 1), sol[0]))},
    }
    if ex is not None:
        out["config"]["exchange_bytes_per_step"] = {"all_to_all": ex.bytes_a2a // (args.steps + args.warmup),
                                                    "all_reduce": ex.bytes_ar // (args.steps + args.warmup)}
    if world == 1 and (n, m) == (2000, 2000):
        # BASELINE configs 2-3 (mcp100, gpp100 "on 1 MI355X"): parity cases, not the headline -- one timed line each so that
        # the driver-run record carries them: a Phase-A pass call by call through the reference's operator surface, the same
        # pass as ONE fused launch (csrc/small.hip, HMiKKTPhaseA), achieved GB/s against the ~0.2 MB of algorithmic bytes,
        # and the plain-C oracle port on one host core (tools/small_configs.py)
        try:
            kkt.destroy(); cone.destroy()          # give the 100 GB back before the small cones are made
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import small_configs
            out["small_configs"] = [small_configs.measure(nm, reps=30, cpu=not args.no_cpu) for nm in ("mcp100_A", "gpp100_A")]
        except Exception as e:  # never lose the headline line over the extras
            out["small_configs"] = {"error": str(e)}
    if not args.no_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(n, m)
        out["cpu_baseline_blas3"] = cpu_baseline_blas3(n, m)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
