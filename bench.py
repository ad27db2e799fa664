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


# `roofline.traffic`: L2-miss (fabric) bytes per launch of the hot kernels -- FETCH_SIZE / WRITE_SIZE count memory-side requests,
# Infinity-Cache hits INCLUDED (/opt/skills/guides/MI355X_MICROARCH.md), so the figure is an upper bound on HBM bytes, not HBM
# bytes.  It comes from the newest profiles/*_traffic.json taken at this (n, m) (tools/gpu_profile.sh <tag> [--m 8000]: separate
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command, condensed by tools/prof_summary.py --traffic;
# FETCH_SIZE x 2 for the gfx950 wide-read correction + WRITE_SIZE, KiB).  bench.py cannot run the profiler on itself, so
# the file is matched on the kernel's NAME (role and variant are template arguments): a kernel that was renamed or
# re-templated since the profile was taken yields `traffic: null` instead of a stale number.
PERSIST = os.environ.get("HDM_PERSIST", "1") != "0"   # persistent workgroups (the default) or one workgroup per tile
KERNEL_NAME = "hdm_gemm_persist_kernel" if PERSIST else "hdm_gemm_kernel"
KERNEL_SYMBOL = {1: KERNEL_NAME + "<false, true, 1, 64>", 2: KERNEL_NAME + "<false, false, 2, 64>",
                 3: KERNEL_NAME + "<true, true, 3, 64>"}


def _profile_size(d):
    """(n, m) a traffic file was taken at; files from before round 5 carry no size and were all taken at n = m = 2000"""
    return (int(d.get("n", 2000)), int(d.get("m", 2000)))


def traffic_bytes_per_launch(role, n=2000, m=2000):
    """(bytes per launch or None, description of the source).  The newest profiles/*_traffic.json BY ITS OWN TIMESTAMP
    (`taken_utc`, written by tools/prof_summary.py --traffic) is used, and only if it was measured on the kernel sources
    this build was made from (`kernel_source_sha`) with the same launch form (`persist`): anything else gives None and
    says why, so a stale number never reaches the bench line."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from prof_summary import kernel_source_sha
    if os.environ.get("HDM_VAR"):
        return None, "diagnostic variant (HDM_VAR set)"
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except Exception:
            continue
        if "taken_utc" in d and _profile_size(d) == (n, m) and (best is None or d["taken_utc"] > best[1]["taken_utc"]):
            best = (f, d)
    if best is None:
        return None, "no profiles/*_traffic.json taken at n=%d m=%d carries a timestamp and a kernel source hash" % (n, m)
    f, d = best
    tag = "%s (taken %s on kernel sources %s)" % (os.path.basename(f), d["taken_utc"], d.get("kernel_source_sha"))
    if d.get("kernel_source_sha") != kernel_source_sha(ROOT):
        return None, "stale: " + tag + ", this build is " + kernel_source_sha(ROOT)
    if bool(d.get("persist", True)) != PERSIST:
        return None, "other launch form: " + tag
    hits = [v for k, v in d["kernels"].items() if KERNEL_SYMBOL[role].replace(" ", "") in k.replace(" ", "")]
    if len(hits) != 1:
        return None, "kernel symbol not found in " + tag
    return hits[0]["bytes_per_launch"], tag


def survey_flops(n, m):
    """SURVEY.md 8(d) operation count of one iteration (reference M3 count, kept for comparability)"""
    return 3.0 * m * n ** 3 + 0.5 * m * (m + 1) * n * (n + 1) + n ** 3 + m ** 3 / 3.0 + 6.0 * m * m


def executed_flops(n, m):
    """what our formulation needs on valid data: congruence n^3 per row (triangular factors), Gram over the packed index"""
    return m * 1.0 * n ** 3 + 0.5 * (m + 3) * (m + 4) * n * (n + 1) + n ** 3 + m ** 3 / 3.0 + 6.0 * m * m


def algorithmic_bytes(role, n, m, world, kln, steps):
    """least HBM bytes of ONE launch of a GEMM role: operands read once, results written once"""
    rows = (m + world - 1) // world
    per_launch = rows / max(1, int(kln[role]) // max(1, steps)) if role in (1, 2) else rows
    low = n * (n + 1) / 2 * 8.0                       # one lower triangle
    if role == 1:                                     # A_L in (skyline ~ lower triangle), U's lower tiles out
        return per_launch * 2 * low
    if role == 2:                                     # U in, blocked lower triangle out
        return per_launch * 2 * low
    R = m + 3                                         # Gram: the operand once (this rank's share of the packed index) + the lower triangle of the result
    launches = max(1, int(kln[3]) // max(1, steps))   # (more K splits than slabs run as several launches: each takes its share of the operand)
    return (R * low / world + 0.5 * R * R * 8.0) / launches   # (the split-K slabs are traffic, not algorithm)


def survey_equiv(n, m, world, kms, kln, steps):
    """achieved TFLOP/s if the work is priced with the reference's operation count; can exceed the
    MFMA peak because the triangular-factor congruence executes n^3, not 3 n^3, flops per constraint"""
    rows = (m + world - 1) // world
    cong_ms = (kms[1] + kms[2] + kms[4]) / steps
    gram_ms = kms[3] / steps
    return {"congruence": {"flops_per_unit": 3.0 * n ** 3, "unit": "constraint", "units_per_step": rows,
                           "tflops": round(3.0 * n ** 3 * rows / max(cong_ms, 1e-9) / 1e9, 2)},
            "gram": {"flops_per_unit": float(n) * (n + 1), "unit": "row pair", "units_per_step": 0.5 * m * (m + 1) / world,
                     "tflops": round(0.5 * m * (m + 1) / world * n * (n + 1) / max(gram_ms, 1e-9) / 1e9, 2)}}


def workload_label(n, m, shards=8, streamed=False):
    """BASELINE.json's configs by (n, m): nothing else may call itself configs[3] or configs[4]"""
    fam = "synthetic dense SDP n=%d m=%d (SURVEY 8(d) splitmix64 family, state y=0 tau=1 Rd=-10n), one Phase-A pass per step" % (n, m)
    if (n, m) == (2000, 2000):
        return "configs[3]: " + fam
    if (n, m) == (2000, 8000):
        if shards > 1:
            return "configs[4]: " + fam + ", constraint rows sharded over %d GPUs" % shards
        return "configs[4]: " + fam + ", all 8000 rows on ONE device (constraint data %s)" % (
            "streamed: regenerated per congruence batch" if streamed else "resident")
    return "custom (not a BASELINE config): " + fam


def golden_check(n, m, kkt, sol, rhs):
    """compare the run with the independent host-fp64 fixtures (tests/golden/, read-only data): n = m = 2000 against
    full2000.npz (the two checksums of the Phase-A solutions), n = 2000, m = 8000 against full8000_rows.npz (rows of M,
    both vectors, and rows of the residual M d = rhs with the fixture's rows of M and the device's solutions).  Returns
    (details, ok) or (None, None) when no fixture exists for this size."""
    gdir = os.path.join(ROOT, "tests", "golden")
    if (n, m) == (2000, 2000) and os.path.exists(os.path.join(gdir, "full2000.npz")):
        g = np.load(os.path.join(gdir, "full2000.npz"))
        d = {"sum_d2": float(np.sum(sol[1])), "sum_d1w": float(np.dot(np.arange(1, m + 1), sol[0])),
             "golden_sum_d2": float(g["bench_sum_d2"]), "golden_sum_d1w": float(g["bench_sum_d1w"]), "fixture": "full2000.npz"}
        ok = (abs(d["sum_d2"] - d["golden_sum_d2"]) <= 1e-9 * abs(d["golden_sum_d2"]) and
              abs(d["sum_d1w"] - d["golden_sum_d1w"]) <= 1e-9 * abs(d["golden_sum_d1w"]))
        return d, bool(ok)
    if (n, m) == (2000, 8000) and os.path.exists(os.path.join(gdir, "full8000_rows.npz")):
        g = np.load(os.path.join(gdir, "full8000_rows.npz"))
        rows = np.asarray(g["bench_rows"])
        Mg = np.asarray(g["bench_M_rows"])
        Md = kkt.rows(rows)
        ex = kkt.export()
        scale = float(np.max(np.abs(Mg)))
        d = {"fixture": "full8000_rows.npz", "rows": int(rows.size),
             "sum_M_rows": float(np.sum(Md)), "golden_sum_M_rows": float(g["bench_sum_M_rows"]),
             "max_err_M_rows": float(np.max(np.abs(Md - Mg)) / scale),
             "max_err_ASinv": float(np.max(np.abs(ex["ASinv"] - g["bench_ASinv"])) / np.max(np.abs(g["bench_ASinv"]))),
             "max_err_ASinvRdSinv": float(np.max(np.abs(ex["ASinvRdSinv"] - g["bench_ASinvRdSinv"])) /
                                          np.max(np.abs(g["bench_ASinvRdSinv"]))),
             # rows of M d - rhs with the FIXTURE's rows of M: an independent check of the factorisation and the solves
             "max_residual_rows": float(max(np.max(np.abs(Mg @ x - np.asarray(r)[rows])) / np.max(np.abs(r))
                                            for x, r in zip(sol, rhs)))}
        ok = (d["max_err_M_rows"] < 1e-10 and d["max_err_ASinv"] < 1e-10 and d["max_err_ASinvRdSinv"] < 1e-10 and
              d["max_residual_rows"] < 1e-8)
        return d, bool(ok)
    return None, None


def cpu_baseline(n, m, budget_cols=16):
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


def sharded_step_profile(profiles, stage_ms_ranks):
    """Condense the per-shard profiles of the LAST timed step's build (api.SDPCone.build_profile, one per rank) into what a
    multi-GPU line needs to diagnose itself: for every stage [min, max] over the ranks; per exchange piece the same for step 2
    of its tile columns, the exchange WAIT (engine stream idle because the piece had not arrived), its Gram splits, the
    bytes a rank sent and the achieved GB/s (bytes / time from hand-over to arrival, host clock: a lower bound).
    `stage_ms_ranks`: per rank the host-timed stages of the step (assemble+factor S, build-up, factor M, three solves)."""
    prof = [p for p in profiles if p]
    if not prof:
        return None
    mm = lambda xs: [round(float(min(xs)), 3), round(float(max(xs)), 3)]
    P = prof[0]["pieces"]
    out = {"ranks": len(prof), "pieces": P, "staged": prof[0]["staged"], "min_max_over_ranks_ms": {}}
    keys = [k for k in ("invert_ms", "congruence_step1_ms", "congruence_ms", "slab_reduce_ms", "allreduce_ms", "extract_ms") if k in prof[0]]
    for k in keys:
        out["min_max_over_ranks_ms"][k] = mm([p[k] for p in prof])
    for k, name in (("step2_piece_ms", "congruence_step2_by_piece"), ("exchange_wait_ms", "exchange_wait_by_piece"),
                    ("exchange_wait_host_ms", "exchange_wait_host_by_piece"), ("gram_piece_ms", "gram_by_piece")):
        out["min_max_over_ranks_ms"][name] = [mm([p[k][j] for p in prof]) for j in range(P)]
        out["min_max_over_ranks_ms"][name.replace("_by_piece", "_total")] = mm([sum(p[k]) for p in prof])
    out["bytes_sent_per_rank"] = mm([sum(p["piece_bytes_sent"]) for p in prof])
    out["piece_bytes_sent"] = [round(float(np.mean([p["piece_bytes_sent"][j] for p in prof]))) for j in range(P)]
    out["piece_gb_per_s"] = [mm([p["piece_bytes_sent"][j] / max(p["piece_flight_ms"][j], 1e-6) * 1e-6 for p in prof]) for j in range(P)]
    if stage_ms_ranks:
        out["replicated_ms"] = {k: mm([s[k] for s in stage_ms_ranks]) for k in ("assemble_S+chol_S", "factor_M", "solve3")}
    return out


def device_pci_bus(index):
    """'dddd:bb:dd.f' of a visible device, or None"""
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def mfma_busy_from_profiles(role, n=2000, m=2000):
    """MFMA-busy share of one role's kernel from the newest kept PMC summary (profiles/*_summary_pmc_SQ_VALU_MFMA_BUSY*.txt)
    whose traffic.json names the kernel sources of this build and this (n, m); None otherwise"""
    import glob
    import re
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from prof_summary import kernel_source_sha
        sha = kernel_source_sha(ROOT)
        best = None
        for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
            try:
                dj = json.load(open(tj))
                if dj.get("kernel_source_sha") == sha and _profile_size(dj) == (n, m):
                    best = tj
            except Exception:
                pass
        if not best:
            return None
        f = glob.glob(best.replace("_traffic.json", "_summary_pmc_SQ_VALU_MFMA_BUSY*.txt"))
        if not f:
            return None
        pat = {1: "<false, true, 1", 2: "<false, false, 2", 3: "<true, true, 3"}[role]
        for line in open(f[0]):
            if "hdm_gemm_persist_kernel" + pat in line and "SQ_VALU_MFMA_BUSY_CYCLES" in line:
                g = float(re.search(r"GRBM_GUI_ACTIVE=([0-9.e+]+)", line).group(1))
                b = float(re.search(r"SQ_VALU_MFMA_BUSY_CYCLES=([0-9.e+]+)", line).group(1))
                return round(b / (g / 8.0 * 1024.0), 4)
    except Exception:
        return None
    return None


def under_profiler():
    """rocprofv3 preloads its tool library into this process AND into every child: a child started from here would exec with
    the GPU already initialised by that library, which this pool refuses.  Under a profiler bench.py starts no child processes."""
    return "rocprof" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCPROFILER_LIBRARY_CTOR") or os.environ.get("ROCP_TOOL_LIBRARIES"))


class PowerClockSampler:
    """Board power and shader clock sampled in a background thread while the timed region runs: the fp64 matrix pipe at this
    density is power-limited (profiles/r03_f_power_clock.txt), so the line carries what the board drew and clocked at.  Read
    from the amdgpu hwmon files of the card this process computes on (no child process, nothing beside the timed region but two file reads
    every 0.2 s); where those are not readable, rocm-smi in a child process -- never under a profiler.  No source, no samples."""

    def __init__(self, pci_bus=None):
        import glob
        import shutil
        import threading
        self.samples = []
        self.source = None
        self._pw = self._fq = None
        # the card THIS process computes on, by its PCI address (a box shows the hwmon files of every GPU of its host)
        dev = os.path.join("/sys/bus/pci/devices", pci_bus) if pci_bus else None
        if dev and os.path.isdir(dev):
            pw = [f for pat in ("power1_average", "power1_input") for f in glob.glob(os.path.join(dev, "hwmon", "hwmon*", pat))]
            fq = glob.glob(os.path.join(dev, "hwmon", "hwmon*", "freq1_input"))
            if pw and os.access(pw[0], os.R_OK):
                self._pw, self._fq = pw[0], (fq[0] if fq and os.access(fq[0], os.R_OK) else None)
                self.source = "amdgpu hwmon of %s (%s%s)" % (pci_bus, os.path.basename(pw[0]), ", freq1_input" if self._fq else "")
        self.exe = None if (self._pw or under_profiler()) else shutil.which("rocm-smi")
        if self.exe:
            self.source = "rocm-smi card0"
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True) if (self._pw or self.exe) else None

    def _run(self):
        import re
        import time
        while not self._stop.is_set():
            if self._pw:
                try:
                    w = float(open(self._pw).read()) * 1e-6
                    f = float(open(self._fq).read()) * 1e-6 if self._fq else float("nan")
                    self.samples.append((w, f))
                except (OSError, ValueError):
                    return
                time.sleep(0.2)
                continue
            try:
                out = subprocess.run([self.exe, "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            rows = [l for l in out.splitlines() if l.startswith("card0")]
            if rows:
                mhz = re.findall(r"\((\d+)Mhz\)", rows[0])
                try:
                    self.samples.append((float(rows[0].rstrip(", ").split(",")[-1]), float(mhz[2]) if len(mhz) > 2 else float("nan")))
                except ValueError:
                    pass

    def start(self):
        if self._th:
            self._th.start()

    def stop(self):
        if not self._th:
            return None
        self._stop.set()
        self._th.join(timeout=15)
        if not self.samples:
            return None
        w, f = np.array([x[0] for x in self.samples]), np.array([x[1] for x in self.samples])
        return {"samples": len(self.samples), "board_power_w": [float(w.min()), float(w.mean()), float(w.max())],
                "shader_clock_mhz": [float(np.nanmin(f)), float(np.nanmean(f)), float(np.nanmax(f))],
                "source": "%s, [min, mean, max] over the timed region" % self.source}


def rccl_self_test_in_child(ids, timeout_s=240):
    """HMiRcclGroupSelfTest over `ids` in a FRESH child process with a timeout (RCCL between these devices may never have
    run on this machine; a hang or a crash there must not take the bench down, and a process that has touched the GPU is
    never re-executed).  Returns (passed, reason)."""
    code = ("import sys; sys.path.insert(0, %r); from hdsdp_amd import api; "
            "sys.exit(api.rccl_group_self_test(%r, 120000))" % (ROOT, list(ids)))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return False, "RCCL self-test over devices %s did not finish in %d s" % (list(ids), timeout_s)
    if r.returncode == 0:
        return True, None
    tail = (r.stderr or "").strip().splitlines()[-1:] or [""]
    return False, "RCCL self-test over devices %s failed at stage %d (%s)" % (list(ids), r.returncode, tail[0][:160])


def plan_devices(gpus, ndev, world, loopback, self_test=rccl_self_test_in_child):
    """How `--gpus N` is driven, as data (no GPU call; tests/test_abi_cpu.py runs it for ndev in {1, 2, 8}):
      under torchrun (world > 1)      one process per GPU, the exchange through torch.distributed (nccl = RCCL)
      --gpus N alone, N devices there the in-process device group over RCCL if the whole-group self-test passes in a child
                                      process, else over device copies with the reason on the bench line
      --gpus N --loopback, fewer      shards share devices: device copies (RCCL needs one device per rank); a rehearsal
    A request that cannot be met is {"error": ...}, never a smaller run under an N-GPU label."""
    from hdsdp_amd import api
    if world > 1:
        if world != gpus:
            return {"error": f"--gpus {gpus} but WORLD_SIZE={world}"}
        if ndev < 1:
            return {"error": "no MI355X visible: bench.py has no CPU fallback"}
        return {"mode": "torchrun", "shards": world, "ids": None, "devices_used": world, "transport_request": None,
                "transport": "torch.distributed", "transport_fallback_reason": None, "rccl_ranks_if_nccl": world}
    if gpus > 1 and ndev < gpus and not loopback:
        return {"error": f"--gpus {gpus} requested but {ndev} device(s) visible (use --loopback for a rehearsal on shared "
                         f"devices, or launch with torchrun)"}
    if ndev < 1:
        return {"error": "no MI355X visible: bench.py has no CPU fallback"}
    if gpus == 1:
        return {"mode": "single", "shards": 1, "ids": [0], "devices_used": 1, "transport_request": None,
                "transport": "none", "transport_fallback_reason": None, "rccl_ranks": 0}
    ids = [r % ndev for r in range(gpus)]
    plan = {"mode": "in-process device group (HMiSetDevicesEx)", "shards": gpus, "ids": ids, "devices_used": len(set(ids))}
    if len(set(ids)) < gpus:
        plan.update(transport_request=api.TRANSPORT_COPY, transport="device copies", rccl_ranks=0,
                    transport_fallback_reason="shards share devices (--loopback): RCCL needs one device per rank")
        return plan
    if under_profiler():
        # the profiler's preloaded library has the GPU initialised in any child before its exec: no self-test child
        ok, why = False, "under a profiler: RCCL self-test child not started"
    else:
        ok, why = self_test(ids)
    if ok:
        plan.update(transport_request=api.TRANSPORT_RCCL, transport="rccl", rccl_ranks=gpus, transport_fallback_reason=None)
    else:
        plan.update(transport_request=api.TRANSPORT_COPY, transport="device copies", rccl_ranks=0, transport_fallback_reason=why)
    return plan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--m", type=int, default=2000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--backend", default=None, help="collective backend for N>1 (default nccl = RCCL)")
    ap.add_argument("--loopback", action="store_true",
                    help="--gpus N on fewer than N devices: shards share devices (a rehearsal; n_gpus then reports the devices "
                         "really used, never N)")
    args = ap.parse_args()

    import torch
    from hdsdp_amd import api, dist as hdist

    rank, world, local = hdist.init_process_group_from_env(args.backend)
    if world > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("LOCAL_RANK", str(local))
    lib = api.load_library()
    n, m = args.n, args.m
    # How N GPUs are driven.  Under torchrun (WORLD_SIZE set): one process per GPU, collectives through torch.distributed
    # (hdsdp_amd/dist.py).  `--gpus N` WITHOUT torchrun: the in-process device group behind the C ABI (HMiSetDevices: the
    # path the reference's single-threaded driver uses), N shards on N devices over RCCL.  Either way the line reports
    # the devices really used, and a request that cannot be met is an error, not a one-GPU run with an N-GPU label.
    ndev = torch.cuda.device_count()        # (does not initialise the GPU: the self-test's child process comes first)
    plan = plan_devices(args.gpus, ndev, world, args.loopback)
    if "error" in plan:
        raise SystemExit(plan["error"])
    shards, mode, devices_used = plan["shards"], plan["mode"], plan["devices_used"]
    if mode.startswith("in-process"):
        api.set_devices(plan["ids"], shard_min_dim=0, transport=plan["transport_request"])
        gids, got = api.device_group()
        if gids != plan["ids"] or got != plan["transport_request"]:
            raise SystemExit(f"device group is {gids} over transport {got}, asked for {plan['ids']} over {plan['transport_request']}")
    elif lib.HMiDeviceInit(local) != 0:
        raise SystemExit("no MI355X visible: bench.py has no CPU fallback")
    torch.cuda.set_device(local % ndev)
    if devices_used != args.gpus and not args.loopback:
        raise SystemExit(f"--gpus {args.gpus} requested, {devices_used} device(s) in use")

    # BASELINE configs 2-3 (mcp100, gpp100 "on 1 MI355X"): parity cases, not the headline -- one timed line each so that the
    # driver-run record carries them (tools/small_configs.py).  Measured FIRST, on a quiet device: round 2 measured them
    # right after giving 100 GB back to the runtime and the first config came out at 0.44 ms instead of 0.17.
    small = None
    if world == 1 and shards == 1 and (n, m) == (2000, 2000):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import small_configs
            small = [small_configs.measure(nm, reps=30, cpu=not args.no_cpu) for nm in ("mcp100_A", "gpp100_A")]
        except Exception as e:  # never lose the headline line over the extras
            small = {"error": str(e)}
        lib.HMiDeviceSynchronize()

    t0 = time.time()
    cone = api.SDPCone.synthetic(n, m, rank=rank, world=world)
    if mode.startswith("in-process") and cone.shard_count() != shards:
        raise SystemExit(f"the block was not sharded ({cone.shard_count()} shard(s), {shards} requested)")
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

    stage = {"assemble_S+chol_S": 0.0, "buildup": 0.0, "factor_M": 0.0, "solve3": 0.0}
    # The cone answers a request for the very point its buffer already holds from the buffer (engine_cone.h: exact short-cut),
    # so a bench that asked for (tau, y) = (1, 0) every step would time the Cholesky alone after the first.  tau therefore
    # alternates between 1 and 1 + 2^-52: every step really assembles S = tau C - Rd I (one pass over C at y = 0) and factors
    # it; the LAST timed step is at tau = 1 exactly, which is the state the checksum and the fixtures belong to.
    tau_of = lambda k: 1.0 if (k % 2 == 0) else float(np.nextafter(1.0, 2.0))

    def step(timed, k=0):
        t = time.perf_counter()
        ok = cone.check_is_interior(tau_of(k), y)
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
            stage["assemble_S+chol_S"] += t1 - t; stage["buildup"] += t2 - t1; stage["factor_M"] += t3 - t2; stage["solve3"] += t4 - t3
        return d1, d2, d3

    # What THIS box's matrix pipe sustains: a register-only fp64 MFMA loop on full-range operands, two workgroups per CU, long
    # enough (about 0.3 s) for the power management to settle.  The load is power-limited (profiles/r03_f_power_clock.txt) and
    # boxes differ by several per cent; the bench line carries the number so that a run can be read against its own box.
    probe_tf = None
    if rank == 0 and not os.environ.get("HDM_BENCH_NO_PROBE"):
        probe_tf = float(lib.HMiMfmaIssueProbe(300, 2, 350000))
        lib.HMiDeviceSynchronize()
    for w in range(args.warmup):
        step(False, args.steps + args.warmup - 1 - w)
    lib.HMiSetKernelTiming(1)
    sampler = PowerClockSampler(device_pci_bus(local % ndev)) if (rank == 0 and world == 1 and shards == 1) else None
    barrier()
    if sampler:
        sampler.start()
    t_start = time.perf_counter()
    for k in range(args.steps):
        sol = step(True, args.steps - 1 - k)
    barrier()
    elapsed = time.perf_counter() - t_start
    power_clock = sampler.stop() if sampler else None
    lib.HMiSetKernelTiming(0)
    import ctypes as C
    kms, kfl, kis = np.zeros(5), np.zeros(5), np.zeros(5)   # roles 0-3 + [4] = the full diagonal tiles of congruence step 2 (their own kernel)
    kln = np.zeros(5, dtype=np.int64)
    lib.HMiGetKernelTimingEx(kms.ctypes.data_as(C.POINTER(C.c_double)), kfl.ctypes.data_as(C.POINTER(C.c_double)),
                             kis.ctypes.data_as(C.POINTER(C.c_double)), kln.ctypes.data_as(C.POINTER(C.c_int64)))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    # where the last step's sharded build went, per rank: in-process device group -> one profile per shard; torchrun -> gathered
    stage_now = {k: v / args.steps * 1e3 for k, v in stage.items()}
    shard_prof = None
    if world == 1 and shards > 1:
        shard_prof = sharded_step_profile([cone.build_profile(r) for r in range(shards)], [stage_now])
    elif world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, (cone.build_profile(0), stage_now))
        shard_prof = sharded_step_profile([g_[0] for g_ in gathered], [g_[1] for g_ in gathered])
    streamed_ranks = None
    if world > 1:   # how many ranks stream their constraint rows (engine_create.h decides from the device's capacity: all or none)
        ts = torch.tensor([1.0 if cone.streaming()[0] else 0.0], dtype=torch.float64,
                          device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(ts)
        streamed_ranks = int(ts.item())
    if rank != 0:
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed
    # dominant kernel = the role with the largest share of the timed region (all three are fp64-MFMA bound)
    names = {1: KERNEL_NAME + "<false,true,1> (congruence step 1: U = Linv*A_L, triangular x triangular)",
             2: KERNEL_NAME + "<false,false,2> (congruence step 2: At = U*Linv^T + Linv*U^T, SYR2K form; all tiles but the full diagonal ones)",
             3: KERNEL_NAME + "<true,true,3> (Gram: M = Ahat*Ahat^T over the packed index)"}
    short = {1: "congruence_step1", 2: "congruence_step2", 3: "gram"}
    dom = max((1, 2, 3), key=lambda r: kms[r])
    dom_ms = kms[dom] / max(1, kln[dom])
    achieved = (kfl[dom] / max(1, kln[dom])) / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    traffic = traffic_bytes_per_launch(dom, n, m) if (world, shards) == (1, 1) else (None, "measured on one GPU only")
    roofline = {
        "bound": "mfma", "kernel": names[dom],
        "achieved": round(achieved, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
        "traffic": traffic[0], "traffic_source": traffic[1],
        "traffic_is": "L2-miss (fabric) bytes per launch, Infinity-Cache hits included: an upper bound on HBM bytes",
        # what one launch has to move at the least: step 1 reads A_L (skyline, 8.5 n^2/2... per constraint) and writes
        # U's lower tiles; step 2 reads U and writes the blocked lower triangle; the Gram product reads its operand once and writes its
        # slabs.  traffic / algorithmic is the re-read factor: operand panels are re-read per tile and only partly served by L2 --
        # energy, not time (the L2-resident ablation bounds its cost at 0.7-2.3 %)
        "algorithmic_bytes_per_launch": float(algorithmic_bytes(dom, n, m, world, kln, args.steps)),
        "traffic_over_algorithmic": (None if not traffic[0] else round(traffic[0] / algorithmic_bytes(dom, n, m, world, kln, args.steps), 2)),
        "power_clock": power_clock,
        "avg_launch_ms": round(float(dom_ms), 4), "launches": int(kln[dom]),
        "flops_per_launch": float(kfl[dom] / max(1, kln[dom])),
        # flops the dominant kernel's MFMA instructions EXECUTED over its algorithmic flops (host count from the tile lists and
        # the kernel's stage sequences, csrc/gemm_f64.hip: issued_mfma_flops; held to SQ_VALU_MFMA_BUSY_CYCLES / 64 of the kept
        # profile to 2e-5): the granularity loss -- 16 x 16 sub-blocks straddling a diagonal, live ranges of triangular K
        # blocks, padding rows.  `achieved_issued` prices the launches with the executed count (what the matrix pipe really did)
        "issued_over_valid": round(float(kis[dom] / max(kfl[dom], 1.0)), 4),
        "achieved_issued": round(float(kis[dom] / max(kms[dom], 1e-9) / 1e9), 3),
        "issued_over_valid_by_role": {short[r]: round(float(kis[r] / max(kfl[r], 1.0)), 4) for r in (1, 2, 3)},
        "issued_over_valid_congruence_step2_whole": round(float((kis[2] + kis[4]) / max(kfl[2] + kfl[4], 1.0)), 4),
        # the same launch priced with SURVEY 8(d)'s per-unit figure (the reference's M3 count: 3 n^3 per constraint
        # for S^-1 A S^-1, n(n+1) per row pair for the trace part): congruence = steps 1+2 together
        "survey_count": survey_equiv(n, m, world, kms, kln, args.steps),
        "kernels": {short[r]: {"ms_per_step": round(float(kms[r]) / args.steps, 3),
                               "tflops": round(float(kfl[r] / max(kms[r], 1e-9) / 1e9), 2),
                               "launches_per_step": int(kln[r] // args.steps)} for r in (1, 2, 3)},
        # congruence step 2 as a whole = the kernel above + its full diagonal tiles (P + P^T from one product, own kernel)
        "congruence_step2_whole": {"ms_per_step": round(float(kms[2] + kms[4]) / args.steps, 3),
                                   "tflops": round(float((kfl[2] + kfl[4]) / max(kms[2] + kms[4], 1e-9) / 1e9), 2),
                                   "diagonal_tiles_ms_per_step": round(float(kms[4]) / args.steps, 3),
                                   "diagonal_tiles_tflops": round(float(kfl[4] / max(kms[4], 1e-9) / 1e9), 2)},
        "helper_gemms_ms_per_step": round(float(kms[0]) / args.steps, 3),
        # this box's own ceiling (register-only fp64 MFMA loop on random operands, measured just before the timed region) and the
        # dominant kernel against it; `frac` above stays against the vendor peak
        "box_mfma_loop_tflops": None if probe_tf is None else round(probe_tf, 2),
        "frac_of_box_mfma_loop": None if not probe_tf else round(achieved / probe_tf, 4),
        # the board runs this load at its power limit: against what the matrix pipe can do at the shader clock sampled over the
        # timed region (peak x clock / 2.4 GHz); `frac` above stays against the vendor peak at 2.4 GHz
        "frac_at_sampled_clock": (None if not (power_clock and power_clock["shader_clock_mhz"][1] > 0) else
                                  round(achieved / (FP64_MFMA_PEAK_TFLOPS * power_clock["shader_clock_mhz"][1] / 2400.0), 4)),
        # busy share of the matrix pipe in the dominant kernel, from the PMC pass of the newest kept profile set of these kernel
        # sources (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)); None when that set was taken on other sources
        "mfma_busy_profiled": mfma_busy_from_profiles(dom, n, m) if (world, shards) == (1, 1) else None,
    }
    out = {
        "metric": "IPM iterations/sec (Schur build+factor+solve), n=%d m=%d dense SDP" % (n, m),
        "value": round(value, 4), "unit": "it/s", "n_gpus": devices_used, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload_label(n, m, shards, cone.streaming()[0]),
                   "n": n, "m": m, "parallelism": "rows%d" % shards,
                   # how the GPUs were driven and what really ran: N requested, shards of the row deal, distinct devices in
                   # use, ranks of the RCCL communicator that carried the exchange (0 = device copies or one GPU)
                   "gpus_requested": args.gpus, "shards": shards, "devices_used": devices_used, "driver": mode,
                   "transport": (("torch.distributed/" + torch.distributed.get_backend()) if world > 1 else plan["transport"]),
                   "transport_fallback_reason": plan["transport_fallback_reason"],
                   "rccl_ranks": ((world if torch.distributed.get_backend() == "nccl" else 0) if world > 1 else plan["rccl_ranks"]),
                   "stage_ms": {k: round(v / args.steps * 1e3, 3) for k, v in stage.items()},
                   "constraint_data": ("streamed: regenerated %d rows at a time" % cone.streaming()[1]) if cone.streaming()[0] else "resident",
                   "streamed_ranks": streamed_ranks,
                   "setup_s": round(setup_s, 1),
                   "whole_step_tflops_survey_count": round(survey_flops(n, m) / (ms_per_step * 1e-3) / 1e12, 2),
                   "whole_step_tflops_executed": round(executed_flops(n, m) / (ms_per_step * 1e-3) / 1e12, 2)},
        "roofline": roofline,
        "checksum": {"sum_d2": float(np.sum(sol[1])), "sum_d1w": float(np.dot(np.arange(1, m + 1), sol[0]))},
    }
    try:
        rhs_last = (b, kkt.export()["ASinv"], kkt.export()["ASinvRdSinv"])
        det, ok = golden_check(n, m, kkt, sol, rhs_last)
    except Exception as e:
        det, ok = {"error": str(e)}, False
    if det is not None:
        out["checksum"].update(det)
    out["checksum_ok"] = ok          # None: no independent fixture exists for this (n, m)
    if ex is not None:
        out["config"]["exchange_bytes_per_step"] = {"all_to_all": ex.bytes_a2a // (args.steps + args.warmup),
                                                    "all_reduce": ex.bytes_ar // (args.steps + args.warmup)}
    if shard_prof is not None:
        out["sharded_step"] = shard_prof
    if small is not None:
        out["small_configs"] = small
    if not args.no_cpu and world == 1 and not under_profiler():
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
