"""CPU tests of the drop-in boundary: the shared library loads without a GPU, exports every symbol the
header declares, keeps the reference's struct layouts, and its host-side presolve reproduces the
reference's classification / ordering / strategy plan.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hdsdp_mi355x.h")
REFERENCE = "/root/reference"


def _declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(H(?:KKT|FpLinsys|Mi)\w+)\s*\(", txt)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_library_exports_every_declared_symbol():
    from hdsdp_amd import api
    lib = api.load_library()
    declared = _declared_functions()
    assert len(declared) >= 50
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(api.EXPORTS) == declared, "api.EXPORTS and the header disagree"
    # the reference operator surface is complete: 12 HKKT* + 13 HFpLinsys*
    assert len([n for n in declared if n.startswith("HKKT")]) == 12
    assert len([n for n in declared if n.startswith("HFpLinsys")]) == 13
    assert lib.HMiVersion().startswith(b"hdsdp-mi355x")


def test_no_cpu_fallback_without_gpu():
    """on a box without a GPU the compute entry points must fail loudly, not fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hdsdp_amd import api
    lib = api.load_library()
    h = C.c_void_p()
    assert lib.HFpLinsysCreate(C.byref(h), 8, api.HDSDP_LINSYS_DENSE_DIRECT) != api.RETCODE_OK
    with pytest.raises(api.HDSDPError):
        api.SDPCone.synthetic(8, 4)


@pytest.mark.skipif(not os.path.isdir(REFERENCE) or shutil.which("gcc") is None, reason="reference headers absent")
def test_struct_layout_matches_reference(tmp_path):
    """offsetof/sizeof of hdsdp_kkt, hdsdp_linsys_fp and hdsdp_cone: our header vs the reference's"""
    fields = {
        "hdsdp_kkt": ["nRow", "nCones", "maxConeDim", "cones", "isKKTSparse", "kktM", "invBuffer", "kktBuffer",
                      "kktBuffer2", "kktMatBeg", "kktMatIdx", "kktMatElem", "kktDiag", "dASinvVec", "dASinvCSinvVec",
                      "dASinvRdSinvVec", "dCSinvCSinv", "dCSinvRdSinv", "dCSinv", "dTraceSinv", "dPrimalX"],
        "hdsdp_linsys_fp": ["nCol", "chol", "LinType", "cholCreate", "cholSetParam", "cholSymbolic", "cholNumeric",
                            "cholPsdCheck", "cholFSolve", "cholBSolve", "cholSolve", "cholGetDiag", "cholInvert",
                            "cholDestroy", "nSolves", "nFactorizes"],
        "hdsdp_cone": ["iCone", "cone", "usrData", "coneData", "coneCreate", "coneProcData", "conePresolveData",
                       "coneDestroyData", "coneSetStart", "coneUpdate", "coneRatioTest", "coneGetSymNnz", "coneGetDim",
                       "coneAddSymNz", "coneGetKKTMap", "coneBuildSchur", "coneBuildSchurFixed",
                       "coneBuildPrimalDirection", "coneInteriorCheck", "coneInteriorCheckExpert", "coneGetBarrier",
                       "coneAxpyBufferAndCheck", "coneReduceResi", "coneSetPerturb", "conePRecover", "coneDRecover",
                       "coneATimesXpy", "coneTraceCX", "coneXDotS", "coneGetCoeffNorm", "coneGetObjNorm", "coneScal",
                       "coneView", "getstat"],
    }
    body = "".join(
        f'printf("{s} %zu\\n", sizeof({s}));\n' + "".join(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));\n' for f in fl)
        for s, fl in fields.items())
    head, tail = "#include <stdio.h>\n#include <stddef.h>\n", "\nint main(void){\n" + body + "return 0;}\n"
    outs = []
    for tag, inc, flags in (("ours", f'#include "{HEADER}"', []),
                            ("ref", '#include "interface/hdsdp_schur.h"', ["-DHEADERPATH", f"-I{REFERENCE}"])):
        src = tmp_path / f"{tag}.c"
        src.write_text(head + inc + tail)
        exe = tmp_path / tag
        subprocess.check_call(["gcc", "-w", "-std=gnu99"] + flags + ["-o", str(exe), str(src)])
        outs.append(subprocess.check_output([str(exe)], text=True))
    assert outs[0] == outs[1]


@pytest.mark.parametrize("name", ["theta1_A", "mcp100_A", "gpp100_A", "mix40_A", "mix40_B"])
def test_host_presolve_matches_reference(name):
    from hdsdp_amd import api
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    p = api.presolve_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    for k in ("coef_type", "coef_rank", "coef_nnz", "kkt_perm", "kkt_strategy"):
        assert np.array_equal(p[k], g[k]), k
    assert p["obj_type"] == int(g["obj_type"][0])


def test_presolve_refuses_malformed_columns():
    """the ingest's presolve (csrc/coeff.cpp: mi_coeff_build) takes the caller's columns as they come: column pointers that
    run backwards and a packed position given twice are refused (RETCODE_FAILED), not read as a zero matrix or as a matrix
    whose value depends on the coefficient class"""
    from hdsdp_amd import api
    n, m = 6, 2
    P = n * (n + 1) // 2
    diag = np.array([0, 6, 11, 15, 18, 20], dtype=np.int32)          # packed positions of the diagonal
    good_beg = np.array([0, 6, 8, 10], dtype=np.int32)
    idx = np.concatenate([diag, [0, 7], [3, 9]]).astype(np.int32)
    val = np.ones(idx.size)
    p = api.presolve_csc(n, m, good_beg, idx, val)                      # the well-formed block passes
    assert p["coef_nnz"].tolist() == [2, 2]
    with pytest.raises(RuntimeError):                                   # column 2 ends before it begins
        api.presolve_csc(n, m, np.array([0, 6, 10, 8], dtype=np.int32), idx, val)
    dup = idx.copy()
    dup[7] = 0                                                          # A_1 names packed position 0 twice
    with pytest.raises(RuntimeError):
        api.presolve_csc(n, m, good_beg, dup, val)
    far = idx.copy()
    far[9] = P                                                          # one past the last packed position
    with pytest.raises(RuntimeError):
        api.presolve_csc(n, m, good_beg, far, val)


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under hdsdp_amd/ may import, link or exec it"""
    pkg = os.path.join(ROOT, "hdsdp_amd")
    for dp, dn, fn in os.walk(pkg):
        dn[:] = [d for d in dn if d not in ("build", "__pycache__")]
        for f in fn:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                for needle in ("oracle", "libhdsdp_ref", "ref_dump"):
                    assert needle not in txt, (os.path.join(dp, f), needle)


@pytest.mark.parametrize("inst", ["theta1", "mcp100", "gpp100"])
def test_sdpa_reader_matches_reference_reader(inst):
    """our .dat-s reader vs the CSC the reference's HReadSDPA produced (stored in the golden files)"""
    from hdsdp_amd import api
    from util import GOLDEN
    d = api.read_sdpa(os.path.join(GOLDEN, inst + ".dat-s"))
    g = load_golden(inst + "_A")
    assert d["m"] == int(g["dims"][1]) and len(d["blocks"]) == 1 and d["blocks"][0]["n"] == int(g["dims"][0])
    blk = d["blocks"][0]
    assert np.array_equal(blk["beg"], g["csc_beg"])
    assert np.array_equal(blk["idx"], g["csc_idx"])
    assert np.array_equal(blk["val"], g["csc_val"])
    assert np.array_equal(d["b"], g["b"])


def test_sdpa_reader_multi_block_and_errors(tmp_path):
    from hdsdp_amd import api
    from util import GOLDEN
    d = api.read_sdpa(os.path.join(GOLDEN, "truss1.dat-s"))   # 6 blocks of 2 + one of 1
    assert d["m"] == 6 and [b["n"] for b in d["blocks"]] == [2, 2, 2, 2, 2, 2, 1]
    assert d["blocks"][6]["beg"][1] == 1 and d["blocks"][6]["val"][0] == 1.0   # C = -F0: "0 7 1 1 -1.0"
    bad = tmp_path / "bad.dat-s"
    bad.write_text("2\n1\n3\n1.0 2.0\n1 1 1 9 1.0\n")
    with pytest.raises(api.HDSDPError):
        api.read_sdpa(str(bad))
    with pytest.raises(api.HDSDPError):
        api.read_sdpa(str(tmp_path / "missing.dat-s"))
    ragged = tmp_path / "ragged.dat-s"
    ragged.write_text('"comment\n* another\n 2 = m\n 1\n {3}\n {1.0, 2.0}\n0 1 1 1 2.0\n1 1 2 1 0.5\n1 1 3 3 1e-13\n2 1 3 3 -4\n')
    r = api.read_sdpa(str(ragged))
    blk = r["blocks"][0]
    assert list(blk["beg"]) == [0, 1, 2, 3] and list(blk["idx"]) == [0, 1, 5] and list(blk["val"]) == [-2.0, 0.5, -4.0]


def test_lanczos_start_vector_matches_libc():
    """the engine reproduces glibc's srand()/rand() stream for the reference's Lanczos start vector
    (linalg/hdsdp_lanczos.c:33-42) without touching libc state: compare with libc itself"""
    import ctypes
    import math
    from hdsdp_amd import api
    libc = ctypes.CDLL("libc.so.6")
    for n in (3, 50, 257):
        libc.srand(ctypes.c_uint(n))
        ref = np.zeros(n)
        for i in range(n):
            libc.srand(ctypes.c_uint(libc.rand()))
            a = libc.rand() % 1627
            b = libc.rand() % 2
            ref[i] = math.sqrt(math.sqrt(a)) * (b - 0.5)
        assert np.array_equal(api.lanczos_start_vector(n), ref)


def test_reference_minus_two_files_links_against_the_library():
    """the drop-in claim at link level: the reference without interface/hdsdp_schur.c and linalg/hdsdp_linsolver.c,
    linked with --no-undefined against the product library (make -C oracle drop), takes every HKKT* / HFpLinsys* symbol
    it calls from the library and has no unresolved symbol left"""
    import subprocess
    so = os.path.join(ROOT, "oracle", "_ref", "libhdsdp_ref_minus.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libhdsdp_ref_minus.so not built (needs /root/reference: make -C oracle drop)")
    nm = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    wanted = sorted({ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith(("HKKT", "HFpLinsys"))})
    assert len(wanted) >= 20, wanted          # the driver and the cones really call into the replaced files
    lib = os.path.join(ROOT, "hdsdp_amd", "libhdsdp_mi355x.so")
    ours = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    have = {ln.split()[-1] for ln in ours.splitlines() if ln.split()}
    assert not [w for w in wanted if w not in have]
    r = subprocess.run(["ldd", "-r", so], capture_output=True, text=True)
    assert "undefined symbol" not in (r.stdout + r.stderr), (r.stdout + r.stderr)[-2000:]


def test_rcm_order_recovers_a_band_and_is_a_permutation():
    """host logic of the sparse Schur operator (engine.hip: rcm_order, no device call): a banded pattern whose rows were renumbered
    at random -- two disconnected chains and an isolated row -- comes back with a bandwidth close to the original one, and
    the result is a permutation whatever the pattern (empty, diagonal only, dense)"""
    import ctypes as C
    from hdsdp_amd import api
    lib = api.load_library()
    ip = C.POINTER(C.c_int)

    def order(m, pairs):
        cols = [[] for _ in range(m)]
        for r, c in set((max(a, b), min(a, b)) for a, b in pairs) | set((i, i) for i in range(m)):
            cols[c].append(r)
        beg, idx = [0], []
        for c in range(m):
            idx += sorted(cols[c]); beg.append(len(idx))
        beg, idx = np.array(beg, dtype=np.int32), np.array(idx, dtype=np.int32)
        perm = np.zeros(m, dtype=np.int32)
        assert lib.HMiRcmOrder(m, beg.ctypes.data_as(ip), idx.ctypes.data_as(ip), perm.ctypes.data_as(ip)) == 0
        assert sorted(perm.tolist()) == list(range(m))
        return perm

    rng = np.random.default_rng(3)
    m, half = 601, 300
    band = [(i, j) for i in range(half) for j in range(max(0, i - 6), i)]                     # chain 1: rows 0..299, bandwidth 6
    band += [(half + i, half + j) for i in range(half) for j in range(max(0, i - 3), i)]      # chain 2: rows 300..599, bandwidth 3
    renum = rng.permutation(m)                                                                 # row 600 stays isolated
    scr = [(int(renum[a]), int(renum[b])) for a, b in band]
    assert max(abs(a - b) for a, b in scr) > 300
    perm = order(m, scr)
    assert max(abs(int(perm[a]) - int(perm[b])) for a, b in scr) <= 16
    for pairs in ([], [(i, j) for i in range(40) for j in range(i)]):
        order(40, pairs)


def test_bench_labels_follow_the_size_and_a_gpu_count_it_cannot_meet_is_an_error():
    """bench.py may call a run configs[3] / configs[4] only at those sizes, and `--gpus N` without N devices (and without
    torchrun) must fail instead of timing one GPU under an N-GPU label"""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.workload_label(2000, 2000).startswith("configs[3]")
    assert bench.workload_label(2000, 8000).startswith("configs[4]")
    assert bench.workload_label(2000, 4000).startswith("custom")
    assert bench.workload_label(1000, 8000).startswith("custom")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--no-cpu"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "--gpus 8 requested" in (r.stderr + r.stdout), (r.stderr + r.stdout)[-500:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_device_plan_labels_and_the_gpu_tests_expectations_agree():
    """bench.plan_devices (pure: no GPU call) for 1, 2 and 8 visible devices, with the whole-group RCCL self-test faked to
    pass and to fail: what the bench line will say (transport, rccl_ranks, fallback reason) and what bench.py will ask
    HMiSetDevicesEx for are one statement; and tests/test_gpu_group.py's at-size test asks for exactly the two transports
    the header's switch table names as its coverage."""
    import importlib.util
    from hdsdp_amd import api
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    passed = lambda ids: (True, None)
    failed = lambda ids: (False, "RCCL self-test over devices %s failed at stage 4 (fake)" % list(ids))
    never = lambda ids: (_ for _ in ()).throw(AssertionError("the self-test must not run here"))
    for ndev in (1, 2, 8):
        one = bench.plan_devices(1, ndev, 1, False, never)
        assert one["mode"] == "single" and one["rccl_ranks"] == 0 and one["devices_used"] == 1
        for gpus in (2, 8):
            if gpus > ndev:
                assert "error" in bench.plan_devices(gpus, ndev, 1, False, never)
                lb = bench.plan_devices(gpus, ndev, 1, True, never)      # shared devices: copies, and the line says why
                assert lb["transport_request"] == api.TRANSPORT_COPY and lb["rccl_ranks"] == 0
                assert lb["devices_used"] == ndev < gpus and "share devices" in lb["transport_fallback_reason"]
                assert lb["ids"] == [r % ndev for r in range(gpus)]
                continue
            ok = bench.plan_devices(gpus, ndev, 1, False, passed)
            assert ok["ids"] == list(range(gpus)) and ok["devices_used"] == gpus
            assert (ok["transport"], ok["transport_request"], ok["rccl_ranks"], ok["transport_fallback_reason"]) == \
                   ("rccl", api.TRANSPORT_RCCL, gpus, None)
            bad = bench.plan_devices(gpus, ndev, 1, False, failed)
            assert (bad["transport"], bad["transport_request"], bad["rccl_ranks"]) == ("device copies", api.TRANSPORT_COPY, 0)
            assert "stage 4" in bad["transport_fallback_reason"]
            tr = bench.plan_devices(gpus, ndev, gpus, False, never)      # under torchrun: torch.distributed carries it
            assert tr["mode"] == "torchrun" and tr["rccl_ranks_if_nccl"] == gpus
    assert "error" in bench.plan_devices(4, 8, 2, False, never)          # WORLD_SIZE disagrees with --gpus
    assert "error" in bench.plan_devices(1, 0, 1, False, never)          # no device: no CPU fallback
    # the at-size test states its transport (it used to assert RCCL after a call that defaulted to copies)
    src = open(os.path.join(ROOT, "tests", "test_gpu_group.py")).read()
    body = src.split("def test_config5_at_size_on_eight_devices")[1].split("\ndef ")[0]
    assert 'parametrize("transport", ["rccl", "copy"])' in src and "transport=want" in body and "got == want" in body
    table = open(HEADER).read().split("environment switches")[1]
    assert "test_config5_at_size_on_eight_devices[rccl|copy]" in table


def test_a_multi_gpu_bench_line_says_where_its_step_went():
    """bench.sharded_step_profile (pure) condenses the per-rank profiles of the last sharded build (HMiConeGetBuildProfile) into
    the `sharded_step` object of the bench line: every stage and every exchange piece as [min, max] over the ranks, the exchange
    WAIT per piece, bytes sent and achieved GB/s -- so that the first 8-GPU line diagnoses itself.  Here: two fake ranks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def rank(scale):
        return {"pieces": 2, "staged": True, "invert_ms": 0.6 * scale, "congruence_step1_ms": 40.0 * scale, "slab_reduce_ms": 1.0,
                "allreduce_ms": 0.8 * scale, "extract_ms": 0.1, "world": 2, "step2_piece_ms": [30.0, 50.0 * scale],
                "exchange_wait_ms": [0.0, 12.0 * scale], "exchange_wait_host_ms": [1.0, 20.0], "gram_piece_ms": [25.0, 26.0],
                "piece_bytes_sent": [4.0e9, 4.0e9], "piece_flight_ms": [80.0, 100.0 * scale]}

    stages = [{"assemble_S+chol_S": 2.0, "buildup": 200.0, "factor_M": 1.4, "solve3": 1.2},
              {"assemble_S+chol_S": 2.1, "buildup": 201.0, "factor_M": 1.5, "solve3": 1.2}]
    d = bench.sharded_step_profile([rank(1.0), rank(1.5)], stages)
    mmx = d["min_max_over_ranks_ms"]
    assert d["ranks"] == 2 and d["pieces"] == 2 and d["staged"] is True
    assert mmx["congruence_step1_ms"] == [40.0, 60.0] and mmx["allreduce_ms"] == [0.8, 1.2]
    assert mmx["exchange_wait_by_piece"] == [[0.0, 0.0], [12.0, 18.0]] and mmx["exchange_wait_total"] == [12.0, 18.0]
    assert mmx["congruence_step2_by_piece"][1] == [50.0, 75.0] and mmx["gram_total"] == [51.0, 51.0]
    assert d["bytes_sent_per_rank"] == [8.0e9, 8.0e9] and d["piece_bytes_sent"] == [4000000000, 4000000000]
    assert d["piece_gb_per_s"][0] == [50.0, 50.0] and d["piece_gb_per_s"][1][0] == pytest.approx(26.667, abs=1e-3)
    assert d["replicated_ms"]["factor_M"] == [1.4, 1.5]
    assert bench.sharded_step_profile([None, None], stages) is None          # one GPU: no sharded build, no object
    # the line carries it in both multi-GPU modes, and the fields come from the C ABI's profile
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'out["sharded_step"] = shard_prof' in src and "all_gather_object" in src and "cone.build_profile(r) for r in range(shards)" in src
    assert "HMiConeGetBuildProfile" in open(HEADER).read()


def test_library_exports_nothing_but_the_declared_surface():
    """-fvisibility=hidden: the dynamic symbol table holds the header's functions and no internal symbol (round 2's library
    exported 197 symbols for a boundary of 25 + helpers)"""
    from hdsdp_amd import api
    out = subprocess.run(["nm", "-D", "--defined-only", api.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == _declared_functions(), sorted(set(exported) ^ set(_declared_functions()))


def test_every_environment_switch_is_listed_in_the_header_and_covered():
    """the header's table of environment switches against the sources (both ways), and every switch that selects a code path
    against the settings tests/test_gpu_switches.py runs or the test the table names"""
    csrc = os.path.join(ROOT, "hdsdp_amd", "csrc")
    read = set()
    diag_only = set()
    for f in os.listdir(csrc):
        if not f.endswith((".hip", ".h", ".cpp")):
            continue
        txt = open(os.path.join(csrc, f)).read()
        # variables read only inside #ifdef HDM_DIAGNOSTICS blocks do not exist in the product
        parts = re.split(r"#ifdef HDM_DIAGNOSTICS(.*?)#e(?:ndif|lse)", txt, flags=re.S)
        for k, part in enumerate(parts):
            for v in re.findall(r'getenv\("([A-Za-z_0-9]+)"\)', part):
                (diag_only if k % 2 == 1 else read).add(v)
    diag_only -= read
    table = open(HEADER).read().split("environment switches")[1].split("utilities")[0]
    listed = set(re.findall(r"^ \*  ([A-Z][A-Z_0-9]+)\s", table, flags=re.M))
    assert read == listed, (sorted(read - listed), sorted(listed - read))
    assert diag_only <= {"HDM_VAR", "HDM_CONG2_DIRECT", "HDM_DBG_SYNC"}
    import importlib.util
    spec = importlib.util.spec_from_file_location("switch_tests", os.path.join(ROOT, "tests", "test_gpu_switches.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    run = set(k for s in mod.SETTINGS for k in s)
    for line in table.splitlines():
        m = re.match(r"^ \*  ([A-Z][A-Z_0-9]+)\s", line)
        if m and "test_gpu_switches.py" in line:
            assert m.group(1) in run, m.group(1) + " is said to be covered by test_gpu_switches.py but no setting names it"


def test_full_stages_of_the_gemm_roles_carry_no_vector_alu_instruction(tmp_path):
    """every vector ALU instruction between the MFMAs of the fp64 GEMM K loop costs the matrix pipe about 11 cycles
    (profiles/r04_h_issue_mix.txt), so the stage body's addressing is arranged to need none.  That is a property of the machine
    code, which a compiler update or an innocent edit of gemm_tile.h can take away without any test of results noticing: the
    built library's gfx950 code object is disassembled and every full stage of the four persistent kernels -- a barrier-to-barrier
    stretch with exactly 64 MFMAs and nothing else of a tile in it -- is held to zero vector ALU instructions."""
    lib = os.path.join(ROOT, "hdsdp_amd", "libhdsdp_mi355x.so")
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(lib) and shutil.which("objcopy") and os.path.exists(os.path.join(llvm, "clang-offload-bundler"))):
        pytest.skip("library or binutils / ROCm LLVM tools not available")
    fat = str(tmp_path / "fatbin.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    dis = None
    for k, st in enumerate(starts):          # one bundle per translation unit: find the persistent kernels' code object
        en = starts[k + 1] if k + 1 < len(starts) else len(data)
        b, o = str(tmp_path / ("b%d.bin" % k)), str(tmp_path / ("co%d.o" % k))
        open(b, "wb").write(data[st:en])
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + b, "--output=" + o], capture_output=True)
        if not os.path.exists(o):
            continue
        syms = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-t", o], capture_output=True, text=True).stdout
        if "hdm_gemm_persist_kernel" in syms:
            dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--no-show-raw-insn", o], capture_output=True, text=True).stdout
            break
    assert dis, "no code object with the persistent GEMM kernels in the library"
    funcs, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur and line.startswith("\t"):
            funcs[cur].append(line.split()[0])
    kernels = {k: v for k, v in funcs.items() if "hdm_gemm_persist_kernel" in k}
    assert len(kernels) == 4, sorted(kernels)
    for name, ins in kernels.items():
        windows, w = [], []
        for op in ins:
            if op == "s_barrier":
                windows.append(w)
                w = []
            else:
                w.append(op)
        full = [x for x in windows if sum(o.startswith("v_mfma") for o in x) == 64 and len(x) <= 160]
        assert len(full) >= 1, (name, len(full))
        for x in full:
            valu = [o for o in x if o.startswith("v_") and not o.startswith("v_mfma")]
            assert not valu, (name, valu)
            assert sum(o.startswith("ds_read") for o in x) == 32 and sum(o.startswith("buffer_load_dwordx4") for o in x) == 8, name
