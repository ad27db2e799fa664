"""Several GPUs behind the C ABI (include/hdsdp_mi355x.h: HMiSetDevices; csrc/group_impl.h), rehearsed on the one GPU of
the test box: all shards live on device 0 and exchange by device-to-device copies (the loopback form of the in-process
device group).  Everything goes through the C entry points only -- no torch.distributed, no Python callbacks: the same
calls the reference's single-threaded driver makes.  The sharded answers must equal the one-device answers and the
reference's goldens; the real-link RCCL transport is the same code path with ncclSend/ncclRecv/ncclAllReduce in place of
the copies (exercised here as far as one device allows: HMiRcclSelfTest)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from util import check_close, load_golden, lower_mask, y_of

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def group():
    """configure a loopback device group of the requested size; back to the plain engine afterwards"""
    from hdsdp_amd import api

    def make(world, min_dim=0):
        api.set_devices([0] * world, shard_min_dim=min_dim)
        ids, transport = api.device_group()
        assert ids == [0] * world and transport == 0
    yield make
    api.set_devices([0])


def _phase_a(api, cone, kkt, Rd, y, hsd=True):
    """every quantity of the operator at one state, as numpy copies"""
    m = cone.m
    cone.set_start(Rd)
    assert cone.check_is_interior(1.0, y)
    out = {"logdet": cone.log_barrier(1.0), "S": cone.dual_matrix()}
    kkt.build_up(api.KKT_TYPE_INFEASIBLE)
    ex = kkt.export()
    out["M_inf"] = kkt.M.copy()
    out["ASinv"], out["ASinvRdSinv"], out["TraceSinv"] = ex["ASinv"].copy(), ex["ASinvRdSinv"].copy(), ex["TraceSinv"]
    kkt.factorize()
    out["d1"], out["d2"] = kkt.solve(cone.traces()), kkt.solve(ex["ASinv"])
    kkt.build_up(api.KKT_TYPE_CORRECTOR)
    exc = kkt.export()
    out["cor_ASinv"], out["cor_ASinvRdSinv"] = exc["ASinv"].copy(), exc["ASinvRdSinv"].copy()
    if hsd:
        kkt.build_up(api.KKT_TYPE_HOMOGENEOUS)
        exh = kkt.export()
        out["M_hsd"] = kkt.M.copy()
        out["ASinvCSinv"] = exh["ASinvCSinv"].copy()
        out["hsd_scal"] = np.array([exh["CSinv"], exh["CSinvCSinv"], exh["CSinvRdSinv"], exh["TraceSinv"]])
    dy = 0.3 * np.cos(0.7 * np.arange(m) + 0.2)
    out["step"] = cone.ratio_test(0.0, dy, 0.0)
    # primal recovery works on S = C - sum y_i A_i without the residual term: near the generator's strictly feasible y0
    from test_gpu_parity import _splitmix_u
    P = cone.n * (cone.n + 1) // 2
    y0 = _splitmix_u(np.uint64(2 * m * P) + np.arange(m, dtype=np.uint64))
    out["X"] = cone.get_primal(0.37, y0 + 1e-3 * np.sin(np.arange(m)), 0.01 * dy)
    assert out["X"] is not None and np.isfinite(out["X"]).all()
    Xs = np.cos(0.01 * np.add.outer(np.arange(cone.n), np.arange(cone.n)))
    out["AX"] = cone.a_times_x(Xs)
    out["norms"] = np.array([cone.coeff_norm(1), cone.coeff_norm(2), cone.obj_norm(1), cone.obj_norm(2)])
    return out


def _compare(a, b, m, tol=1e-11):
    msk = lower_mask(m)
    for k in a:
        x, yv = np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64)
        if k.startswith("M_"):
            check_close(yv[msk], x[msk], k)
            continue
        if k == "S":
            ms = lower_mask(x.shape[0])
            x, yv = x[ms], yv[ms]
        den = max(1e-300, float(np.max(np.abs(x))))
        err = float(np.max(np.abs(x - yv))) / den
        assert err < (1e-8 if k == "step" else tol), (k, err)


@pytest.mark.parametrize("n,m,world", [(96, 50, 2), (200, 131, 3), (130, 77, 8), (520, 45, 2), (400, 41, 3)])
def test_in_process_shards_match_one_device(n, m, world, group):
    """world = 2, 3, 8 on one device through the C entry points == world = 1: all four build types, the Phase-A solves,
    the ratio test, the barrier, primal recovery, A*X and the data norms"""
    from hdsdp_amd import api
    Rd = -2.5 * n
    y = 0.02 * np.sin(1.3 * np.arange(m) + 0.4)
    cone = api.SDPCone.synthetic(n, m)
    assert cone.shard_count() == 1
    kkt = api.KKT(m, [cone])
    ref = _phase_a(api, cone, kkt, Rd, y)
    kkt.destroy(); cone.destroy()
    group(world)
    cone = api.SDPCone.synthetic(n, m)
    try:
        assert cone.shard_count() == world
        kkt = api.KKT(m, [cone])
        got = _phase_a(api, cone, kkt, Rd, y)
        pieces, staged = cone.exchange_stats()
        if n >= 400:
            assert pieces > 1 and staged > 1, (pieces, staged)      # step 2 ran by packed-index range, pieces left in between
        a2a, ar = cone.group_traffic()
        assert a2a > 0 and ar > 0
        _compare(ref, got, m)
        kkt.destroy()
    finally:
        cone.destroy()


def test_streamed_shards_match_one_resident_device(group, monkeypatch):
    """streamed constraint data (MiCone::streamed) under sharding: every shard regenerates ITS rows (cyclic deal: one launch of the
    generator per row) batch by batch; three shards with batches of 8 rows against one device with resident data"""
    from hdsdp_amd import api
    n, m, world = 200, 70, 3
    Rd = -2.5 * n
    y = 0.02 * np.sin(1.3 * np.arange(m) + 0.4)
    cone = api.SDPCone.synthetic(n, m)
    kkt = api.KKT(m, [cone])
    ref = _phase_a(api, cone, kkt, Rd, y)
    assert cone.streaming() == (False, 0)
    kkt.destroy(); cone.destroy()
    group(world)
    monkeypatch.setenv("HDSDP_MI355X_STREAM_A", "1")
    monkeypatch.setenv("HDM_BC", "8")
    cone = api.SDPCone.synthetic(n, m)
    try:
        assert cone.shard_count() == world and cone.streaming() == (True, 8)
        kkt = api.KKT(m, [cone])
        got = _phase_a(api, cone, kkt, Rd, y)
        _compare(ref, got, m)
        kkt.destroy()
    finally:
        cone.destroy()


def test_config5_shape_world8(group):
    """BASELINE configs[4] in shape: m = 8000 constraint rows dealt over world = 8 shards (1000 rows each, as on the
    8-GPU node), at n = 256 so that eight shards fit the one GPU of the test box; equal to world = 1 to 1e-11"""
    from hdsdp_amd import api
    n, m, world = 256, 8000, 8
    Rd = -10.0 * n
    y = np.zeros(m)
    res = []
    for w in (1, world):
        if w > 1:
            group(w)
        cone = api.SDPCone.synthetic(n, m)
        try:
            assert cone.shard_count() == w
            kkt = api.KKT(m, [cone], host_mirror=True)
            cone.set_start(Rd)
            assert cone.check_is_interior(1.0, y)
            kkt.build_up(api.KKT_TYPE_INFEASIBLE)
            ex = kkt.export()
            M = kkt.M.copy()
            kkt.factorize()
            d2 = kkt.solve(ex["ASinv"])
            res.append((M, ex["ASinv"].copy(), ex["ASinvRdSinv"].copy(), d2, cone.log_barrier(1.0)))
            kkt.destroy()
        finally:
            cone.destroy()
    msk = lower_mask(m)
    check_close(res[1][0][msk], res[0][0][msk], "M")
    for k in (1, 2, 3):
        err = np.max(np.abs(res[0][k] - res[1][k])) / np.max(np.abs(res[0][k]))
        assert err < 1e-11, (k, err)
    assert abs(res[0][4] - res[1][4]) <= 1e-12 * abs(res[0][4])


@pytest.mark.parametrize("name,world", [("syn64", 2), ("syn100", 3), ("syn200", 2), ("syn96x40_B", 2), ("mix40_A", 2),
                                        ("theta1_B", 2), ("mcp100_A", 3), ("syn2000x32", 2)])
def test_reference_goldens_through_sharded_blocks(name, world, group, monkeypatch):
    """the complete golden check of tests/test_gpu_parity.py::test_schur_against_reference -- S, log det, both ratio tests,
    the checker buffer, all four build types, the Phase-A solves, fixed-strategy builds, primal recovery, the cone
    utilities, KKT_TYPE_PRIMAL -- with the block sharded over `world` in-process shards: what the compiled reference
    computed must come out of the group cone as it comes out of the plain one"""
    from hdsdp_amd import api
    import test_gpu_parity
    monkeypatch.setenv("HDSDP_MI355X_FORCE_GEMM", "1")      # theta1 / mcp100 would take the gather / rank-one path
    group(world)
    seen = []
    real = test_gpu_parity._make_cone

    def spy(nm, g):
        cone, n, m = real(nm, g)
        seen.append(cone.shard_count())
        return cone, n, m
    monkeypatch.setattr(test_gpu_parity, "_make_cone", spy)
    test_gpu_parity.test_schur_against_reference(name)
    assert seen == [world], seen


@pytest.mark.parametrize("name,world", [("theta1_B", 2), ("mix40_A", 3)])
def test_sharded_ingested_blocks_with_rows_in_the_compressed_copy(name, world, group, monkeypatch):
    """the same complete golden check with the shards' INGESTED rows streamed (HDSDP_MI355X_STREAM_A=1, round 5): every shard keeps
    its own rows only as the zero-suppressed copy and expands them a few at a time for the congruence -- sharded builds, staged
    exchange, sweeps summed over the shards, all on expanded batches"""
    import test_gpu_parity
    monkeypatch.setenv("HDSDP_MI355X_FORCE_GEMM", "1")
    monkeypatch.setenv("HDSDP_MI355X_STREAM_A", "1")
    monkeypatch.setenv("HDM_BC", "8")
    group(world)
    seen = []
    real = test_gpu_parity._make_cone

    def spy(nm, g):
        cone, n, m = real(nm, g)
        seen.append((cone.shard_count(), cone.streaming()[0]))
        return cone, n, m
    monkeypatch.setattr(test_gpu_parity, "_make_cone", spy)
    test_gpu_parity.test_schur_against_reference(name)
    assert seen == [(world, True)], seen


def test_two_sharded_blocks_in_one_operator(group):
    """two group cones (and a plain one) accumulate into ONE Schur operator, every build type: the sum over the shards
    must touch each block's own contribution only (a corrector build that reduced the operator-wide accumulator would
    count the first block's part `world` times)"""
    from hdsdp_amd import api
    m, dims, world = 60, (96, 130, 40), 3
    y = 0.03 * np.cos(0.9 * np.arange(m))

    def run():
        cones = [api.SDPCone.synthetic(n, m, iCone=i) for i, n in enumerate(dims)]
        try:
            kkt = api.KKT(m, cones)
            for c, n in zip(cones, dims):
                c.set_start(-3.0 * n)
                assert c.check_is_interior(1.0, y)
            out = {}
            for tag, t in (("inf", api.KKT_TYPE_INFEASIBLE), ("cor", api.KKT_TYPE_CORRECTOR), ("hsd", api.KKT_TYPE_HOMOGENEOUS)):
                kkt.build_up(t)
                ex = kkt.export()
                out["M_" + tag] = kkt.M.copy()
                for k in ("ASinv", "ASinvRdSinv", "ASinvCSinv"):
                    out[k + "_" + tag] = np.asarray(ex[k]).copy()
            shards = [c.shard_count() for c in cones]
            kkt.destroy()
            return out, shards
        finally:
            for c in cones:
                c.destroy()

    ref, s1 = run()
    assert s1 == [1, 1, 1]
    group(world, min_dim=90)          # the 40 x 40 block stays a plain single-device cone
    got, sw = run()
    assert sw == [world, world, 1]
    msk = lower_mask(m)
    for k in ref:
        if k.startswith("M_"):
            check_close(got[k][msk], ref[k][msk], k)
        else:
            den = max(1e-300, float(np.max(np.abs(ref[k]))))
            assert float(np.max(np.abs(ref[k] - got[k]))) / den < 1e-11, k


@pytest.mark.parametrize("world,need_gib", [(2, 200), (8, 235)])
def test_full_size_shards_against_host_fp64(world, need_gib, group, monkeypatch):
    """n = m = 2000 (BASELINE configs[3]) sharded over two and over EIGHT in-process shards on one GPU, against the
    independent host fp64 fixture (tests/golden/full2000.npz): the exchange layout, the K-sharded Gram and the all-reduce
    at the headline size, where segment edges, 16 tile columns and eight exchange pieces all come into play -- with eight
    shards exactly the staging an 8-GPU node runs (250 rows per shard in one launch group, step 2 by packed-index range,
    eight pieces), only the transport differs (device copies instead of RCCL)"""
    import torch
    from hdsdp_amd import api
    from test_gpu_parity import check_full_size_state
    free, total = torch.cuda.mem_get_info()
    if free < need_gib * (1 << 30):
        pytest.skip(f"needs about {need_gib} GiB of free HBM for {world} shards of the full-size problem on one device")
    monkeypatch.setenv("HDM_TCAP_GIB", "8")        # congruence batches of 250 rows: the intermediates of all shards fit
    g = load_golden("full2000")
    n, m = int(g["n"]), int(g["m"])
    group(world)
    cone = api.SDPCone.synthetic(n, m)
    try:
        assert cone.shard_count() == world
        kkt = api.KKT(m, [cone])
        check_full_size_state(cone, kkt, g, "bench")
        pieces, staged = cone.exchange_stats()
        assert pieces > 1, pieces
        if world == 8:      # a shard's 250 rows are one launch group: step 2 ran by packed-index range, piece by piece
            assert staged > 1, (pieces, staged)
        check_full_size_state(cone, kkt, g, "hard")
        kkt.destroy()
    finally:
        cone.destroy()


def test_config5_rows_eight_shards_against_host_fp64(group, monkeypatch):
    """m = 8000 constraint rows over eight in-process shards at n = 1000 -- the largest block dimension at which
    BASELINE configs[4]'s row count fits the one GPU of the test box -- against an INDEPENDENT host answer
    (tests/golden/rows_1000x8000.npz, oracle/row_subset_golden.py: rows of M from B_i = S^-1 A_i S^-1 and <B_i, A_j>,
    no congruence, no Gram product, no sharding): rows of M from every shard's share, both vectors in full, log det S,
    and rows of the residual of the three Phase-A solves"""
    import torch
    from hdsdp_amd import api
    from test_gpu_parity import check_row_subset_state
    free, total = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs about 200 GiB of free HBM for eight shards of n = 1000, m = 8000 on one device")
    monkeypatch.setenv("HDM_TCAP_GIB", "8")
    g = load_golden("rows_1000x8000")
    n, m = int(g["n"]), int(g["m"])
    group(8)
    cone = api.SDPCone.synthetic(n, m)
    try:
        assert cone.shard_count() == 8
        kkt = api.KKT(m, [cone], host_mirror=False)
        check_row_subset_state(cone, kkt, g, "bench")
        check_row_subset_state(cone, kkt, g, "hard")
        kkt.destroy()
    finally:
        cone.destroy()


@pytest.mark.parametrize("transport", ["rccl", "copy"])
def test_config5_at_size_on_eight_devices(transport):
    """BASELINE configs[4] itself: n = 2000, m = 8000, constraint rows over 8 MI355X through the in-process device group,
    once with the exchange carried by RCCL (requested explicitly: HMiSetDevicesEx(.., 1), after the whole-group self-test
    has passed) and once by device copies (the library's default), against tests/golden/full8000_rows.npz (host fp64,
    oracle/row_subset_golden.py: rows [0, 64) -- eight per rank -- and the tile-edge rows of M, both vectors in full,
    log det S, residual rows of the three solves, at the bench state and at cond(S) = 1e3).  Needs eight devices (136 GB of
    constraint data, 130 GB of transformed rows); skipped by name on a smaller box -- the one-GPU rehearsals above cover
    the same code with m = 8000 at n = 1000 and eight shards at n = m = 2000, test_config5_at_size_streamed_on_one_device
    the workload itself."""
    import torch
    from hdsdp_amd import api
    from test_gpu_parity import check_row_subset_state
    ndev = torch.cuda.device_count()
    if ndev < 8:
        pytest.skip(f"BASELINE configs[4] (n=2000, m=8000) over 8 devices: {ndev} visible -- never run on this pool")
    g = load_golden("full8000_rows")
    n, m = int(g["n"]), int(g["m"])
    want = api.TRANSPORT_RCCL if transport == "rccl" else api.TRANSPORT_COPY
    if want == api.TRANSPORT_RCCL:
        assert api.rccl_group_self_test(list(range(8)), 120000) == 0
    api.set_devices(list(range(8)), shard_min_dim=0, transport=want)
    try:
        ids, got = api.device_group()
        assert ids == list(range(8)) and got == want, (ids, got, want)
        cone = api.SDPCone.synthetic(n, m)
        try:
            assert cone.shard_count() == 8
            kkt = api.KKT(m, [cone], host_mirror=False)
            check_row_subset_state(cone, kkt, g, "bench")
            check_row_subset_state(cone, kkt, g, "hard")
            kkt.destroy()
        finally:
            cone.destroy()
    finally:
        api.set_devices([0])


def test_transport_request_is_honoured():
    """HMiSetDevicesEx states the transport; HMiGetDeviceGroup reports what the group really uses.  On the one device of the
    test box: shards that share a device cannot use RCCL (one communicator rank per device) and say so by falling back to
    copies; a one-device "group" is no group; the whole-group self-test refuses repeated ids with its own code (8) and passes
    on the one device there is."""
    from hdsdp_amd import api
    try:
        api.set_devices([0, 0], transport=api.TRANSPORT_RCCL)
        assert api.device_group() == ([0, 0], api.TRANSPORT_COPY)
        api.set_devices([0, 0], transport=api.TRANSPORT_COPY)
        assert api.device_group() == ([0, 0], api.TRANSPORT_COPY)
        api.set_devices([0], transport=api.TRANSPORT_RCCL)
        assert api.device_group() == ([0], -1)
    finally:
        api.set_devices([0])
    assert api.rccl_group_self_test([0, 0]) == 8
    assert api.rccl_group_self_test([0]) == 0
    assert api.rccl_group_self_test([99]) == 1


def test_shard_plan_predicts_what_the_engine_allocates(group):
    """hdsdp_amd.dist.ShardPlan.hbm_bytes -- the statement that BASELINE configs[4] fits 8 x 288 GB
    (tests/test_dist_cpu.py) -- against the device memory the engine really takes, one device and two shards"""
    import torch
    from hdsdp_amd import api
    from hdsdp_amd.dist import ShardPlan
    n, m = 640, 900
    for world in (1, 2):
        if world > 1:
            group(world)
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        cone = api.SDPCone.synthetic(n, m)
        kkt = api.KKT(m, [cone])
        cone.set_start(-10.0 * n)
        assert cone.check_is_interior(1.0, np.zeros(m))
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        kkt.factorize()
        free1, _ = torch.cuda.mem_get_info()
        used = free0 - free1
        plan = sum(ShardPlan(n, m, world).hbm_bytes(r)["total"] for r in range(world))
        if world > 1:   # M and its factor exist once (the caller's operator), not per shard
            plan -= (world - 1) * ShardPlan(n, m, world).hbm_bytes(0)["Schur matrix M + its factor (replicated)"]
        if cone.sweep_info()[0]:   # the zero-suppressed sweep copy is made at creation where it pays: outside the plan's total
            plan += world * (cone.sweep_info()[1] * 8 + cone.sweep_info()[2] // 1024 * 192)   # (a group cone answers for its shard 0)
        kkt.destroy(); cone.destroy()
        assert 0.8 * plan <= used <= 1.25 * plan + (256 << 20), (world, used / 1e9, plan / 1e9)


def test_rccl_is_linked_and_works_in_process():
    """the library links librccl (the group's transport between distinct devices); one device cannot form a group, so
    the transport's three calls -- communicator, all-reduce, grouped send/receive -- run on a one-rank communicator and
    are checked for their results"""
    from hdsdp_amd import api
    lib = api.load_library()
    assert lib.HMiRcclSelfTest(-1) == 0
    out = subprocess.run(["ldd", api.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in out


def test_unchanged_driver_shards_by_environment(tmp_path):
    """the reference's own solver main (oracle/_ref/sdpasolve_mi355x, engine cones attached at presolve) with nothing but
    HDSDP_MI355X_GPUS=2 HDSDP_MI355X_LOOPBACK=1 in its environment: the dense block is sharded behind HKKTBuildUp and the
    solve reaches the pure reference's optimum"""
    exe = os.path.join(ROOT, "oracle", "_ref", "sdpasolve_mi355x")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/sdpasolve_mi355x not built (needs /root/reference at build time)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from synth_sdpa import write_synth_sdpa
    fname = str(tmp_path / "syn120.dat-s")
    write_synth_sdpa(120, 120, fname)
    env = dict(os.environ, HDSDP_DROP_ATTACH="1", HDSDP_MI355X_GPUS="2", HDSDP_MI355X_LOOPBACK="1",
               HDSDP_MI355X_SHARD_MIN_N="64")
    r = subprocess.run([exe, fname], capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "device group: 2 shards" in out, out[-3000:]
    assert "SDP Status: Primal dual optimal" in out, out[-3000:]
    import re
    dobj = float(re.search(r"dObj\s+([-+0-9.eE]+)", out).group(1))
    assert abs(dobj - (-36.746433644)) <= 1e-6 * 36.746433644, dobj


def test_a_sharded_build_says_where_its_time_went(group):
    """HMiConeGetBuildProfile: after a sharded Schur build every shard reports its stages -- triangular inverse, congruence step
    1, step 2 per exchange piece, the exchange WAIT per piece (engine stream idle because the piece had not arrived), the Gram
    splits per piece, slab reduction, all-reduce -- and the bytes it sent per piece; bench.py puts min / max over the ranks on a
    multi-GPU line (`sharded_step`).  On one device (loopback) the transfers are device copies: the waits of the shard that
    queued first are next to nothing, and the bytes add up to what the exchange moves."""
    from hdsdp_amd import api
    n, m, world = 384, 96, 2
    y = 0.02 * np.cos(0.3 * np.arange(m))
    group(world)
    cone = api.SDPCone.synthetic(n, m)
    try:
        assert cone.shard_count() == world
        cone.set_start(-500.0)
        assert cone.check_is_interior(1.0, y)
        assert cone.build_profile(0) is None                      # no sharded build yet
        kkt = api.KKT(m, [cone])
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        profs = [cone.build_profile(r) for r in range(world)]
        pieces, launches = cone.exchange_stats()
        for p in profs:
            assert p is not None and p["world"] == world and p["pieces"] == pieces and p["staged"] == (launches > 0)
            for k in ("step2_piece_ms", "exchange_wait_ms", "exchange_wait_host_ms", "gram_piece_ms", "piece_bytes_sent", "piece_flight_ms"):
                assert len(p[k]) == pieces and all(np.isfinite(v) and v >= 0.0 for v in p[k]), k
            assert p["invert_ms"] > 0 and p["allreduce_ms"] > 0 and sum(p["gram_piece_ms"]) > 0
            assert (p["congruence_step1_ms"] if p["staged"] else p["congruence_ms"]) > 0
            if p["staged"]:
                assert sum(p["step2_piece_ms"]) > 0
        # every shard sends (world - 1) / world of its transformed rows: the group's own byte count agrees
        a2a_bytes, _ = cone.group_traffic()
        assert abs(sum(profs[0]["piece_bytes_sent"]) - a2a_bytes) <= 1e-9 * a2a_bytes
        # the engine stream never sat idle for long before a piece's Gram splits on the shard that was served first
        assert min(sum(p["exchange_wait_ms"]) for p in profs) < 5.0
        assert cone.build_profile(world) is None                  # no such shard
        kkt.destroy()
    finally:
        cone.destroy()
