"""Streamed constraint data (include/hdsdp_mi355x.h: HMiConeGetStreaming; csrc/engine_cone.h: MiCone::streamed): the synthetic
family's A_L forms regenerated per batch instead of kept resident -- what lets BASELINE configs[4] (n = 2000, m = 8000: 136 GB
of constraint data beside 130 GB of transformed rows) run on ONE MI355X.  Small blocks are run both ways and must agree; the
workload itself is held to the independent host-fp64 fixture tests/golden/full8000_rows.npz."""
import numpy as np
import pytest

from util import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,bc", [(96, 50, 16), (200, 131, 32), (300, 70, 1024)])
@pytest.mark.parametrize("sweep_copy", [False, True])
def test_streamed_rows_give_what_resident_rows_give(n, m, bc, sweep_copy, monkeypatch):
    """every quantity of the operator (all build types, Phase-A solves, ratio test, barrier, primal recovery, A X, data norms)
    from regenerated batches == from resident data: several batches with a ragged last one (m = 50 in 16s, 131 in 32s) and
    one batch that holds everything; the S / dS sweeps and corrector dots once from the zero-suppressed copy (built batch by
    batch in two passes) and once by regenerating the rows"""
    from hdsdp_amd import api
    from test_gpu_group import _phase_a, _compare
    Rd = -2.5 * n
    y = 0.02 * np.sin(1.3 * np.arange(m) + 0.4)
    monkeypatch.setenv("HDM_BC", str(bc))
    out = []
    for stream in ("0", "1"):
        monkeypatch.setenv("HDSDP_MI355X_STREAM_A", stream)
        cone = api.SDPCone.synthetic(n, m)
        try:
            on, rows = cone.streaming()
            assert on == (stream == "1") and (rows == min(bc, -(-m // -(-m // bc))) if on else rows == 0), (on, rows)
            cone.use_sweep_copy(sweep_copy)
            assert cone.sweep_info()[0] == sweep_copy
            kkt = api.KKT(m, [cone])
            out.append(_phase_a(api, cone, kkt, Rd, y))
            kkt.destroy()
        finally:
            cone.destroy()
    _compare(out[0], out[1], m, tol=1e-12)


def test_the_sweep_copy_of_streamed_rows_is_the_copy_of_resident_rows(monkeypatch):
    """the zero-suppressed copy built from regenerated batches (counts, then values, batch after batch) holds the same number
    of values as the one built from resident data, and the sweeps from it give the same BITS (same sums in the same order)"""
    from hdsdp_amd import api
    n, m = 260, 90
    y = 0.05 * np.cos(0.9 * np.arange(m))
    monkeypatch.setenv("HDM_BC", "24")
    got = []
    for stream in ("0", "1"):
        monkeypatch.setenv("HDSDP_MI355X_STREAM_A", stream)
        cone = api.SDPCone.synthetic(n, m)
        try:
            cone.use_sweep_copy(True)
            cone.set_start(-3.0 * n)
            assert cone.check_is_interior(0.7, y)
            got.append((cone.sweep_info(), cone.dual_matrix().copy()))
        finally:
            cone.destroy()
    from util import lower_mask
    assert got[0][0] == got[1][0]
    assert np.array_equal(got[0][1][lower_mask(n)], got[1][1][lower_mask(n)])      # (S is lower-valid: nothing writes the other half)


@pytest.mark.parametrize("mode", ["auto", "resident"])
def test_config5_at_size_on_one_device(mode, monkeypatch):
    """BASELINE configs[4] itself -- n = 2000, m = 8000 -- on ONE device against tests/golden/full8000_rows.npz (host fp64,
    oracle/row_subset_golden.py: 72 complete rows of M incl. the tile-edge rows, both vectors in full, log det S, tr S^-1,
    residual rows of the three Phase-A solves; bench state and a cond(S) = 1e3 state with y != 0, whose S assembly sweeps all
    8000 constraint matrices).  "auto": the cone decides from the memory it finds -- 136 GB of constraint data beside 130 GB
    of transformed rows and 33 GB of work buffers leave no slack on a 288 GiB device, so it streams (regenerates the rows per
    congruence batch) and the sweeps read the zero-suppressed copy, for which there is room then; "resident": streaming
    forbidden (HDSDP_MI355X_STREAM_A=0) -- it fits when nothing else is on the device (285 of 287 GiB), skipped when not."""
    import torch
    from hdsdp_amd import api
    from test_gpu_parity import check_row_subset_state
    free, total = torch.cuda.mem_get_info()
    if free < 250 * (1 << 30):
        pytest.skip(f"BASELINE configs[4] on one device needs about 250 GiB of free HBM, {free >> 30} GiB free")
    if mode == "resident":
        monkeypatch.setenv("HDSDP_MI355X_STREAM_A", "0")
    g = load_golden("full8000_rows")
    n, m = int(g["n"]), int(g["m"])
    try:
        cone = api.SDPCone.synthetic(n, m)
    except api.HDSDPError:
        if mode == "resident":
            pytest.skip("136 GB of resident constraint data do not fit next to the rest on this device today")
        raise
    try:
        on, rows = cone.streaming()
        assert (on, rows) == ((True, 1000) if mode == "auto" else (False, 0)), (mode, on, rows)
        assert cone.shard_count() == 1
        kkt = api.KKT(m, [cone], host_mirror=False)
        try:
            check_row_subset_state(cone, kkt, g, "bench")
        except api.HDSDPError:
            if mode == "resident":
                pytest.skip("the resident form ran out of device memory in its first build (285 of 287 GiB needed)")
            raise
        check_row_subset_state(cone, kkt, g, "hard")
        if mode == "auto":
            assert cone.sweep_info()[0], "room for the zero-suppressed sweep copy is what streaming buys"
        kkt.destroy()
    finally:
        cone.destroy()


@pytest.mark.parametrize("name,bc", [("syn96x40_B", 16), ("syn200", 64), ("mix40_B", 8)])
def test_ingested_rows_kept_only_as_the_compressed_copy(name, bc, monkeypatch):
    """round 5: INGESTED rows that do not stay resident live in the zero-suppressed copy alone (csrc/engine_create.h:
    upload_streamed_rows) and a batch of A_L forms is expanded from it wherever the dense form is read (congruence, data norms,
    A X, indefinite-X builds) -- what the counter-based generator does for the synthetic family.  A CSC block taken in both ways
    (HDSDP_MI355X_STREAM_A = 0 / 1, several batches with a ragged last one) gives the same operator in every quantity, with the
    sweeps reading the copy and with the sweeps expanding batches; and the reference's numbers (the golden's M)."""
    import os
    import sys
    from hdsdp_amd import api
    from test_gpu_group import _phase_a, _compare
    from util import check_close, lower_mask, y_of
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    if "csc_beg" in g:
        beg, idx, val = g["csc_beg"], g["csc_idx"], g["csc_val"]
    else:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import oracle_py
        beg, idx, val, _ = oracle_py.synth_csc(n, m)
    monkeypatch.setenv("HDM_BC", str(bc))
    monkeypatch.setenv("HDSDP_MI355X_FORCE_GEMM", "1")      # (mix40: its five classes through the congruence + Gram path)
    Rd, y = float(g["Rd"][0]), y_of(g)
    out = []
    for stream, sweep_copy in (("0", True), ("1", True), ("1", False)):
        monkeypatch.setenv("HDSDP_MI355X_STREAM_A", stream)
        cone = api.SDPCone.from_csc(n, m, beg, idx, val)
        try:
            on, rows = cone.streaming()
            assert on == (stream == "1") and (rows > 0) == on, (on, rows)
            cone.use_sweep_copy(sweep_copy)
            kkt = api.KKT(m, [cone])
            if name != "mix40_B":
                out.append(_phase_a(api, cone, kkt, Rd, y))
            else:                      # (a constraint that is zero: singular M, and S = C - sum y A is not definite without the residual)
                cone.set_start(Rd)
                assert cone.check_is_interior(float(g["tau"][0]), y)
                kkt.build_up(api.KKT_TYPE_HOMOGENEOUS)
                ex = kkt.export()
                out.append({"M_hsd": kkt.M.copy(), "ASinv": ex["ASinv"].copy(), "ASinvCSinv": ex["ASinvCSinv"].copy(),
                            "norms": np.array([cone.coeff_norm(1), cone.coeff_norm(2)])})
                check_close(kkt.M[lower_mask(m)], g["M_hsd"][lower_mask(m)], name + " streamed ingest")
            kkt.destroy()
        finally:
            cone.destroy()
    _compare(out[0], out[1], m, tol=1e-12)
    _compare(out[0], out[2], m, tol=1e-12)


def test_an_ingested_config5_block_fits_one_device():
    """BASELINE configs[4]'s shape as INGESTED data -- n = 2000, m = 8000, 6.7e9 entries (81 GB of CSC) pushed through the
    column-by-column builder from the SURVEY 8(d) stream -- on ONE device: its 136 GB of A_L forms never exist, the rows live in
    the 57 GB compressed copy and are expanded 1000 at a time.  Held to tests/golden/full8000_rows.npz like the synthetic cone
    (rows of M, both vectors, log det S, residual rows of the solves, at the bench state and at the cond(S) = 1e3 state)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from hdsdp_amd import api
    from test_gpu_parity import _splitmix_u, check_row_subset_state
    from test_gpu_ingest import _packed_positions
    free, total = torch.cuda.mem_get_info()
    if free < 250 * (1 << 30):
        pytest.skip(f"needs about 250 GiB of free HBM, {free >> 30} GiB free")
    g = load_golden("full8000_rows")
    n, m = int(g["n"]), int(g["m"])
    P = n * (n + 1) // 2
    ii, jj = _packed_positions(n)
    diag = ii == jj
    k = np.arange(P, dtype=np.uint64)
    y0 = _splitmix_u(np.uint64(2 * m * P) + np.arange(m, dtype=np.uint64))

    def column(c):
        base = np.uint64(2 * c * P)
        v = _splitmix_u(base + np.uint64(2) * k)
        w = _splitmix_u(base + np.uint64(2) * k + np.uint64(1))
        keep = diag | (w >= 0.2)
        return np.flatnonzero(keep).astype(np.int32), v[keep]

    Cp = diag.astype(np.float64)
    import os
    import time
    # (several minutes without a dot from pytest: progress goes to a file under gpurun_out/, which the GPU box watches)
    pdir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
    t0 = time.time()

    def note(msg):
        try:                                   # (a progress note must never be the reason the test fails)
            os.makedirs(pdir, exist_ok=True)
            with open(os.path.join(pdir, "ingest8000_progress.txt"), "a") as f:
                f.write("%7.1f s  %s\n" % (time.time() - t0, msg))
        except OSError:
            pass

    def columns():
        with ThreadPoolExecutor(12) as pool:
            for c, (pi, pv) in enumerate(pool.map(column, range(m))):
                Cp[pi] += y0[c] * pv
                if c % 500 == 499:
                    note("column %d handed to the builder" % (c + 1))
                yield c + 1, pi, pv
        yield 0, np.arange(P, dtype=np.int32), Cp

    cone = api.SDPCone.from_columns(n, m, columns())
    try:
        note("cone created")
        on, rows = cone.streaming()
        assert (on, rows) == (True, 1000) and cone.path == 0 and cone.stored_entries > 6_400_000_000
        assert cone.sweep_info()[0]
        kkt = api.KKT(m, [cone], host_mirror=False)
        check_row_subset_state(cone, kkt, g, "bench")
        note("bench state checked")
        check_row_subset_state(cone, kkt, g, "hard")
        note("hard state checked")
        kkt.destroy()
    finally:
        cone.destroy()
