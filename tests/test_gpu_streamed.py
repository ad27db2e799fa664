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
