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


@pytest.mark.parametrize("mode", ["auto", "streamed"])
def test_config5_at_size_on_one_device(mode, monkeypatch, capfd):
    """BASELINE configs[4] itself -- n = 2000, m = 8000 -- on ONE device against tests/golden/full8000_rows.npz (host fp64,
    oracle/row_subset_golden.py: 72 complete rows of M incl. the tile-edge rows, both vectors in full, log det S, tr S^-1,
    residual rows of the three Phase-A solves; bench state and a cond(S) = 1e3 state with y != 0, whose S assembly sweeps all
    8000 constraint matrices).  "auto": whatever the cone chooses from the memory it finds (resident if 136 GB of constraint
    data fit next to 130 GB of transformed rows and the work buffers, else streamed); "streamed": regeneration forced, which
    is also what leaves room for the zero-suppressed sweep copy."""
    import torch
    from hdsdp_amd import api
    from test_gpu_parity import check_row_subset_state
    free, total = torch.cuda.mem_get_info()
    if free < 250 * (1 << 30):
        pytest.skip(f"BASELINE configs[4] on one device needs about 250 GiB of free HBM, {free >> 30} GiB free")
    if mode == "streamed":
        monkeypatch.setenv("HDSDP_MI355X_STREAM_A", "1")
    g = load_golden("full8000_rows")
    n, m = int(g["n"]), int(g["m"])
    cone = api.SDPCone.synthetic(n, m)
    try:
        on, rows = cone.streaming()
        if mode == "streamed":
            assert on and rows == 1000, (on, rows)
        assert cone.shard_count() == 1
        kkt = api.KKT(m, [cone], host_mirror=False)
        check_row_subset_state(cone, kkt, g, "bench")
        check_row_subset_state(cone, kkt, g, "hard")
        print(f"configs[4] on one device [{mode}]: free HBM before {free / 2**30:.1f} of {total / 2**30:.1f} GiB, streamed={on}, "
              f"sweep copy in use={cone.sweep_info()[0]}, after the builds {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB free")
        kkt.destroy()
    finally:
        cone.destroy()
