"""world_size-2 (and 3) gloo tests on CPU of the N > 1 path: the sharding plan and the two collectives
(all-to-all transpose + all-reduce) reproduce the unsharded Schur quantities."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "dist_worker.py")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(mode, world, n, m, out, timeout=600):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(n), str(m), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_schur_choreography_gloo(world, tmp_path):
    n, m = 20, 7  # n not a multiple of 16, m not a multiple of world: ragged shards and padding
    out1 = str(tmp_path / "w1.npy")
    outw = str(tmp_path / "ww.npy")
    launch("numpy", 1, n, m, out1)
    launch("numpy", world, n, m, outw)
    assert np.allclose(np.load(out1), np.load(outw), rtol=1e-12, atol=1e-14)


def test_shard_plan_invariants():
    from hdsdp_amd.dist import ShardPlan
    for n, m, w in [(2000, 2000, 1), (2000, 2000, 8), (2000, 8000, 8), (50, 104, 2), (100, 101, 4), (17, 3, 2)]:
        p = ShardPlan(n, m, w)
        rs = p.rows_seg()
        assert sorted(rs[rs >= 0].tolist()) == list(range(m))          # every constraint exactly once
        assert (rs == -2).sum() == (rs == -3).sum() == (rs == -4).sum() == 1
        assert p.npb_loc * w >= p.npb and p.R == w * p.Lr
        assert sum(len(p.owned(r)) for r in range(w)) == m
        if w > 1:
            assert p.Lr % 128 == 0                                      # Gram tiles never straddle segments
        # balanced by construction: shard sizes differ by at most one row
        sizes = [len(p.owned(r)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    # transpose volume: each rank sends (w-1)/w of its rows' data -- 1/w of an all-gather
    p = ShardPlan(2000, 2000, 8)
    assert p.chunk * 8 * 7 < 0.15 * (p.npb * 16 * 2000 * 8)


def test_config5_fits_the_hbm_of_eight_mi355x():
    """BASELINE configs[4] (n = 2000, m = 8000 over 8 GPUs): what one rank allocates, by the plan that mirrors the engine's
    allocation rules, against 288 GB of HBM3E per MI355X -- half of it; two ranks could not hold the problem"""
    from hdsdp_amd.dist import ShardPlan
    HBM = 288e9
    for rank in (0, 7):
        parts = ShardPlan(2000, 8000, 8).hbm_bytes(rank)
        assert parts["total"] < 0.6 * HBM, parts
        assert 0.52 * 1000 * 2000 * 2000 * 8 < parts["A (A_L form, skyline)"] < 0.55 * 1000 * 2000 * 2000 * 8
    one = ShardPlan(2000, 2000, 1).hbm_bytes(0)
    assert 90e9 < one["total"] < 110e9, one           # about 100 GB for the 32 GB problem
    assert ShardPlan(2000, 8000, 2).hbm_bytes(0)["total"] < HBM < ShardPlan(2000, 8000, 1).hbm_bytes(0)["total"] * 2
