"""GPU rehearsal of world > 1: two ranks share the one visible MI355X, collectives staged through gloo.
Every rank runs the real HIP path through the C ABI; the sharded answer must equal the one-GPU answer."""
import numpy as np
import pytest

from test_dist_cpu import launch
from util import check_close, lower_mask

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,world", [(96, 50, 2), (200, 131, 2), (130, 77, 3), (520, 45, 2), (400, 41, 3)])
def test_sharded_ranks_match_one(n, m, world, tmp_path):
    out1, out2 = str(tmp_path / "w1.npz"), str(tmp_path / "w2.npz")
    launch("gpu", 1, n, m, out1)
    launch("gpu", world, n, m, out2)
    a, b = np.load(out1), np.load(out2)
    if n >= 400:
        # several tile columns: the second congruence step really ran by packed-index range, in more than one launch,
        # with the exchange pieces leaving in between
        pieces, staged = (int(v) for v in b["xstats"])
        assert pieces > 1 and staged > 1, (pieces, staged)
    msk = lower_mask(m)
    for tag in ("inf", "hsd"):
        check_close(b["M_" + tag][msk], a["M_" + tag][msk], "M_" + tag)
    for k in a.files:
        if k.startswith(("ASinv", "scal", "sol")):
            err = np.max(np.abs(a[k] - b[k])) / max(1e-300, np.max(np.abs(a[k])))
            assert err < 1e-11, (k, err)
    check_close(b["S"][lower_mask(n)], a["S"][lower_mask(n)], "S")


def test_rccl_collectives_on_engine_buffers(tmp_path):
    """the nccl (= RCCL) flavour of hdsdp_amd.dist.Exchange on a one-rank group: the same torch calls the engine's
    callbacks make on N GPUs (all_to_all_single on the torch-owned exchange buffers, all_reduce on a raw engine
    pointer wrapped through the CUDA array interface), checked for values.  Runs in a child process so that the
    process group does not leak into the other tests."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
from hdsdp_amd import api, dist as hdist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
n, m = 96, 40
cone = api.SDPCone.synthetic(n, m)
ex = hdist.Exchange(cone)
assert ex.backend == "nccl" and ex.world == 1
ex.send.copy_(torch.arange(ex.send.numel(), dtype=torch.float64, device="cuda"))
assert ex._alltoall(None) == 0
assert torch.equal(ex.recv, ex.send)
# the piecewise flavour (list all_to_all, asynchronous on the side stream): two pieces of the one chunk
half = ex.plan.chunk // 2
assert ex._alltoall_piece(None, 0, half, 0) == 0 and ex._alltoall_piece(None, half, ex.plan.chunk - half, 1) == 0
assert ex._alltoall_wait(None, 0) == 0 and ex._alltoall_wait(None, 1) == 0
assert torch.equal(ex.recv, ex.send)
buf = torch.full((1000,), 2.5, dtype=torch.float64, device="cuda")
assert ex._allreduce(None, buf.data_ptr(), 1000) == 0
assert float(buf.sum()) == 2500.0
# the cone still builds with the torch-owned buffers behind it
kkt = api.KKT(m, [cone])
cone.set_start(-10.0 * n)
assert cone.check_is_interior(1.0, np.zeros(m))
kkt.build_up(api.KKT_TYPE_INFEASIBLE)
M = kkt.M.copy()
assert np.isfinite(M).all() and M[0, 0] > 0
dist.destroy_process_group()
print("RCCL_OK")
''' % root
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
