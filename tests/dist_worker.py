"""worker for the multi-process tests (launched by test_dist_*.py, one process per rank)

mode "numpy": CPU only -- numpy stands in for the kernels, torch.distributed(gloo) for RCCL; checks the
              sharding plan + collective choreography of hdsdp_amd/dist.py end to end.
mode "gpu"  : every rank drives the real HIP path through the C ABI on the (single) visible GPU, with the
              collectives staged through gloo; checks world > 1 against the world == 1 answer.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_problem(n, m, seed=7):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n, n))
    A = A + A.transpose(0, 2, 1)
    C = rng.standard_normal((n, n))
    C = C + C.T
    G = rng.standard_normal((n, n))
    S = G @ G.T + n * np.eye(n)
    return A, C, S


def run_numpy(rank, world, n, m, out):
    import torch
    import torch.distributed as dist
    from hdsdp_amd.dist import ShardPlan
    A, C, S = make_problem(n, m)
    Rd = -3.5
    plan = ShardPlan(n, m, world)
    Linv = np.linalg.inv(np.linalg.cholesky(S))
    own = plan.owned(rank)
    loc = np.zeros((world * plan.npb_loc, plan.Lr, 16))
    for q, i in enumerate(own):
        loc[:, q, :] = plan.to_blocked(Linv @ A[i] @ Linv.T)
    if rank == 0:
        loc[:, len(own) + 0, :] = plan.to_blocked(Linv @ Linv.T)          # "I row"
        loc[:, len(own) + 1, :] = plan.to_blocked(np.eye(n))              # "S row"
        loc[:, len(own) + 2, :] = plan.to_blocked(Linv @ C @ Linv.T)      # "C row"
    send = torch.from_numpy(loc.reshape(world, -1).copy())
    assert send.shape[1] == plan.chunk
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send) if world > 1 else recv.copy_(send)
    X = recv.numpy().reshape(world, plan.npb_loc, plan.Lr, 16).transpose(0, 2, 1, 3).reshape(plan.R, -1)
    G = torch.from_numpy(X @ X.T)
    if world > 1:
        dist.all_reduce(G)
    G = G.numpy()
    rs = plan.rows_seg()
    M = np.zeros((m, m))
    asinv, asinvrd, asinvc = np.zeros(m), np.zeros(m), np.zeros(m)
    pI = plan.pI
    for a in range(plan.R):
        if rs[a] < 0:
            continue
        asinvrd[rs[a]] = Rd * G[a, pI]
        asinv[rs[a]] = G[a, pI + 1]
        asinvc[rs[a]] = G[a, pI + 2]
        for b in range(plan.R):
            if rs[b] >= 0:
                M[rs[a], rs[b]] = G[a, b]
    if rank == 0:
        Sinv = np.linalg.inv(S)
        W = np.einsum("ab,ibc->iac", Sinv, A)                              # S^-1 A_i
        Mref = np.einsum("iab,jba->ij", W, W)
        assert np.allclose(M, Mref, rtol=1e-11, atol=1e-13)
        assert np.allclose(asinv, np.einsum("iaa->i", W), rtol=1e-11)
        assert np.allclose(asinvrd, Rd * np.einsum("iab,ba->i", W, Sinv), rtol=1e-11)
        assert np.allclose(asinvc, np.einsum("iab,bc,ca->i", W, Sinv, C), rtol=1e-11)
        assert np.isclose(G[pI + 1, pI], np.trace(Sinv), rtol=1e-12)        # TraceSinv
        assert np.isclose(G[pI + 2, pI + 1], np.trace(C @ Sinv), rtol=1e-11)  # CSinv
        np.save(out, M)


def run_gpu(rank, world, n, m, out):
    from hdsdp_amd import api, dist as hdist
    os.environ["LOCAL_RANK"] = "0"  # every rank shares the one visible GPU in this rehearsal
    Rd = -10.0 * n
    y = 0.02 * np.sin(1.7 * np.arange(1, m + 1))
    cone = api.SDPCone.synthetic(n, m, rank=rank, world=world)
    ex = hdist.Exchange(cone) if world > 1 else None
    kkt = api.KKT(m, [cone])
    cone.set_start(Rd)
    assert cone.check_is_interior(0.9, y)      # sharded S assembly + all-reduce
    res = {}
    for typ, tag in ((api.KKT_TYPE_INFEASIBLE, "inf"), (api.KKT_TYPE_HOMOGENEOUS, "hsd"), (api.KKT_TYPE_CORRECTOR, "cor")):
        kkt.build_up(typ)
        e = kkt.export()
        res["M_" + tag] = kkt.M.copy()
        for k in ("ASinv", "ASinvRdSinv", "ASinvCSinv"):
            res[k + "_" + tag] = e[k]
        res["scal_" + tag] = np.array([e["CSinv"], e["CSinvCSinv"], e["CSinvRdSinv"], e["TraceSinv"]])
    kkt.build_up(api.KKT_TYPE_INFEASIBLE)
    res["xstats"] = np.array(cone.exchange_stats())
    kkt.factorize()
    res["sol"] = kkt.solve(cone.traces())
    res["S"] = cone.dual_matrix()
    if rank == 0:
        np.savez(out, **res)
    kkt.destroy()
    cone.destroy()


def main():
    mode, n, m, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    (run_numpy if mode == "numpy" else run_gpu)(rank, world, n, m, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
