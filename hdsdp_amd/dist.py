"""Multi-GPU choreography of the Schur build: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm) over xGMI.  New design -- the reference has no distributed code.

Sharding (see DESIGN.md "Multi-GPU"):
  1. constraint rows are dealt cyclically: row i lives on rank i % world (balanced by construction);
     every rank congruence-transforms only its own rows (the 4/3 m n^3 part of the work);
  2. ONE all-to-all transposes the transformed data from "by constraint" to "by packed-index range"
     (each rank sends (world-1)/world of its m/world rows: 1/world of what an all-gather would move);
  3. every rank forms the Gram partial sum over its index range for ALL row pairs (the m^2 n^2 / 2 part,
     again 1/world each -- a distributed split-K);
  4. ONE all-reduce(sum) of the (m+3)^2 augmented Gram matrix assembles M, ASinv, ASinvRdSinv,
     ASinvCSinv and the scalars on every rank; Cholesky + solves are replicated (m^3/3 is noise).

`ShardPlan` is the pure-Python statement of the layout the engine uses (hdsdp_amd/csrc/engine.hip
cone_alloc_common); the CPU gloo tests drive it with numpy standing in for the kernels.
"""
import ctypes as C
import os

import numpy as np


def _roundup(x, q):
    return (x + q - 1) // q * q


class ShardPlan:
    """Index bookkeeping of the sharded Schur build for one SDP block."""

    def __init__(self, n, m, world):
        self.n, self.m, self.world = n, m, world
        self.n16 = _roundup(n, 16)
        self.nblk = self.n16 // 16
        self.npb = self.nblk * (self.nblk + 1) // 2 * 16       # 16-wide p-blocks of the blocked layout
        self.npb_loc = (self.npb + world - 1) // world          # p-blocks per rank (Gram K range)
        maxloc = (m + world - 1) // world
        self.Lr = _roundup(maxloc + 3, 8) if world == 1 else _roundup(maxloc + 3, 128)
        self.R = world * self.Lr                                # rows of the segment-ordered Gram matrix
        self.pI = maxloc                                        # augmented rows sit in segment 0
        self.chunk = self.npb_loc * self.Lr * 16                # doubles per all-to-all chunk

    def hbm_bytes(self, rank=0, tcap_gib=32, slab_cap_gib=40):
        """device memory one rank of the sharded GEMM path allocates, mirroring engine.hip (cone_alloc_common,
        cone_alloc_gemm_work, make_synth_cone / upload_dense_rows) and chol.hip: constraint data, congruence
        intermediates, both exchange buffers, Gram slabs, and the small replicated matrices.  Returns a dict of parts
        plus "total"."""
        n16, nn = self.n16, self.n16 * self.n16 * 8
        npad = _roundup(self.n, 128)
        mloc = len(self.owned(rank))
        # constraint matrices: A_L form in skyline storage (csrc/hdm_common.h): 128-column panels from the diagonal down
        t = (n16 + 127) // 128 - 1
        sky = 128 * (t * n16 - 64 * t * (t - 1)) + (n16 - 128 * t) ** 2
        parts = {"A (A_L form, skyline)": mloc * sky * 8}
        bc = max(1, min(int(tcap_gib * (1 << 30) / nn), 1024))
        launches = -(-max(1, mloc) // bc)
        bc = -(-max(1, mloc) // launches)
        parts["congruence intermediates T"] = bc * nn
        ahat = self.world * self.npb_loc * self.Lr * 16 * 8
        parts["exchange buffers (send + recv)"] = ahat * (1 if self.world == 1 else 2)
        kblocks = self.npb_loc
        RT = -(-self.R // 128)
        tiles = RT * (RT + 1) // 2
        slab_cap = max(1, int((4 << 30) / (8.0 * self.R * self.R)))
        kcap = max(1, kblocks // 64)
        ns, best = 1, -1.0
        for cand in range(1, 65):
            if cand > slab_cap or cand > kcap:
                break
            if cand > 8 and cand % 8:
                continue
            rounds = tiles * cand / 512.0
            eff = rounds / np.ceil(rounds)
            if rounds < 2.0:
                eff *= 0.5 + 0.25 * rounds
            if cand < 8 and kcap >= 8 and slab_cap >= 8:
                eff *= 0.5
            if eff > best + 1e-9:
                best, ns = eff, cand
        big_cap = int((slab_cap_gib << 30) / (8.0 * self.R * self.R))
        byk = kblocks // 96
        splits = ns
        if byk >= 128 and self.world == 1:
            # one device, long packed index (round 5): K splits sized so that one split's operand panel fills the 256 MiB
            # memory-side cache, run in groups over at most 8 GiB of slabs (engine_cone.h: cone_alloc_gemm_work)
            kst = min(int((1 << 28) / (128.0 * self.R)), int(np.sqrt(tiles * kblocks / 128.0)))
            kst = max(96, min(kst, 2048))
            splits = max(8, -(-kblocks // kst))
            ns = max(ns, min(splits, max(8, int(max(8 << 30, parts["congruence intermediates T"]) / (8.0 * self.R * self.R)))))
            splits = max(splits, ns)
        elif byk >= 128:
            # sharded: as many splits as before (a multiple of 8: the exchange pieces are groups of them), at most 8 GiB of slabs
            kst = min(int((1 << 28) / (128.0 * self.R)), int(np.sqrt(tiles * kblocks / 128.0)))
            kst = max(96, min(kst, 2048))
            big = min(1024, (-(-kblocks // kst) + 7) & ~7)
            if big > ns:
                splits = big
                ns = max(ns, min(big, max(8, int((8 << 30) / (8.0 * self.R * self.R)))))
        slab = ns * self.R * self.R * 8
        if self.world == 1:
            # one GPU: T is dead when the Gram product starts, the two share one buffer of the larger size
            tb = parts.pop("congruence intermediates T")
            parts["congruence intermediates T / Gram slabs (%d slabs for %d splits), one shared buffer" % (ns, splits)] = max(tb, slab)
        else:
            parts["Gram slabs (%d slabs for %d splits)" % (ns, splits)] = slab
        parts["Gram matrix"] = self.R * self.R * 8
        parts["S, checker, C, dS + Cholesky / inverse of S"] = 4 * nn + 4 * npad * npad * 8
        mpad = _roundup(self.m, 128)
        parts["Schur matrix M + its factor (replicated)"] = 3 * mpad * mpad * 8
        parts["total"] = sum(parts.values())
        # not in the total: the zero-suppressed copy of A the S / dS sweeps read (csrc/schur.h, made at cone creation and given
        # back if the work buffers need the room; data dependent -- 8 bytes per non-zero + 192 bytes per 1024 positions; 40 % fill assumed here)
        parts["optional: zero-suppressed sweep copy of A at 40 % fill"] = int(mloc * sky * (0.4 * 8 + 192.0 / 1024))
        return parts

    def owned(self, rank):
        return list(range(rank, self.m, self.world))

    def rows_seg(self):
        """segment-ordered Gram row -> global constraint (-1 pad, -2/-3/-4 = I/S/C rows)"""
        rs = -np.ones(self.R, dtype=np.int64)
        for g in range(self.world):
            own = self.owned(g)
            rs[g * self.Lr:g * self.Lr + len(own)] = own
            if g == 0:
                rs[len(own):len(own) + 3] = (-2, -3, -4)
        return rs

    def blocked_index(self):
        """(p-block, slot) position of every lower-triangular 16x16 sub-block entry and its weight:
        returns arrays (r, c, pb, q, w) with value(pb*16+q) = w * At[r, c]"""
        nb = self.nblk
        out = []
        for bj in range(nb):
            for bi in range(bj, nb):
                sub = bj * nb - bj * (bj - 1) // 2 + (bi - bj)
                w = 1.0 if bi == bj else np.sqrt(2.0)
                for cl in range(16):
                    for rl in range(16):
                        out.append((bi * 16 + rl, bj * 16 + cl, sub * 16 + cl, rl, w))
        a = np.array(out)
        return a[:, 0].astype(int), a[:, 1].astype(int), a[:, 2].astype(int), a[:, 3].astype(int), a[:, 4]

    def to_blocked(self, At):
        """n x n symmetric matrix -> blocked vector of length world*npb_loc*16 (zero padded)"""
        r, c, pb, q, w = self.blocked_index()
        P = np.zeros((At.shape[0] if At.ndim == 3 else 1, self.world * self.npb_loc, 16))
        Ap = np.zeros((self.n16, self.n16))
        Ap[:self.n, :self.n] = At
        P[0, pb, q] = w * Ap[r, c]
        return P[0]


class _DevPtr:
    """raw device pointer -> torch tensor via the CUDA array interface (no copy)"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2}


def device_view(ptr, count):
    import torch
    return torch.as_tensor(_DevPtr(ptr, count), device="cuda")


class Exchange:
    """Registers the two collectives the engine needs on a sharded cone.

    backend "nccl": collectives run directly on the device buffers (RCCL over xGMI).
    backend "gloo": device buffers are staged through host memory (used to rehearse world > 1 with every
    rank on one GPU, and on CPU-only boxes for the layout tests)."""

    def __init__(self, cone, group=None, pieces=8):
        import torch
        import torch.distributed as dist
        from . import api
        self.dist, self.torch, self.group = dist, torch, group
        self.backend = dist.get_backend(group)
        self.world = dist.get_world_size(group)
        lib = api.load_library()
        self.plan = ShardPlan(cone.n, cone.m, self.world)
        count = self.plan.chunk * self.world
        # torch owns the exchange buffers so RCCL sees registered allocations; the engine's Gram kernel reads them
        # with unmasked tile loads, hence the slack behind the payload (hdm_common.h: HDM_OPERAND_PAD_DOUBLES)
        pad = 8192
        self._send_full = torch.zeros(count + pad, dtype=torch.float64, device="cuda")
        self._recv_full = torch.zeros(count + pad, dtype=torch.float64, device="cuda") if self.world > 1 else self._send_full
        self.send, self.recv = self._send_full[:count], self._recv_full[:count]
        torch.cuda.synchronize()
        rc = lib.HMiConeSetExchangeBuffers(cone._h, self.send.data_ptr(), self.recv.data_ptr())
        if rc != 0:
            raise api.HDSDPError("HMiConeSetExchangeBuffers failed")
        self._a2a = C.CFUNCTYPE(C.c_int, C.c_void_p)(self._alltoall)
        self._ar = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)(self._allreduce)
        lib.HMiConeSetExchange(cone._h, C.cast(self._a2a, C.c_void_p), C.cast(self._ar, C.c_void_p), None)
        # piecewise exchange: the engine starts piece k as soon as the congruence has finished that piece's index range
        # (the later ranges are still being computed), then waits for piece k and launches the Gram splits of its
        # range while pieces k+1.. are still on the links (RCCL runs them on its own stream)
        self._a2a_start = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int)(self._alltoall_piece)
        self._a2a_wait = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)(self._alltoall_wait)
        self._works = {}
        self._comm = torch.cuda.Stream() if self.backend != "gloo" else None
        lib.HMiConeSetExchangePieces(cone._h, C.cast(self._a2a_start, C.c_void_p), C.cast(self._a2a_wait, C.c_void_p),
                                     pieces if self.world > 1 else 1)
        self.bytes_a2a = 0
        self.bytes_ar = 0

    def _alltoall(self, ctx):
        try:
            if self.backend == "gloo":
                s = self.send.cpu()
                r = self.torch.empty_like(s)
                self.dist.all_to_all_single(r, s, group=self.group)
                self.recv.copy_(r)
            else:
                self.dist.all_to_all_single(self.recv, self.send, group=self.group)
            self.torch.cuda.synchronize()
            self.bytes_a2a += self.send.numel() * 8
            return 0
        except Exception as e:  # a Python exception must not unwind through the C frame
            print("[hdsdp_amd.dist] all_to_all failed:", e, flush=True)
            return 1

    def _alltoall_piece(self, ctx, off, cnt, piece):
        try:
            chunk = self.plan.chunk
            ins = [self.send[d * chunk + off:d * chunk + off + cnt] for d in range(self.world)]
            outs = [self.recv[s * chunk + off:s * chunk + off + cnt] for s in range(self.world)]
            if self.backend == "gloo":
                si = self.torch.cat([t.cpu() for t in ins])
                ro = self.torch.empty_like(si)
                self.dist.all_to_all_single(ro, si, group=self.group)
                for s_, t in enumerate(outs):
                    t.copy_(ro[s_ * cnt:(s_ + 1) * cnt])
                self.torch.cuda.synchronize()
                self._works[piece] = None
            else:
                with self.torch.cuda.stream(self._comm):
                    self._works[piece] = self.dist.all_to_all(outs, ins, group=self.group, async_op=True)
            self.bytes_a2a += int(cnt) * 8 * self.world
            return 0
        except Exception as e:
            print("[hdsdp_amd.dist] all_to_all piece failed:", e, flush=True)
            return 1

    def _alltoall_wait(self, ctx, piece):
        try:
            w = self._works.pop(piece, None)
            if w is not None:
                with self.torch.cuda.stream(self._comm):
                    w.wait()
                self._comm.synchronize()
            return 0
        except Exception as e:
            print("[hdsdp_amd.dist] all_to_all wait failed:", e, flush=True)
            return 1

    def _allreduce(self, ctx, buf, count):
        try:
            t = device_view(buf, count)
            if self.backend == "gloo":
                h = t.cpu()
                self.dist.all_reduce(h, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, group=self.group)
            self.torch.cuda.synchronize()
            self.bytes_ar += int(count) * 8
            return 0
        except Exception as e:
            print("[hdsdp_amd.dist] all_reduce failed:", e, flush=True)
            return 1


def init_process_group_from_env(backend=None):
    """one process per GPU, launched by torch.distributed.run; returns (rank, world, local_rank)"""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl"
        if torch.cuda.is_available():
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
