#!/usr/bin/env python3
"""oracle/row_subset_golden.py -- TEST INFRASTRUCTURE: host fp64 answers for BASELINE configs[4] (n = 2000, m = 8000).

SURVEY.md 8(d): the m = 8000 instance exceeds the reference's int32 CSC and no single host (or single GPU) forms its full
Schur matrix in reasonable time, so reference-style parity is checked on a ROW SUBSET.  Row i of M needs only A_i, S^-1
and every A_j:

    B_i = S^-1 A_i S^-1,   M_ij = <B_i, A_j>,   ASinv_j = <S^-1, A_j>,   ASinvRdSinv_j = Rd <S^-2, A_j>,   b_j = tr A_j

i.e. one pass over the m constraint matrices with |rows| + 2 inner products each (level-3 BLAS on the lower triangles,
fp64, numpy / scipy OpenBLAS), on the instance of the SURVEY 8(d) generator as restated in oracle/hdsdp_oracle.c
(orc_synth_matrix / orc_synth_objective).  The formulation shares nothing with the engine's (no Cholesky congruence, no
Gram product over the packed index, no sharding).  Two states as in oracle/full_size_golden.py: "bench" (y = 0, Rd = -10 n)
with rows [0, 64) -- eight rows per rank of the 8-way cyclic deal -- plus rows at the 128-tile edges of M, and "hard"
(y != 0, cond(S) ~ 1e3) with a handful of rows.  Both vectors are complete (all m entries).  The three Phase-A solutions need
the whole M and are not affordable here; the fixture instead lets a test check rows of the residual: (M d)_i = rhs_i on
the golden rows, with M's rows from this file and d from the device.

    python oracle/row_subset_golden.py [--n 2000 --m 8000] [--procs 8] [--out tests/golden/full8000_rows.npz]

About 15 minutes on 8 cores at the default size, 6 GB of memory.
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_py  # noqa: E402
from full_size_golden import hard_y  # noqa: E402

_G = {}   # inherited by the forked workers: n, the packed index, one (rows + 2, P) weight matrix per state


def bench_rows(m):
    base = list(range(min(64, m)))
    edges = [127, 128, m // 2 - 1, m // 2, 128 * ((m - 1) // 128) - 1, 128 * ((m - 1) // 128), m - 2, m - 1]
    return sorted(set(base + [r for r in edges if 0 <= r < m]))


def hard_rows(m):
    edges = [0, 1, 63, 127, 128, m // 2, 128 * ((m - 1) // 128), m - 1]
    return sorted(set(r for r in edges if 0 <= r < m))


def _weights(n, S, Rd, rows):
    """(len(rows) + 2, P) matrix whose product with the packed lower triangle of A_j gives M[rows, j], ASinv_j and
    ASinvRdSinv_j: off-diagonal entries carry the factor 2 of the symmetric inner product"""
    L = np.linalg.cholesky(S)
    Linv = sla.solve_triangular(L, np.eye(n), lower=True)
    Sinv = Linv.T @ Linv
    Sinv = 0.5 * (Sinv + Sinv.T)
    Sinv2 = Sinv @ Sinv
    il = _G["il"]
    w = np.where(il[0] == il[1], 1.0, 2.0)
    G = np.empty((len(rows) + 2, il[0].size))
    for k, r in enumerate(rows):
        A = oracle_py.synth_matrix(n, int(r))
        B = Sinv @ A @ Sinv
        G[k] = (0.5 * (B + B.T))[il] * w
    G[len(rows)] = Sinv[il] * w
    G[len(rows) + 1] = Rd * Sinv2[il] * w
    info = {"logdetS": 2.0 * float(np.sum(np.log(np.diag(L)))), "TraceSinv": float(np.trace(Sinv)),
            "condS": float(np.linalg.cond(S))}
    return G, info


def _one_blas_thread():
    """pool initializer: the workers run side by side, each with a single-threaded BLAS"""
    try:
        from threadpoolctl import threadpool_limits
        _G["_tp"] = threadpool_limits(1)
    except Exception:
        pass


def _partial_sum(args):
    """sum_c coef_c A_c over a range of constraints (pass 1, "hard" state's S)"""
    n, c0, c1, coef = args
    acc = np.zeros((n, n))
    for c in range(c0, c1):
        acc += coef[c - c0] * oracle_py.synth_matrix(n, c)
    return acc


def _dots(args):
    """pass 2: the columns j0..j1-1 of every state's weight matrix times the packed constraint matrices"""
    j0, j1 = args
    n, il = _G["n"], _G["il"]
    X = np.empty((il[0].size, j1 - j0))
    tr = np.empty(j1 - j0)
    for j in range(j0, j1):
        A = oracle_py.synth_matrix(n, j)
        X[:, j - j0] = A[il]
        tr[j - j0] = np.trace(A)
    return j0, j1, tr, [G @ X for G in _G["W"]]


def generate(n, m, procs, log, states=("bench", "hard"), max_rows=0):
    t0 = time.time()
    _G["n"], _G["il"] = n, np.tril_indices(n)
    y0, C = oracle_py.synth_objective(n, m)
    log(f"objective generated {time.time() - t0:.1f} s")
    res, W, rowsets = {"n": n, "m": m}, [], []
    for st in states:
        if st == "bench":
            Rd, y = -10.0 * n, np.zeros(m)
            S = C - Rd * np.eye(n)
            rows = bench_rows(m)
            if max_rows and len(rows) > max_rows:      # a smaller fixture: the first rows (one per rank and more) + the tile edges
                rows = sorted(set(rows[:max_rows - 8] + rows[-8:]))
        else:
            y = hard_y(m)
            chunks = [(n, c0, min(m, c0 + 50), y[c0:min(m, c0 + 50)]) for c0 in range(0, m, 50)]
            B = C.copy()
            with mp.get_context("fork").Pool(procs, initializer=_one_blas_thread) as pool:
                for part in pool.imap(_partial_sum, chunks):      # in constraint order: the sum does not depend on `procs`
                    B -= part
            ev = np.linalg.eigvalsh(B)
            Rd = float(np.float32(ev[0] - 1e-3 * (ev[-1] - ev[0])))
            S = B - Rd * np.eye(n)
            rows = hard_rows(m)
            log(f"state hard: spectrum of C - sum y_i A_i = [{ev[0]:.4f}, {ev[-1]:.4f}], Rd = {Rd!r}  {time.time() - t0:.0f} s")
        G, info = _weights(n, S, Rd, rows)
        W.append(G)
        rowsets.append(rows)
        res[st + "_Rd"], res[st + "_y"], res[st + "_rows"] = Rd, y, np.array(rows)
        for k, v in info.items():
            res[st + "_" + k] = v
        log(f"state {st}: {len(rows)} rows of S^-1 A S^-1 formed, cond(S) = {info['condS']:.3g}  {time.time() - t0:.0f} s")
    _G["W"] = W
    outs = [np.empty((G.shape[0], m)) for G in W]
    b = np.empty(m)
    step = 16
    with mp.get_context("fork").Pool(procs, initializer=_one_blas_thread) as pool:
        done = 0
        for j0, j1, tr, parts in pool.imap_unordered(_dots, [(j, min(m, j + step)) for j in range(0, m, step)]):
            b[j0:j1] = tr
            for o, p in zip(outs, parts):
                o[:, j0:j1] = p
            done += j1 - j0
            if done % 800 < step:
                log(f"  inner products {done}/{m}  {time.time() - t0:.0f} s")
    for st, rows, o in zip(states, rowsets, outs):
        k = len(rows)
        res[st + "_M_rows"] = o[:k].copy()
        res[st + "_ASinv"], res[st + "_ASinvRdSinv"] = o[k].copy(), o[k + 1].copy()
        res[st + "_b"] = b.copy()
        # size-independent digests of the same rows (bench.py prints the first one for m = 8000)
        res[st + "_sum_M_rows"] = float(np.sum(o[:k]))
        res[st + "_sum_ASinv"] = float(np.sum(o[k]))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--m", type=int, default=8000)
    ap.add_argument("--procs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--states", default="bench,hard")
    ap.add_argument("--max-rows", type=int, default=0, help="cap on the bench state's row count (0 = rows [0, 64) + edges)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden", "full8000_rows.npz"))
    a = ap.parse_args()

    def log(s):
        print(s, flush=True)

    t0 = time.time()
    res = generate(a.n, a.m, a.procs, log, tuple(a.states.split(",")), a.max_rows)
    np.savez_compressed(a.out, **res)
    log(f"wrote {a.out}  ({os.path.getsize(a.out) / 1024:.0f} KiB)  total {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
