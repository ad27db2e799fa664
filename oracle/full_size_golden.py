#!/usr/bin/env python3
"""oracle/full_size_golden.py -- TEST INFRASTRUCTURE: host-side fp64 answers at the BASELINE size n = m = 2000.

The compiled reference cannot produce a golden at this size (its int32 CSC cannot hold the instance and one Schur build
takes hours on a core), so the answers come from the same mathematics as level-3 BLAS on the host (numpy / scipy OpenBLAS,
fp64), on the instance of the SURVEY.md 8(d) generator as restated in oracle/hdsdp_oracle.c (orc_synth_matrix /
orc_synth_objective, bit-equal to orc_synth_csc at the sizes where both exist, tests/test_oracle.py):

    S = tau*C - sum y_i A_i - Rd*I = L L^T,   At_i = L^-1 A_i L^-T,   M_ij = <At_i, At_j> = tr(A_i S^-1 A_j S^-1)
    ASinv_i = tr(At_i),   ASinvRdSinv_i = Rd * <A_i, S^-2>,   d1 = M^-1 b,  d2 = M^-1 ASinv,  d3 = M^-1 ASinvRdSinv

for two states: "bench" (y = 0, Rd = -10 n: the state bench.py times) and "hard" (small non-zero y, Rd just below the smallest eigenvalue of C - sum y_i A_i: cond(S) ~ 1e3).
The full M (32 MB) is not committed; the fixture keeps 16 full rows of it spread over all 128-row tiles, the vectors, the
solutions, checksums over the whole lower triangle and four bilinear probes u^T M v with seeded dense u, v -- every entry
of M enters the probes and the checksums.  Runs about 15 minutes on 8 cores and needs 32 GB of scratch under --scratch.

    python oracle/full_size_golden.py [--n 2000 --m 2000] [--scratch /tmp] [--out tests/golden/full2000.npz]
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_py  # noqa: E402

ROWS = [0, 1, 127, 128, 129, 255, 256, 640, 777, 1023, 1024, 1500, 1919, 1920, 1998, 1999]


def hard_y(m):
    """closed-form multipliers of the "hard" state (the fixture also stores them)"""
    c = np.arange(m, dtype=np.float64)
    return 0.05 * np.cos(0.9 * c + 0.3)


def probe_vectors(m, k):
    rng = np.random.RandomState(20260 + k)
    return rng.uniform(-1.0, 1.0, m), rng.uniform(-1.0, 1.0, m)


def solve_state(n, m, S, Rd, W, log):
    """everything the Schur path produces at the dual matrix S (full symmetric); W is the (m, P) scratch array"""
    P = n * (n + 1) // 2
    L = np.linalg.cholesky(S)
    logdet = 2.0 * float(np.sum(np.log(np.diag(L))))
    Linv = sla.solve_triangular(L, np.eye(n), lower=True)
    Sinv = Linv.T @ Linv
    Sinv2 = Sinv @ Sinv
    il = np.tril_indices(n)                       # row-major lower triangle; any fixed order of the packed index will do
    scale = np.where(il[0] == il[1], 1.0, np.sqrt(2.0))
    asinv, asinvrd, b = np.zeros(m), np.zeros(m), np.zeros(m)
    t0 = time.time()
    for i in range(m):
        A = oracle_py.synth_matrix(n, i)
        T = sla.blas.dtrmm(1.0, Linv, A, side=0, lower=1)              # Linv * A
        At = sla.blas.dtrmm(1.0, Linv, T, side=1, lower=1, trans_a=1)  # (Linv A) Linv^T
        At = 0.5 * (At + At.T)
        W[i, :] = At[il] * scale
        b[i] = np.trace(A)
        asinv[i] = np.trace(At)
        asinvrd[i] = Rd * float(np.sum(A * Sinv2))
        if i % 100 == 99:
            log(f"  congruence {i + 1}/{m}  {time.time() - t0:.0f} s")
    M = np.zeros((m, m))
    step = 65536
    for c0 in range(0, P, step):
        Wc = np.ascontiguousarray(W[:, c0:c0 + step])
        M += Wc @ Wc.T
    M = 0.5 * (M + M.T)
    log(f"  gram done {time.time() - t0:.0f} s")
    cf = sla.cho_factor(M, lower=True)
    d1, d2, d3 = (sla.cho_solve(cf, v) for v in (b, asinv, asinvrd))
    out = {"logdetS": logdet, "TraceSinv": float(np.trace(Sinv)), "b": b, "ASinv": asinv, "ASinvRdSinv": asinvrd,
           "rows": np.array([r for r in ROWS if r < m]), "d1": d1, "d2": d2, "d3": d3,
           "sum_d2": float(np.sum(d2)), "sum_d1w": float(np.dot(np.arange(1, m + 1), d1)),
           "sum_M_lower": float(np.sum(np.tril(M))), "sumsq_M_lower": float(np.sum(np.tril(M) ** 2)),
           "diag_M": np.diag(M).copy(), "condS": float(np.linalg.cond(S))}
    out["M_rows"] = M[out["rows"], :].copy()
    pr = []
    for k in range(4):
        u, v = probe_vectors(m, k)
        pr.append(float(u @ M @ v))
    out["probes"] = np.array(pr)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--m", type=int, default=2000)
    ap.add_argument("--scratch", default="/tmp")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden", "full2000.npz"))
    a = ap.parse_args()
    n, m = a.n, a.m
    P = n * (n + 1) // 2

    def log(s):
        print(s, flush=True)

    t0 = time.time()
    y0, C = oracle_py.synth_objective(n, m)
    log(f"objective generated {time.time() - t0:.1f} s")
    scratch = os.path.join(a.scratch, f"full_size_golden_{os.getpid()}.f64")
    W = np.lib.format.open_memmap(scratch, mode="w+", dtype=np.float64, shape=(m, P))
    res = {}
    try:
        # state "bench": y = 0, tau = 1, Rd = -10 n  (bench.py)
        Rd = -10.0 * n
        S = C - Rd * np.eye(n)
        log("state bench")
        for k, v in solve_state(n, m, S, Rd, W, log).items():
            res["bench_" + k] = v
        res["bench_Rd"] = Rd
        res["bench_y"] = np.zeros(m)
        # state "hard": small non-zero multipliers (the S assembly runs over all constraint matrices) and a residual
        # term that puts the smallest eigenvalue of S at 1e-3 of the spectrum's width: cond(S) ~ 1e3, nothing like a
        # multiple of the identity.  No cancellation in forming S (y0 itself would cancel C's O(10) entries down to
        # O(1e-2) and make the answer depend on the summation order at the 1e-9 level).
        yh = hard_y(m)
        B = C.copy()
        for i in range(m):
            B -= yh[i] * oracle_py.synth_matrix(n, i)
        ev = np.linalg.eigvalsh(B)
        Rd = float(np.float32(ev[0] - 1e-3 * (ev[-1] - ev[0])))   # a short mantissa: the test passes exactly this number
        S = B - Rd * np.eye(n)
        log(f"state hard: spectrum of C - sum y_i A_i = [{ev[0]:.4f}, {ev[-1]:.4f}], Rd = {Rd!r}")
        for k, v in solve_state(n, m, S, Rd, W, log).items():
            res["hard_" + k] = v
        res["hard_Rd"] = Rd
        res["hard_y"] = yh
        res["n"] = n
        res["m"] = m
        np.savez_compressed(a.out, **res)
        log(f"wrote {a.out}  ({os.path.getsize(a.out) / 1024:.0f} KiB)  total {time.time() - t0:.0f} s  "
            f"condS bench {res['bench_condS']:.3g} hard {res['hard_condS']:.3g}")
    finally:
        del W
        os.unlink(scratch)


if __name__ == "__main__":
    main()
