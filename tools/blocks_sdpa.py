"""Write a small, strictly primal-dual feasible multi-block SDP in SDPA sparse format: three SDP blocks (21 / 34 / 9) on
which most constraints are zero -- sparse rows on the first, dense rows on the second, single-diagonal-entry (rank-one) rows
on the third -- and, optionally, an LP block of 5 variables.  Feasible by construction like the SURVEY.md 8(d) family:
b_i = <A_i, I> (X = I is primal feasible), C = I + sum_i y0_i A_i (S = I at y0).  Deterministic (numpy default_rng(7)).
usage: tools/blocks_sdpa.py out.dat-s [lp]"""
import sys

import numpy as np


def write_blocks_sdpa(out, with_lp=False):
    rng = np.random.default_rng(7)
    m, dims = 24, [21, 34, 9]
    A = [[np.zeros((n, n)) for n in dims] for _ in range(m)]     # A[c][blk], symmetric
    lp = np.zeros((m, 5))

    def put(c, b, i, j, v):
        A[c][b][i, j] = v
        A[c][b][j, i] = v
    touched = set()
    for c in (0, 5, 6, 14, 21):
        for _ in range(6):
            i, j = rng.integers(0, dims[0], 2)
            put(c, 0, i, j, 0.4 * rng.uniform(-1, 1))
        touched.add(c)
    for c in (1, 2, 8, 15, 23):
        n = dims[1]
        for j in range(n):
            for i in range(j + 1):
                if i == j or rng.uniform() < 0.45:
                    put(c, 1, i, j, 0.25 * rng.uniform(-1, 1))
        touched.add(c)
    for c in (5, 10):
        for _ in range(9):
            i, j = rng.integers(0, dims[1], 2)
            put(c, 1, i, j, 0.4 * rng.uniform(-1, 1))
        touched.add(c)
    for k, c in enumerate((3, 4, 7, 9, 11, 12, 13, 16)):
        put(c, 2, k, k, 0.5 + rng.uniform(0, 1))
        touched.add(c)
    for c in range(m):
        if c not in touched:
            for _ in range(6):
                i, j = rng.integers(0, dims[0], 2)
                put(c, 0, i, j, 0.4 * rng.uniform(-1, 1))
    if with_lp:
        for c in (0, 2, 4, 7, 12, 20, 23):
            lp[c, rng.integers(0, 5)] = 0.5 * rng.uniform(-1, 1)
    y0 = 0.3 * rng.uniform(-1, 1, m)
    C = [np.eye(n) + sum(y0[c] * A[c][b] for c in range(m)) for b, n in enumerate(dims)]
    clp = 1.0 + lp.T @ y0
    bvec = np.array([sum(np.trace(A[c][b]) for b in range(len(dims))) + lp[c].sum() for c in range(m)])
    with open(out, "w") as f:
        f.write("%d\n%d\n%s%s\n" % (m, len(dims) + (1 if with_lp else 0), " ".join(str(d) for d in dims), " -5" if with_lp else ""))
        f.write(" ".join("%.17g" % v for v in bvec) + "\n")
        for mat in range(m + 1):
            for b, n in enumerate(dims):
                M = -C[b] if mat == 0 else A[mat - 1][b]          # the reader turns F0 into C = -F0
                for i in range(n):
                    for j in range(i, n):
                        if M[i, j] != 0.0:
                            f.write("%d %d %d %d %.17g\n" % (mat, b + 1, i + 1, j + 1, M[i, j]))
            if with_lp:
                v = -clp if mat == 0 else lp[mat - 1]
                for i in range(5):
                    if v[i] != 0.0:
                        f.write("%d %d %d %d %.17g\n" % (mat, len(dims) + 1, i + 1, i + 1, v[i]))


if __name__ == "__main__":
    write_blocks_sdpa(sys.argv[1], len(sys.argv) > 2)
