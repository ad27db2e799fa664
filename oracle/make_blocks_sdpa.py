#!/usr/bin/env python3
"""oracle/make_blocks_sdpa.py -- TEST INFRASTRUCTURE: writes tests/golden/blocks3.dat-s, a small many-block SDP in SDPA
sparse format whose blocks look like the blocks of a real multi-block problem: most constraints are ZERO on a given block.

  block 1 (n = 21): sparse constraints                            -> the engine's sparse-gather path, 14 of 24 rows zero
  block 2 (n = 34): five dense constraints + two sparse ones      -> the congruence + Gram path,       17 of 24 rows zero
  block 3 (n =  9): single-diagonal-entry (rank-one) constraints  -> the rank-one path,                16 of 24 rows zero

Every constraint touches at least one block; the objective has entries in every block.  The file is data for both sides:
the compiled reference reads it through its own SDPA reader (oracle/gen_golden.py, case blocks3_A), the engine through
hdsdp_amd/csrc/sdpa.cpp.  Deterministic (numpy default_rng(7)); values are written with 17 significant digits.
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "blocks3.dat-s")


def main():
    rng = np.random.default_rng(7)
    m, dims = 24, (21, 34, 9)
    ent = []   # (matno, blk, i, j, v), 1-based, i <= j (upper triangle as the format wants)

    def add(mat, blk, i, j, v):
        if i > j:
            i, j = j, i
        ent.append((mat, blk, i + 1, j + 1, float(v)))

    # objective F0 (the reader turns it into C = -F0): diagonal-dominant entries in every block
    for b, n in enumerate(dims, start=1):
        for i in range(n):
            add(0, b, i, i, -(2.0 + rng.uniform(0, 1)))
            if i + 1 < n and rng.uniform() < 0.5:
                add(0, b, i, i + 1, 0.3 * rng.uniform(-1, 1))
    touched = set()
    # block 1: sparse rows on constraints 1, 6, 7, 15, 22
    for c in (1, 6, 7, 15, 22):
        n = dims[0]
        for _ in range(6):
            i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
            add(c, 1, i, j, 0.4 * rng.uniform(-1, 1))
        touched.add(c)
    # block 2: dense rows on constraints 2, 3, 9, 16, 24 (> 0.3 of the packed entries), sparse on 6 and 11
    n = dims[1]
    for c in (2, 3, 9, 16, 24):
        for j in range(n):
            for i in range(j + 1):
                if i == j or rng.uniform() < 0.45:
                    add(c, 2, i, j, 0.25 * rng.uniform(-1, 1))
        touched.add(c)
    for c in (6, 11):
        for _ in range(9):
            i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
            add(c, 2, i, j, 0.4 * rng.uniform(-1, 1))
        touched.add(c)
    # block 3: one diagonal entry each (rank one) on constraints 4, 5, 8, 10, 12, 13, 14, 17
    for k, c in enumerate((4, 5, 8, 10, 12, 13, 14, 17)):
        add(c, 3, k % dims[2], k % dims[2], 0.5 + rng.uniform(0, 1))
        touched.add(c)
    # whatever is left gets a sparse row on block 1 (random positions: linearly independent of the others)
    for c in range(1, m + 1):
        if c not in touched:
            for _ in range(6):
                i, j = int(rng.integers(0, dims[0])), int(rng.integers(0, dims[0]))
                add(c, 1, i, j, 0.4 * rng.uniform(-1, 1))
    # duplicates (same matrix, block, position) would be summed by one reader and overwritten by another: keep the first
    seen, uniq = set(), []
    for e in ent:
        if e[:4] not in seen:
            seen.add(e[:4])
            uniq.append(e)
    uniq.sort(key=lambda e: (e[0], e[1], e[2], e[3]))
    bvec = rng.uniform(-1, 1, m)
    with open(OUT, "w") as f:
        f.write("%d\n%d\n%s\n" % (m, len(dims), " ".join(str(d) for d in dims)))
        f.write(" ".join("%.17g" % v for v in bvec) + "\n")
        for mat, blk, i, j, v in uniq:
            f.write("%d %d %d %d %.17g\n" % (mat, blk, i, j, v))
    print("wrote", OUT, len(uniq), "entries")


if __name__ == "__main__":
    main()
