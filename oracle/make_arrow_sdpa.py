#!/usr/bin/env python3
"""oracle/make_arrow_sdpa.py -- TEST INFRASTRUCTURE: writes tests/golden/arrow128.dat-s, a many-block SDP with LINKING
constraints: 128 small blocks (dimension 8..11), block b carrying its own eight constraints 8b+1 .. 8b+8 and ALL of the
last 32 ("linking") constraints, m = 1056.  The aggregated pattern of the Schur matrix is an arrow -- 128 diagonal 8 x 8
blocks, 32 dense last rows, 0.034 m^2 entries: far under the reference's 0.3 m^2 switch, so the reference runs its SPARSE
Schur operator (interface/hdsdp_schur.c:46-139) with sparse SDP cones (each block has data on 40 <= 0.3 m constraints).
Every bandwidth-reducing order of an arrow has a full envelope (round 2's device path paid the dense m^3 / 3 for it); with
the linking rows ordered last its Cholesky factor has no fill at all, which is what the tile form (hdsdp_amd/csrc/bsparse.h)
exploits.  Deterministic (numpy default_rng(23)); 17 significant digits.  Read by the compiled reference
(oracle/gen_golden.py, case arrow128_A) and by the engine's own reader."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "arrow128.dat-s")


def main():
    rng = np.random.default_rng(23)
    nb, loc, link = 128, 8, 32
    m = nb * loc + link
    dims = [8 + (b * 3) % 4 for b in range(nb)]
    ent = {}

    def add(mat, blk, i, j, v):
        if i > j:
            i, j = j, i
        ent.setdefault((mat, blk, i + 1, j + 1), float(v))

    for b, n in enumerate(dims, start=1):
        for i in range(n):                                        # F0 = -C, C diagonally dominant
            add(0, b, i, i, -(3.0 + rng.uniform(0, 1)))
            if i + 1 < n and rng.uniform() < 0.5:
                add(0, b, i, i + 1, 0.3 * rng.uniform(-1, 1))
        for q in range(loc):                                      # the block's own constraints: a diagonal entry and one more
            c = (b - 1) * loc + q + 1
            add(c, b, q % n, q % n, 0.5 + 0.5 * rng.uniform())                 # (n >= 8: eight different diagonal positions)
            add(c, b, q % n, (q + 1 + int(rng.integers(0, n - 1))) % n, 0.4 * rng.uniform(-1, 1))
        for q in range(link):                                     # linking constraints: every block has data on each of them
            c = nb * loc + q + 1
            add(c, b, (q + b) % n, (q + b) % n, 0.05 + 0.1 * rng.uniform())
            if q % 4 == 0:
                add(c, b, int(rng.integers(0, n)), int(rng.integers(0, n)), 0.05 * rng.uniform(-1, 1))
    keys = sorted(ent)
    # b_c = sum over the blocks of tr(A_c): X = I is primal feasible, y = 0 (S = C > 0) dual feasible -- a solvable problem for the
    # reference's driver (tests/test_gpu_reference_driver.py)
    bvec = np.zeros(m)
    for (mat, blk, i, j), v in ent.items():
        if mat > 0 and i == j:
            bvec[mat - 1] += v
    with open(OUT, "w") as f:
        f.write("%d\n%d\n%s\n" % (m, nb, " ".join(str(d) for d in dims)))
        f.write(" ".join("%.17g" % v for v in bvec) + "\n")
        for k in keys:
            f.write("%d %d %d %d %.17g\n" % (k + (ent[k],)))
    print("wrote", OUT, len(keys), "entries; m =", m)


if __name__ == "__main__":
    main()
