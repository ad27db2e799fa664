#!/usr/bin/env python3
"""oracle/make_chain_sdpa.py -- TEST INFRASTRUCTURE: writes tests/golden/chain16.dat-s, a many-small-block SDP whose
Schur matrix is SPARSE (the case the reference's aggregated-pattern CSC operator, interface/hdsdp_schur.c:46-139, and its
sparse SDP cone, hdsdp_conic_sdp.c:1814-1886, exist for): m = 48 constraints, 16 blocks of dimension 6..13, block b
touching the five constraints 3b+1 .. 3b+5 (wrapping round), so neighbouring blocks overlap in two constraints and the
aggregated pattern is a cyclic band of about 8 entries per column -- 0.15 m^2, under the reference's 0.3 m^2 switch.
Coefficient kinds rotate over the blocks: sparse triplets, dense (> 0.3 of the packed entries), single diagonal entries
(rank one).  Deterministic (numpy default_rng(11)); values are written with 17 significant digits.  Read by the compiled
reference (oracle/gen_golden.py, case chain16_A) and by the engine's own reader (hdsdp_amd/csrc/sdpa.cpp)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "chain16.dat-s")


def main():
    rng = np.random.default_rng(11)
    m, nb = 48, 16
    dims = [6 + (b * 5) % 8 for b in range(nb)]
    ent = {}

    def add(mat, blk, i, j, v):
        if i > j:
            i, j = j, i
        ent.setdefault((mat, blk, i + 1, j + 1), float(v))     # a repeated position keeps its first value

    for b, n in enumerate(dims, start=1):
        for i in range(n):                                        # objective F0 = -C: C diagonally dominant
            add(0, b, i, i, -(2.0 + rng.uniform(0, 1)))
            if i + 1 < n and rng.uniform() < 0.5:
                add(0, b, i, i + 1, 0.3 * rng.uniform(-1, 1))
        rows = [(3 * (b - 1) + q) % m + 1 for q in range(5)]
        kind = (b - 1) % 3
        for q, c in enumerate(rows):
            if kind == 0:                                         # sparse triplets
                for _ in range(4):
                    add(c, b, int(rng.integers(0, n)), int(rng.integers(0, n)), 0.4 * rng.uniform(-1, 1))
                add(c, b, q % n, q % n, 0.3 + 0.2 * rng.uniform())
            elif kind == 1:                                       # dense
                for j in range(n):
                    for i in range(j + 1):
                        if i == j or rng.uniform() < 0.6:
                            add(c, b, i, j, 0.25 * rng.uniform(-1, 1))
            else:                                                 # one diagonal entry: rank one
                add(c, b, q % n, q % n, 0.5 + rng.uniform(0, 1))
    keys = sorted(ent)
    bvec = rng.uniform(-1, 1, m)
    with open(OUT, "w") as f:
        f.write("%d\n%d\n%s\n" % (m, nb, " ".join(str(d) for d in dims)))
        f.write(" ".join("%.17g" % v for v in bvec) + "\n")
        for k in keys:
            f.write("%d %d %d %d %.17g\n" % (k + (ent[k],)))
    print("wrote", OUT, len(keys), "entries; block dimensions", dims)


if __name__ == "__main__":
    main()
