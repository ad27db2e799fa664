"""Write the SURVEY.md 8(d) synthetic dense SDP (n, m) as an SDPA sparse file (F0 = -C, F_i = A_i, c = b), so that a
driver reading it through its SDPA reader (C = -F0, b = c) works on exactly the generator's (C, A_i, b).
usage: tools/synth_sdpa.py n m out.dat-s"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_synth_sdpa(n, m, out):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    beg, idx, val, b = oracle_py.synth_csc(n, m)
    # packed lower column-major index -> (row, col)
    col_start = np.array([(2 * n - j - 1) * j // 2 + j for j in range(n)])   # packed index of (j, j)
    with open(out, "w") as f:
        f.write("%d\n1\n%d\n" % (m, n))
        f.write(" ".join("%.17g" % v for v in b) + "\n")
        for c in range(m + 1):
            lo, hi = int(beg[c]), int(beg[c + 1])
            pk = np.asarray(idx[lo:hi], dtype=np.int64)
            j = np.searchsorted(col_start, pk, side="right") - 1
            i = j + (pk - col_start[j])
            v = np.asarray(val[lo:hi]) * (-1.0 if c == 0 else 1.0)
            lines = ["%d 1 %d %d %.17g\n" % (c, jj + 1, ii + 1, vv) for ii, jj, vv in zip(i, j, v)]
            f.write("".join(lines))


if __name__ == "__main__":
    write_synth_sdpa(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
