#!/usr/bin/env python3
"""oracle/blas3_baseline.py -- TEST INFRASTRUCTURE (bench.py's second CPU leg, SURVEY.md 8(d) "for fairness").

The reference's hot path is level-1/2 BLAS on one core.  This script times the SAME math written the way a CPU would
want it -- level-3 BLAS on all the host cores the box gives us (numpy/scipy -> OpenBLAS) -- so that the GPU number can be
read against a strong CPU formulation too, not only against the reference's own loop nest:

    L = chol(S);   At_i = L^-1 A_i L^-T  (two dtrsm per constraint);   M = Ahat Ahat^T  (one dsyrk over the packed,
    sqrt(2)-weighted lower triangles);   chol(M);   three solves.

It is a bounded sample: `sample_m` constraints for the congruence and a `gram_rows`-row slice for the Gram product,
each extrapolated by its flop count to m rows.  Prints one JSON line.  Never imported by the product.
"""
import json
import os
import sys
import time

import numpy as np


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    sample_m = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    gram_rows = int(sys.argv[4]) if len(sys.argv) > 4 else 192
    threads = int(sys.argv[5]) if len(sys.argv) > 5 else min(16, os.cpu_count() or 1)
    from threadpoolctl import threadpool_limits
    from scipy.linalg import cholesky, solve_triangular, cho_factor, cho_solve
    rng = np.random.default_rng(1)
    with threadpool_limits(limits=threads):
        # same shapes and density as the SURVEY 8(d) family; the values do not matter for timing
        A = rng.uniform(-1, 1, (sample_m, n, n))
        A = 0.5 * (A + A.transpose(0, 2, 1))
        G = rng.uniform(-1, 1, (n, n))
        S = 10.0 * n * np.eye(n) + 0.5 * (G + G.T)
        il = np.tril_indices(n)
        w = np.where(il[0] == il[1], 1.0, np.sqrt(2.0))
        t0 = time.perf_counter()
        L = cholesky(S, lower=True)
        t_chol = time.perf_counter() - t0
        rows = np.empty((sample_m, il[0].size))
        t0 = time.perf_counter()
        for i in range(sample_m):
            T = solve_triangular(L, A[i], lower=True)
            At = solve_triangular(L, T.T, lower=True)
            rows[i] = At[il] * w
        t_cong = time.perf_counter() - t0
        del A
        # Gram rate on a gram_rows x P slice (dsyrk-shaped product)
        X = np.ascontiguousarray(np.resize(rows, (gram_rows, rows.shape[1])))
        t0 = time.perf_counter()
        Gm = X @ X.T
        t_gram = time.perf_counter() - t0
        gram_flops_sample = 2.0 * gram_rows * gram_rows * X.shape[1]
        gram_flops_full = 1.0 * m * (m + 1) * X.shape[1]          # symmetric half
        # Schur factor + three solves at full m (cheap)
        Mm = rng.uniform(-1, 1, (m, m))
        Mm = Mm @ Mm.T + m * np.eye(m)
        t0 = time.perf_counter()
        cf = cho_factor(Mm, lower=True)
        for _ in range(3):
            cho_solve(cf, np.ones(m))
        t_m = time.perf_counter() - t0
    t_full = t_chol + t_cong * (m / sample_m) + t_gram * (gram_flops_full / gram_flops_sample) + t_m
    print(json.dumps({
        "value": 1.0 / t_full, "unit": "it/s", "cores": threads, "kind": "port-blas3",
        "sample": ("same math as level-3 BLAS on %d threads (numpy/scipy OpenBLAS): chol(S) %.2f s; congruence of %d of %d "
                   "constraints %.2f s (2 dtrsm each, extrapolated x%.0f); Gram on a %d-row slice %.2f s = %.0f GFLOP/s "
                   "(extrapolated by flops); chol(M)+3 solves %.2f s" % (
                       threads, t_chol, sample_m, m, t_cong, m / sample_m, gram_rows, t_gram,
                       gram_flops_sample / t_gram / 1e9, t_m)),
        "sample_seconds": t_chol + t_cong + t_gram + t_m, "full_step_seconds_extrapolated": t_full}))


if __name__ == "__main__":
    main()
