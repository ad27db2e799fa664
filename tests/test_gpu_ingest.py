"""64-bit ingest (include/hdsdp_mi355x.h: HMiConeCreateSDP64, HMiConeBuilder*; SURVEY finding 6, row f3): the reference's user
data is one CSC with `int` column pointers (interface/def_hdsdp_user_data.h:22-32) and cannot hold a block with more than
2^31 - 1 entries -- a fully dense n = m = 2000 instance has 4.0e9.  The two 64-bit entries must give exactly the device data
of the 32-bit one where all three apply, carry the headline family column by column, and take a block no int32 CSC can hold."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from util import load_golden, lower_mask, y_of

pytestmark = pytest.mark.gpu

CSC_GOLDENS = ["theta1_A", "mcp100_A", "gpp100_A", "mix40_A", "mix40_B", "syn64", "syn100", "syn200", "syn96x40_B", "syn2000x32"]


def _state(api, cone, m, g):
    """dual matrix, log det, Schur matrix and vectors at the golden's state, as numpy copies"""
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    cone.set_start(Rd)
    assert cone.check_is_interior(tau, y)
    kkt = api.KKT(m, [cone])
    kkt.build_up(api.KKT_TYPE_INFEASIBLE)
    ex = kkt.export()
    out = {"S": cone.dual_matrix().copy(), "logdet": cone.log_barrier(tau), "M": kkt.M.copy(), "ASinv": ex["ASinv"].copy(),
           "ASinvRdSinv": ex["ASinvRdSinv"].copy(), "traces": cone.traces().copy(), "presolve": cone.presolve()}
    kkt.destroy()
    return out


@pytest.mark.parametrize("name", CSC_GOLDENS)
def test_the_three_ways_in_give_the_same_bits(name):
    """HMiConeCreateSDP (int32 CSC), HMiConeCreateSDP64 (int64 column pointers) and the column-by-column builder (columns in a
    shuffled order, entries of every column shuffled too, zero columns not given at all) on every CSC golden: dual matrix,
    log det S, Schur matrix, both vectors, traces and the presolve arrays bit for bit"""
    from hdsdp_amd import api
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    if "csc_beg" in g:
        beg, idx, val = np.asarray(g["csc_beg"]), np.asarray(g["csc_idx"]), np.asarray(g["csc_val"])
    else:                                    # the synthetic goldens carry no CSC: the oracle's restatement of the generator makes it
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import oracle_py
        beg, idx, val, _ = oracle_py.synth_csc(n, m)
    rng = np.random.default_rng(7)

    def columns():
        for c in rng.permutation(m + 1):
            lo, hi = int(beg[c]), int(beg[c + 1])
            if hi == lo:
                continue
            p = rng.permutation(hi - lo)
            yield int(c), idx[lo:hi][p], val[lo:hi][p]

    got = []
    for make in (lambda: api.SDPCone.from_csc(n, m, beg, idx, val), lambda: api.SDPCone.from_csc64(n, m, beg, idx, val),
                 lambda: api.SDPCone.from_columns(n, m, columns())):
        cone = make()
        try:
            got.append(_state(api, cone, m, g))
        finally:
            cone.destroy()
    if "csc_beg" not in g:                   # ... and the cone generated in HBM holds the same data
        cone = api.SDPCone.synthetic(n, m)
        try:
            syn = _state(api, cone, m, g)
        finally:
            cone.destroy()
        # (the constraint matrices are the same bits; the objective C = I + sum y0_c A_c is summed with fused multiply-adds by
        # the device generator and without by the host restatement: equal to rounding, not to the bit)
        a, b = syn["S"][lower_mask(n)], got[0]["S"][lower_mask(n)]
        assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(b))
        assert np.max(np.abs(syn["traces"] - got[0]["traces"])) <= 1e-13 * np.max(np.abs(got[0]["traces"]))
        assert np.max(np.abs(syn["M"][lower_mask(m)] - got[0]["M"][lower_mask(m)])) <= 1e-11 * np.max(np.abs(got[0]["M"][lower_mask(m)]))
    if "M_inf" in g and "csc_beg" in g:      # every way in gives the compiled reference's numbers, not just each other's
        from util import check_close
        for st in got:
            check_close(st["M"][lower_mask(m)], g["M_inf"][lower_mask(m)], name + " M")
            check_close(st["ASinv"], g["ASinv_inf"], name + " ASinv")
            check_close(st["ASinvRdSinv"], g["ASinvRdSinv_inf"], name + " ASinvRdSinv")
    for other in got[1:]:
        msk = {"M": lower_mask(m), "S": lower_mask(n)}
        for k in ("S", "M", "ASinv", "ASinvRdSinv", "traces"):
            a, b = (got[0][k][msk[k]], other[k][msk[k]]) if k in msk else (got[0][k], other[k])
            assert np.array_equal(a, b), k
        assert got[0]["logdet"] == other["logdet"]
        for k in got[0]["presolve"]:
            assert np.array_equal(got[0]["presolve"][k], other["presolve"][k]), k


def _packed_positions(n):
    jj = np.repeat(np.arange(n, dtype=np.int64), np.arange(n, 0, -1))
    ii = np.concatenate([np.arange(j, n, dtype=np.int64) for j in range(n)])
    return ii, jj


def test_the_headline_family_column_by_column_equals_the_synthetic_cone():
    """BASELINE configs[3] -- n = m = 2000, 1.65e9 entries, 40 % filled -- pushed through the builder one column at a time
    (generated here from the SURVEY 8(d) stream with numpy, eight columns in flight; the caller never holds more than those)
    against tests/golden/full2000.npz at the bench state: the two Phase-A checksums bench.py prints and the 16 committed rows
    of M, i.e. what HMiConeCreateSynthetic's cone is held to"""
    from hdsdp_amd import api
    from test_gpu_parity import _splitmix_u, check_full_size_state
    g = load_golden("full2000")
    n, m = int(g["n"]), int(g["m"])
    P = n * (n + 1) // 2
    ii, jj = _packed_positions(n)
    diag = ii == jj
    k = np.arange(P, dtype=np.uint64)
    y0 = _splitmix_u(np.uint64(2 * m * P) + np.arange(m, dtype=np.uint64))

    def column(c):
        base = np.uint64(2 * c * P)
        v = _splitmix_u(base + np.uint64(2) * k)
        w = _splitmix_u(base + np.uint64(2) * k + np.uint64(1))
        keep = diag | (w >= 0.2)
        return np.flatnonzero(keep).astype(np.int32), v[keep]

    Cp = diag.astype(np.float64)            # C = I + sum_c y0_c A_c, every entry summed in constraint order

    def columns():
        with ThreadPoolExecutor(8) as pool:
            for c, (pi, pv) in enumerate(pool.map(column, range(m))):
                Cp[pi] += y0[c] * pv
                yield c + 1, pi, pv
        yield 0, np.arange(P, dtype=np.int32), Cp

    cone = api.SDPCone.from_columns(n, m, columns())
    try:
        assert cone.stored_entries > 1_600_000_000 and cone.path == 0
        kkt = api.KKT(m, [cone])
        b = cone.traces()
        assert np.max(np.abs(b - g["bench_b"])) <= 1e-12 * np.max(np.abs(b))
        cs = check_full_size_state(cone, kkt, g, "bench")
        assert abs(cs["sum_d2"] - float(g["bench_sum_d2"])) <= 1e-9 * abs(float(g["bench_sum_d2"]))
        assert abs(cs["sum_d1w"] - float(g["bench_sum_d1w"])) <= 1e-9 * abs(float(g["bench_sum_d1w"]))
        kkt.destroy()
    finally:
        cone.destroy()


def test_a_block_beyond_int32_streams_in_and_builds():
    """an UNMASKED fully dense block, n = 2000, m = 1100: 2 001 000 entries per matrix, 2.2e9 > INT_MAX in all, so the column
    pointer of the last column overflows the reference's int CSC -- column by column through the builder, then rows of M, both
    vectors and log det S against the definition on the host (B_i = S^-1 A_i S^-1, M_ij = <B_i, A_j>; numpy fp64)"""
    from hdsdp_amd import api
    from test_gpu_parity import _splitmix_u
    n, m = 2000, 1100
    P = n * (n + 1) // 2
    ii, jj = _packed_positions(n)
    k = np.arange(P, dtype=np.uint64)
    every = np.arange(P, dtype=np.int32)

    def packed(c):                           # matrix c (0 = the objective): draw 2 (cP + k) of the stream, all positions kept
        return _splitmix_u(np.uint64(2 * c * P) + np.uint64(2) * k)

    def columns():
        with ThreadPoolExecutor(8) as pool:
            for c, pv in enumerate(pool.map(packed, range(m + 1))):
                yield c, every, pv

    def dense(c):
        A = np.zeros((n, n))
        A[ii, jj] = packed(c)
        return A + np.tril(A, -1).T

    cone = api.SDPCone.from_columns(n, m, columns())
    try:
        assert cone.stored_entries == (m + 1) * P > 2 ** 31 - 1
        assert cone.path == 0
        Rd, tau = -10.0 * n, 1.0
        y = 0.3 * np.sin(0.37 * np.arange(m))
        cone.set_start(Rd)
        assert cone.check_is_interior(tau, y)
        kkt = api.KKT(m, [cone])
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        ex = kkt.export()
        M = kkt.M.copy()
        ld = cone.log_barrier(tau)
        kkt.destroy()
    finally:
        cone.destroy()
    # the definition, on the host
    S = tau * dense(0) - Rd * np.eye(n)
    rows = [0, 549, 1099]
    cols = [0, 1, 548, 549, 777, 1098, 1099]
    for i in range(m):                      # (S needs every constraint: 1100 regenerated columns)
        S -= y[i] * dense(i + 1)
    sign, logdet = np.linalg.slogdet(S)
    assert sign > 0 and abs(ld - logdet) <= 1e-12 * abs(logdet)
    Sinv = np.linalg.inv(S)
    Sinv = 0.5 * (Sinv + Sinv.T)
    Aj = {j: dense(j + 1) for j in set(rows) | set(cols)}
    scale = None
    for i in rows:
        B = Sinv @ Aj[i] @ Sinv
        assert abs(ex["ASinv"][i] - np.sum(Sinv * Aj[i])) <= 1e-10 * abs(np.sum(np.abs(Sinv * Aj[i])))
        assert abs(ex["ASinvRdSinv"][i] - Rd * np.trace(B)) <= 1e-10 * abs(Rd) * np.sum(np.abs(np.diag(B)))
        for j in cols:
            ref = float(np.sum(B * Aj[j]))
            scale = scale or abs(float(np.sum(B * Aj[i])))
            got = M[min(i, j), max(i, j)]          # (C-order view of the column-major matrix: lower triangle at [col, row])
            assert abs(got - ref) <= 1e-10 * scale, (i, j, got, ref)
