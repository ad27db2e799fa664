"""CPU tests (no GPU): pin the oracle (oracle/hdsdp_oracle.c) against the golden vectors that were
generated from the compiled reference.  This is what makes the oracle trustworthy as a checker."""
import os
import sys

import numpy as np
import pytest

from util import golden_schur_dense, KKT_TOL, RATIO_TOL, check_close, load_golden, lower_mask, primal_X, y_of

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_py  # noqa: E402

CSC_CASES = ["theta1_A", "theta1_B", "mcp100_A", "mcp100_B", "gpp100_A", "gpp100_B", "mix40_A", "mix40_B"]
SYN_SMALL = ["syn64", "syn96x40_B"]


def _block(name, g):
    n, m = int(g["dims"][0]), int(g["dims"][1])
    if "csc_beg" in g:
        beg, idx, val = g["csc_beg"], g["csc_idx"], g["csc_val"]
    else:
        beg, idx, val, b = oracle_py.synth_csc(n, m)
        assert int(beg[-1]) == int(g["csc_nnz"][0])
        assert np.array_equal(b, g["b"])
    return oracle_py.Block(n, m, beg, idx, val), n, m


@pytest.mark.parametrize("name", CSC_CASES + SYN_SMALL)
def test_oracle_matches_reference(name):
    g = load_golden(name)
    blk, n, m = _block(name, g)
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    p = blk.presolve()
    for k in ("coef_type", "coef_rank", "coef_nnz", "kkt_perm", "kkt_strategy"):
        assert np.array_equal(p[k], g[k]), k
    assert p["obj_type"] == int(g["obj_type"][0])
    S = blk.assemble_S(tau, y, Rd)
    msk = lower_mask(n)
    check_close(S[msk], g["S"][msk], "S")
    Lf, info = blk.factor(S)
    assert info == 0
    assert np.allclose(np.diag(Lf), g["Ldiag"], rtol=1e-12)
    assert abs(blk.logdet(Lf) - float(g["logdet"][0])) <= 1e-12 * abs(float(g["logdet"][0]))
    if "rt_step1" in g:   # ratio test: fresh call, then the warm-started one (dense dual matrices only, see ref_dump.c)
        for tag in ("1", "2"):
            par = g["rt_par" + tag]
            step = blk.ratio_test(Lf, float(par[0]), g["rt_dy" + tag], float(par[1]) * Rd)
            ref = float(g["rt_step" + tag][0])
            assert abs(step - ref) <= RATIO_TOL * abs(ref), (tag, step, ref)
    Sinv = blk.inverse(Lf)
    check_close(Sinv, g["Sinv"], "Sinv")
    mm = lower_mask(m)
    k = blk.kkt_build(Sinv, Rd, 0)
    check_close(k["M"][mm], g["M_inf"][mm], "M_inf")
    check_close(k["ASinv"], g["ASinv_inf"], "ASinv")
    check_close(k["ASinvRdSinv"], g["ASinvRdSinv_inf"], "ASinvRdSinv")
    check_close([k["TraceSinv"]], g["TraceSinv_inf"], "TraceSinv")
    h = blk.kkt_build(Sinv, Rd, 2)
    check_close(h["M"][mm], g["M_hsd"][mm], "M_hsd")
    check_close(h["ASinvCSinv"], g["ASinvCSinv_hsd"], "ASinvCSinv")
    for got, ref in zip((h["CSinv"], h["CSinvCSinv"], h["CSinvRdSinv"], h["TraceSinv"]), g["hsd_scalars"]):
        check_close([got], [ref], "hsd scalar")
    c = blk.kkt_build(Sinv, Rd, 1)
    check_close(c["ASinv"], g["ASinv_cor"], "cor ASinv")
    check_close(c["ASinvRdSinv"], g["ASinvRdSinv_cor"], "cor ASinvRdSinv")
    # fixed strategies: M3 and M4 always apply (HKKTBuildUpFixed, reference cross-strategy invariant)
    for strat, key in ((2, "M_inf_fixedM3"), (3, "M_inf_fixedM4")):
        f = blk.kkt_build(Sinv, Rd, 0, fixed=strat)
        check_close(f["M"][mm], g[key][mm], key)
    if "pr_X" in g:   # primal recovery (hdsdp_conic_sdp.c:2393-2446)
        X = blk.get_primal(float(g["pr_mu"][0]), g["pr_y"], g["pr_dy"])
        assert X is not None
        check_close(X, g["pr_X"], "primal recovery")
    # KKT_TYPE_PRIMAL: the same builder on the registered primal matrix (hdsdp_conic_sdp.c:1745-1753)
    pk = blk.kkt_build(primal_X(n), Rd, 3)
    check_close(pk["M"][mm], g["M_pri"][mm], "M_pri")
    check_close(pk["ASinv"], g["ASinv_pri"], "ASinv_pri")
    check_close(pk["ASinvRdSinv"], g["ASinvRdSinv_pri"], "ASinvRdSinv_pri")
    check_close([pk["TraceSinv"]], g["TraceSinv_pri"], "TraceSinv_pri")
    # state "C" (ref_dump.c): a dual perturbation on the diagonal of S (hdsdp_conic_sdp.c:383, :2237-2241), then a reduced
    # residual (:2225-2229), then Phase B's form -- residual 0, the shift carried by the perturbation (hdsdp_algo.c:1698-1704)
    for kc in (1, 2, 3):
        pert, Rdc = float(g["c_par"][2 * kc - 2]), float(g["c_par"][2 * kc - 1])
        Sc = blk.assemble_S(tau, y, Rdc - pert)
        Lc, infoc = blk.factor(Sc)
        assert (infoc == 0) == bool(g["c%d_interior" % kc][0]), kc
        if infoc != 0:
            continue
        ldc = float(g["c%d_logdet" % kc][0])
        assert abs(blk.logdet(Lc) - ldc) <= 1e-12 * abs(ldc)
        Sic = blk.inverse(Lc)
        kk = blk.kkt_build(Sic, Rdc, 0)
        check_close(kk["M"][mm], g["c%d_M_inf" % kc][mm], "state C M")
        check_close(kk["ASinv"], g["c%d_ASinv_inf" % kc], "state C ASinv")
        check_close(kk["ASinvRdSinv"], g["c%d_ASinvRdSinv_inf" % kc], "state C ASinvRdSinv")
        check_close([kk["TraceSinv"]], g["c%d_TraceSinv_inf" % kc], "state C TraceSinv")
        hh = blk.kkt_build(Sic, Rdc, 2)
        check_close(hh["ASinvCSinv"], g["c%d_ASinvCSinv_hsd" % kc], "state C ASinvCSinv")
        for got, ref in zip((hh["CSinv"], hh["CSinvCSinv"], hh["CSinvRdSinv"], hh["TraceSinv"]), g["c%d_hsd_scalars" % kc]):
            check_close([got], [ref], "state C hsd scalar")
        cc = blk.kkt_build(Sic, Rdc, 1)
        check_close(cc["ASinv"], g["c%d_ASinv_cor" % kc], "state C cor ASinv")
        check_close(cc["ASinvRdSinv"], g["c%d_ASinvRdSinv_cor" % kc], "state C cor ASinvRdSinv")
        Mc = kk["M"].copy()
        Mc[np.arange(m), np.arange(m)] += float(g["diag_add"][0])
        xc = oracle_py.pcg_solve(Mc, g["c%d_ASinv_inf" % kc])
        assert np.linalg.norm(xc - g["c%d_sol_ASinv" % kc]) <= 1e-8 * np.linalg.norm(g["c%d_sol_ASinv" % kc])
    # the Schur solves (Jacobi PCG to the reference's tolerances)
    Ms = k["M"].copy()
    Ms[np.arange(m), np.arange(m)] += float(g["diag_add"][0])
    for rhs, key in ((g["b"], "sol_b"), (g["ASinv_inf"], "sol_ASinv"), (g["ASinvRdSinv_inf"], "sol_ASinvRdSinv")):
        x = oracle_py.pcg_solve(Ms, rhs)
        assert np.linalg.norm(x - g[key]) <= 1e-8 * np.linalg.norm(g[key]), key
    blk.close()


def test_oracle_generator_checksums():
    """SURVEY.md 8(c) known answers of the synthetic family at n=m=100 (full reference run)"""
    g = load_golden("syn100")
    beg, idx, val, b = oracle_py.synth_csc(100, 100)
    assert int(beg[-1]) == 212782
    blk = oracle_py.Block(100, 100, beg, idx, val)
    S = blk.assemble_S(1.0, np.zeros(100), -1000.0)
    Lf, info = blk.factor(S)
    assert info == 0
    assert abs(blk.logdet(Lf) - 6.909052338645e+02) < 1e-9
    k = blk.kkt_build(blk.inverse(Lf), -1000.0, 0)
    mm = lower_mask(100)
    check_close(k["M"][mm], g["M_inf"][mm], "syn100 M")
    assert abs(k["M"][mm].sum() - 1.367732857163390e-01) < 1e-12
    blk.close()


def test_potrf_reports_first_bad_pivot():
    n = 12
    A = np.eye(n) * 2.0
    A[7, 7] = -1.0
    Lf = A.copy()
    info = oracle_py.lib().orc_potrf(n, Lf.ctypes.data_as(oracle_py.C.POINTER(oracle_py.C.c_double)))
    assert info == 8


@pytest.mark.parametrize("name", ["truss1_A", "blocks3_A", "chain16_A"])
def test_oracle_multi_block_matches_reference(name):
    """truss1 (six 2 x 2 blocks and a 1 x 1) and blocks3 (21 / 34 / 9, most constraints zero on each block): the Schur
    operator is the SUM of the per-cone contributions (interface/hdsdp_schur.c:256-268); the oracle builds each block
    and the parts are added here"""
    g = load_golden(name)
    nb, m = int(g["mb_dims"][0]), int(g["mb_dims"][1])
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    M = np.zeros((m, m)); a = np.zeros(m); r = np.zeros(m); c = np.zeros(m); sc = np.zeros(4); ld = 0.0
    Mi = np.zeros((m, m)); ai = np.zeros(m); ri = np.zeros(m)
    for k in range(nb):
        n = int(g["mb_blkdims"][k])
        blk = oracle_py.Block(n, m, g["mb%d_beg" % k], g["mb%d_idx" % k], g["mb%d_val" % k])
        S = blk.assemble_S(tau, y, Rd)
        Lf, info = blk.factor(S)
        assert info == 0
        ld += blk.logdet(Lf)
        Sinv = blk.inverse(Lf)
        h = blk.kkt_build(Sinv, Rd, 2)
        M += h["M"]; a += h["ASinv"]; r += h["ASinvRdSinv"]; c += h["ASinvCSinv"]
        sc += [h["CSinv"], h["CSinvCSinv"], h["CSinvRdSinv"], h["TraceSinv"]]
        f = blk.kkt_build(Sinv, Rd, 0)
        Mi += f["M"]; ai += f["ASinv"]; ri += f["ASinvRdSinv"]
        blk.close()
    mm = lower_mask(m)
    assert abs(ld - float(g["logdet"][0])) <= 1e-12 * abs(float(g["logdet"][0]))
    # chain16: the reference ran its SPARSE Schur operator (aggregated CSC pattern); outside the pattern M is exactly zero
    Mh, Mf = golden_schur_dense(g, "M_hsd"), golden_schur_dense(g, "M_inf")
    if name == "chain16_A":
        assert int(g["kkt_sparse"][0]) == 1 and not M[mm][Mh[mm] == 0.0].any()
    check_close(M[mm], Mh[mm], "M_hsd")
    check_close(a, g["ASinv_hsd"], "ASinv"); check_close(r, g["ASinvRdSinv_hsd"], "ASinvRdSinv")
    check_close(c, g["ASinvCSinv_hsd"], "ASinvCSinv"); check_close(sc, g["hsd_scalars"], "scalars")
    check_close(Mi[mm], Mf[mm], "M_inf")
    check_close(ai, g["ASinv_inf"], "ASinv_inf"); check_close(ri, g["ASinvRdSinv_inf"], "ASinvRdSinv_inf")
    x = oracle_py.pcg_solve(Mi, g["b"])
    assert np.linalg.norm(x - g["sol_b"]) <= 1e-8 * np.linalg.norm(g["sol_b"])


@pytest.mark.parametrize("name", ["indef96", "indef150"])
def test_oracle_indefinite_schur_fallback_matches_reference(name):
    """the Schur system object on a matrix that is not positive definite (strongly indefinite; a few eigenvalues just
    below zero): the reference's PCG gives up, HFpLinsysSwitchToIndefinite replaces it by LDL^T for good
    (linalg/hdsdp_linsolver.c:1827-1857, 2029-2110); the restatement must take the same way and land on the same
    solution, in the first round and in the next one"""
    g = load_golden(name)
    M, b = g["indef_M"], g["indef_b"]
    rc_num, rc_sol, lt, rc_num2, rc_sol2, lt2 = (int(v) for v in g["indef_codes"])
    assert (rc_num, rc_sol, rc_num2, rc_sol2) == (0, 0, 0, 0) and lt == 6 and lt2 == 6
    x, lin_type = oracle_py.schur_solve(M, b, lin_type=5)
    assert lin_type == lt
    assert np.linalg.norm(x - g["indef_x1"]) <= 1e-11 * np.linalg.norm(g["indef_x1"])
    M2 = M.copy()
    i = np.arange(M.shape[0])
    M2[i, i] += 0.125
    x2, lin_type = oracle_py.schur_solve(M2, b, lin_type=lin_type)
    assert lin_type == lt2
    assert np.linalg.norm(x2 - g["indef_x2"]) <= 1e-11 * np.linalg.norm(g["indef_x2"])
    # a positive definite matrix does not switch
    A = np.triu(M) + np.triu(M, 1).T
    P = A @ A.T / M.shape[0] + np.eye(M.shape[0])
    xp, lin_type = oracle_py.schur_solve(np.triu(P), b, lin_type=5)
    assert lin_type == 5 and np.linalg.norm(P @ xp - b) <= 1e-9 * np.linalg.norm(b)


def test_streamed_generator_is_the_csc_generator():
    """orc_synth_matrix / orc_synth_objective (no CSC: they serve the sizes the int32 CSC cannot hold) are bit-equal to
    orc_synth_csc where both exist -- which in turn is pinned to the compiled reference by the syn* goldens"""
    n, m = 37, 23
    beg, idx, val, b = oracle_py.synth_csc(n, m)
    P = n * (n + 1) // 2

    def unpack(col):
        pk = np.zeros(P)
        pk[idx[beg[col]:beg[col + 1]]] = val[beg[col]:beg[col + 1]]
        A = np.zeros((n, n))
        k = 0
        for j in range(n):
            for i in range(j, n):
                A[i, j] = A[j, i] = pk[k]
                k += 1
        return A

    y0, Cm = oracle_py.synth_objective(n, m, threads=3)
    assert np.array_equal(Cm, unpack(0))
    for c in (0, 7, m - 1):
        A = oracle_py.synth_matrix(n, c)
        assert np.array_equal(A, unpack(c + 1))
        assert abs(np.trace(A) - b[c]) < 1e-13


def test_full_size_golden_is_what_the_generator_and_fp64_blas_say():
    """tests/golden/full2000.npz (n = m = 2000, made by oracle/full_size_golden.py) spot-checked on the CPU: the objective
    is regenerated, S^-1 formed, and a handful of entries of M, ASinv and ASinvRdSinv recomputed from the definition
    M_ij = tr(A_i S^-1 A_j S^-1) -- a different formula from the congruence + Gram form the fixture was built with"""
    g = load_golden("full2000")
    n, m = int(g["n"]), int(g["m"])
    y0, Cm = oracle_py.synth_objective(n, m)
    rows = [int(r) for r in g["bench_rows"]]
    for st, pick in (("bench", (0, 5, 15)), ("hard", (3, 12))):
        y, Rd = np.asarray(g[st + "_y"]), float(g[st + "_Rd"])
        S = Cm - Rd * np.eye(n)
        if np.any(y != 0.0):
            for i in range(m):
                S -= y[i] * oracle_py.synth_matrix(n, i)
        Sinv = np.linalg.inv(S)
        Sinv = 0.5 * (Sinv + Sinv.T)
        for q in pick:
            i = rows[q]
            Ai = oracle_py.synth_matrix(n, i)
            B = Sinv @ Ai @ Sinv
            assert abs(np.sum(Ai * Sinv) - g[st + "_ASinv"][i]) <= 1e-9 * np.max(np.abs(g[st + "_ASinv"]))
            assert abs(Rd * np.trace(B) - g[st + "_ASinvRdSinv"][i]) <= 1e-9 * np.max(np.abs(g[st + "_ASinvRdSinv"]))
            for j in (0, 128, 1023, 1999):
                ref = float(np.sum(oracle_py.synth_matrix(n, j) * B))
                assert abs(ref - g[st + "_M_rows"][q, j]) <= 1e-9 * np.max(np.abs(g[st + "_diag_M"])), (st, i, j)
            assert abs(g[st + "_M_rows"][q, i] - g[st + "_diag_M"][i]) == 0.0


def test_row_subset_generator_agrees_with_the_pinned_oracle():
    """oracle/row_subset_golden.py (the generator of tests/golden/full8000_rows.npz and rows_1000x8000.npz) at a size the
    pinned oracle can check in full: its rows of M, both vectors, log det S and the traces equal the reference-faithful
    column builders' (oracle/hdsdp_oracle.c) at both states"""
    import row_subset_golden as rs
    n, m = 96, 200
    res = rs.generate(n, m, 2, lambda s: None)
    beg, idx, val, b = oracle_py.synth_csc(n, m)
    blk = oracle_py.Block(n, m, beg, idx, val)
    try:
        for st in ("bench", "hard"):
            y, Rd = res[st + "_y"], res[st + "_Rd"]
            Lf, info = blk.factor(blk.assemble_S(1.0, y, Rd))
            assert info == 0
            ref = blk.kkt_build(blk.inverse(Lf), Rd, 0)
            M = ref["M"]
            Mf = np.triu(M) + np.triu(M, 1).T          # C-order view of a column-major lower triangle
            rows = res[st + "_rows"]
            assert np.max(np.abs(Mf[rows, :] - res[st + "_M_rows"])) <= 1e-12 * np.max(np.abs(Mf))
            for k in ("ASinv", "ASinvRdSinv"):
                assert np.max(np.abs(ref[k] - res[st + "_" + k])) <= 1e-12 * np.max(np.abs(ref[k]))
            assert abs(res[st + "_logdetS"] - blk.logdet(Lf)) <= 1e-12 * abs(blk.logdet(Lf))
            assert np.max(np.abs(b - res[st + "_b"])) <= 1e-12
    finally:
        blk.close()


def test_config5_row_fixture_is_what_the_definition_says():
    """tests/golden/full8000_rows.npz (BASELINE configs[4], n = 2000, m = 8000) spot-checked at the bench state: the
    objective of the 8000-constraint instance is regenerated, S^-1 formed by a plain inverse, and entries of two fixture
    rows recomputed as tr(A_i S^-1 A_j S^-1), with the vectors' entries beside them"""
    g = load_golden("full8000_rows")
    n, m = int(g["n"]), int(g["m"])
    assert (n, m) == (2000, 8000)
    rows = [int(r) for r in g["bench_rows"]]
    assert rows[:64] == list(range(64))              # row subset [0, 64), eight rows per rank of the cyclic deal
    y0, Cm = oracle_py.synth_objective(n, m)
    Rd = float(g["bench_Rd"])
    Sinv = np.linalg.inv(Cm - Rd * np.eye(n))
    Sinv = 0.5 * (Sinv + Sinv.T)
    scale = float(np.max(np.abs(g["bench_M_rows"])))
    for q in (5, len(rows) - 1):
        i = rows[q]
        Ai = oracle_py.synth_matrix(n, i)
        B = Sinv @ Ai @ Sinv
        for j in (0, 4097, m - 1):
            Aj = oracle_py.synth_matrix(n, j)
            assert abs(float(np.sum(Aj * B)) - g["bench_M_rows"][q, j]) <= 1e-10 * scale, (i, j)
            assert abs(float(np.sum(Aj * Sinv)) - g["bench_ASinv"][j]) <= 1e-10 * np.max(np.abs(g["bench_ASinv"]))
            assert abs(float(np.trace(Aj)) - g["bench_b"][j]) <= 1e-12 * n


def test_the_sparse_operators_indefinite_round_is_what_the_golden_says():
    """arrow128_A's indefinite round (oracle/ref_dump.c, sdpam mode: the compiled reference's SPARSE Schur operator with its
    diagonal lowered by indef_shift, factored and solved by its LDL' without pivoting): the dumped solution solves
    (M_inf - shift I) x = b for the dumped M_inf, five eigenvalues of that matrix are negative, and the reference's own codes
    say "factorised, solved, not positive definite" -- what tests/test_gpu_parity.py holds the tile form to."""
    import scipy.sparse as sp
    g = load_golden("arrow128_A")
    m = int(g["mb_dims"][1])
    A = sp.csc_matrix((g["M_inf"], g["kkt_idx"], g["kkt_beg"]), shape=(m, m)).toarray()
    A = A + np.tril(A, -1).T - float(g["indef_shift"]) * np.eye(m)
    assert int(np.sum(np.linalg.eigvalsh(A) < 0)) == 5
    assert list(g["indef_codes"]) == [0, 0, 0]
    assert np.linalg.norm(A @ g["indef_sol"] - g["b"]) <= 1e-12 * np.linalg.norm(g["b"])
