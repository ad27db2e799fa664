"""GPU parity tests proper: the HIP path, called through the C ABI, against golden vectors dumped
from the compiled reference (oracle/gen_golden.py).  Tolerance = the reference's own HUtilKKTCheck
bar (|a-b|/(|a|+1e-4) < 1e-8, interface/hdsdp_utils.c:621-641) for M / ASinv / ASinvRdSinv /
ASinvCSinv; 1e-8 relative 2-norm for the Schur solves; 1e-12 relative for logdet(S)."""
import os

import numpy as np
import pytest

from util import golden_schur_dense, primal_X, KKT_TOL, RATIO_TOL, check_close, kkt_err, load_golden, lower_mask, y_of

pytestmark = pytest.mark.gpu

CSC_CASES = ["theta1_A", "theta1_B", "mcp100_A", "mcp100_B", "gpp100_A", "gpp100_B", "mix40_A", "mix40_B"]
SYN_CASES = ["syn64", "syn96x40_B", "syn100", "syn200", "syn2000x32"]


def _make_cone(name, g):
    from hdsdp_amd import api
    n, m = int(g["dims"][0]), int(g["dims"][1])
    if "csc_beg" in g:
        cone = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    else:
        cone = api.SDPCone.synthetic(n, m)
    return cone, n, m


@pytest.mark.parametrize("name", CSC_CASES + SYN_CASES)
def test_schur_against_reference(name):
    from hdsdp_amd import api
    g = load_golden(name)
    cone, n, m = _make_cone(name, g)
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    try:
        cone.set_start(Rd)
        assert cone.check_is_interior(tau, y)
        # S assembly (next-row f1) and its factor
        if "S" in g:
            S = cone.dual_matrix()
            msk = lower_mask(n)
            assert kkt_err(S[msk], g["S"][msk]) < 1e-12
        logdet = cone.log_barrier(tau)
        assert abs(logdet - float(g["logdet"][0])) <= 1e-12 * abs(float(g["logdet"][0]))
        # --- ratio test on the resident factor (next-row f2): same Lanczos recurrence, start vector and acceptance
        # rule as HLanczosSolve; the first call starts fresh, the second is warm-started like the reference's
        if "rt_step1" in g:
            for tag in ("1", "2"):
                par = g["rt_par" + tag]
                step = cone.ratio_test(float(par[0]), g["rt_dy" + tag], float(par[1]))
                ref = float(g["rt_step" + tag][0])
                assert abs(step - ref) <= RATIO_TOL * abs(ref), (name, tag, step, ref)
            if "ck_axpy_half" in g:
                # the checker buffer of the line search: trial point S + 0.5*step*dS, its log-barrier, a ratio test
                # from there (third Lanczos call), a trial point beyond the boundary, an expert check
                ok_h, ld_h, st3 = g["ck_axpy_half"]
                assert cone.axpy_buffer_and_check(0.5 * step, api.BUFFER_DUALCHECK) == bool(ok_h)
                assert abs(cone.log_barrier_of(api.BUFFER_DUALCHECK) - ld_h) <= 1e-11 * abs(ld_h)
                got3 = cone.ratio_test(float(par[0]), g["rt_dy2"], float(par[1]), api.BUFFER_DUALCHECK)
                assert abs(got3 - st3) <= RATIO_TOL * abs(st3), (name, got3, st3)
                assert cone.axpy_buffer_and_check(1.5 * step, api.BUFFER_DUALCHECK) == bool(g["ck_axpy_beyond"][0])
                ok_e, ld_e = g["ck_expert"]
                assert cone.check_is_interior_expert(tau, -1.0, y, -0.5 * Rd, api.BUFFER_DUALCHECK) == bool(ok_e)
                if ok_e:   # (a failed factorisation has no log-barrier: the reference's value is NaN there)
                    assert abs(cone.log_barrier_of(api.BUFFER_DUALCHECK) - ld_e) <= 1e-11 * abs(ld_e)
                # none of this may have touched the factor of S itself
                assert abs(cone.log_barrier_of(api.BUFFER_DUALVAR) - float(g["logdet"][0])) <= 1e-12 * abs(float(g["logdet"][0]))

        # --- the getstat slot: the reference's feature detection from the engine's own presolve (hdsdp_conic_sdp.c:2651-2745)
        if "feat_int" in g:
            fi, fd = cone.detect_feature(g["b"] if "b" in g else cone.traces())
            assert np.array_equal(fi, g["feat_int"]), (name, fi.tolist(), g["feat_int"].tolist())
            assert np.allclose(fd, g["feat_dbl"], rtol=1e-12, atol=0.0), (name, fd.tolist(), g["feat_dbl"].tolist())
        kkt = api.KKT(m, [cone])
        msk = lower_mask(m)
        # --- KKT_TYPE_INFEASIBLE
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        ex = kkt.export()
        check_close(kkt.M[msk], g["M_inf"][msk], name)
        check_close(ex["ASinv"], g["ASinv_inf"], name)
        check_close(ex["ASinvRdSinv"], g["ASinvRdSinv_inf"], name)
        check_close([ex["TraceSinv"]], g["TraceSinv_inf"], name)
        # --- the three Phase-A solves (interface/hdsdp_algo.c:1099-1101)
        b = cone.traces() if "b" not in g else g["b"]
        if "b" in g:
            assert kkt_err(cone.traces(), g["b"]) < 1e-12 or "csc_beg" in g  # synthetic: b_i = tr(A_i)
        kkt.add_to_diag(float(g["diag_add"][0]))
        kkt.factorize()
        for rhs, key in ((b, "sol_b"), (g["ASinv_inf"], "sol_ASinv"), (g["ASinvRdSinv_inf"], "sol_ASinvRdSinv")):
            x = kkt.solve(np.array(rhs, dtype=np.float64))
            ref = g[key]
            assert np.linalg.norm(x - ref) <= 1e-8 * np.linalg.norm(ref), key
        # --- KKT_TYPE_HOMOGENEOUS
        if "ASinvCSinv_hsd" in g:
            kkt.build_up(api.KKT_TYPE_HOMOGENEOUS)
            ex = kkt.export()
            if "M_hsd" in g:
                check_close(kkt.M[msk], g["M_hsd"][msk], name)
            check_close(ex["ASinvCSinv"], g["ASinvCSinv_hsd"], name)
            sc = g["hsd_scalars"]  # CSinv, CSinvCSinv, CSinvRdSinv, TraceSinv
            for got, ref in zip((ex["CSinv"], ex["CSinvCSinv"], ex["CSinvRdSinv"], ex["TraceSinv"]), sc):
                check_close([got], [ref], name)
        # --- KKT_TYPE_CORRECTOR leaves M alone, refreshes the two vectors (hdsdp_schur.c:156-162)
        Mbefore = kkt.M.copy()
        kkt.build_up(api.KKT_TYPE_CORRECTOR)
        ex = kkt.export()
        check_close(ex["ASinv"], g["ASinv_cor"], name)
        check_close(ex["ASinvRdSinv"], g["ASinvRdSinv_cor"], name)
        assert np.array_equal(Mbefore, kkt.M)
        # --- fixed-strategy builds give the same matrix (reference invariant, hdsdp_utils.c:536-707)
        kkt.build_up_fixed(api.KKT_TYPE_INFEASIBLE, api.KKT_M4)
        check_close(kkt.M[msk], g["M_inf"][msk], name)
        # --- primal recovery X = mu L^-T (sym(L^-1 dS L^-T) + I) L^-1 (next-row f4, hdsdp_conic_sdp.c:2393-2446)
        if "pr_checks" in g:
            X = cone.get_primal(float(g["pr_mu"][0]), g["pr_y"], g["pr_dy"])
            assert X is not None
            if "pr_X" in g:
                check_close(X, g["pr_X"], name + " primal recovery")
            ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")   # X is symmetric: index order is moot
            got = np.array([np.trace(X), X.sum(), (X * np.cos(0.013 * ii + 0.007 * jj)).sum()])
            ref = g["pr_checks"]
            assert np.max(np.abs(got - ref)) <= 1e-9 * np.max(np.abs(ref)), (got, ref)
            if "ut_scalars" in g:
                # the remaining cone utilities on that X: y += A X, <C, X>, <X, S>, data norms, objective scaling
                ax = cone.a_times_x(g["pr_X"] if "pr_X" in g else X, 0.25 * np.arange(m))
                check_close(ax, g["ut_atimesx"], name + " A times X")
                us = g["ut_scalars"]
                Xr = g["pr_X"] if "pr_X" in g else X
                for gotv, refv, what in ((cone.trace_cx(Xr), us[0], "tr CX"), (cone.x_dot_s(Xr), us[1], "X.S"),
                                         (cone.coeff_norm(1), us[2], "rows abs"), (cone.coeff_norm(2), us[3], "rows fro"),
                                         (cone.obj_norm(1), us[4], "obj abs"), (cone.obj_norm(2), us[5], "obj fro")):
                    assert abs(gotv - refv) <= 1e-11 * max(abs(refv), 1e-300), (name, what, gotv, refv)
                cone.scal_by_constant(0.5)
                assert abs(cone.obj_norm(2) - us[6]) <= 1e-11 * max(abs(us[6]), 1e-300)
                cone.scal_by_constant(2.0)
                Sd = cone.get_dual()
                assert np.array_equal(Sd, Sd.T)
                if "S" in g:
                    assert kkt_err(Sd[lower_mask(n)], g["S"][lower_mask(n)]) < 1e-12
        # --- KKT_TYPE_PRIMAL: the builder on a registered primal matrix (hdsdp_conic_sdp.c:1745-1753)
        if "M_pri" in g:
            kkt.register_psdp([primal_X(n)])
            kkt.build_up(api.KKT_TYPE_PRIMAL)
            ex = kkt.export()
            check_close(kkt.M[msk], g["M_pri"][msk], name + " primal")
            check_close(ex["ASinv"], g["ASinv_pri"], name + " primal")
            check_close(ex["ASinvRdSinv"], g["ASinvRdSinv_pri"], name + " primal")
            check_close([ex["TraceSinv"]], g["TraceSinv_pri"], name + " primal")
        # --- state "C": the two cone slots a solve moves between builds (oracle/ref_dump.c) -- a dual perturbation on the
        # diagonal of S and of the checker (hdsdp_conic_sdp.c:383, :441, :2237-2241), a reduced residual (:2225-2229;
        # hdsdp_algo.c:244), and Phase B's form: residual 0, the shift carried by the perturbation (hdsdp_algo.c:1698-1704)
        if "c_par" in g:
            _state_c(name, g, cone, kkt, tau, y, m)
        kkt.destroy()
    finally:
        cone.destroy()


def _state_c(name, g, cone, kkt, tau, y, m):
    from hdsdp_amd import api
    msk = lower_mask(m)
    for kc in (1, 2, 3):
        pert, Rdc = float(g["c_par"][2 * kc - 2]), float(g["c_par"][2 * kc - 1])
        tag = "%s state c%d" % (name, kc)
        cone.set_perturb(pert)
        cone.reduce_resi(Rdc)
        assert cone.check_is_interior(tau, y) == bool(g["c%d_interior" % kc][0]), tag
        if not g["c%d_interior" % kc][0]:
            continue
        ld = float(g["c%d_logdet" % kc][0])
        assert abs(cone.log_barrier(tau) - ld) <= 1e-12 * abs(ld), tag
        kkt.build_up(api.KKT_TYPE_HOMOGENEOUS)
        ex = kkt.export()
        check_close(ex["ASinvCSinv"], g["c%d_ASinvCSinv_hsd" % kc], tag)
        for got, ref in zip((ex["CSinv"], ex["CSinvCSinv"], ex["CSinvRdSinv"], ex["TraceSinv"]), g["c%d_hsd_scalars" % kc]):
            check_close([got], [ref], tag)
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        ex = kkt.export()
        check_close(kkt.M[msk], g["c%d_M_inf" % kc][msk], tag)
        check_close(ex["ASinv"], g["c%d_ASinv_inf" % kc], tag)
        check_close(ex["ASinvRdSinv"], g["c%d_ASinvRdSinv_inf" % kc], tag)
        check_close([ex["TraceSinv"]], g["c%d_TraceSinv_inf" % kc], tag)
        kkt.add_to_diag(float(g["diag_add"][0]))
        kkt.factorize()
        x = kkt.solve(np.array(g["c%d_ASinv_inf" % kc], dtype=np.float64))
        ref = g["c%d_sol_ASinv" % kc]
        assert np.linalg.norm(x - ref) <= 1e-8 * np.linalg.norm(ref), tag
        Mbefore = kkt.M.copy()
        kkt.build_up(api.KKT_TYPE_CORRECTOR)
        ex = kkt.export()
        check_close(ex["ASinv"], g["c%d_ASinv_cor" % kc], tag)
        check_close(ex["ASinvRdSinv"], g["c%d_ASinvRdSinv_cor" % kc], tag)
        assert np.array_equal(Mbefore, kkt.M)
        # the checker takes the perturbation too, on top of the caller's own diagonal term
        ok_e, ld_e = g["c%d_ck_expert" % kc]
        assert cone.check_is_interior_expert(tau, -1.0, y, -0.5 * Rdc, api.BUFFER_DUALCHECK) == bool(ok_e), tag
        if ok_e:
            assert abs(cone.log_barrier_of(api.BUFFER_DUALCHECK) - ld_e) <= 1e-11 * abs(ld_e), tag
    cone.set_perturb(0.0)


@pytest.mark.parametrize("name", ["mcp100_A", "gpp100_B", "theta1_A"])
def test_forced_dense_path_matches(name, monkeypatch):
    """rank-one instances pushed through the dense congruence + Gram kernels give the same numbers"""
    from hdsdp_amd import api
    monkeypatch.setenv("HDSDP_MI355X_FORCE_GEMM", "1")
    g = load_golden(name)
    cone, n, m = _make_cone(name, g)
    try:
        assert cone.path == 0
        cone.set_start(float(g["Rd"][0]))
        assert cone.check_is_interior(float(g["tau"][0]), y_of(g))
        kkt = api.KKT(m, [cone])
        kkt.build_up(api.KKT_TYPE_HOMOGENEOUS)
        ex = kkt.export()
        msk = lower_mask(m)
        check_close(kkt.M[msk], g["M_hsd"][msk], name)
        check_close(ex["ASinv"], g["ASinv_hsd"], name)
        check_close(ex["ASinvRdSinv"], g["ASinvRdSinv_hsd"], name)
        check_close(ex["ASinvCSinv"], g["ASinvCSinv_hsd"], name)
        kkt.destroy()
    finally:
        cone.destroy()


@pytest.mark.parametrize("name,path", [("theta1_A", 2), ("theta1_B", 2), ("theta1_B", 0), ("mcp100_B", 2), ("mcp100_B", 0),
                                       ("gpp100_B", 2), ("mix40_B", 2), ("mix40_B", 0)])
def test_every_device_path_gives_the_same_numbers(name, path, monkeypatch):
    """the three device paths (0 dense congruence + Gram, 1 rank-one, 2 sparse gather) are interchangeable"""
    from hdsdp_amd import api
    monkeypatch.setenv("HDSDP_MI355X_FORCE_PATH", str(path))
    g = load_golden(name)
    cone, n, m = _make_cone(name, g)
    try:
        assert cone.path == path
        cone.set_start(float(g["Rd"][0]))
        assert cone.check_is_interior(float(g["tau"][0]), y_of(g))
        kkt = api.KKT(m, [cone])
        msk = lower_mask(m)
        kkt.build_up(api.KKT_TYPE_HOMOGENEOUS)
        ex = kkt.export()
        check_close(kkt.M[msk], g["M_hsd"][msk], name)
        check_close(ex["ASinv"], g["ASinv_hsd"], name)
        check_close(ex["ASinvRdSinv"], g["ASinvRdSinv_hsd"], name)
        check_close(ex["ASinvCSinv"], g["ASinvCSinv_hsd"], name)
        for got, ref in zip((ex["CSinv"], ex["CSinvCSinv"], ex["CSinvRdSinv"], ex["TraceSinv"]), g["hsd_scalars"]):
            check_close([got], [ref], name)
        kkt.build_up(api.KKT_TYPE_CORRECTOR)
        ex = kkt.export()
        check_close(ex["ASinv"], g["ASinv_cor"], name)
        check_close(ex["ASinvRdSinv"], g["ASinvRdSinv_cor"], name)
        kkt.destroy()
    finally:
        cone.destroy()


def test_default_path_selection():
    """theta1 (sparse rows) -> gather path, mcp100/gpp100 (rank-one rows) -> rank-one path, dense rows -> MFMA path"""
    from hdsdp_amd import api
    for name, want in (("theta1_A", 2), ("mcp100_A", 1), ("gpp100_A", 1), ("mix40_A", 0), ("syn64", 0)):
        g = load_golden(name)
        cone, n, m = _make_cone(name, g)
        try:
            assert cone.path == want, name
        finally:
            cone.destroy()


def test_presolve_plan_matches_reference():
    """classification, ordering and strategy plan (host logic) on the GPU box build of the library"""
    from hdsdp_amd import api
    for name in CSC_CASES:
        g = load_golden(name)
        cone, n, m = _make_cone(name, g)
        try:
            p = cone.presolve()
            assert np.array_equal(p["coef_type"], g["coef_type"]), name
            assert np.array_equal(p["coef_rank"], g["coef_rank"]), name
            assert np.array_equal(p["coef_nnz"], g["coef_nnz"]), name
            assert np.array_equal(p["kkt_perm"], g["kkt_perm"]), name
            assert np.array_equal(p["kkt_strategy"], g["kkt_strategy"]), name
            assert p["obj_type"] == int(g["obj_type"][0]), name
        finally:
            cone.destroy()


# ----------------------------------------------------------------------------------------------
# BASELINE.json full size (configs[3]: n = m = 2000) through size-independent properties
# ----------------------------------------------------------------------------------------------
def _splitmix_u(t):
    """draw number t (0-based, numpy uint64 array) of the SURVEY.md 8(d) stream -> U(-1, 1)"""
    g = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = g + (t.astype(np.uint64) + np.uint64(1)) * g
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return 2.0 * ((z >> np.uint64(11)).astype(np.float64) / 9007199254740992.0) - 1.0


def _synth_matrix(n, c):
    """constraint matrix c (0-based) of the synthetic family as a dense symmetric numpy array"""
    P = n * (n + 1) // 2
    k = np.arange(P, dtype=np.uint64)
    base = np.uint64(2 * c * P)
    v = _splitmix_u(base + np.uint64(2) * k)
    w = _splitmix_u(base + np.uint64(2) * k + np.uint64(1))
    jj = np.repeat(np.arange(n), np.arange(n, 0, -1))
    ii = np.concatenate([np.arange(j, n) for j in range(n)])
    keep = (ii == jj) | (w >= 0.2)
    A = np.zeros((n, n))
    A[ii, jj] = np.where(keep, v, 0.0)
    return A + np.tril(A, -1).T


def test_full_size_known_answers():
    """n = m = 2000 at y = y0 (the generator's strictly feasible point): S = I exactly, so
    ASinv_i = tr(A_i) = b_i, M_ij = <A_i, A_j>, logdet S = 0; and at Rd = -3 (S = 4 I) everything scales by
    powers of 4 (linearity).  Then the Schur system is solved and the residual checked on the host."""
    from hdsdp_amd import api
    n = m = 2000
    P = n * (n + 1) // 2
    y0 = _splitmix_u(np.uint64(2 * m * P) + np.arange(m, dtype=np.uint64))
    cone = api.SDPCone.synthetic(n, m)
    try:
        b = cone.traces()
        kkt = api.KKT(m, [cone])
        cone.set_start(0.0)
        assert cone.check_is_interior(1.0, y0)
        assert abs(cone.log_barrier(1.0)) < 1e-7
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        ex1 = kkt.export()
        M1 = kkt.M.copy()
        assert np.max(np.abs(ex1["ASinv"] - b)) <= 1e-9 * np.max(np.abs(b))
        assert np.all(ex1["ASinvRdSinv"] == 0.0)
        rows = [0, 1, 777, 1999]
        mats = {i: _synth_matrix(n, i) for i in rows}
        for i in rows:
            assert abs(np.trace(mats[i]) - b[i]) < 1e-9
            for j in rows:
                ref = float(np.sum(mats[i] * mats[j]))
                got = M1[min(i, j), max(i, j)]      # C-order view: lower triangle sits at [col, row]
                assert abs(got - ref) <= 1e-9 * abs(ref), (i, j, got, ref)
        # S = 4 I
        cone.set_start(-3.0)
        assert cone.check_is_interior(1.0, y0)
        assert abs(cone.log_barrier(1.0) - n * np.log(4.0)) < 1e-6
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        ex4 = kkt.export()
        msk = lower_mask(m)
        assert np.max(np.abs(kkt.M[msk] * 16.0 - M1[msk])) <= 1e-9 * np.max(np.abs(M1[msk]))
        assert np.max(np.abs(ex4["ASinv"] * 4.0 - b)) <= 1e-9 * np.max(np.abs(b))
        assert np.max(np.abs(ex4["ASinvRdSinv"] - (-3.0) * b / 16.0)) <= 1e-9 * np.max(np.abs(b))
        assert abs(ex4["TraceSinv"] - n / 4.0) < 1e-8
        # solve and check the residual with the host copy of M
        kkt.factorize()
        x = kkt.solve(b)
        Mfull = np.triu(kkt.M) + np.triu(kkt.M, 1).T
        r = Mfull @ x - b
        assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(b)
        # corrector leaves M alone and reproduces the vectors
        kkt.build_up(api.KKT_TYPE_CORRECTOR)
        exc = kkt.export()
        assert np.max(np.abs(exc["ASinv"] - ex4["ASinv"])) <= 1e-10 * np.max(np.abs(ex4["ASinv"]))
        assert np.max(np.abs(exc["ASinvRdSinv"] - ex4["ASinvRdSinv"])) <= 1e-10 * np.max(np.abs(ex4["ASinvRdSinv"]))
        kkt.destroy()
    finally:
        cone.destroy()


def check_full_size_state(cone, kkt, g, st, tol_scale=1.0):
    """one state of tests/golden/full2000.npz (host BLAS-3 fp64 answers at n = m = 2000, oracle/full_size_golden.py)
    against the engine: 16 full rows of M spread over every 128-row tile (tile edges 127|128, 1919|1920 and the ragged
    last tile included), the whole diagonal, checksums and four dense bilinear probes over the WHOLE lower triangle,
    both vectors, log det S, tr S^-1 and the three Phase-A solutions."""
    from hdsdp_amd import api
    m = int(g["m"])
    y, Rd = np.asarray(g[st + "_y"]), float(g[st + "_Rd"])
    cone.set_start(Rd)
    assert cone.check_is_interior(1.0, y), st
    ld = cone.log_barrier(1.0)
    assert abs(ld - float(g[st + "_logdetS"])) <= 1e-12 * abs(float(g[st + "_logdetS"])) * tol_scale + 1e-9, st
    kkt.build_up(api.KKT_TYPE_INFEASIBLE)
    ex = kkt.export()
    Mc = kkt.M                                    # C-order view of the column-major matrix: lower triangle at [col, row]
    Mf = np.triu(Mc) + np.triu(Mc, 1).T           # full symmetric
    rows = np.asarray(g[st + "_rows"])
    check_close(Mf[rows, :], g[st + "_M_rows"], st + " sampled rows of M")
    check_close(np.diag(Mf), g[st + "_diag_M"], st + " diag M")
    check_close(ex["ASinv"], g[st + "_ASinv"], st + " ASinv")
    check_close(ex["ASinvRdSinv"], g[st + "_ASinvRdSinv"], st + " ASinvRdSinv")
    assert abs(ex["TraceSinv"] - float(g[st + "_TraceSinv"])) <= 1e-10 * abs(float(g[st + "_TraceSinv"])), st
    low = np.tril(Mf)
    assert abs(np.sum(low) - float(g[st + "_sum_M_lower"])) <= 1e-10 * np.sum(np.abs(low)), st
    assert abs(np.sum(low ** 2) - float(g[st + "_sumsq_M_lower"])) <= 1e-10 * float(g[st + "_sumsq_M_lower"]), st
    for k in range(4):
        rng = np.random.RandomState(20260 + k)    # oracle/full_size_golden.py: probe_vectors
        u, v = rng.uniform(-1.0, 1.0, m), rng.uniform(-1.0, 1.0, m)
        got = float(u @ Mf @ v)
        assert abs(got - float(g[st + "_probes"][k])) <= 1e-10 * float(np.abs(u) @ np.abs(Mf) @ np.abs(v)), (st, k)
    kkt.factorize()
    sols = [kkt.solve(np.asarray(g[st + "_b"])), kkt.solve(ex["ASinv"]), kkt.solve(ex["ASinvRdSinv"])]
    for x, name in zip(sols, ("d1", "d2", "d3")):
        ref = np.asarray(g[st + "_" + name])
        assert np.linalg.norm(x - ref) <= 1e-8 * np.linalg.norm(ref), (st, name)
    return {"sum_d2": float(np.sum(sols[1])), "sum_d1w": float(np.dot(np.arange(1, m + 1), sols[0]))}


def check_row_subset_state(cone, kkt, g, st):
    """one state of a row-subset fixture (oracle/row_subset_golden.py: tests/golden/full8000_rows.npz for BASELINE
    configs[4], rows_1000x8000.npz for its one-GPU rehearsal) against the engine: log det S, complete rows of M taken
    from the DEVICE copy (HMiKKTGetRows; the operator may run with the host mirror off), both vectors in full, and --
    since the host cannot afford the three solutions at m = 8000 -- rows of the residual M d = rhs formed with the
    FIXTURE's rows of M and the device's solutions."""
    from hdsdp_amd import api
    m = int(g["m"])
    y, Rd = np.asarray(g[st + "_y"]), float(g[st + "_Rd"])
    cone.set_start(Rd)
    assert cone.check_is_interior(1.0, y), st
    ld = cone.log_barrier(1.0)
    assert abs(ld - float(g[st + "_logdetS"])) <= 1e-12 * abs(float(g[st + "_logdetS"])) + 1e-9, st
    b = cone.traces()
    assert np.max(np.abs(b - g[st + "_b"])) <= 1e-12 * np.max(np.abs(b))
    kkt.build_up(api.KKT_TYPE_INFEASIBLE)
    ex = kkt.export()
    rows = np.asarray(g[st + "_rows"])
    Mg = np.asarray(g[st + "_M_rows"])
    check_close(kkt.rows(rows), Mg, st + " rows of M")
    check_close(ex["ASinv"], g[st + "_ASinv"], st + " ASinv")
    check_close(ex["ASinvRdSinv"], g[st + "_ASinvRdSinv"], st + " ASinvRdSinv")
    assert abs(ex["TraceSinv"] - float(g[st + "_TraceSinv"])) <= 1e-10 * abs(float(g[st + "_TraceSinv"])), st
    kkt.factorize()
    for rhs, name in ((b, "d1"), (ex["ASinv"], "d2"), (ex["ASinvRdSinv"], "d3")):
        x = kkt.solve(rhs)
        res = Mg @ x - np.asarray(rhs)[rows]
        assert np.max(np.abs(res)) <= 1e-8 * np.max(np.abs(rhs)), (st, name, float(np.max(np.abs(res))))


def test_full_size_against_host_fp64_at_nontrivial_states():
    """BASELINE configs[3], n = m = 2000, at two states whose dual matrix is NOT a multiple of the identity: the state
    bench.py times (y = 0, Rd = -10 n: S = C + 20000 I with the dense objective C) and a harder one (non-zero y, Rd just
    below the smallest eigenvalue of C - sum y_i A_i: cond(S) ~ 1e3).  The expected values are an independent host
    computation in fp64 (tests/golden/full2000.npz, generated by oracle/full_size_golden.py from the oracle's restatement
    of the SURVEY 8(d) generator: level-3 BLAS congruence + Gram, 15 min on 8 cores); tolerances are the tests/util.py
    bars.  Also pins bench.py's `checksum` (sum_d2, sum_d1w at the bench state) to that independent value."""
    from hdsdp_amd import api
    g = load_golden("full2000")
    n, m = int(g["n"]), int(g["m"])
    cone = api.SDPCone.synthetic(n, m)
    try:
        kkt = api.KKT(m, [cone])
        b = cone.traces()
        assert np.max(np.abs(b - g["bench_b"])) <= 1e-12 * np.max(np.abs(b))
        cs = check_full_size_state(cone, kkt, g, "bench")
        # bench.py prints these two numbers as its `checksum`; BENCH_r01.json carried sum_d2 = -0.7590183915452648
        assert abs(cs["sum_d2"] - float(g["bench_sum_d2"])) <= 1e-9 * abs(float(g["bench_sum_d2"]))
        assert abs(cs["sum_d1w"] - float(g["bench_sum_d1w"])) <= 1e-9 * abs(float(g["bench_sum_d1w"]))
        assert abs(float(g["bench_sum_d2"]) - (-0.7590183915452648)) <= 1e-9
        check_full_size_state(cone, kkt, g, "hard")
        kkt.destroy()
    finally:
        cone.destroy()


def test_ratio_test_bracket_at_full_dimension():
    """n = 2000 (the BASELINE dimension; too large for a reference dump in the golden set): the Lanczos step must
    bracket the cone boundary -- S + 0.99*step*dS is still positive definite, S + 1.05*step*dS is not -- which is
    checked with the device Cholesky itself (size-independent property of the ratio test)."""
    from hdsdp_amd import api
    n, m = 2000, 48
    cone = api.SDPCone.synthetic(n, m)
    try:
        cone.set_start(-10.0 * n)
        y0 = np.zeros(m)
        assert cone.check_is_interior(1.0, y0)
        dy = 40.0 * np.cos(0.7 * np.arange(m) + 0.2)
        step = cone.ratio_test(0.0, dy, 0.0)
        assert np.isfinite(step) and step > 0
        assert cone.check_is_interior(1.0, 0.99 * step * dy)
        assert not cone.check_is_interior(1.0, 1.05 * step * dy)
        # a direction that only moves S into the cone is unbounded
        assert cone.check_is_interior(1.0, y0)
        assert cone.ratio_test(1.0, np.zeros(m), -1.0) == np.inf or cone.ratio_test(1.0, np.zeros(m), -1.0) > 1e6
    finally:
        cone.destroy()


def test_many_constraints_known_answers():
    """the row count of BASELINE config 5 (m = 8000) at a block size that fits one GPU beside it (n = 128): same
    size-independent known answers as above (S = I at y0 => ASinv = b, M_ij = <A_i, A_j>), sampled over the whole
    row range, plus the S = 4 I scaling and a host-side residual of the solve"""
    from hdsdp_amd import api
    n, m = 128, 8000
    P = n * (n + 1) // 2
    y0 = _splitmix_u(np.uint64(2 * m * P) + np.arange(m, dtype=np.uint64))
    cone = api.SDPCone.synthetic(n, m)
    try:
        b = cone.traces()
        kkt = api.KKT(m, [cone])
        cone.set_start(0.0)
        assert cone.check_is_interior(1.0, y0)
        assert abs(cone.log_barrier(1.0)) < 1e-8
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        ex1 = kkt.export()
        assert np.max(np.abs(ex1["ASinv"] - b)) <= 1e-10 * np.max(np.abs(b))
        rows = [0, 1, 127, 128, 4095, 7871, 7999]
        mats = {i: _synth_matrix(n, i) for i in rows}
        M1 = kkt.M
        for i in rows:
            assert abs(np.trace(mats[i]) - b[i]) < 1e-10
            for j in rows:
                ref = float(np.sum(mats[i] * mats[j]))
                got = M1[min(i, j), max(i, j)]
                assert abs(got - ref) <= 1e-10 * abs(ref), (i, j, got, ref)
        d1 = np.diag(M1).copy()
        cone.set_start(-3.0)
        assert cone.check_is_interior(1.0, y0)
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        assert np.max(np.abs(np.diag(kkt.M) * 16.0 - d1)) <= 1e-10 * np.max(d1)
        assert abs(kkt.export()["TraceSinv"] - n / 4.0) < 1e-9
        kkt.factorize()
        x = kkt.solve(b)
        Mfull = np.triu(kkt.M) + np.triu(kkt.M, 1).T
        assert np.linalg.norm(Mfull @ x - b) <= 1e-9 * np.linalg.norm(b)
        kkt.destroy()
    finally:
        cone.destroy()


def test_cones_accumulate_into_one_schur_matrix():
    """HKKTBuildUp loops over the cones and every cone ADDS its contribution (hdsdp_schur.c:262-264): two blocks with
    the same row space -- here the same block twice, once on the sparse-gather path and once forced onto the GEMM
    path -- must give exactly twice the one-block matrix, vectors and scalars"""
    import os
    from hdsdp_amd import api
    g = load_golden("theta1_B")
    n, m = int(g["dims"][0]), int(g["dims"][1])
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    c1 = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    os.environ["HDSDP_MI355X_FORCE_GEMM"] = "1"
    try:
        c2 = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    finally:
        del os.environ["HDSDP_MI355X_FORCE_GEMM"]
    try:
        assert c1.path != c2.path
        for c in (c1, c2):
            c.set_start(Rd)
            assert c.check_is_interior(tau, y)
        kkt = api.KKT(m, [c1, c2])
        msk = lower_mask(m)
        kkt.build_up(api.KKT_TYPE_HOMOGENEOUS)
        ex = kkt.export()
        check_close(kkt.M[msk], 2.0 * g["M_hsd"][msk], "two cones: M")
        check_close(ex["ASinv"], 2.0 * g["ASinv_hsd"], "two cones: ASinv")
        check_close(ex["ASinvCSinv"], 2.0 * g["ASinvCSinv_hsd"], "two cones: ASinvCSinv")
        for got, ref in zip((ex["CSinv"], ex["CSinvCSinv"], ex["CSinvRdSinv"], ex["TraceSinv"]), g["hsd_scalars"]):
            check_close([got], [2.0 * ref], "two cones: scalar")
        kkt.build_up(api.KKT_TYPE_CORRECTOR)
        check_close(kkt.export()["ASinvRdSinv"], 2.0 * g["ASinvRdSinv_cor"], "two cones: corrector")
        kkt.destroy()
    finally:
        c1.destroy()
        c2.destroy()


@pytest.mark.parametrize("n,m", [(1, 1), (2, 3), (17, 5), (33, 70), (129, 10), (130, 257), (257, 9)])
def test_odd_shapes_against_the_oracle(n, m, monkeypatch):
    """ragged sizes (1 x 1 blocks, dimensions just above a tile boundary, more rows than packed entries) on the GEMM path,
    checked against the pinned oracle on the synthetic family: edge tiles, single-tile problems, padded batches"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    from hdsdp_amd import api
    beg, idx, val, b = oracle_py.synth_csc(n, m)
    blk = oracle_py.Block(n, m, beg, idx, val)
    monkeypatch.setenv("HDSDP_MI355X_FORCE_GEMM", "1")
    cone = api.SDPCone.from_csc(n, m, beg, idx, val)
    try:
        Rd, tau = -10.0 * n - 5.0, 0.9
        y = 0.05 * np.cos(np.arange(m) + 0.3)
        S = blk.assemble_S(tau, y, Rd)
        Lf, info = blk.factor(S)
        assert info == 0
        Sinv = blk.inverse(Lf)
        cone.set_start(Rd)
        assert cone.check_is_interior(tau, y)
        assert abs(cone.log_barrier(tau) - blk.logdet(Lf)) <= 1e-11 * max(1.0, abs(blk.logdet(Lf)))
        kkt = api.KKT(m, [cone])
        msk = lower_mask(m)
        for typ, key in ((api.KKT_TYPE_INFEASIBLE, 0), (api.KKT_TYPE_HOMOGENEOUS, 2)):
            ref = blk.kkt_build(Sinv, Rd, key)
            kkt.build_up(typ)
            ex = kkt.export()
            check_close(kkt.M[msk], ref["M"][msk], "M %dx%d" % (n, m))
            check_close(ex["ASinv"], ref["ASinv"], "ASinv")
            check_close(ex["ASinvRdSinv"], ref["ASinvRdSinv"], "ASinvRdSinv")
            if key == 2:
                check_close(ex["ASinvCSinv"], ref["ASinvCSinv"], "ASinvCSinv")
                for k in ("CSinv", "CSinvCSinv", "CSinvRdSinv", "TraceSinv"):
                    check_close([ex[k]], [ref[k]], k)
        ref = blk.kkt_build(Sinv, Rd, 1)
        kkt.build_up(api.KKT_TYPE_CORRECTOR)
        check_close(kkt.export()["ASinvRdSinv"], ref["ASinvRdSinv"], "corrector")
        if n > 1:
            dy = 3.0 * np.sin(0.7 * np.arange(m) + 0.1)
            got = cone.ratio_test(-0.02, dy, 0.5)
            want = blk.ratio_test(Lf, -0.02, dy, 0.5 * Rd)
            assert abs(got - want) <= RATIO_TOL * abs(want), (got, want)
        kkt.destroy()
    finally:
        cone.destroy()
        blk.close()


@pytest.mark.parametrize("inst,opt,y_start", [("theta1", 23.0, "first"), ("mcp100", 226.1574, "all")])
def test_dual_barrier_method_reaches_the_sdplib_optimum(inst, opt, y_start):
    """end to end through the C ABI only: a feasible-start dual barrier method (Newton step dy = M^-1 (b/mu - ASinv),
    the Phase-B direction of interface/hdsdp_algo.c:1334-1340, step length from the device ratio test) on the reference's
    own SDPLIB examples, read with the engine's SDPA reader; the dual objective must reach the published optimum
    (theta1 23, mcp100 226.1574; SURVEY 6 reproduces them with the reference driver).  gpp100 is left out: its primal
    has no interior (X . 11' = 0), the dual central path runs off to infinity in y_0, and the reference copes with that
    through its homogeneous embedding and the box on y, which are driver logic outside this engine."""
    import os
    from hdsdp_amd import api
    here = os.path.dirname(os.path.abspath(__file__))
    prob = api.read_sdpa(os.path.join(here, "golden", inst + ".dat-s"))
    n, m = prob["blocks"][0]["n"], prob["m"]
    blk = prob["blocks"][0]
    cone = api.SDPCone.from_csc(n, m, blk["beg"], blk["idx"], blk["val"])
    b = np.asarray(prob["b"], dtype=np.float64)
    try:
        kkt = api.KKT(m, [cone])
        cone.set_start(0.0)                       # no dual residual: S = C - sum y_i A_i
        y = np.zeros(m)
        if y_start == "first":
            y[0] = -100.0
        else:
            y[:] = -100.0
        assert cone.check_is_interior(1.0, y)
        mu = 1.0
        for outer in range(40):
            for inner in range(4):
                assert cone.check_is_interior(1.0, y)
                kkt.build_up(api.KKT_TYPE_INFEASIBLE)
                kkt.factorize()
                ex = kkt.export()
                dy = kkt.solve(b / mu - ex["ASinv"])
                step = cone.ratio_test(0.0, dy, 0.0)
                alpha = min(1.0, 0.9 * step)
                y = y + alpha * dy
                # Newton decrement small -> this mu is done
                if alpha == 1.0 and np.sqrt(abs(np.dot(dy, b / mu - ex["ASinv"]))) < 0.1:
                    break
            if n * mu < 1e-8 * opt:
                break
            mu *= 0.25
        assert cone.check_is_interior(1.0, y)
        dobj = float(np.dot(b, y))
        assert abs(abs(dobj) - opt) <= 2e-6 * opt, (inst, dobj, opt, mu)
        kkt.destroy()
    finally:
        cone.destroy()


def test_cpu_extra_cone_writes_through_the_host_mirror():
    """the driver's HKKTBuildUpExtraCone (hdsdp_algo.c:1087,:1735): a CPU cone -- here a stand-in for the y-box cone of
    interface/hdsdp_conic_bound.c:201-249 -- adds to diag(M) through kktDiag[] and to dASinvVec on the HOST fields after
    the device cones have been pulled back; HKKTFactorize must then factor the host matrix, not the stale device copy"""
    import ctypes as C
    from hdsdp_amd import api
    g = load_golden("syn64")
    n, m = int(g["dims"][0]), int(g["dims"][1])
    cone = api.SDPCone.synthetic(n, m)
    lib = api.load_library()

    class HostCone(C.Structure):   # hdsdp_cone, interface/def_hdsdp_conic.h:60-100: 2 ints, 2 pointers, 30 slots
        _fields_ = [("iCone", C.c_int), ("cone", C.c_int), ("usrData", C.c_void_p), ("coneData", C.c_void_p),
                    ("slots", C.c_void_p * 30)]
    add_diag = 0.125 + 0.01 * np.arange(m)
    add_asinv = np.cos(np.arange(m))
    calls = []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int)
    def build_schur(cone_data, icone, kkt_ptr, type_kkt):
        k = C.cast(kkt_ptr, C.POINTER(api.hdsdp_kkt)).contents
        for i in range(k.nRow):
            k.kktDiag[i][0] += add_diag[i]
            k.dASinvVec[i] += add_asinv[i]
        calls.append((icone, type_kkt))
        return 0
    extra = HostCone()
    extra.iCone = 7
    extra.slots[11] = C.cast(build_schur, C.c_void_p)   # coneBuildSchur is the 12th slot
    try:
        kkt = api.KKT(m, [cone])                          # host mirror on: the reference boundary
        cone.set_start(float(g["Rd"][0]))
        assert cone.check_is_interior(float(g["tau"][0]), y_of(g))
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        M0 = kkt.M.copy()
        a0 = kkt.export()["ASinv"].copy()
        assert lib.HKKTBuildUpExtraCone(kkt._k, C.cast(C.pointer(extra), C.c_void_p), api.KKT_TYPE_INFEASIBLE) == 0
        assert calls == [(7, api.KKT_TYPE_INFEASIBLE)]
        M1 = kkt.M
        assert np.allclose(np.diag(M1) - np.diag(M0), add_diag, rtol=0, atol=1e-15)
        assert np.allclose(kkt.export()["ASinv"] - a0, add_asinv, rtol=0, atol=1e-15)
        kkt.factorize()
        rhs = np.sin(np.arange(m) + 1.0)
        x = kkt.solve(rhs)
        Mfull = np.triu(M1) + np.triu(M1, 1).T
        assert np.linalg.norm(Mfull @ x - rhs) <= 1e-11 * np.linalg.norm(rhs)
        # and it is the modified matrix that was factored: the unmodified one gives a different answer
        M0full = np.triu(M0) + np.triu(M0, 1).T
        assert np.linalg.norm(M0full @ x - rhs) > 1e-6 * np.linalg.norm(rhs)
        kkt.destroy()
    finally:
        cone.destroy()


def test_regularize_follows_the_reference_rule():
    """HKKTRegularize (interface/hdsdp_schur.c:348-373): diag(M) += min(reg * min diag(M), 1e-5), nothing below 1e-14"""
    from hdsdp_amd import api
    g = load_golden("syn64")
    n, m = int(g["dims"][0]), int(g["dims"][1])
    cone = api.SDPCone.synthetic(n, m)
    try:
        kkt = api.KKT(m, [cone])
        cone.set_start(float(g["Rd"][0]))
        assert cone.check_is_interior(float(g["tau"][0]), y_of(g))
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        for reg in (1e-6, 1e3, 1e12):
            d0 = np.diag(kkt.M).copy()
            want = min(reg * d0.min(), 1e-5)
            if want < 1e-14:
                want = 0.0
            kkt.regularize(reg)
            assert np.array_equal(np.diag(kkt.M), d0 + want)
        # the device-resident flavour (host mirror off) applies the same rule to the device copy of M: seen through
        # the solve, against the mirrored operator regularised the same way
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        kkt.regularize(1e3)
        Mh = kkt.M
        A = np.triu(Mh) + np.triu(Mh, 1).T
        b = cone.traces()
        kdev = api.KKT(m, [cone], host_mirror=False)
        kdev.build_up(api.KKT_TYPE_INFEASIBLE)
        kdev.factorize()
        x_plain = kdev.solve(b)
        kdev.regularize(1e3)
        kdev.factorize()
        x_reg = kdev.solve(b)
        ref = np.linalg.solve(A, b)
        assert np.linalg.norm(x_reg - ref) <= 1e-10 * np.linalg.norm(ref)
        assert np.linalg.norm(x_plain - ref) > 1e-6 * np.linalg.norm(ref)   # the shift is visible in the solution
        kdev.destroy()
        kkt.destroy()
    finally:
        cone.destroy()


def test_tile_form_operator_factors_an_indefinite_matrix_like_the_reference():
    """arrow128's sparse Schur operator (tile form on the device) with its diagonal lowered -- through kktDiag, how the bound
    cone and HKKTRegularize write into it -- until FIVE eigenvalues are negative.  The reference's sparse operator is factored
    by an LDL' without pivoting (linalg/hdsdp_linsolver.c:596-626 over external/qdldl.c): HKKTFactorize and HKKTSolve go
    through (indef_codes of the golden: 0 0) and only the PSD check says no.  Round 3's tile form stopped at the first
    non-positive pivot.  Now: same solution as the compiled reference's, five negative pivots (inertia), and the operator is
    positive definite again after the diagonal is restored."""
    import os
    from hdsdp_amd import api
    g = load_golden("arrow128_A")
    assert list(g["indef_codes"]) == [0, 0, 0]                 # reference: factorize ok, solve ok, isPsd = 0
    nb, m = int(g["mb_dims"][0]), int(g["mb_dims"][1])
    Rd, tau, y, shift = float(g["Rd"][0]), float(g["tau"][0]), y_of(g), float(np.asarray(g["indef_shift"]).ravel()[0])
    prob = api.read_sdpa(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arrow128.dat-s"))
    cones = [api.SDPCone.from_csc(blk["n"], m, blk["beg"], blk["idx"], blk["val"], iCone=k) for k, blk in enumerate(prob["blocks"])]
    try:
        for c in cones:
            c.set_start(Rd)
            assert c.check_is_interior(tau, y)
        kkt = api.KKT(m, cones)
        assert kkt.is_sparse and kkt.tile_info() is not None
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        kkt.factorize()
        assert kkt.negative_pivots() == 0
        x0 = kkt.solve(g["b"])
        kkt.add_to_diag(-shift)
        A = np.triu(kkt.M) + np.triu(kkt.M, 1).T
        assert int(np.sum(np.linalg.eigvalsh(A) < 0)) == 5
        kkt.factorize()                                         # (raised in round 3: "not positive definite")
        assert kkt.negative_pivots() == 5
        x = kkt.solve(g["b"])
        assert np.linalg.norm(x - g["indef_sol"]) <= 1e-8 * np.linalg.norm(g["indef_sol"])
        assert np.linalg.norm(A @ x - g["b"]) <= 1e-10 * np.linalg.norm(g["b"])
        kkt.add_to_diag(shift)
        kkt.factorize()
        assert kkt.negative_pivots() == 0
        x1 = kkt.solve(g["b"])
        assert np.linalg.norm(x1 - x0) <= 1e-9 * np.linalg.norm(x0)
        kkt.destroy()
    finally:
        for c in cones:
            c.destroy()


@pytest.mark.parametrize("name,fname", [("truss1_A", "truss1.dat-s"), ("blocks3_A", "blocks3.dat-s"), ("chain16_A", "chain16.dat-s"),
                                        ("arrow128_A", "arrow128.dat-s")])
def test_multi_block_instance_against_reference(name, fname):
    """multi-block instances through the engine's own SDPA reader, all cones in one KKT object, against the reference's
    numbers for the same file (the reference run with one dense-SDP cone per block): truss1 (2 x 2 blocks and a 1 x 1),
    and blocks3 (21 / 34 / 9), where most constraints are zero on each block and the three blocks take the three
    different device paths; chain16 and arrow128 (oracle/make_arrow_sdpa.py: 128 small blocks sharing 32 linking constraints,
    m = 1056) come up with the SPARSE Schur operator in the reference, and the goldens are its aggregated CSC, entry for
    entry.  arrow128's pattern is dense in every envelope; the engine keeps it in tile form (csrc/bsparse.h): 17 of 45 tiles."""
    import os
    from hdsdp_amd import api
    g = load_golden(name)
    nb, m = int(g["mb_dims"][0]), int(g["mb_dims"][1])
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    prob = api.read_sdpa(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fname))
    assert prob["m"] == m and len(prob["blocks"]) == nb
    assert np.array_equal(prob["b"], g["b"])
    cones = []
    try:
        for k, blk in enumerate(prob["blocks"]):
            assert blk["n"] == int(g["mb_blkdims"][k])
            if "mb%d_beg" % k in g:      # (arrow128's golden does not carry the 128 per-block input CSCs)
                assert np.array_equal(blk["beg"], g["mb%d_beg" % k]) and np.array_equal(blk["idx"], g["mb%d_idx" % k])
                assert np.array_equal(blk["val"], g["mb%d_val" % k])
            cones.append(api.SDPCone.from_csc(blk["n"], m, blk["beg"], blk["idx"], blk["val"], iCone=k))
        if name == "blocks3_A":
            assert [c.path for c in cones] == [2, 0, 1]    # sparse gather, congruence + Gram, rank one
        ld = 0.0
        for c in cones:
            c.set_start(Rd)
            assert c.check_is_interior(tau, y)
            ld += c.log_barrier(tau)
        assert abs(ld - float(g["logdet"][0])) <= 1e-12 * abs(float(g["logdet"][0]))
        kkt = api.KKT(m, cones)
        msk = lower_mask(m)
        # chain16: sixteen blocks of five constraints each -- the operator must come up SPARSE like the reference's
        # (aggregated CSC pattern, hdsdp_schur.c:46-139), with the reference's pattern, entry for entry
        assert kkt.is_sparse == bool(int(g["kkt_sparse"][0]))
        if kkt.is_sparse:
            beg, idx, val = kkt.csc()
            assert np.array_equal(beg, g["kkt_beg"]) and np.array_equal(idx, g["kkt_idx"])
        if name == "arrow128_A":
            ti = kkt.tile_info()
            assert ti is not None, "the arrow pattern should be kept in tile form"
            tiles, dense_tiles, levels, nbytes = ti
            assert dense_tiles == 45 and tiles <= 20 and levels <= 3, ti
        kkt.build_up(api.KKT_TYPE_HOMOGENEOUS)
        ex = kkt.export()
        if kkt.is_sparse:
            check_close(kkt.csc()[2], g["M_hsd"], name + " M_hsd (CSC values)")
        check_close(kkt.M[msk], golden_schur_dense(g, "M_hsd")[msk], name + " M_hsd")
        check_close(ex["ASinv"], g["ASinv_hsd"], "truss1 ASinv")
        check_close(ex["ASinvRdSinv"], g["ASinvRdSinv_hsd"], "truss1 ASinvRdSinv")
        check_close(ex["ASinvCSinv"], g["ASinvCSinv_hsd"], "truss1 ASinvCSinv")
        check_close([ex["CSinv"], ex["CSinvCSinv"], ex["CSinvRdSinv"], ex["TraceSinv"]], g["hsd_scalars"], "truss1 scalars")
        kkt.build_up(api.KKT_TYPE_CORRECTOR)
        exc = kkt.export()
        check_close(exc["ASinv"], g["ASinv_cor"], name + " ASinv_cor")
        check_close(exc["ASinvRdSinv"], g["ASinvRdSinv_cor"], name + " ASinvRdSinv_cor")
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        check_close(kkt.M[msk], golden_schur_dense(g, "M_inf")[msk], name + " M_inf")
        kkt.factorize()
        x = kkt.solve(g["b"])
        assert np.linalg.norm(x - g["sol_b"]) <= 1e-8 * np.linalg.norm(g["sol_b"])
        if kkt.is_sparse:
            # what a CPU cone / the driver does to a sparse operator: through kktDiag into the CSC, then factor again
            kkt.add_to_diag(0.25)
            kkt.factorize()
            A = np.triu(kkt.M) + np.triu(kkt.M, 1).T
            x2 = kkt.solve(g["b"])
            assert np.linalg.norm(A @ x2 - g["b"]) <= 1e-11 * np.linalg.norm(g["b"])
            assert np.linalg.norm(x2 - x) > 1e-6 * np.linalg.norm(x)
            d0 = kkt.csc()[2][kkt.csc()[0][:-1]].copy()
            kkt.regularize(1e3)                                   # HKKTRegularize goes through kktDiag as well
            assert np.array_equal(kkt.csc()[2][kkt.csc()[0][:-1]], d0 + min(1e3 * d0.min(), 1e-5))
            # the device-resident flavour of the same operator (no host round trip of M)
            kdev = api.KKT(m, cones, host_mirror=False)
            assert kdev.is_sparse
            kdev.build_up(api.KKT_TYPE_INFEASIBLE)
            kdev.factorize()
            assert np.linalg.norm(kdev.solve(g["b"]) - g["sol_b"]) <= 1e-8 * np.linalg.norm(g["sol_b"])
            kdev.destroy()
        kkt.destroy()
    finally:
        for c in cones:
            c.destroy()


@pytest.mark.parametrize("name", ["mcp100_A", "mcp100_B", "gpp100_A", "gpp100_B"])
def test_fused_phase_a_pass_of_small_rank_one_blocks(name):
    """HMiKKTPhaseA (csrc/small.hip): BASELINE configs 2-3 in ONE launch -- S assembly, Cholesky + triangular inverse + S^-1
    in registers, the rank-one Schur build, Cholesky of M, three solves -- against the compiled reference's goldens for the
    same state (M, both vectors, log det S, tr S^-1), against LAPACK for the three solutions, and against the call-by-call
    path of the same library; afterwards the factor objects must serve later calls (HKKTSolve, the barrier, a ratio test)
    exactly as after the separate calls"""
    from hdsdp_amd import api
    g = load_golden(name)
    cone, n, m = _make_cone(name, g)
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    try:
        assert cone.path == 1
        kkt = api.KKT(m, [cone])
        assert kkt.phase_a_eligible()
        cone.set_start(Rd)
        b = np.asarray(g["b"], dtype=np.float64)
        ok, logdet, d1, d2, d3 = kkt.phase_a(tau, y, b)
        assert ok
        assert abs(logdet - float(g["logdet"][0])) <= 1e-12 * abs(float(g["logdet"][0]))
        msk = lower_mask(m)
        ex = kkt.export()
        check_close(kkt.M[msk], g["M_inf"][msk], name + " M")
        check_close(ex["ASinv"], g["ASinv_inf"], name + " ASinv")
        check_close(ex["ASinvRdSinv"], g["ASinvRdSinv_inf"], name + " ASinvRdSinv")
        check_close([ex["TraceSinv"]], g["TraceSinv_inf"], name + " TraceSinv")
        Mg = g["M_inf"]
        A = np.triu(Mg) + np.triu(Mg, 1).T
        for x, rhs in ((d1, b), (d2, g["ASinv_inf"]), (d3, g["ASinvRdSinv_inf"])):
            ref = np.linalg.solve(A, np.asarray(rhs, dtype=np.float64))
            assert np.linalg.norm(x - ref) <= 1e-8 * np.linalg.norm(ref)
        # the objects behind the fused call are in the state the separate calls leave them in
        x4 = kkt.solve(b)
        assert np.linalg.norm(x4 - d1) <= 1e-13 * np.linalg.norm(d1)
        assert abs(cone.log_barrier_of(api.BUFFER_DUALVAR) - logdet) <= 1e-13 * abs(logdet)
        if "rt_step1" in g:
            par = g["rt_par1"]
            step = cone.ratio_test(float(par[0]), g["rt_dy1"], float(par[1]))
            assert abs(step - float(g["rt_step1"][0])) <= RATIO_TOL * abs(float(g["rt_step1"][0]))
        # ... and the call-by-call path gives the same numbers
        assert cone.check_is_interior(tau, y)
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        ex2 = kkt.export()
        assert np.max(np.abs(ex2["ASinv"] - ex["ASinv"])) <= 1e-12 * np.max(np.abs(ex["ASinv"]))
        kkt.factorize()
        y2 = kkt.solve(ex2["ASinv"])
        assert np.linalg.norm(y2 - d2) <= 1e-10 * np.linalg.norm(d2)
        # a state outside the cone is a value, not an error
        cone.set_start(1e6)
        ok, _, _, _, _ = kkt.phase_a(tau, y, b)
        assert ok is False
        kkt.destroy()
    finally:
        cone.destroy()


def _rank_one_block(n, m, rng, ndense=1, nnz=(1, 2, 3)):
    """CSC (column 0 = C) of a block whose constraints are all sigma a a' -- sparse factors with a few entries, `ndense` dense"""
    P = n * (n + 1) // 2

    def pk(i, j):
        return (2 * n - j - 1) * j // 2 + i

    cols = []
    Cm = np.diag(3.0 + rng.uniform(0, 1, n))
    Cm += 0.05 * np.tril(rng.uniform(-1, 1, (n, n)), -1)
    cols.append({pk(i, j): Cm[i, j] for j in range(n) for i in range(j, n) if Cm[i, j] != 0.0})
    for q in range(m):
        a = np.zeros(n)
        if q < ndense:
            a[:] = rng.uniform(-1, 1, n)
        else:
            k = int(rng.choice(nnz))
            idx = rng.choice(n, min(k, n), replace=False)
            a[idx] = rng.uniform(0.3, 1.0, len(idx)) * rng.choice([-1.0, 1.0], len(idx))
        sig = float(rng.choice([-1.0, 1.0])) * rng.uniform(0.5, 1.5)
        nz = np.nonzero(a)[0]
        cols.append({pk(max(i, j), min(i, j)): sig * a[i] * a[j] for i in nz for j in nz if i >= j})
    beg, idx, val = [0], [], []
    for c in cols:
        for k in sorted(c):
            idx.append(k); val.append(c[k])
        beg.append(len(idx))
    return np.array(beg, dtype=np.int32), np.array(idx, dtype=np.int32), np.array(val)


@pytest.mark.parametrize("n,m,ndense", [(128, 128, 4), (100, 128, 0), (37, 5, 1), (2, 3, 0), (1, 1, 0), (64, 90, 2)])
def test_fused_phase_a_on_generated_rank_one_blocks(n, m, ndense):
    """the fused pass at its limits (n = m = 128, four dense factors), on ragged and tiny shapes, signs of both kinds, factors
    with one to three entries: against the plain-C oracle's column-by-column build (the reference's M2 formulas) and LAPACK"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    from hdsdp_amd import api
    rng = np.random.default_rng(1000 * n + m)
    beg, idx, val = _rank_one_block(n, m, rng, ndense=ndense)
    cone = api.SDPCone.from_csc(n, m, beg, idx, val)
    blk = oracle_py.Block(n, m, beg, idx, val)
    try:
        if cone.path != 1:
            pytest.skip("not classified rank one")
        kkt = api.KKT(m, [cone])
        assert kkt.phase_a_eligible()
        Rd, tau = -2.0, 0.9
        y = 0.05 * np.cos(np.arange(m) + 0.3)
        cone.set_start(Rd)
        b = np.sin(np.arange(m) + 1.0)
        ok, logdet, d1, d2, d3 = kkt.phase_a(tau, y, b)
        S = blk.assemble_S(tau, y, Rd)
        Lf, info = blk.factor(S)
        assert ok == (info == 0) and ok
        assert abs(logdet - blk.logdet(Lf)) <= 1e-12 * max(1.0, abs(blk.logdet(Lf)))
        ref = blk.kkt_build(blk.inverse(Lf), Rd, 0)
        msk = lower_mask(m)
        ex = kkt.export()
        check_close(kkt.M[msk], ref["M"][msk], "M")
        check_close(ex["ASinv"], ref["ASinv"], "ASinv")
        check_close(ex["ASinvRdSinv"], ref["ASinvRdSinv"], "ASinvRdSinv")
        check_close([ex["TraceSinv"]], [ref["TraceSinv"]], "TraceSinv")
        A = np.triu(ref["M"]) + np.triu(ref["M"], 1).T
        if np.linalg.cond(A) < 1e10:
            for x, rhs in ((d1, b), (d2, ref["ASinv"]), (d3, ref["ASinvRdSinv"])):
                xr = np.linalg.solve(A, rhs)
                assert np.linalg.norm(x - xr) <= 1e-8 * max(np.linalg.norm(xr), 1e-300) * max(1.0, np.linalg.cond(A) * 1e-6)
        kkt.destroy()
    finally:
        cone.destroy()
        blk.close()


def test_fused_phase_a_refuses_what_it_cannot_do():
    from hdsdp_amd import api
    cone = api.SDPCone.synthetic(64, 40)          # dense block: congruence path
    try:
        kkt = api.KKT(40, [cone])
        assert not kkt.phase_a_eligible()
        with pytest.raises(api.HDSDPError):
            kkt.phase_a(1.0, np.zeros(40), np.ones(40))
        kkt.destroy()
    finally:
        cone.destroy()


def _block_with_rows(n, m, keep):
    """a block of the synthetic family on which only the constraints in `keep` have data (CSC, column 0 = C)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    beg0, idx0, val0, _ = oracle_py.synth_csc(n, m)
    beg, idx, val = [0], [], []
    for col in range(m + 1):
        lo, hi = int(beg0[col]), int(beg0[col + 1])
        if col == 0 or (col - 1) in keep:
            idx += [int(v) for v in idx0[lo:hi]]; val += [float(v) for v in val0[lo:hi]]
        beg.append(len(idx))
    return np.array(beg, dtype=np.int32), np.array(idx, dtype=np.int32), np.array(val)


@pytest.mark.parametrize("nblocks,per,expect_sparse", [(12, 7, False), (12, 3, True)])
def test_sparse_operator_decision_and_its_dense_way_back(nblocks, per, expect_sparse):
    """HKKTInit's two-stage rule (interface/hdsdp_schur.c:229-238, :104-108) on blocks that are each sparse candidates
    (at most 0.3 m constraints with data): twelve blocks of three scattered constraints aggregate to a sparse pattern;
    twelve blocks of seven scattered constraints aggregate past 0.3 m^2, so the operator must fall back to the dense matrix
    in the middle of the pattern walk -- and be right either way: M, the vectors and a solve against the oracle's
    block-by-block sum"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    from hdsdp_amd import api
    m = 24
    rng = np.random.default_rng(100 + nblocks)
    dims = [12 + 3 * (b % 4) for b in range(nblocks)]
    # block b: constraints 2b, 2b+1 (so every constraint has data somewhere) and per-2 scattered ones
    keeps = []
    for b in range(nblocks):
        base = {(2 * b) % m, (2 * b + 1) % m}
        rest = [int(v) for v in rng.permutation(m) if int(v) not in base][:per - 2]
        keeps.append(sorted(base | set(rest)))
    Rd, tau = -40.0, 1.0
    y = 0.05 * np.cos(np.arange(m) + 0.3)
    cones, Mref, aref = [], np.zeros((m, m)), np.zeros(m)
    try:
        for b, (n, keep) in enumerate(zip(dims, keeps)):
            beg, idx, val = _block_with_rows(n, m, keep)
            blk = oracle_py.Block(n, m, beg, idx, val)
            Lf, info = blk.factor(blk.assemble_S(tau, y, Rd))
            assert info == 0
            ref = blk.kkt_build(blk.inverse(Lf), Rd, 0)
            Mref += ref["M"]; aref += ref["ASinv"]
            blk.close()
            c = api.SDPCone.from_csc(n, m, beg, idx, val, iCone=b)
            c.set_start(Rd)
            assert c.check_is_interior(tau, y)
            cones.append(c)
        kkt = api.KKT(m, cones)
        assert kkt.is_sparse == expect_sparse
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        msk = lower_mask(m)
        check_close(kkt.M[msk], Mref[msk], "M")
        check_close(kkt.export()["ASinv"], aref, "ASinv")
        kkt.factorize()
        rhs = np.sin(np.arange(m) + 1.0)
        A = np.triu(Mref) + np.triu(Mref, 1).T
        x = kkt.solve(rhs)
        assert np.linalg.norm(A @ x - rhs) <= 1e-10 * np.linalg.norm(rhs)
        kkt.destroy()
    finally:
        for c in cones:
            c.destroy()


@pytest.mark.parametrize("path", [0, 1, 2])
@pytest.mark.parametrize("keep", [(), (4,), (0, 3, 4, 11, 29)])
def test_constraints_that_are_zero_on_a_block(path, keep, monkeypatch):
    """a block on which most constraints vanish (what every block of a many-block SDP looks like; the reference's sparse
    SDP cone loops over the non-zero ones only, interface/hdsdp_conic_sdp.c:1814-1886): the engine leaves the zero
    rows out of its device data, and their rows of M, ASinv, ... must come out exactly zero -- on every device path,
    also when NO constraint touches the block (then only the objective's HSD scalars are non-zero)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    from hdsdp_amd import api
    n, m = 40, 30
    beg0, idx0, val0, _ = oracle_py.synth_csc(n, m)
    beg, idx, val = [0], [], []
    for col in range(m + 1):                      # column 0 = C, column i = A_i
        lo, hi = int(beg0[col]), int(beg0[col + 1])
        if col == 0 or (col - 1) in keep:
            if path != 0 and col > 0:             # rank-one / sparse paths need sparse data: keep 3 diagonal-ish entries
                sel = [k for k in range(lo, hi) if idx0[k] in (0, n, 2 * n - 1)]
                if path == 1:
                    sel = sel[:1]                 # one diagonal entry = rank one
                idx += [int(idx0[k]) for k in sel]; val += [float(val0[k]) for k in sel]
            else:
                idx += [int(v) for v in idx0[lo:hi]]; val += [float(v) for v in val0[lo:hi]]
        beg.append(len(idx))
    beg, idx, val = np.array(beg, dtype=np.int32), np.array(idx, dtype=np.int32), np.array(val)
    blk = oracle_py.Block(n, m, beg, idx, val)
    monkeypatch.setenv("HDSDP_MI355X_FORCE_PATH", str(path))
    cone = api.SDPCone.from_csc(n, m, beg, idx, val)
    kkt = api.KKT(m, [cone])
    try:
        Rd, tau = -10.0 * n, 0.9
        y = 0.05 * np.cos(np.arange(m) + 0.3)
        S = blk.assemble_S(tau, y, Rd)
        Lf, info = blk.factor(S)
        assert info == 0
        Sinv = blk.inverse(Lf)
        cone.set_start(Rd)
        assert cone.check_is_interior(tau, y)
        check_close(cone.dual_matrix()[lower_mask(n)], S[lower_mask(n)], "S")
        msk = lower_mask(m)
        zero = np.array([i not in keep for i in range(m)])
        for typ, key in ((api.KKT_TYPE_INFEASIBLE, 0), (api.KKT_TYPE_HOMOGENEOUS, 2), (api.KKT_TYPE_CORRECTOR, 1)):
            ref = blk.kkt_build(Sinv, Rd, key)
            kkt.build_up(typ)
            ex = kkt.export()
            if key != 1:
                M = kkt.M
                check_close(M[msk], ref["M"][msk], "M")
                assert not M[zero, :].any() and not M[:, zero].any()
            check_close(ex["ASinv"], ref["ASinv"], "ASinv")
            check_close(ex["ASinvRdSinv"], ref["ASinvRdSinv"], "ASinvRdSinv")
            assert not ex["ASinv"][zero].any() and not ex["ASinvRdSinv"][zero].any()
            if key == 2:
                check_close(ex["ASinvCSinv"], ref["ASinvCSinv"], "ASinvCSinv")
                for k in ("CSinv", "CSinvCSinv", "CSinvRdSinv", "TraceSinv"):
                    check_close([ex[k]], [ref[k]], k)
    finally:
        kkt.destroy()
        cone.destroy()
        blk.close()


def test_cpu_cone_inside_the_operator_accumulates_with_the_engine_cones():
    """HKKTBuildUp lets EVERY cone of HKKT->cones[] accumulate (interface/hdsdp_schur.c:256-268).  A cone that is not
    the engine's -- the reference's own CPU cones in a partly converted solver; here a stand-in that adds a known
    symmetric matrix and vector to the host fields -- must end up summed with the device-built part, whatever its
    position in the array; with only such cones the operator is still a working Schur system (the engine then only
    factors and solves).  This is what the reference's unmodified driver relies on (tests/test_gpu_reference_driver.py)."""
    import ctypes as C
    from hdsdp_amd import api
    g = load_golden("syn64")
    n, m = int(g["dims"][0]), int(g["dims"][1])
    lib = api.load_library()

    class HostCone(C.Structure):   # hdsdp_cone, interface/def_hdsdp_conic.h:60-100: 2 ints, 2 pointers, 30 slots
        _fields_ = [("iCone", C.c_int), ("cone", C.c_int), ("usrData", C.c_void_p), ("coneData", C.c_void_p),
                    ("slots", C.c_void_p * 30)]
    rng = np.random.default_rng(5)
    G = rng.uniform(-1, 1, (m, m))
    P = G @ G.T / m + np.eye(m)                       # what the CPU cone adds to M (lower triangle, column-major)
    add_asinv = np.cos(np.arange(m))

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int)
    def build_schur(cone_data, icone, kkt_ptr, type_kkt):
        k = C.cast(kkt_ptr, C.POINTER(api.hdsdp_kkt)).contents
        Mh = np.ctypeslib.as_array(k.kktMatElem, shape=(m, m))       # C-order view: Mh[j, i] = element (row i, col j)
        if type_kkt != api.KKT_TYPE_CORRECTOR:
            Mh += np.triu(P)
        for i in range(k.nRow):
            k.dASinvVec[i] += add_asinv[i]
        return 0

    @C.CFUNCTYPE(C.c_int, C.c_void_p)
    def get_dim(cone_data):
        return 3

    @C.CFUNCTYPE(C.c_int64, C.c_void_p)
    def get_nnz(cone_data):
        return m * m
    host = HostCone()
    host.iCone = 1
    host.slots[7] = C.cast(get_nnz, C.c_void_p)       # coneGetSymNnz
    host.slots[8] = C.cast(get_dim, C.c_void_p)       # coneGetDim
    host.slots[11] = C.cast(build_schur, C.c_void_p)  # coneBuildSchur
    host_ptr = C.cast(C.pointer(host), C.c_void_p).value
    cone = api.SDPCone.synthetic(n, m)
    msk = lower_mask(m)
    try:
        cone.set_start(float(g["Rd"][0]))
        assert cone.check_is_interior(float(g["tau"][0]), y_of(g))
        ref = api.KKT(m, [cone])
        ref.build_up(api.KKT_TYPE_INFEASIBLE)
        M0, a0 = ref.M.copy(), ref.export()["ASinv"].copy()
        ref.destroy()
        for order in ("engine first", "host first", "host only"):
            k = C.POINTER(api.hdsdp_kkt)()
            assert lib.HKKTCreate(C.byref(k)) == 0
            ptrs = {"engine first": [cone._h, host_ptr], "host first": [host_ptr, cone._h], "host only": [host_ptr]}[order]
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            assert lib.HKKTInit(k, m, len(ptrs), arr) == 0
            lib.HMiKKTSetHostMirror(k, 0)             # must be refused: a host cone is in the operator
            assert lib.HKKTBuildUp(k, api.KKT_TYPE_INFEASIBLE) == 0
            Mh = np.ctypeslib.as_array(k.contents.kktMatElem, shape=(m, m)).copy()
            asinv = np.ctypeslib.as_array(k.contents.dASinvVec, shape=(m,)).copy()
            want_M = np.triu(P) + (M0 * np.triu(np.ones((m, m))) if order != "host only" else 0.0)
            want_a = add_asinv + (a0 if order != "host only" else 0.0)
            assert np.allclose(Mh[msk], want_M[msk], rtol=1e-14, atol=1e-300), order
            assert np.allclose(asinv, want_a, rtol=1e-14, atol=1e-300), order
            assert lib.HKKTFactorize(k) == 0
            rhs = np.sin(np.arange(m) + 1.0)
            x = np.zeros(m)
            assert lib.HKKTSolve(k, api._dptr(rhs), api._dptr(x)) == 0
            full = np.triu(want_M) + np.triu(want_M, 1).T
            assert np.linalg.norm(full @ x - rhs) <= 1e-11 * np.linalg.norm(rhs), order
            lib.HKKTDestroy(C.byref(k))
    finally:
        cone.destroy()


@pytest.mark.parametrize("name", ["theta1_B", "mix40_B"])
def test_primal_xsx_direction(name):
    """the cone's coneBuildPrimalDirection slot (sdpDenseConeBuildPrimalXSXDirection, interface/hdsdp_conic_sdp.c:2021-2040
    -> fds_trimultiply, linalg/dense_opts.c:102-132): XSX += X^T D X, accumulated into the caller's buffer, with D the
    dual matrix or the dual step of the last ratio test; checked against the same product in numpy on the oracle's S"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    from hdsdp_amd import api
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    blk = oracle_py.Block(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    cone = api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
    try:
        cone.set_start(Rd)
        assert cone.check_is_interior(tau, y)
        def full(T):    # the oracle's matrices are column-major with the lower triangle valid = numpy's upper triangle
            T = np.asarray(T)
            return np.triu(T) + np.triu(T, 1).T
        S = full(blk.assemble_S(tau, y, Rd))
        rng = np.random.default_rng(n)
        G = rng.uniform(-1, 1, (n, n))
        X = G + 0.3 * G.T                                        # not symmetric on purpose: X^T D X, not X D X
        out = np.full((n, n), 0.25)
        cone.build_primal_xsx(X.T.copy(), out, dual_matrix=True)  # column-major X == C-order X^T
        want = X.T @ S @ X
        assert np.allclose(out.T - 0.25, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
        # the dual step of the last ratio test: dS = dtau * C - sum dy_i A_i
        dy = 3.0 * np.sin(0.7 * np.arange(m) + 0.1)
        cone.ratio_test(0.0, dy, 0.0)
        D = full(blk.assemble_S(0.0, dy, 0.0))
        out2 = np.zeros((n, n))
        cone.build_primal_xsx(X.T.copy(), out2, dual_matrix=False)
        want2 = X.T @ D @ X
        assert np.allclose(out2.T, want2, rtol=1e-12, atol=1e-12 * np.abs(want2).max())
    finally:
        cone.destroy()
        blk.close()


@pytest.mark.parametrize("name", ["theta1_B", "mix40_A", "syn64"])
def test_primal_build_with_an_indefinite_registered_matrix(name):
    """KKT_TYPE_PRIMAL when the registered matrix is NOT positive definite -- the primal refinement does hand such iterates
    over (seen with the reference's driver on the synthetic family at n = 30, m = 100), and the reference's formulas
    M_ij = tr(A_i X A_j X) do not care.  The engine then has no triangular factor of X to lean on and falls back to the
    row-by-row product X A_i X; checked against the pinned oracle run with X in S^-1's place (its typeKKT = 3 branch,
    pinned for positive definite X by the M_pri goldens)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    from hdsdp_amd import api
    g = load_golden(name)
    n, m = int(g["dims"][0]), int(g["dims"][1])
    if "csc_beg" in g:
        beg, idx, val = g["csc_beg"], g["csc_idx"], g["csc_val"]
    else:
        beg, idx, val, _ = oracle_py.synth_csc(n, m)
    blk = oracle_py.Block(n, m, beg, idx, val)
    cone = api.SDPCone.from_csc(n, m, beg, idx, val)
    kkt = api.KKT(m, [cone])
    try:
        Rd = float(g["Rd"][0])
        cone.set_start(Rd)
        assert cone.check_is_interior(float(g["tau"][0]), y_of(g))
        rng = np.random.default_rng(n + m)
        G = rng.uniform(-1, 1, (n, n))
        X = 0.5 * (G + G.T)                       # symmetric, eigenvalues of both signs
        w = np.linalg.eigvalsh(X)
        assert w[0] < -0.1 and w[-1] > 0.1
        ref = blk.kkt_build(X, Rd, 3)
        kkt.register_psdp([X])
        kkt.build_up(api.KKT_TYPE_PRIMAL)
        ex = kkt.export()
        msk = lower_mask(m)
        check_close(kkt.M[msk], ref["M"][msk], name + " M, indefinite X")
        check_close(ex["ASinv"], ref["ASinv"], "ASinv")
        check_close(ex["ASinvRdSinv"], ref["ASinvRdSinv"], "ASinvRdSinv")
        check_close([ex["TraceSinv"]], [ref["TraceSinv"]], "TraceSinv")
        # and a positive definite matrix still takes the congruence path and gives the oracle's numbers too
        P = X @ X + np.eye(n)
        ref = blk.kkt_build(P, Rd, 3)
        kkt.register_psdp([P])
        kkt.build_up(api.KKT_TYPE_PRIMAL)
        check_close(kkt.M[msk], ref["M"][msk], name + " M, definite X")
    finally:
        kkt.destroy()
        cone.destroy()
        blk.close()


@pytest.mark.parametrize("scrambled", [False, True], ids=["band-in-the-drivers-order", "band-after-reordering"])
def test_sparse_operator_with_a_banded_pattern_over_several_blocks_of_M(scrambled):
    """a chain of fifty small blocks, block b holding constraints 8b .. 8b+15 (m = 408): the operator comes up sparse, its
    pattern is a band, and the factorisation of the 512 x 512 device matrix runs on the pattern's block envelope
    (HdmChol::set_envelope: block row 3 does not reach block column 0 ...).  Scrambled: the same chain with the constraints
    renumbered at random -- the pattern is a band only after a reverse Cuthill-McKee reordering, which HKKTInit finds and
    keeps inside the factor object (P M P' is factored, right-hand sides and solutions are permuted).  M against the oracle's
    block-by-block sum, the solve against LAPACK on that sum, and twice -- the second factorisation replays the captured
    launch chain"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    from hdsdp_amd import api
    nblocks, m = 50, 408
    Rd, tau = -30.0, 1.0
    y = 0.02 * np.cos(np.arange(m) + 0.3)
    cones, Mref = [], np.zeros((m, m))
    renum = np.random.default_rng(7).permutation(m) if scrambled else np.arange(m)
    try:
        for b in range(nblocks):
            n = 10 + (b % 3)
            keep = sorted(int(renum[k]) for k in range(8 * b, 8 * b + 16))
            beg, idx, val = _block_with_rows(n, m, keep)
            blk = oracle_py.Block(n, m, beg, idx, val)
            Lf, info = blk.factor(blk.assemble_S(tau, y, Rd))
            assert info == 0
            Mref += blk.kkt_build(blk.inverse(Lf), Rd, 0)["M"]
            blk.close()
            c = api.SDPCone.from_csc(n, m, beg, idx, val, iCone=b)
            c.set_start(Rd)
            assert c.check_is_interior(tau, y)
            cones.append(c)
        kkt = api.KKT(m, cones)
        assert kkt.is_sparse
        A = np.triu(Mref) + np.triu(Mref, 1).T
        permuted, fraction = kkt.envelope_info()
        if os.environ.get("HDSDP_MI355X_KKT_ENVELOPE", "1") != "0" and os.environ.get("HDSDP_MI355X_KKT_RCM", "1") != "0":
            assert permuted == scrambled and fraction <= 0.8, (permuted, fraction)
        if not scrambled:
            assert not A[300:, :128].any() and A[127, 120] != 0.0      # a band: the far corner is structurally empty
        else:
            assert A[300:, :128].any()                                   # no band in this numbering
        for rep in range(2):
            kkt.build_up(api.KKT_TYPE_INFEASIBLE)
            msk = lower_mask(m)
            check_close(kkt.M[msk], Mref[msk], "M")
            kkt.factorize()
            rhs = np.sin(np.arange(m) + 1.0 + rep)
            x = kkt.solve(rhs)
            assert np.linalg.norm(x - np.linalg.solve(A, rhs)) <= 1e-10 * np.linalg.cond(A) * np.linalg.norm(rhs)
        kkt.destroy()
    finally:
        for c in cones:
            c.destroy()
