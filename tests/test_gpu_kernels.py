"""Kernel-level GPU tests through the C ABI: fp64 MFMA GEMM variants, blocked Cholesky, solves."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _colmajor(a):
    """numpy (r, c) matrix -> buffer holding it column-major"""
    return np.ascontiguousarray(a.T)


@pytest.mark.parametrize("akm,bkm", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 48), (264, 392, 512), (16, 8, 16)])
def test_gemm_nt_layouts(akm, bkm, M, N, K):
    """C = alpha A B^T + beta C for every operand storage; A = I-style asymmetry check is implicit in
    random non-symmetric operands (a swapped C layout cannot pass)."""
    import torch
    from hdsdp_amd import api
    lib = api.load_library()
    rng = np.random.default_rng(M * 1000 + N + K + akm * 7 + bkm * 13)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((N, K))
    C0 = rng.standard_normal((M, N))
    # M-major: element (i,k) at i + k*ld -> column-major A ; K-major: element (i,k) at i*ld + k -> row-major A
    dA = _dev(A if akm else _colmajor(A))
    dB = _dev(B if bkm else _colmajor(B))
    dC = _dev(_colmajor(C0))
    lda = K if akm else M
    ldb = K if bkm else N
    rc = lib.HMiGemmNT(dA.data_ptr(), lda, akm, dB.data_ptr(), ldb, bkm, dC.data_ptr(), M, M, N, K, 1.5, -0.5, 0, 0)
    assert rc == 0
    got = dC.cpu().numpy().T
    ref = 1.5 * A @ B.T - 0.5 * C0
    assert np.max(np.abs(got - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))) * K


def test_gemm_triangular_k_limits():
    """K loop cut by the row tile (lower-triangular A) and by the column tile (lower-triangular B)"""
    import torch
    from hdsdp_amd import api
    lib = api.load_library()
    rng = np.random.default_rng(5)
    n = 400
    L = np.tril(rng.standard_normal((n, n)))
    X = rng.standard_normal((n, n))
    dL, dX = _dev(_colmajor(L)), _dev(_colmajor(X))
    dC = _dev(np.zeros((n, n)))
    assert lib.HMiGemmNT(dL.data_ptr(), n, 0, dX.data_ptr(), n, 0, dC.data_ptr(), n, n, n, n, 1.0, 0.0, 1, 0) == 0
    assert np.allclose(dC.cpu().numpy().T, L @ X.T, rtol=0, atol=1e-10)
    dC.zero_()
    assert lib.HMiGemmNT(dX.data_ptr(), n, 0, dL.data_ptr(), n, 0, dC.data_ptr(), n, n, n, n, 1.0, 0.0, 2, 1) == 0
    got = dC.cpu().numpy().T
    ref = X @ L.T
    msk = np.tril(np.ones((n, n), dtype=bool))
    assert np.allclose(got[msk], ref[msk], rtol=0, atol=1e-10)
    assert np.all(got[~msk] == 0.0)  # lower_only leaves the strict upper part untouched


@pytest.mark.parametrize("n", [50, 128, 300, 400, 517, 1000])
def test_dense_direct_linsys(n):
    """HFpLinsys* dense-direct surface vs LAPACK semantics (linalg/hdsdp_linsolver.c:1082-1260); 1, 2, 3, 4, 5 and 8
    diagonal blocks: the triangular inverse runs by recursive doubling for 2, 4, 8 and by the block-column sweep otherwise"""
    from hdsdp_amd import api
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n))
    S = G @ G.T + n * np.eye(n)
    ls = api.LinSys(n)
    try:
        lower_only = np.triu(S)  # buffer is column-major: C-order upper == column-major lower
        assert ls.psd_check(lower_only)
        Lref = np.linalg.cholesky(S)
        assert np.allclose(ls.get_diag(), np.diag(Lref), rtol=1e-12)
        rhs = rng.standard_normal((3, n))
        x = ls.solve(rhs)
        assert np.allclose(S @ x.T, rhs.T, atol=1e-9)
        f = ls.fsolve(rhs[0])
        assert np.allclose(Lref @ f, rhs[0], atol=1e-10)
        bsol = ls.bsolve(rhs[1])
        assert np.allclose(Lref.T @ bsol, rhs[1], atol=1e-10)
        Sinv = ls.invert()
        assert np.allclose(Sinv, np.linalg.inv(S), rtol=1e-9, atol=1e-12)
        assert np.array_equal(Sinv, Sinv.T) or np.allclose(Sinv, Sinv.T, atol=1e-15)
        # "not PSD" is a value, not an error (hdsdp_linsolver.c:1133-1140)
        bad = S.copy()
        bad[n // 2, n // 2] = -1.0
        assert ls.psd_check(np.triu(bad)) is False
        with pytest.raises(api.HDSDPError):
            ls.numeric(np.triu(bad))
    finally:
        ls.destroy()


def test_mfma_probe_reports_a_rate():
    from hdsdp_amd import api
    tf = api.load_library().HMiMfmaPeakProbe(20000)
    print("fp64 MFMA register-loop rate: %.1f TFLOP/s" % tf)
    assert tf > 10.0


@pytest.mark.parametrize("name", ["indef96", "indef150"])
def test_schur_system_indefinite_fallback_matches_reference(name):
    """HFpLinsysNumeric / HFpLinsysSolve on the Schur system object (DENSE_ITERATIVE) with a matrix that is not
    positive definite: the object switches to the symmetric-indefinite solver and stays switched
    (linalg/hdsdp_linsolver.c:1827-1857, 2029-2110); values against the reference's own LDL^T solves.
    Tolerance 1e-9 relative (the reference's bar for Schur solves is 1e-8)."""
    import os
    from hdsdp_amd import api
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz")))
    M, b = g["indef_M"], g["indef_b"]
    m = M.shape[0]
    ls = api.LinSys(m, api.HDSDP_LINSYS_DENSE_ITERATIVE)
    try:
        assert ls.lin_type == api.HDSDP_LINSYS_DENSE_ITERATIVE
        ls.numeric(M)
        assert ls.lin_type == int(g["indef_codes"][2]) == api.HDSDP_LINSYS_DENSE_INDEFINITE
        x1 = ls.solve(b)
        assert np.linalg.norm(x1 - g["indef_x1"]) <= 1e-9 * np.linalg.norm(g["indef_x1"])
        M2 = M.copy()
        i = np.arange(m)
        M2[i, i] += 0.125
        ls.numeric(M2)
        assert ls.lin_type == int(g["indef_codes"][5])
        x2 = ls.solve(b)
        assert np.linalg.norm(x2 - g["indef_x2"]) <= 1e-9 * np.linalg.norm(g["indef_x2"])
        # in-place solve and several right-hand sides
        rhs = np.stack([b, 2.0 * b, b[::-1].copy()])
        xs = ls.solve(rhs)
        assert np.linalg.norm(xs[1] - 2.0 * x2) <= 1e-9 * np.linalg.norm(xs[1])
        A2 = np.triu(M2) + np.triu(M2, 1).T
        assert np.linalg.norm(A2 @ xs[2] - rhs[2]) <= 1e-10 * np.linalg.norm(rhs[2]) * np.linalg.cond(A2)
        # the pivoted object is no PSD oracle and has no half solves (hdsdp_linsolver.c:1729-1797)
        with pytest.raises(api.HDSDPError):
            ls.get_diag()
    finally:
        ls.destroy()
    # the dual-matrix object (DENSE_DIRECT) has no such way out: not positive definite is a failure there
    ld = api.LinSys(m, api.HDSDP_LINSYS_DENSE_DIRECT)
    try:
        with pytest.raises(api.HDSDPError):
            ld.numeric(M)
        assert ld.lin_type == api.HDSDP_LINSYS_DENSE_DIRECT
    finally:
        ld.destroy()


@pytest.mark.parametrize("m", [1, 2, 33, 129, 500, 2000])
def test_schur_system_indefinite_fallback_sizes(m):
    """size-independent property at sizes the goldens do not cover (one panel + identity padding, several panels, the
    bench size): M x = b to rounding for a symmetric indefinite M, against numpy's LAPACK solve; a singular matrix
    fails like dsytrf's info > 0"""
    from hdsdp_amd import api
    rng = np.random.default_rng(m)
    G = rng.uniform(-1, 1, (m, m))
    A = 0.5 * (G + G.T)
    A[0, 0] = -abs(A[0, 0]) - 0.1          # never positive definite, also at m = 1
    b = rng.uniform(-1, 1, m)
    ls = api.LinSys(m, api.HDSDP_LINSYS_DENSE_ITERATIVE)
    try:
        ls.numeric(np.triu(A))
        assert ls.lin_type == api.HDSDP_LINSYS_DENSE_INDEFINITE
        x = ls.solve(b)
        ref = np.linalg.solve(A, b)
        cond = np.linalg.cond(A)
        assert np.linalg.norm(x - ref) <= 1e-13 * cond * np.linalg.norm(ref)
        assert np.linalg.norm(A @ x - b) <= 1e-11 * np.linalg.norm(A, 2) * np.linalg.norm(x)
        Z = A.copy()
        Z[:, m // 2] = 0.0
        Z[m // 2, :] = 0.0
        with pytest.raises(api.HDSDPError):
            ls.numeric(np.triu(Z))
    finally:
        ls.destroy()


def test_kkt_factorize_falls_back_when_the_schur_matrix_is_indefinite():
    """HKKTFactorize / HKKTSolve with a Schur matrix pushed indefinite through kktDiag (what a cone or the driver can do
    to the host matrix): the operator keeps solving, through the pivoted solver, and stays on it afterwards"""
    from hdsdp_amd import api
    n = m = 64
    cone = api.SDPCone.synthetic(n, m)
    kkt = api.KKT(m, [cone])
    try:
        cone.set_start(-10.0 * n)
        assert cone.check_is_interior(1.0, np.zeros(m))
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        assert kkt.lin_type == api.HDSDP_LINSYS_DENSE_ITERATIVE
        Mh = kkt.M
        A = np.triu(Mh) + np.triu(Mh, 1).T
        shift = -0.5 * (np.linalg.eigvalsh(A)[0] + np.linalg.eigvalsh(A)[-1])
        kkt.add_to_diag(shift)                      # eigenvalues now straddle zero
        A = A + shift * np.eye(m)
        assert np.linalg.eigvalsh(A)[0] < 0 < np.linalg.eigvalsh(A)[-1]
        kkt.factorize()
        assert kkt.lin_type == api.HDSDP_LINSYS_DENSE_INDEFINITE
        b = cone.traces()
        x = kkt.solve(b)
        ref = np.linalg.solve(A, b)
        assert np.linalg.norm(x - ref) <= 1e-12 * np.linalg.cond(A) * np.linalg.norm(ref)
        # next iteration: a positive definite matrix again, the object stays on the pivoted solver and is still right
        kkt.build_up(api.KKT_TYPE_INFEASIBLE)
        kkt.factorize()
        assert kkt.lin_type == api.HDSDP_LINSYS_DENSE_INDEFINITE
        Mh = kkt.M
        A = np.triu(Mh) + np.triu(Mh, 1).T
        x = kkt.solve(b)
        assert np.linalg.norm(A @ x - b) <= 1e-11 * np.linalg.norm(b)
    finally:
        kkt.destroy()
        cone.destroy()


def test_sparse_direct_linsys_takes_a_csc_and_behaves_like_the_cholesky_backend():
    """HDSDP_LINSYS_SPARSE_DIRECT (the reference's QDLDL backend for a sparse dual matrix, hdsdp_linsolver.c:509-809): a
    lower-triangular CSC goes in; every observable -- diag (sqrt(D) there), forward / backward solves (with the
    D^-1/2 scaling there), solve, full inverse, the PSD verdict -- equals the dense Cholesky object's on the same matrix"""
    from hdsdp_amd import api
    n = 150
    rng = np.random.default_rng(3)
    A = np.zeros((n, n))
    for _ in range(5 * n):                             # a sparse symmetric pattern + a dominant diagonal
        i, j = rng.integers(0, n, 2)
        A[max(i, j), min(i, j)] += rng.uniform(-1, 1)
    A = np.tril(A, -1)
    S = A + A.T + np.diag(4.0 + np.abs(A).sum(0) + np.abs(A).sum(1))
    beg, idx, val = [0], [], []
    for j in range(n):
        rows = [i for i in range(j, n) if S[i, j] != 0.0]
        idx += rows; val += [S[i, j] for i in rows]; beg.append(len(idx))
    sp = api.LinSys(n, api.HDSDP_LINSYS_SPARSE_DIRECT)
    de = api.LinSys(n, api.HDSDP_LINSYS_DENSE_DIRECT)
    try:
        sp.symbolic(beg, idx)
        assert sp.psd_check_csc(val) and de.psd_check(np.triu(S))
        assert sp.lin_type == api.HDSDP_LINSYS_SPARSE_DIRECT      # callers branch on it (hdsdp_sdpdata.c:1035)
        assert np.array_equal(sp.get_diag(), de.get_diag())
        b = rng.uniform(-1, 1, n)
        for f in ("solve", "fsolve", "bsolve"):
            assert np.array_equal(getattr(sp, f)(b), getattr(de, f)(b)), f
        assert np.array_equal(sp.invert(), de.invert())
        assert np.linalg.norm(S @ sp.solve(b) - b) <= 1e-12 * np.linalg.norm(b) * np.linalg.cond(S)
        # ... and equals LAPACK on the dense matrix, observable by observable (not only the library's other object):
        # sum log diag = half the log-determinant (all the reference uses GetDiag for), the solve, the inverse, and the
        # forward / backward halves composing to the solve (L L^T = S, whatever the elimination order)
        Lc = np.linalg.cholesky(S)
        assert abs(2.0 * np.sum(np.log(sp.get_diag())) - np.linalg.slogdet(S)[1]) <= 1e-12 * abs(np.linalg.slogdet(S)[1])
        assert np.linalg.norm(sp.solve(b) - np.linalg.solve(S, b)) <= 1e-13 * np.linalg.cond(S) * np.linalg.norm(b)
        assert np.max(np.abs(sp.invert() - np.linalg.inv(S))) <= 1e-13 * np.linalg.cond(S) * np.max(np.abs(np.linalg.inv(S)))
        assert np.linalg.norm(sp.bsolve(sp.fsolve(b)) - np.linalg.solve(S, b)) <= 1e-13 * np.linalg.cond(S) * np.linalg.norm(b)
        assert np.linalg.norm(sp.fsolve(b) - np.linalg.solve(Lc, b)) <= 1e-13 * np.linalg.cond(S) * np.linalg.norm(b)
        bad = np.array(val)
        bad[beg[n // 2]] = -1.0                                    # a negative diagonal entry
        assert sp.psd_check_csc(bad) is False
    finally:
        sp.destroy()
        de.destroy()


FALLBACK_SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from hdsdp_amd import api
import ctypes as C
lib = api.load_library()
rng = np.random.default_rng(11)
dhash = []
for n in (130, 300, 1000):
    ls = api.LinSys(n, api.HDSDP_LINSYS_DENSE_DIRECT)
    for trial in range(3):
        # a new matrix every time through ONE factor object: the first factorisation runs eagerly, the second is captured into a
        # hipGraph, the third replays it (chol.hip)
        G = rng.uniform(-1, 1, (n, n))
        S = G @ G.T / n + (1.0 + trial) * np.eye(n)
        ls.numeric(np.triu(S))
        Lc = np.linalg.cholesky(S)
        d = ls.get_diag()
        assert np.max(np.abs(d - np.diag(Lc))) <= 1e-13 * np.max(d), ("diag", n, trial)
        dhash.append(d.tobytes())
        b = rng.uniform(-1, 1, n)
        assert np.linalg.norm(ls.solve(b) - np.linalg.solve(S, b)) <= 1e-12 * np.linalg.cond(S) * np.linalg.norm(b), ("solve", n)
        assert np.linalg.norm(ls.fsolve(b) - np.linalg.solve(Lc, b)) <= 1e-12 * np.linalg.cond(S) * np.linalg.norm(b), ("fsolve", n)
        assert np.linalg.norm(ls.bsolve(b) - np.linalg.solve(Lc.T, b)) <= 1e-12 * np.linalg.cond(S) * np.linalg.norm(b), ("bsolve", n)
        # in place, the way the reference's driver solves (solVec == NULL, hdsdp_algo.c:452-454)
        x = b.copy()
        assert lib.HFpLinsysSolve(ls._h, 1, x.ctypes.data_as(C.POINTER(C.c_double)), None) == 0
        assert np.linalg.norm(x - np.linalg.solve(S, b)) <= 1e-12 * np.linalg.cond(S) * np.linalg.norm(b), ("in place", n)
        B2 = rng.uniform(-1, 1, (3, n))
        X2 = ls.solve(B2)
        assert np.linalg.norm(X2 - np.linalg.solve(S, B2.T).T) <= 1e-12 * np.linalg.cond(S) * np.linalg.norm(B2), ("3 rhs", n)
    ls.destroy()
n = m = 300
cone = api.SDPCone.synthetic(n, m)
kkt = api.KKT(m, [cone])
cone.set_start(-10.0 * n)
assert cone.check_is_interior(1.0, np.zeros(m))
kkt.build_up(api.KKT_TYPE_INFEASIBLE)
kkt.factorize()
Mh = kkt.M
A = np.triu(Mh) + np.triu(Mh, 1).T
b = cone.traces()
x = kkt.solve(b.copy(), inplace=True)
assert np.linalg.norm(A @ x - b) <= 1e-11 * np.linalg.norm(b)
import hashlib
print("MHASH", hashlib.sha256(np.ascontiguousarray(np.tril(Mh)).tobytes()).hexdigest())
print("DHASH", hashlib.sha256(b"".join(dhash)).hexdigest())
print("FALLBACK_OK")
'''


@pytest.mark.parametrize("env", [{"HDM_TRSV_FLOW": "0"}, {"HDM_GRAPHS": "1"}, {"HDM_GRAPHS": "2", "HDM_TRSV_FLOW": "0"},
                                 {"HDM_TRSV_FLOW_FAIL_ONCE": "1"}, {"HDM_DIAG_SWEEP": "0"}, {"HDM_PERSIST": "0"}, {"HDM_CHOL_K128": "0"}],
                         ids=["per-block-substitution", "graph-replayed-factorisation", "graph-replayed-substitution", "flow-gives-up-once",
                              "lds-panel-diagonal-block", "one-tile-per-workgroup", "general-gemm-panel-and-update"])
def test_fallback_chains_of_the_factor_and_solve_kernels(env):
    """the paths behind the defaults stay covered: the per-block substitution launches (HDM_TRSV_FLOW=0, also what a
    timed-out single-launch substitution falls back to), the LDS-panel diagonal-block kernel (HDM_DIAG_SWEEP=0), the
    one-tile-per-workgroup GEMM launches (HDM_PERSIST=0), graph-replayed instead of eager factorisation chains
    (HDM_GRAPHS=1), graph-replayed substitutions (HDM_GRAPHS=2), and the give-up path itself (HDM_TRSV_FLOW_FAIL_ONCE=1
    throws the first single-launch result away): in-place solves included -- a retry must start from the caller's
    untouched right-hand side -- all against LAPACK.  Child process: the switches are read once per process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", FALLBACK_SCRIPT % root], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env))
    assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    if "HDM_TRSV_FLOW_FAIL_ONCE" in env:
        assert "single-launch substitution timed out" in r.stderr


def test_persistent_and_one_tile_per_workgroup_launches_give_the_same_bits():
    """which workgroup computes a tile, and when, must not show in the result: the Schur matrix of a 300 x 300 block built by the
    persistent kernels (tiles drawn from per-XCD counters, stealing between XCDs) and by one workgroup per tile is the same to
    the last bit (fixed-order slab reduction, no atomics on data)"""
    import os, subprocess, sys, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = []
    # (the third run leaves the persistent launches 12 workgroups instead of 512: most of every XCD's queue is then drained by
    # workgroups of other XCDs -- the stealing path, and what a partition that shows fewer XCDs would look like)
    for v, extra in (("1", {}), ("0", {}), ("1", {"HDM_PERSIST_RESERVE_CUS": "250"})):
        r = subprocess.run([sys.executable, "-c", FALLBACK_SCRIPT % root], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, HDM_PERSIST=v, **extra))
        assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        h.append(re.search(r"MHASH (\w+)", r.stdout).group(1))
    assert h[0] == h[1] == h[2], h


def test_graph_replayed_factorisations_give_the_eager_bits():
    """one factor object, a new matrix every time (first factorisation eager, second captured into a hipGraph, third replayed):
    the factors' diagonals of the nine factorisations of the script above are the same bits with and without graph replay"""
    import os, subprocess, sys, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = []
    for extra in ({}, {"HDM_GRAPHS": "1"}):
        r = subprocess.run([sys.executable, "-c", FALLBACK_SCRIPT % root], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **extra))
        assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        h.append(re.search(r"DHASH (\w+)", r.stdout).group(1))
    assert h[0] == h[1], h


def test_cholesky_small_tile_products_give_the_same_bits():
    """the panel and trailing-update products of the blocked Cholesky as 64-row tiles with fragments straight from global memory
    (hdm_k128_kernel, the default) and through the general GEMM kernel (HDM_CHOL_K128=0) sum the same terms in the same order:
    the factors' diagonals of the nine factorisations of the script above, and the Schur matrix, are the same bits"""
    import os, subprocess, sys, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = []
    for extra in ({}, {"HDM_CHOL_K128": "0"}):
        r = subprocess.run([sys.executable, "-c", FALLBACK_SCRIPT % root], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **extra))
        assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        h.append((re.search(r"DHASH (\w+)", r.stdout).group(1), re.search(r"MHASH (\w+)", r.stdout).group(1)))
    assert h[0] == h[1], h


def _spd_lower_csc(m, pairs, rng):
    """a symmetric positive definite matrix with the given off-diagonal pattern: (beg, idx, val) of its lower triangle with the
    diagonal entry first in every column, and the dense matrix"""
    A = np.zeros((m, m))
    for i, j in pairs:
        if i != j:
            v = rng.uniform(-1.0, 1.0)
            A[max(i, j), min(i, j)] = v
    A = A + A.T
    A[np.arange(m), np.arange(m)] = np.sum(np.abs(A), axis=1) + 1.0 + rng.uniform(0, 1, m)
    beg, idx, val = [0], [], []
    for c in range(m):
        rows = [c] + [int(r) for r in np.nonzero(A[c + 1:, c])[0] + c + 1]
        idx += rows
        val += [A[r, c] for r in rows]
        beg.append(len(idx))
    return np.array(beg, dtype=np.int32), np.array(idx, dtype=np.int32), np.array(val), A


@pytest.mark.parametrize("kind", ["band", "arrow", "chain_arrow", "random", "block_diagonal"])
def test_tile_form_cholesky_against_lapack(kind):
    """csrc/bsparse.hip (the device counterpart of the reference's sparse direct solver for a sparse Schur matrix): reordering,
    block symbolic factorisation, level-scheduled left-looking numeric factorisation on 128 x 128 tiles and the two
    substitutions, against a dense LAPACK solve -- a band, an arrow (block diagonal plus dense linking rows: dense in every
    envelope, two levels here), a chain with linking rows, random fill, and independent blocks; also a matrix that is not
    positive definite"""
    import ctypes as C
    from hdsdp_amd import api
    lib = api.load_library()
    rng = np.random.RandomState(7)
    if kind == "band":
        m = 1100
        pairs = [(i, j) for i in range(m) for j in range(max(0, i - 37), i)]
    elif kind == "arrow":
        m = 1500
        pairs = [(i, j) for i in range(m - 40) for j in range(i - i % 12, i)] + [(i, j) for i in range(m - 40, m) for j in range(i)]
    elif kind == "chain_arrow":
        m = 900
        pairs = [(i, i - 1) for i in range(1, m)] + [(i, i - 9) for i in range(9, m)] + [(m - 1 - q, j) for q in range(5) for j in range(m - 1 - q)]
    elif kind == "random":
        m = 700
        pairs = [(int(rng.randint(0, m)), int(rng.randint(0, m))) for _ in range(3 * m)]
    else:
        m = 1000
        pairs = [(i, j) for i in range(m) for j in range(i - i % 50, i)]
    perm0 = rng.permutation(m)                                   # the driver's numbering is arbitrary
    pairs = [(int(perm0[i]), int(perm0[j])) for i, j in pairs]
    beg, idx, val, A = _spd_lower_csc(m, pairs, rng)
    b = rng.uniform(-1, 1, m)
    x = np.zeros(m)
    info, stats, ms = C.c_int(-1), (C.c_int * 4)(), C.c_double(0.0)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    rc = lib.HMiBspSolve(m, beg.ctypes.data_as(ip), idx.ctypes.data_as(ip), val.ctypes.data_as(dp), b.ctypes.data_as(dp),
                         x.ctypes.data_as(dp), C.byref(info), stats, C.byref(ms))
    assert rc == 0 and info.value == 0, (rc, info.value)
    ref = np.linalg.solve(A, b)
    assert np.linalg.norm(x - ref) <= 1e-11 * np.linalg.norm(ref), np.linalg.norm(x - ref) / np.linalg.norm(ref)
    nb, ntiles, nlevels = stats[0], stats[1], stats[2]
    assert nb == (m + 127) // 128
    if kind == "arrow":
        assert ntiles <= 3 * nb and nlevels <= nb                # diagonal tiles, a neighbour where a small block straddles two tiles, the linking rows: no fill
    if kind == "block_diagonal":
        assert ntiles <= 2 * nb
    assert stats[3] == 0                                         # positive definite: no negative pivot
    # not positive definite: the factorisation is an LDL' without pivoting like the reference's sparse direct solver
    # (external/qdldl.c) -- it goes through, counts the negative pivots (= negative eigenvalues, by inertia) and solves
    val2 = val.copy()
    for c in (m // 2, m // 3, m - 1):
        val2[beg[c]] = -3.0
    A2 = A.copy()
    for c in (m // 2, m // 3, m - 1):
        A2[c, c] = -3.0
    x2 = np.zeros(m)
    rc = lib.HMiBspSolve(m, beg.ctypes.data_as(ip), idx.ctypes.data_as(ip), val2.ctypes.data_as(dp), b.ctypes.data_as(dp),
                         x2.ctypes.data_as(dp), C.byref(info), stats, None)
    assert rc == 0 and info.value == 0, (rc, info.value)
    assert stats[3] == int(np.sum(np.linalg.eigvalsh(A2) < 0)) >= 3, stats[3]
    ref2 = np.linalg.solve(A2, b)
    assert np.linalg.norm(x2 - ref2) <= 1e-9 * np.linalg.norm(ref2), np.linalg.norm(x2 - ref2) / np.linalg.norm(ref2)
    # an exactly zero pivot is the one thing that fails (qdldl.c:109, :212)
    val3 = np.zeros_like(val)
    rc = lib.HMiBspSolve(m, beg.ctypes.data_as(ip), idx.ctypes.data_as(ip), val3.ctypes.data_as(dp), None, None, C.byref(info), stats, None)
    assert rc == 0 and info.value > 0


@pytest.mark.parametrize("case", ["syn200x37", "syn136x5", "gpp100_B", "mix40_A"])
def test_sweeps_from_the_zero_suppressed_copy_give_the_same_bits(case):
    """S = tau C - sum y_i A_i - Rd I and the ratio test's dS, swept from the zero-suppressed copy of the constraint data
    (csrc/schur.h: HdmZs -- occupancy masks + the non-zero values, chunk-major) and from the dense skyline storage: the same
    sums in the same order without the exact-zero terms, so the same bits.  Sizes with a ragged last chunk, matrix counts
    that are not a multiple of the four the sweep takes per trip, and data that is almost all zeros (rank-one rows)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import load_golden, y_of
    from hdsdp_amd import api
    rng = np.random.default_rng(5)
    if case.startswith("syn"):
        n, m = (int(v) for v in case[3:].split("x"))
        mk = lambda: api.SDPCone.synthetic(n, m)
        Rd, tau, y = -10.0 * n, 1.0, 0.05 * rng.uniform(-1, 1, m)
    else:
        g = load_golden(case)
        n, m = int(g["dims"][0]), int(g["dims"][1])
        mk = lambda: api.SDPCone.from_csc(n, m, g["csc_beg"], g["csc_idx"], g["csc_val"])
        Rd, tau, y = float(g["Rd"][0]), float(g["tau"][0]), y_of(g)
    dy = rng.uniform(-1, 1, m)
    got = []
    for on in (0, 1):
        cone = mk()
        cone.set_start(Rd)
        cone.use_sweep_copy(on)
        used, vals, pos = cone.sweep_info()
        assert used == bool(on)
        if on:
            assert 0 < vals <= pos
            if case.startswith("syn"):
                assert 0.2 < vals / pos < 0.45          # keep-probability 0.4 on the lower triangle; stored zeros above it and in the padding
        assert cone.check_is_interior(tau, y)
        S = cone.get_dual().copy()
        step = cone.ratio_test(0.3, dy, 0.1)
        ok = cone.check_is_interior(tau, y + 0.5 * min(step, 1.0) * dy)     # (a point on the line: axpy short-cut or a sweep, as the block's size says)
        S2 = cone.get_dual().copy()
        # the corrector's <A_i, S^-1>, Rd <A_i, S^-2> read the same copy (other summation order than the dense pass: not the same bits)
        assert cone.check_is_interior(tau, y)
        kkt = api.KKT(m, [cone])
        kkt.build_up(api.KKT_TYPE_CORRECTOR)
        ex = kkt.export()
        cor = np.concatenate([np.asarray(ex["ASinv"]), np.asarray(ex["ASinvRdSinv"])])
        kkt.destroy()
        got.append((S, step, bool(ok), S2, cor))
        cone.destroy()
    assert np.array_equal(got[0][0], got[1][0])
    assert got[0][1] == got[1][1] and got[0][2] == got[1][2]
    assert np.array_equal(got[0][3], got[1][3])
    assert np.max(np.abs(got[0][4] - got[1][4])) <= 1e-12 * np.max(np.abs(got[0][4]))
