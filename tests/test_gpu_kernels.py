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
