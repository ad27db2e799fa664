"""ctypes binding of the plain-C oracle (oracle/hdsdp_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (hdsdp_amd/) never does.
"""
import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhdsdp_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE, "port"], stdout=subprocess.DEVNULL)
        L = C.CDLL(_SO)
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.orc_block_create.restype = vp
        L.orc_block_create.argtypes = [C.c_int, C.c_int, ip, ip, dp]
        L.orc_block_free.argtypes = [vp]
        L.orc_get_presolve.argtypes = [vp, ip, ip, ip, ip, ip, ip]
        L.orc_assemble_S.argtypes = [vp, C.c_double, dp, C.c_double, dp]
        L.orc_potrf.restype = C.c_int
        L.orc_potrf.argtypes = [C.c_int, dp]
        L.orc_potri_sym.argtypes = [C.c_int, dp, dp]
        L.orc_logdet.restype = C.c_double
        L.orc_logdet.argtypes = [C.c_int, dp]
        L.orc_kkt_build.restype = C.c_int
        L.orc_kkt_build.argtypes = [vp, dp, C.c_double, C.c_int, C.c_int, dp, dp, dp, dp, dp]
        L.orc_pcg_solve.restype = C.c_int
        L.orc_pcg_solve.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_int]
        L.orc_schur_solve.restype = C.c_int
        L.orc_schur_solve.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_int, ip]
        L.orc_lanczos_create.restype = vp
        L.orc_lanczos_create.argtypes = [C.c_int]
        L.orc_lanczos_free.argtypes = [vp]
        L.orc_ratio_test.restype = C.c_int
        L.orc_ratio_test.argtypes = [vp, dp, C.c_double, dp, C.c_double, vp, dp]
        L.orc_get_primal.restype = C.c_int
        L.orc_get_primal.argtypes = [vp, C.c_double, dp, dp, dp]
        L.orc_synth_csc.argtypes = [C.c_int, C.c_int, C.POINTER(ip), C.POINTER(ip), C.POINTER(dp), C.POINTER(dp)]
        L.orc_free_csc.argtypes = [ip, ip, dp, dp]
        L.orc_synth_matrix.argtypes = [C.c_int, C.c_int, dp]
        L.orc_synth_objective.argtypes = [C.c_int, C.c_int, dp, dp, C.c_int]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def synth_csc(n, m):
    """SURVEY.md 8(d) generator -> (beg, idx, val, b) numpy arrays"""
    L = lib()
    pb, pi, pv, pbb = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    L.orc_synth_csc(n, m, C.byref(pb), C.byref(pi), C.byref(pv), C.byref(pbb))
    beg = np.ctypeslib.as_array(pb, shape=(m + 2,)).copy()
    nnz = int(beg[-1])
    idx = np.ctypeslib.as_array(pi, shape=(nnz,)).copy()
    val = np.ctypeslib.as_array(pv, shape=(nnz,)).copy()
    b = np.ctypeslib.as_array(pbb, shape=(m,)).copy()
    L.orc_free_csc(pb, pi, pv, pbb)
    return beg, idx, val, b


def synth_matrix(n, c):
    """constraint matrix c (0-based) of the synthetic family, full symmetric; no CSC involved (any n)"""
    A = np.empty((n, n))
    lib().orc_synth_matrix(n, c, _d(A))
    return A


def synth_objective(n, m, threads=0):
    """(y0, C) of the synthetic family, C = I + sum y0_c A_c full symmetric; streams over the m matrices"""
    import os
    y0, Cm = np.empty(m), np.empty((n, n))
    lib().orc_synth_objective(n, m, _d(y0), _d(Cm), threads or min(16, os.cpu_count() or 1))
    return y0, Cm


class Block:
    """reference-faithful CPU evaluation of one SDP block"""

    def __init__(self, n, m, beg, idx, val):
        self.n, self.m = n, m
        self._beg = np.ascontiguousarray(beg, dtype=np.int32)
        self._idx = np.ascontiguousarray(idx, dtype=np.int32)
        self._val = np.ascontiguousarray(val, dtype=np.float64)
        self._h = lib().orc_block_create(n, m, _i(self._beg), _i(self._idx), _d(self._val))

    def presolve(self):
        m = self.m
        out = {k: np.zeros(m, dtype=np.int32) for k in ("coef_type", "coef_rank", "coef_nnz", "kkt_perm", "kkt_strategy")}
        ot = C.c_int(0)
        lib().orc_get_presolve(self._h, _i(out["coef_type"]), _i(out["coef_rank"]), _i(out["coef_nnz"]),
                               _i(out["kkt_perm"]), _i(out["kkt_strategy"]), C.byref(ot))
        out["obj_type"] = ot.value
        return out

    def assemble_S(self, tau, y, Rd):
        S = np.zeros((self.n, self.n))
        y = np.ascontiguousarray(y, dtype=np.float64)
        lib().orc_assemble_S(self._h, float(tau), _d(y), float(Rd), _d(S))
        return S  # column-major lower == C-order upper

    def factor(self, S):
        Lf = np.array(S, dtype=np.float64, copy=True)
        info = lib().orc_potrf(self.n, _d(Lf))
        return Lf, info

    def inverse(self, Lf):
        Sinv = np.zeros((self.n, self.n))
        lib().orc_potri_sym(self.n, _d(Lf), _d(Sinv))
        return Sinv

    def logdet(self, Lf):
        return lib().orc_logdet(self.n, _d(Lf))

    def kkt_build(self, Sinv, Rd, typeKKT=0, fixed=-1):
        m = self.m
        M = np.zeros((m, m))
        a, r, c = np.zeros(m), np.zeros(m), np.zeros(m)
        scal = np.zeros(4)
        Sinv = np.ascontiguousarray(Sinv)
        rc = lib().orc_kkt_build(self._h, _d(Sinv), float(Rd), typeKKT, fixed, _d(M), _d(a), _d(r), _d(c), _d(scal))
        if rc != 0:
            raise RuntimeError("orc_kkt_build: strategy not applicable to this data")
        return {"M": M, "ASinv": a, "ASinvRdSinv": r, "ASinvCSinv": c, "CSinv": scal[0], "CSinvCSinv": scal[1],
                "CSinvRdSinv": scal[2], "TraceSinv": scal[3]}

    def get_primal(self, mu, y, dy):
        X = np.zeros((self.n, self.n))
        rc = lib().orc_get_primal(self._h, float(mu), _d(np.ascontiguousarray(y, dtype=np.float64)),
                                  _d(np.ascontiguousarray(dy, dtype=np.float64)), _d(X))
        return None if rc else X

    def ratio_test(self, Lf, dtau_step, dy, eye_coef):
        """largest alpha with S + alpha*dS >= 0 (reference Lanczos estimate); consecutive calls warm-start"""
        if getattr(self, "_lz", None) is None:
            self._lz = lib().orc_lanczos_create(self.n)
        out = C.c_double(0.0)
        dy = np.ascontiguousarray(dy, dtype=np.float64)
        rc = lib().orc_ratio_test(self._h, _d(np.ascontiguousarray(Lf)), float(dtau_step), _d(dy), float(eye_coef),
                                  self._lz, C.byref(out))
        if rc != 0:
            raise RuntimeError("orc_ratio_test failed")
        return out.value

    def close(self):
        if getattr(self, "_lz", None):
            lib().orc_lanczos_free(self._lz)
            self._lz = None
        if self._h:
            lib().orc_block_free(self._h)
            self._h = None


def pcg_solve(M, rhs, relTol=5e-12, absTol=1e-12, maxIter=-1):
    """the reference's Schur solve (Jacobi PCG, hdsdp_schur.c:21-35 tolerances)"""
    M = np.ascontiguousarray(M, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    x = np.zeros_like(rhs)
    it = lib().orc_pcg_solve(M.shape[0], _d(M), _d(rhs), _d(x), relTol, absTol, maxIter)
    if it < 0:
        raise RuntimeError("orc_pcg_solve failed")
    return x


def schur_solve(M, rhs, lin_type=5, relTol=5e-12, absTol=1e-12, maxIter=-1):
    """factor + solve on the Schur system object with the reference's switch to LDL^T; returns (x, lin_type_after)"""
    M = np.ascontiguousarray(M, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    x = np.zeros_like(rhs)
    lt = C.c_int(lin_type)
    if lib().orc_schur_solve(M.shape[0], _d(M), _d(rhs), _d(x), relTol, absTol, maxIter, C.byref(lt)):
        raise RuntimeError("orc_schur_solve failed")
    return x, lt.value


def bench_sample(n, sample_m):
    """CPU baseline leg (kind "port"): the unit of work of the metric on `sample_m` constraint matrices"""
    beg, idx, val, b = synth_csc(n, sample_m)
    blk = Block(n, sample_m, beg, idx, val)
    y = np.zeros(sample_m)
    t0 = time.perf_counter()
    S = blk.assemble_S(1.0, y, -10.0 * n)
    Lf, info = blk.factor(S)
    Sinv = blk.inverse(Lf)
    t1 = time.perf_counter()
    k = blk.kkt_build(Sinv, -10.0 * n, 0)
    t2 = time.perf_counter()
    for rhs in (b, k["ASinv"], k["ASinvRdSinv"]):
        pcg_solve(k["M"], rhs)
    t3 = time.perf_counter()
    blk.close()
    return {"chol_s": t1 - t0, "buildup_s": t2 - t1, "factor_s": 0.0, "solve3_s": t3 - t2}
