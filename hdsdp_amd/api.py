"""Host-side mirror of the reference operator interface for the Schur path, over the C ABI.

Names, argument meaning and error behaviour follow the reference (interface/hdsdp_schur.h:10-22,
linalg/hdsdp_linsolver.h:16-28, interface/hdsdp_conic.h:27-61): `KKT.build_up(type)` is
HKKTBuildUp, `KKT.factorize()` HKKTFactorize, `KKT.solve(rhs)` HKKTSolve, ... A non-OK
`hdsdp_retcode` raises `HDSDPError` (the reference's `goto exit_cleanup`).

There is NO CPU fallback: importing works without a GPU (symbol checks), any compute call needs
the HIP extension and an MI355X.
"""
import ctypes as C
import os
import sys

import numpy as np

KKT_TYPE_INFEASIBLE, KKT_TYPE_CORRECTOR, KKT_TYPE_HOMOGENEOUS, KKT_TYPE_PRIMAL = 0, 1, 2, 3
KKT_M1, KKT_M2, KKT_M3, KKT_M4, KKT_M5 = 0, 1, 2, 3, 4
HDSDP_LINSYS_DENSE_DIRECT, HDSDP_LINSYS_SPARSE_DIRECT, HDSDP_LINSYS_DENSE_ITERATIVE, HDSDP_LINSYS_DENSE_INDEFINITE = 0, 2, 5, 6
RETCODE_OK, RETCODE_FAILED, RETCODE_MEMORY = 0, 1, 2
BUFFER_DUALVAR, BUFFER_DUALCHECK, BUFFER_DUALSTEP = 0, 1, 2   # interface/hdsdp_conic.h:24-26

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HDSDP_MI355X_LIB") or os.path.join(_PKG, "libhdsdp_mi355x.so")   # override: same-box A/B of two builds

# every symbol include/hdsdp_mi355x.h declares
EXPORTS = [
    "HKKTCreate", "HKKTInit", "HKKTBuildUp", "HKKTBuildUpExtraCone", "HKKTBuildUpFixed", "HKKTExport",
    "HKKTFactorize", "HKKTSolve", "HKKTRegularize", "HKKTRegisterPSDP", "HKKTClear", "HKKTDestroy",
    "HFpLinsysCreate", "HFpLinsysSetParam", "HFpLinsysSymbolic", "HFpLinsysNumeric", "HFpLinsysSwitchToBackUp",
    "HFpLinsysPsdCheck", "HFpLinsysFSolve", "HFpLinsysBSolve", "HFpLinsysSolve", "HFpLinsysGetDiag",
    "HFpLinsysInvert", "HFpLinsysClear", "HFpLinsysDestroy",
    "HMiConeCreateSDP", "HMiConeCreateSDP64", "HMiConeBuilderBegin", "HMiConeBuilderAddColumn", "HMiConeBuilderStored", "HMiConeBuilderFinish", "HMiConeBuilderAbort", "HMiConeCreateSynthetic", "HMiConeDestroy", "HMiConeSetStart", "HMiConeUpdate",
    "HMiConeCheckIsInterior", "HMiConeGetLogBarrier", "HMiConeRatioTest", "HMiLanczosStartVector", "HMiConeGetPrimal", "HMiConeCheckIsInteriorExpert",
    "HMiConeAddStepToBufferAndCheck", "HMiConeReduceResi", "HMiConeSetPerturb", "HMiConeGetCoeffNorm", "HMiConeGetObjNorm",
    "HMiConeScalByConstant", "HMiConeComputeATimesXpy", "HMiConeComputeXDotS", "HMiConeComputeTraceCX", "HMiConeGetDual", "HMiConeGetPresolve", "HMiConeDetectFeature", "HMiConeGetDualMatrix",
    "HMiConeGetTraces", "HMiConeGetPath", "HMiConeSweepInfo", "HMiConeGetStreaming", "HMiConeUseSweepCopy", "HMiKKTSetHostMirror", "HMiConeSetExchange", "HMiConeSetExchangePieces", "HMiConeGetExchangeStats", "HMiConeGetBuildProfile", "HMiConeBuildPrimalXSXDirection",
    "HMiConeGetExchangeBuffers", "HMiConeSetExchangeBuffers", "HMiKKTDeviceMatrix", "HMiKKTGetRows", "HMiDeviceInit",
    "HMiSetDevices", "HMiSetDevicesEx", "HMiRcclGroupSelfTest", "HMiGetDeviceGroup", "HMiSetShardMinDim", "HMiConeGetShardCount", "HMiConeGetGroupTraffic", "HMiRcclSelfTest", "HMiGetCallStats", "HMiCallStatName", "HMiResetCallStats", "HMiKKTPhaseAEligible", "HMiKKTPhaseA",
    "HMiDeviceSynchronize", "HMiStream", "HMiVersion", "HMiGetStageTimes", "HMiGemmNT", "HMiPotrf",
    "HMiMfmaPeakProbe", "HMiDiagBlockProbe", "HMiCholEnvelopeSolve", "HMiCholEnvelopeProbe", "HMiKKTEnvelopeInfo", "HMiKKTTileInfo", "HMiKKTNegativePivots", "HMiBspSolve", "HMiRcmOrder", "HMiSetKernelTiming", "HMiGetKernelTiming", "HMiGetKernelTimingEx", "HMiPresolveCSC", "HMiMfmaIssueProbe", "HMiSetDebugBuffer",
    "HMiReadSDPA", "HMiSDPAGetDims", "HMiSDPAGetBlock", "HMiSDPAGetBlock64", "HMiSDPAGetRHS", "HMiSDPAFree",
]


class HDSDPError(RuntimeError):
    pass


class hdsdp_kkt(C.Structure):  # interface/def_hdsdp_schur.h:32-68
    _fields_ = [
        ("nRow", C.c_int), ("nCones", C.c_int), ("maxConeDim", C.c_int), ("cones", C.c_void_p),
        ("isKKTSparse", C.c_int), ("kktM", C.c_void_p),
        ("invBuffer", C.POINTER(C.c_double)), ("kktBuffer", C.POINTER(C.c_double)),
        ("kktBuffer2", C.POINTER(C.c_double)),
        ("kktMatBeg", C.POINTER(C.c_int)), ("kktMatIdx", C.POINTER(C.c_int)),
        ("kktMatElem", C.POINTER(C.c_double)), ("kktDiag", C.POINTER(C.POINTER(C.c_double))),
        ("dASinvVec", C.POINTER(C.c_double)), ("dASinvCSinvVec", C.POINTER(C.c_double)),
        ("dASinvRdSinvVec", C.POINTER(C.c_double)),
        ("dCSinvCSinv", C.c_double), ("dCSinvRdSinv", C.c_double), ("dCSinv", C.c_double),
        ("dTraceSinv", C.c_double), ("dPrimalX", C.c_void_p),
    ]


class hdsdp_linsys_head(C.Structure):  # linalg/def_hdsdp_linsolver.h:40-63, the fields callers read
    _fields_ = [("nCol", C.c_int), ("chol", C.c_void_p), ("LinType", C.c_int)]


def _lin_type(ptr):
    return C.cast(ptr, C.POINTER(hdsdp_linsys_head)).contents.LinType


_lib = None


def load_library():
    """dlopen the in-tree HIP extension; fails loudly if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HDSDPError(f"{LIB_PATH} is missing: run `python -m hdsdp_amd.build` (hipcc, gfx950). "
                         "There is no CPU fallback for the Schur path.")
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.7 and loads it by path, the engine
    # links the system one by soname.  Engine first, torch later = two runtimes in one process, and torch then finds
    # "No HIP GPUs"; torch first = the engine's DT_NEEDED resolves to the copy torch already loaded.  So when torch is
    # around (tests, bench, the RCCL exchange), let it load first.  HDSDP_MI355X_NO_TORCH=1 skips this for processes
    # that never touch torch (the C ABI itself has no torch dependency).
    if "torch" not in sys.modules and os.environ.get("HDSDP_MI355X_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    kp = C.POINTER(hdsdp_kkt)
    sig = {
        "HKKTCreate": (C.c_int, [C.POINTER(kp)]),
        "HKKTInit": (C.c_int, [kp, C.c_int, C.c_int, C.POINTER(vp)]),
        "HKKTBuildUp": (C.c_int, [kp, C.c_int]),
        "HKKTBuildUpExtraCone": (C.c_int, [kp, vp, C.c_int]),
        "HKKTBuildUpFixed": (C.c_int, [kp, C.c_int, C.c_int]),
        "HKKTExport": (None, [kp, dp, dp, dp, dp, dp, dp, dp]),
        "HKKTFactorize": (C.c_int, [kp]),
        "HKKTSolve": (C.c_int, [kp, dp, dp]),
        "HKKTRegularize": (None, [kp, C.c_double]),
        "HKKTRegisterPSDP": (None, [kp, vp]),
        "HKKTClear": (None, [kp]),
        "HKKTDestroy": (None, [C.POINTER(kp)]),
        "HFpLinsysCreate": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int]),
        "HFpLinsysSetParam": (None, [vp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]),
        "HFpLinsysSymbolic": (C.c_int, [vp, ip, ip]),
        "HFpLinsysNumeric": (C.c_int, [vp, ip, ip, dp]),
        "HFpLinsysSwitchToBackUp": (C.c_int, [vp]),
        "HFpLinsysPsdCheck": (C.c_int, [vp, ip, ip, dp, ip]),
        "HFpLinsysFSolve": (None, [vp, C.c_int, dp, dp]),
        "HFpLinsysBSolve": (None, [vp, C.c_int, dp, dp]),
        "HFpLinsysSolve": (C.c_int, [vp, C.c_int, dp, dp]),
        "HFpLinsysGetDiag": (C.c_int, [vp, dp]),
        "HFpLinsysInvert": (None, [vp, dp, dp]),
        "HFpLinsysClear": (None, [vp]),
        "HFpLinsysDestroy": (None, [C.POINTER(vp)]),
        "HMiConeCreateSDP": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, ip, ip, dp, C.c_int, C.c_int]),
        "HMiConeCreateSynthetic": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "HMiConeDestroy": (None, [C.POINTER(vp)]),
        "HMiConeSetStart": (None, [vp, C.c_double]),
        "HMiConeUpdate": (None, [vp, C.c_double, dp]),
        "HMiConeCheckIsInterior": (C.c_int, [vp, C.c_double, dp, ip]),
        "HMiConeRatioTest": (C.c_int, [vp, C.c_double, dp, C.c_double, C.c_int, dp]),
        "HMiLanczosStartVector": (None, [C.c_int, dp]),
        "HMiConeGetPrimal": (None, [vp, C.c_double, dp, dp, dp, dp]),
        "HMiConeCheckIsInteriorExpert": (C.c_int, [vp, C.c_double, C.c_double, dp, C.c_double, C.c_int, ip]),
        "HMiConeAddStepToBufferAndCheck": (C.c_int, [vp, C.c_double, C.c_int, ip]),
        "HMiConeReduceResi": (None, [vp, C.c_double]),
        "HMiConeSetPerturb": (None, [vp, C.c_double]),
        "HMiConeGetCoeffNorm": (C.c_double, [vp, C.c_int]),
        "HMiConeGetObjNorm": (C.c_double, [vp, C.c_int]),
        "HMiConeScalByConstant": (None, [vp, C.c_double]),
        "HMiConeComputeATimesXpy": (None, [vp, dp, dp]),
        "HMiConeComputeXDotS": (C.c_double, [vp, dp]),
        "HMiConeComputeTraceCX": (C.c_double, [vp, dp]),
        "HMiConeGetDual": (None, [vp, dp, dp]),
        "HMiConeGetLogBarrier": (C.c_int, [vp, C.c_double, dp, C.c_int, dp]),
        "HMiConeGetPresolve": (None, [vp, ip, ip, ip, ip, ip, ip]),
        "HMiConeDetectFeature": (None, [vp, dp, ip, dp]),
        "HMiConeGetDualMatrix": (C.c_int, [vp, dp]),
        "HMiConeGetTraces": (C.c_int, [vp, dp]),
        "HMiConeGetPath": (C.c_int, [vp]),
        "HMiConeSweepInfo": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "HMiConeGetStreaming": (C.c_int, [vp, ip]),
        "HMiConeUseSweepCopy": (C.c_int, [vp, C.c_int]),
        "HMiKKTSetHostMirror": (None, [kp, C.c_int]),
        "HMiConeSetExchange": (None, [vp, vp, vp, vp]),
        "HMiConeSetExchangePieces": (None, [vp, vp, vp, C.c_int]),
        "HMiConeGetExchangeStats": (None, [vp, ip, ip]),
        "HMiConeGetBuildProfile": (C.c_int, [vp, C.c_int, dp, C.c_int]),
        "HMiConeBuildPrimalXSXDirection": (None, [vp, dp, dp, C.c_int]),
        "HMiConeGetExchangeBuffers": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64)]),
        "HMiConeSetExchangeBuffers": (C.c_int, [vp, vp, vp]),
        "HMiKKTDeviceMatrix": (vp, [kp, C.POINTER(C.c_int64)]),
        "HMiKKTGetRows": (C.c_int, [kp, C.c_int, ip, dp]),
        "HMiSetDevices": (C.c_int, [C.c_int, ip]),
        "HMiSetDevicesEx": (C.c_int, [C.c_int, ip, C.c_int]),
        "HMiRcclGroupSelfTest": (C.c_int, [C.c_int, ip, C.c_int]),
        "HMiGetDeviceGroup": (C.c_int, [ip, C.c_int, ip]),
        "HMiSetShardMinDim": (None, [C.c_int]),
        "HMiConeGetShardCount": (C.c_int, [vp]),
        "HMiConeGetGroupTraffic": (None, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "HMiRcclSelfTest": (C.c_int, [C.c_int]),
        "HMiGetCallStats": (C.c_int, [dp, C.POINTER(C.c_int64), C.c_int]),
        "HMiCallStatName": (C.c_char_p, [C.c_int]),
        "HMiResetCallStats": (None, []),
        "HMiKKTPhaseAEligible": (C.c_int, [kp]),
        "HMiKKTPhaseA": (C.c_int, [kp, C.c_double, dp, dp, dp, dp, dp, ip, dp]),
        "HMiDeviceInit": (C.c_int, [C.c_int]),
        "HMiDeviceSynchronize": (C.c_int, []),
        "HMiStream": (vp, []),
        "HMiVersion": (C.c_char_p, []),
        "HMiGetStageTimes": (None, [dp, C.c_int]),
        "HMiGemmNT": (C.c_int, [vp, C.c_int64, C.c_int, vp, C.c_int64, C.c_int, vp, C.c_int64, C.c_int, C.c_int,
                                C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]),
        "HMiPotrf": (C.c_int, [vp, C.c_int, C.c_int64, ip]),
        "HMiMfmaPeakProbe": (C.c_double, [C.c_int]),
        "HMiDiagBlockProbe": (C.c_double, [C.c_int, C.c_int]),
        "HMiCholEnvelopeSolve": (C.c_int, [vp, C.c_int, ip, vp, vp, vp, ip]),
        "HMiCholEnvelopeProbe": (C.c_int, [C.c_int, C.c_int, C.c_int, dp, dp]),
        "HMiKKTEnvelopeInfo": (None, [kp, ip, dp]),
        "HMiKKTTileInfo": (C.c_int, [kp, ip, C.POINTER(C.c_int64), ip, C.POINTER(C.c_int64)]),
        "HMiKKTNegativePivots": (C.c_int, [kp]),
        "HMiBspSolve": (C.c_int, [C.c_int, ip, ip, dp, dp, dp, ip, ip, dp]),
        "HMiRcmOrder": (C.c_int, [C.c_int, ip, ip, ip]),
        "HMiPresolveCSC": (C.c_int, [C.c_int, C.c_int, ip, ip, dp, ip, ip, ip, ip, ip, ip]),
        "HMiReadSDPA": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "HMiSDPAGetDims": (None, [vp, ip, ip, ip]),
        "HMiSDPAGetBlock": (C.c_int, [vp, C.c_int, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(dp)]),
        "HMiSDPAGetBlock64": (C.c_int, [vp, C.c_int, ip, C.POINTER(C.POINTER(C.c_int64)), C.POINTER(ip), C.POINTER(dp)]),
        "HMiSDPAGetRHS": (dp, [vp]),
        "HMiConeCreateSDP64": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), ip, dp, C.c_int, C.c_int]),
        "HMiConeBuilderBegin": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "HMiConeBuilderAddColumn": (C.c_int, [vp, C.c_int, C.c_int64, ip, dp]),
        "HMiConeBuilderStored": (C.c_int64, [vp]),
        "HMiConeBuilderFinish": (C.c_int, [C.POINTER(vp), C.POINTER(vp)]),
        "HMiConeBuilderAbort": (None, [C.POINTER(vp)]),
        "HMiSDPAFree": (None, [C.POINTER(vp)]),
        "HMiMfmaIssueProbe": (C.c_double, [C.c_int, C.c_int, C.c_int]),
        "HMiSetDebugBuffer": (None, [vp, C.c_int]),
        "HMiSetKernelTiming": (None, [C.c_int]),
        "HMiGetKernelTiming": (C.c_int, [dp, dp, C.POINTER(C.c_int64)]),
        "HMiGetKernelTimingEx": (C.c_int, [dp, dp, dp, C.POINTER(C.c_int64)]),
    }
    for name in EXPORTS:
        fn = getattr(lib, name)  # AttributeError == missing export
        fn.restype, fn.argtypes = sig[name]
    _lib = lib
    return lib


TRANSPORT_ENV, TRANSPORT_COPY, TRANSPORT_RCCL = -1, 0, 1


def set_devices(ids, shard_min_dim=None, transport=TRANSPORT_ENV):
    """single-process multi-device mode (include/hdsdp_mi355x.h: HMiSetDevicesEx): cones created afterwards with rank 0 of
    world 1 are sharded over `ids` (repeated ids = shards sharing a device, exchanged by device copies).  transport:
    TRANSPORT_COPY / TRANSPORT_RCCL, or TRANSPORT_ENV = what HDSDP_MI355X_TRANSPORT says (default copies)"""
    lib = load_library()
    arr = (C.c_int * len(ids))(*ids)
    if lib.HMiSetDevicesEx(len(ids), arr, int(transport)) != 0:
        raise HDSDPError("HMiSetDevicesEx failed")
    if shard_min_dim is not None:
        lib.HMiSetShardMinDim(int(shard_min_dim))


def rccl_group_self_test(ids, timeout_ms=60000):
    """HMiRcclGroupSelfTest over `ids` (distinct devices): 0 = passed, else the first failing stage"""
    lib = load_library()
    arr = (C.c_int * len(ids))(*ids)
    return int(lib.HMiRcclGroupSelfTest(len(ids), arr, int(timeout_ms)))


def device_group():
    lib = load_library()
    ids = (C.c_int * 16)()
    tr = C.c_int(-1)
    n = lib.HMiGetDeviceGroup(ids, 16, C.byref(tr))
    return list(ids[:n]), tr.value


def _check(rc, what):
    if rc != RETCODE_OK:
        raise HDSDPError(f"{what} returned hdsdp_retcode {rc}")


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def presolve_csc(n, m, beg, idx, val):
    """host-only presolve of one block (classification, ordering, strategy plan); needs no GPU"""
    lib = load_library()
    beg = np.ascontiguousarray(beg, dtype=np.int32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    out = {k: np.zeros(m, dtype=np.int32) for k in ("coef_type", "coef_rank", "coef_nnz", "kkt_perm", "kkt_strategy")}
    ot = C.c_int(0)
    _check(lib.HMiPresolveCSC(m, n, _iptr(beg), _iptr(idx), _dptr(val), _iptr(out["coef_type"]),
                              _iptr(out["coef_rank"]), _iptr(out["coef_nnz"]), _iptr(out["kkt_perm"]),
                              _iptr(out["kkt_strategy"]), C.byref(ot)), "HMiPresolveCSC")
    out["obj_type"] = ot.value
    return out


def lanczos_start_vector(n):
    v = np.zeros(n)
    load_library().HMiLanczosStartVector(int(n), _dptr(v))
    return v


def read_sdpa(fname):
    """SDPA sparse file -> dict(m, b, blocks=[dict(n, beg, idx, val)], n_lp) in the reference's CSC layout"""
    lib = load_library()
    h = C.c_void_p()
    _check(lib.HMiReadSDPA(os.fsencode(fname), C.byref(h)), f"HMiReadSDPA({fname})")
    try:
        m, nb, nlp = C.c_int(), C.c_int(), C.c_int()
        lib.HMiSDPAGetDims(h, C.byref(m), C.byref(nb), C.byref(nlp))
        b = np.ctypeslib.as_array(lib.HMiSDPAGetRHS(h), shape=(m.value,)).copy()
        blocks = []
        for i in range(nb.value):
            dim = C.c_int()
            pb, pi, pv = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
            _check(lib.HMiSDPAGetBlock(h, i, C.byref(dim), C.byref(pb), C.byref(pi), C.byref(pv)), "HMiSDPAGetBlock")
            beg = np.ctypeslib.as_array(pb, shape=(m.value + 2,)).copy()
            nnz = int(beg[-1])
            idx = np.ctypeslib.as_array(pi, shape=(max(nnz, 1),))[:nnz].copy()
            val = np.ctypeslib.as_array(pv, shape=(max(nnz, 1),))[:nnz].copy()
            blocks.append({"n": dim.value, "beg": beg, "idx": idx, "val": val})
        return {"m": m.value, "b": b, "blocks": blocks, "n_lp": nlp.value}
    finally:
        lib.HMiSDPAFree(C.byref(h))


class SDPCone:
    """One SDP block living in HBM (the reference's hdsdp_cone with the dense-SDP slots of the Schur path)."""

    def __init__(self, handle, n, m, rank=0, world=1):
        self._h, self.n, self.m, self.rank, self.world = handle, n, m, rank, world

    @classmethod
    def from_csc(cls, n, m, beg, idx, val, iCone=0, rank=0, world=1):
        """CSC of shape n(n+1)/2 x (m+1), column 0 = C (interface/def_hdsdp_user_data.h:16-32)."""
        lib = load_library()
        beg = np.ascontiguousarray(beg, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        assert beg.shape[0] == m + 2
        h = C.c_void_p()
        _check(lib.HMiConeCreateSDP(C.byref(h), iCone, m, n, _iptr(beg), _iptr(idx), _dptr(val), rank, world),
               "HMiConeCreateSDP")
        return cls(h, n, m, rank, world)

    @classmethod
    def from_csc64(cls, n, m, beg, idx, val, iCone=0, rank=0, world=1):
        """the same CSC with 64-bit column pointers (HMiConeCreateSDP64): a block may hold more than 2^31 - 1 entries"""
        lib = load_library()
        beg = np.ascontiguousarray(beg, dtype=np.int64)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        assert beg.shape[0] == m + 2
        h = C.c_void_p()
        _check(lib.HMiConeCreateSDP64(C.byref(h), iCone, m, n, beg.ctypes.data_as(C.POINTER(C.c_int64)), _iptr(idx), _dptr(val),
                                      rank, world), "HMiConeCreateSDP64")
        return cls(h, n, m, rank, world)

    @classmethod
    def from_columns(cls, n, m, columns, iCone=0, rank=0, world=1):
        """column-by-column ingest (HMiConeBuilder*): `columns` yields (iCol, packed_idx, values), iCol 0 = the objective,
        i = A_i, in any order; nothing but the column in hand has to exist on the caller's side"""
        lib = load_library()
        b = C.c_void_p()
        _check(lib.HMiConeBuilderBegin(C.byref(b), iCone, m, n, rank, world), "HMiConeBuilderBegin")
        try:
            for iCol, idx, val in columns:
                idx = np.ascontiguousarray(idx, dtype=np.int32)
                val = np.ascontiguousarray(val, dtype=np.float64)
                assert idx.shape == val.shape
                _check(lib.HMiConeBuilderAddColumn(b, int(iCol), int(idx.shape[0]), _iptr(idx), _dptr(val)), "HMiConeBuilderAddColumn")
            stored = int(lib.HMiConeBuilderStored(b))
            h = C.c_void_p()
            _check(lib.HMiConeBuilderFinish(C.byref(b), C.byref(h)), "HMiConeBuilderFinish")
        finally:
            if b:
                lib.HMiConeBuilderAbort(C.byref(b))
        cone = cls(h, n, m, rank, world)
        cone.stored_entries = stored
        return cone

    @classmethod
    def synthetic(cls, n, m, iCone=0, rank=0, world=1):
        """SURVEY.md 8(d) strictly feasible dense family, generated in HBM."""
        lib = load_library()
        h = C.c_void_p()
        _check(lib.HMiConeCreateSynthetic(C.byref(h), iCone, n, m, rank, world), "HMiConeCreateSynthetic")
        return cls(h, n, m, rank, world)

    def set_start(self, Rd):
        load_library().HMiConeSetStart(self._h, float(Rd))

    def check_is_interior(self, tau, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        ok = C.c_int(0)
        _check(load_library().HMiConeCheckIsInterior(self._h, float(tau), _dptr(y), C.byref(ok)),
               "HConeCheckIsInterior")
        return bool(ok.value)

    def ratio_test(self, dtau_step, dy, ada_ratio=0.0, buffer=BUFFER_DUALVAR):
        """HConeRatioTest: largest step keeping S + step*dS in the cone (inf if unbounded); S = the chosen buffer"""
        dy = np.ascontiguousarray(dy, dtype=np.float64)
        out = C.c_double(0.0)
        _check(load_library().HMiConeRatioTest(self._h, float(dtau_step), _dptr(dy), float(ada_ratio), buffer,
                                               C.byref(out)), "HConeRatioTest")
        return out.value

    def check_is_interior_expert(self, c_coef, a_scal, a_coef, eye_coef, buffer=BUFFER_DUALVAR):
        a_coef = np.ascontiguousarray(a_coef, dtype=np.float64)
        ok = C.c_int(0)
        _check(load_library().HMiConeCheckIsInteriorExpert(self._h, float(c_coef), float(a_scal), _dptr(a_coef),
                                                           float(eye_coef), buffer, C.byref(ok)), "HConeCheckIsInteriorExpert")
        return bool(ok.value)

    def axpy_buffer_and_check(self, step, buffer=BUFFER_DUALCHECK):
        ok = C.c_int(0)
        _check(load_library().HMiConeAddStepToBufferAndCheck(self._h, float(step), buffer, C.byref(ok)), "HConeAddStepToBufferAndCheck")
        return bool(ok.value)

    def log_barrier_of(self, buffer):
        out = C.c_double(0.0)
        _check(load_library().HMiConeGetLogBarrier(self._h, 0.0, None, buffer, C.byref(out)), "HConeGetLogBarrier")
        return out.value

    def coeff_norm(self, which):
        return load_library().HMiConeGetCoeffNorm(self._h, int(which))

    def obj_norm(self, which):
        return load_library().HMiConeGetObjNorm(self._h, int(which))

    def scal_by_constant(self, s):
        load_library().HMiConeScalByConstant(self._h, float(s))

    def a_times_x(self, X, y=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        out = np.zeros(self.m) if y is None else np.ascontiguousarray(y, dtype=np.float64).copy()
        load_library().HMiConeComputeATimesXpy(self._h, _dptr(X), _dptr(out))
        return out

    def x_dot_s(self, X):
        return load_library().HMiConeComputeXDotS(self._h, _dptr(np.ascontiguousarray(X, dtype=np.float64)))

    def trace_cx(self, X):
        return load_library().HMiConeComputeTraceCX(self._h, _dptr(np.ascontiguousarray(X, dtype=np.float64)))

    def get_dual(self):
        S = np.zeros((self.n, self.n))
        load_library().HMiConeGetDual(self._h, _dptr(S), None)
        return S

    def reduce_resi(self, v):
        load_library().HMiConeReduceResi(self._h, float(v))

    def set_perturb(self, v):
        load_library().HMiConeSetPerturb(self._h, float(v))

    def get_primal(self, mu, y, dy):
        """HConeGetPrimal: n x n primal recovery matrix, or None if S(y) is not positive definite"""
        y = np.ascontiguousarray(y, dtype=np.float64)
        dy = np.ascontiguousarray(dy, dtype=np.float64)
        X = np.full((self.n, self.n), np.nan)
        load_library().HMiConeGetPrimal(self._h, float(mu), _dptr(y), _dptr(dy), _dptr(X), None)
        return None if np.isnan(X[0, 0]) else X

    def log_barrier(self, tau, y=None):
        out = C.c_double(0.0)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float64)
            yp = _dptr(y)
        _check(load_library().HMiConeGetLogBarrier(self._h, float(tau), yp, BUFFER_DUALVAR, C.byref(out)),
               "HConeGetLogBarrier")
        return out.value

    def build_primal_xsx(self, X, XSX, dual_matrix=True):
        """XSX += X^T D X with D = the dual matrix S (dual_matrix) or the dual step of the last ratio test"""
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert XSX.flags.c_contiguous and XSX.dtype == np.float64
        load_library().HMiConeBuildPrimalXSXDirection(self._h, _dptr(X), _dptr(XSX), 1 if dual_matrix else 0)
        return XSX

    def shard_count(self):
        return int(load_library().HMiConeGetShardCount(self._h))

    def group_traffic(self):
        a, b = C.c_int64(0), C.c_int64(0)
        load_library().HMiConeGetGroupTraffic(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def exchange_stats(self):
        """(pieces of the last build's all-to-all, launches the staged second congruence step was cut into)"""
        p, s = C.c_int(0), C.c_int(0)
        load_library().HMiConeGetExchangeStats(self._h, C.byref(p), C.byref(s))
        return p.value, s.value

    def build_profile(self, shard=0):
        """where the last SHARDED Schur build of one shard spent its time (HMiConeGetBuildProfile; ms, bytes): None when no
        sharded build has run on it"""
        buf = np.zeros(8 + 6 * 64)
        k = int(load_library().HMiConeGetBuildProfile(self._h, int(shard), _dptr(buf), int(buf.size)))
        if k <= 0:
            return None
        P = int(buf[0])
        per = buf[8:8 + 6 * P].reshape(P, 6)
        return {"pieces": P, "staged": bool(buf[1]), "invert_ms": float(buf[2]),
                ("congruence_step1_ms" if buf[1] else "congruence_ms"): float(buf[3]), "slab_reduce_ms": float(buf[4]),
                "allreduce_ms": float(buf[5]), "extract_ms": float(buf[6]), "world": int(buf[7]),
                "step2_piece_ms": per[:, 0].tolist(), "exchange_wait_ms": per[:, 1].tolist(),
                "exchange_wait_host_ms": per[:, 2].tolist(), "gram_piece_ms": per[:, 3].tolist(),
                "piece_bytes_sent": per[:, 4].tolist(), "piece_flight_ms": per[:, 5].tolist()}

    def detect_feature(self, b):
        """the cone's getstat slot (HConeDetectFeature): (int features[20], double features[20]), zero where the cone decides nothing"""
        fi, fd = np.zeros(20, dtype=np.int32), np.zeros(20)
        bb = np.ascontiguousarray(b, dtype=np.float64)
        load_library().HMiConeDetectFeature(self._h, _dptr(bb), _iptr(fi), _dptr(fd))
        return fi, fd

    def presolve(self):
        m = self.m
        out = {k: np.zeros(m, dtype=np.int32) for k in ("coef_type", "coef_rank", "coef_nnz", "kkt_perm", "kkt_strategy")}
        ot = C.c_int(0)
        load_library().HMiConeGetPresolve(self._h, _iptr(out["coef_type"]), _iptr(out["coef_rank"]),
                                          _iptr(out["coef_nnz"]), _iptr(out["kkt_perm"]), _iptr(out["kkt_strategy"]),
                                          C.byref(ot))
        out["obj_type"] = ot.value
        return out

    def dual_matrix(self):
        S = np.zeros((self.n, self.n), dtype=np.float64)  # column-major n x n seen as C-order transposed
        _check(load_library().HMiConeGetDualMatrix(self._h, _dptr(S)), "HMiConeGetDualMatrix")
        return S

    def traces(self):
        t = np.zeros(self.m, dtype=np.float64)
        _check(load_library().HMiConeGetTraces(self._h, _dptr(t)), "HMiConeGetTraces")
        return t

    @property
    def path(self):
        return load_library().HMiConeGetPath(self._h)

    def use_sweep_copy(self, on):
        _check(load_library().HMiConeUseSweepCopy(self._h, int(bool(on))), "HMiConeUseSweepCopy")

    def streaming(self):
        """(streamed?, rows per regenerated batch): include/hdsdp_mi355x.h: HMiConeGetStreaming"""
        b = C.c_int(0)
        on = load_library().HMiConeGetStreaming(self._h, C.byref(b))
        return bool(on), int(b.value)

    def sweep_info(self):
        """(in use, stored values, skyline positions they stand for) of the zero-suppressed copy the S / dS sweeps read"""
        v, p = C.c_int64(0), C.c_int64(0)
        on = load_library().HMiConeSweepInfo(self._h, C.byref(v), C.byref(p))
        return bool(on), int(v.value), int(p.value)

    def destroy(self):
        if self._h:
            load_library().HMiConeDestroy(C.byref(self._h))
            self._h = None


class KKT:
    """hdsdp_kkt: the Schur operator (interface/hdsdp_schur.c)."""

    def __init__(self, m, cones, host_mirror=True):
        lib = load_library()
        self.m = m
        self.cones = list(cones)
        self._k = C.POINTER(hdsdp_kkt)()
        _check(lib.HKKTCreate(C.byref(self._k)), "HKKTCreate")
        self._cone_arr = (C.c_void_p * len(self.cones))(*[c._h for c in self.cones])
        _check(lib.HKKTInit(self._k, m, len(self.cones), self._cone_arr), "HKKTInit")
        if not host_mirror:
            lib.HMiKKTSetHostMirror(self._k, 0)

    def build_up(self, typeKKT=KKT_TYPE_INFEASIBLE):
        _check(load_library().HKKTBuildUp(self._k, typeKKT), "HKKTBuildUp")

    def build_up_fixed(self, typeKKT, strategy):
        _check(load_library().HKKTBuildUpFixed(self._k, typeKKT, strategy), "HKKTBuildUpFixed")

    def register_psdp(self, primal_mats):
        """HKKTRegisterPSDP (interface/hdsdp_schur.c:375-380): borrow one n x n column-major primal matrix per cone;
        build_up(KKT_TYPE_PRIMAL) then runs the builder on them instead of S^-1"""
        self._psdp = [np.ascontiguousarray(x, dtype=np.float64) for x in primal_mats]
        self._psdp_arr = (C.POINTER(C.c_double) * len(self._psdp))(*[_dptr(x) for x in self._psdp])
        load_library().HKKTRegisterPSDP(self._k, C.cast(self._psdp_arr, C.c_void_p))

    def factorize(self):
        _check(load_library().HKKTFactorize(self._k), "HKKTFactorize")

    def solve(self, rhs, inplace=False):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        if inplace:
            _check(load_library().HKKTSolve(self._k, _dptr(rhs), None), "HKKTSolve")
            return rhs
        sol = np.zeros_like(rhs)
        _check(load_library().HKKTSolve(self._k, _dptr(rhs), _dptr(sol)), "HKKTSolve")
        return sol

    def phase_a_eligible(self):
        return bool(load_library().HMiKKTPhaseAEligible(self._k))

    def phase_a(self, tau, y, rhs):
        """HMiKKTPhaseA: interior check + INFEASIBLE build + factorisation + the three solves in ONE launch (small rank-one
        blocks).  Returns (is_interior, logdet S, d1, d2, d3); the operator's host fields are filled as after build_up."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        d1, d2, d3 = (np.zeros(self.m) for _ in range(3))
        ok, ld = C.c_int(0), C.c_double(0.0)
        _check(load_library().HMiKKTPhaseA(self._k, float(tau), _dptr(y), _dptr(rhs), _dptr(d1), _dptr(d2), _dptr(d3),
                                           C.byref(ok), C.byref(ld)), "HMiKKTPhaseA")
        return bool(ok.value), ld.value, d1, d2, d3

    def regularize(self, reg):
        load_library().HKKTRegularize(self._k, float(reg))

    @property
    def lin_type(self):
        """kktM->LinType: DENSE_ITERATIVE until a factorisation fails, DENSE_INDEFINITE afterwards
        (HFpLinsysSwitchToIndefinite, linalg/hdsdp_linsolver.c:1827-1857)"""
        return _lin_type(self._k.contents.kktM)

    def export(self):
        m = self.m
        a, r, c = (np.zeros(m) for _ in range(3))
        s = [C.c_double(0.0) for _ in range(4)]
        load_library().HKKTExport(self._k, _dptr(a), _dptr(r), _dptr(c), C.byref(s[0]), C.byref(s[1]), C.byref(s[2]),
                                  C.byref(s[3]))
        return {"ASinv": a, "ASinvRdSinv": r, "ASinvCSinv": c, "CSinvCSinv": s[0].value, "CSinv": s[1].value,
                "CSinvRdSinv": s[2].value, "TraceSinv": s[3].value}

    @property
    def is_sparse(self):
        """isKKTSparse: the host Schur matrix is the aggregated-pattern CSC of interface/hdsdp_schur.c:46-139"""
        return bool(self._k.contents.isKKTSparse)

    def csc(self):
        """(kktMatBeg, kktMatIdx, kktMatElem) of a sparse operator as numpy views: lower triangle, column by column"""
        k = self._k.contents
        beg = np.ctypeslib.as_array(k.kktMatBeg, shape=(self.m + 1,))
        nnz = int(beg[self.m])
        return beg, np.ctypeslib.as_array(k.kktMatIdx, shape=(nnz,)), np.ctypeslib.as_array(k.kktMatElem, shape=(nnz,))

    @property
    def M(self):
        """kktMatElem as the reference leaves it: m x m column-major, lower triangle valid.
        Returned in numpy C-order, i.e. M[j, i] is element (row i, col j): valid where i >= j.
        For a sparse operator the CSC is expanded into such an array (a copy, not a view)."""
        k = self._k.contents
        if k.isKKTSparse:
            beg, idx, val = self.csc()
            D = np.zeros((self.m, self.m))
            for j in range(self.m):
                D[j, idx[beg[j]:beg[j + 1]]] = val[beg[j]:beg[j + 1]]
            return D
        return np.ctypeslib.as_array(k.kktMatElem, shape=(self.m, self.m))

    def rows(self, rows):
        """full symmetric rows of the DEVICE copy of M after a build (works with the host mirror off)"""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        out = np.zeros((rows.size, self.m))
        _check(load_library().HMiKKTGetRows(self._k, int(rows.size), _iptr(rows), _dptr(out)), "HMiKKTGetRows")
        return out

    def add_to_diag(self, v):
        """what the y-box cone does through kktDiag[] (interface/hdsdp_conic_bound.c:201-229)"""
        if self.is_sparse:
            beg, idx, val = self.csc()
            val[beg[:-1]] += v
            return
        M = self.M
        idx = np.arange(self.m)
        M[idx, idx] += v

    def envelope_info(self):
        """(permuted, fraction): is P M P' factored, and the share of 128-blocks inside the pattern's block envelope"""
        p, f = C.c_int(0), C.c_double(1.0)
        load_library().HMiKKTEnvelopeInfo(self._k, C.byref(p), C.byref(f))
        return bool(p.value), float(f.value)

    def tile_info(self):
        """None, or (tiles stored, tiles of the dense lower triangle, levels, bytes) of a sparse operator kept in tile form"""
        t, lv = C.c_int(0), C.c_int(0)
        dt, by = C.c_int64(0), C.c_int64(0)
        if not load_library().HMiKKTTileInfo(self._k, C.byref(t), C.byref(dt), C.byref(lv), C.byref(by)):
            return None
        return t.value, dt.value, lv.value, by.value

    def negative_pivots(self):
        """negative pivots of the tile-form operator's last LDL' factorisation (-1: not tile form / not factored)"""
        return int(load_library().HMiKKTNegativePivots(self._k))

    def stage_times_ms(self):
        t = np.zeros(8)
        load_library().HMiGetStageTimes(_dptr(t), 8)
        return t

    def destroy(self):
        if self._k:
            load_library().HKKTDestroy(C.byref(self._k))
            self._k = None


class LinSys:
    """hdsdp_linsys_fp, dense direct backend (linalg/hdsdp_linsolver.c:1044-1286)."""

    def __init__(self, n, ltype=HDSDP_LINSYS_DENSE_DIRECT):
        self.n = n
        self._h = C.c_void_p()
        _check(load_library().HFpLinsysCreate(C.byref(self._h), n, ltype), "HFpLinsysCreate")

    def numeric(self, A):
        A = np.ascontiguousarray(A, dtype=np.float64)
        _check(load_library().HFpLinsysNumeric(self._h, None, None, _dptr(A)), "HFpLinsysNumeric")

    @property
    def lin_type(self):
        return _lin_type(self._h)

    def symbolic(self, beg, idx):
        """HFpLinsysSymbolic: the lower-triangular CSC pattern of a SPARSE_DIRECT object"""
        self._beg = np.ascontiguousarray(beg, dtype=np.int32)
        self._idx = np.ascontiguousarray(idx, dtype=np.int32)
        _check(load_library().HFpLinsysSymbolic(self._h, _iptr(self._beg), _iptr(self._idx)), "HFpLinsysSymbolic")

    def psd_check_csc(self, val):
        val = np.ascontiguousarray(val, dtype=np.float64)
        ok = C.c_int(0)
        _check(load_library().HFpLinsysPsdCheck(self._h, _iptr(self._beg), _iptr(self._idx), _dptr(val), C.byref(ok)),
               "HFpLinsysPsdCheck")
        return bool(ok.value)

    def psd_check(self, A):
        A = np.ascontiguousarray(A, dtype=np.float64)
        ok = C.c_int(0)
        _check(load_library().HFpLinsysPsdCheck(self._h, None, None, _dptr(A), C.byref(ok)), "HFpLinsysPsdCheck")
        return bool(ok.value)

    def get_diag(self):
        d = np.zeros(self.n)
        _check(load_library().HFpLinsysGetDiag(self._h, _dptr(d)), "HFpLinsysGetDiag")
        return d

    def _rhs(self, rhs):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        nrhs = 1 if rhs.ndim == 1 else rhs.shape[0]  # C-order (nrhs, n) == column-major n x nrhs
        return rhs, nrhs

    def solve(self, rhs):
        rhs, nrhs = self._rhs(rhs)
        sol = np.zeros_like(rhs)
        _check(load_library().HFpLinsysSolve(self._h, nrhs, _dptr(rhs), _dptr(sol)), "HFpLinsysSolve")
        return sol

    def fsolve(self, rhs):
        rhs, nrhs = self._rhs(rhs)
        sol = np.zeros_like(rhs)
        load_library().HFpLinsysFSolve(self._h, nrhs, _dptr(rhs), _dptr(sol))
        return sol

    def bsolve(self, rhs):
        rhs, nrhs = self._rhs(rhs)
        sol = np.zeros_like(rhs)
        load_library().HFpLinsysBSolve(self._h, nrhs, _dptr(rhs), _dptr(sol))
        return sol

    def invert(self):
        out = np.zeros((self.n, self.n))
        aux = np.zeros(1)
        load_library().HFpLinsysInvert(self._h, _dptr(out), _dptr(aux))
        return out

    def destroy(self):
        if self._h:
            load_library().HFpLinsysDestroy(C.byref(self._h))
            self._h = None
