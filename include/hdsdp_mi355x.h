/* hdsdp_mi355x.h -- C ABI of the MI355X-native Schur-complement / Cholesky engine for HDSDP.
 *
 * Drop-in boundary: this library exports the reference's own operator surface for the hot path,
 *   HKKT*      (interface/hdsdp_schur.h:10-22)      -- Schur operator
 *   HFpLinsys* (linalg/hdsdp_linsolver.h:16-28)     -- dense factor / solve operator
 * with binary-compatible structs (hdsdp_kkt: interface/def_hdsdp_schur.h:32-68; hdsdp_linsys_fp:
 * linalg/def_hdsdp_linsolver.h:40-63; hdsdp_cone: interface/def_hdsdp_conic.h:56-107), so that
 * interface/hdsdp_algo.c keeps calling HKKTBuildUp / HKKTFactorize / HKKTSolve unchanged.
 * Return codes and error behaviour follow interface/hdsdp.h:42-48 and hdsdp_utils.h:32-45
 * ("not PSD" is a value, not an error: linalg/hdsdp_linsolver.c:1133-1140).
 *
 * The per-cone plug point is the reference's `coneBuildSchur` vtable slot
 * (def_hdsdp_conic.h:80-81, wired at interface/hdsdp_conic.c:140-141): HKKTBuildUp calls
 * cone->coneBuildSchur(cone->coneData, cone->iCone, kkt, typeKKT) exactly as
 * HConeBuildSchurComplement does (interface/hdsdp_conic.c:308-328).  HMiCone* below creates a cone
 * whose slot is the GPU builder (replacing sdpDenseConeGetKKT, hdsdp_conic_sdp.c:1726-1812).
 *
 * Plain C: pointers + sizes only.  All exported calls are synchronous on return for host-visible
 * outputs.  One solver instance per process/GPU (as the reference: single-threaded caller).
 */
#ifndef HDSDP_MI355X_H
#define HDSDP_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* Everything declared in this header is the library's exported surface and nothing else is: the library is built with
 * -fvisibility=hidden, these declarations carry default visibility. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ---- interface/hdsdp.h:42-48 ---- */
typedef enum { HDSDP_RETCODE_OK, HDSDP_RETCODE_FAILED, HDSDP_RETCODE_MEMORY } hdsdp_retcode;

/* ---- interface/hdsdp_conic.h:16-19 ---- */
#define KKT_TYPE_INFEASIBLE  (0)
#define KKT_TYPE_CORRECTOR   (1)
#define KKT_TYPE_HOMOGENEOUS (2)
#define KKT_TYPE_PRIMAL      (3)

/* ---- interface/def_hdsdp_schur.h:26-30 ---- */
#define KKT_M1 (0)
#define KKT_M2 (1)
#define KKT_M3 (2)
#define KKT_M4 (3)
#define KKT_M5 (4)

/* ---- linalg/def_hdsdp_linsolver.h:26-38 ---- */
typedef enum {
    HDSDP_LINSYS_DENSE_DIRECT,
    HDSDP_LINSYS_SMALL_DIRECT,
    HDSDP_LINSYS_SPARSE_DIRECT,
    HDSDP_LINSYS_SPARSE_INDEFINITE,
    HDSDP_LINSYS_SPARSE_ITERATIVE,
    HDSDP_LINSYS_DENSE_ITERATIVE,
    HDSDP_LINSYS_DENSE_INDEFINITE
} linsys_type;

/* ---- linalg/def_hdsdp_linsolver.h:40-63 (field order is ABI) ---- */
typedef struct {
    int nCol;
    void *chol;
    linsys_type LinType;
    hdsdp_retcode (*cholCreate)(void **, int);
    void (*cholSetParam)(void *, void *);
    hdsdp_retcode (*cholSymbolic)(void *, int *, int *);
    hdsdp_retcode (*cholNumeric)(void *, int *, int *, double *);
    hdsdp_retcode (*cholPsdCheck)(void *, int *, int *, double *, int *);
    void (*cholFSolve)(void *, int, double *, double *);
    void (*cholBSolve)(void *, int, double *, double *);
    hdsdp_retcode (*cholSolve)(void *, int, double *, double *);
    hdsdp_retcode (*cholGetDiag)(void *, double *);
    void (*cholInvert)(void *, double *, double *);
    void (*cholDestroy)(void **);
    int nSolves;
    int nFactorizes;
} hdsdp_linsys_fp;

/* ---- interface/def_hdsdp_conic.h:21-31, :56-107 (field order is ABI) ---- */
typedef enum {
    HDSDP_CONETYPE_UNKNOWN,
    HDSDP_CONETYPE_LP,
    HDSDP_CONETYPE_BOUND,
    HDSDP_CONETYPE_SCALAR_BOUND,
    HDSDP_CONETYPE_DENSE_SDP,
    HDSDP_CONETYPE_SPARSE_SDP,
    HDSDP_CONETYPE_SOCP
} cone_type;

typedef struct {
    int iCone;
    cone_type cone;
    void *usrData;
    void *coneData;
    hdsdp_retcode (*coneCreate)(void **);
    hdsdp_retcode (*coneProcData)(void *, int, int, int *, int *, double *);
    hdsdp_retcode (*conePresolveData)(void *);
    void (*coneDestroyData)(void **);
    void (*coneSetStart)(void *, double);
    void (*coneUpdate)(void *, double, double *);
    hdsdp_retcode (*coneRatioTest)(void *, double, double *, double, int, double *);
    int64_t (*coneGetSymNnz)(void *);
    int (*coneGetDim)(void *);
    void (*coneAddSymNz)(void *, int, int *);
    void (*coneGetKKTMap)(void *, int, int *);
    hdsdp_retcode (*coneBuildSchur)(void *, int, void *, int);
    hdsdp_retcode (*coneBuildSchurFixed)(void *, int, void *, int, int);
    void (*coneBuildPrimalDirection)(void *, void *, double *, double *, int);
    hdsdp_retcode (*coneInteriorCheck)(void *, double, double *, int *);
    hdsdp_retcode (*coneInteriorCheckExpert)(void *, double, double, double *, double, int, int *);
    hdsdp_retcode (*coneGetBarrier)(void *, double, double *, int, double *);
    hdsdp_retcode (*coneAxpyBufferAndCheck)(void *, double, int, int *);
    void (*coneReduceResi)(void *, double);
    void (*coneSetPerturb)(void *, double);
    void (*conePRecover)(void *, double, double *, double *, double *, double *);
    void (*coneDRecover)(void *, double *, double *);
    void (*coneATimesXpy)(void *, double *, double *);
    double (*coneTraceCX)(void *, double *);
    double (*coneXDotS)(void *, double *);
    double (*coneGetCoeffNorm)(void *, int);
    double (*coneGetObjNorm)(void *, int);
    void (*coneScal)(void *, double);
    void (*coneView)(void *);
    void (*getstat)(void *, double *, int[20], double[20]);
} hdsdp_cone;

/* ---- interface/def_hdsdp_schur.h:32-68 (field order is ABI) ---- */
typedef struct {
    int nRow;
    int nCones;
    int maxConeDim;
    hdsdp_cone **cones;
    int isKKTSparse;
    hdsdp_linsys_fp *kktM;
    double *invBuffer;
    double *kktBuffer;
    double *kktBuffer2;
    int *kktMatBeg;
    int *kktMatIdx;
    double *kktMatElem;
    double **kktDiag;
    double *dASinvVec;
    double *dASinvCSinvVec;
    double *dASinvRdSinvVec;
    double dCSinvCSinv;
    double dCSinvRdSinv;
    double dCSinv;
    double dTraceSinv;
    double **dPrimalX;
} hdsdp_kkt;

/* =====================  Schur operator: interface/hdsdp_schur.h:10-22  ===================== */
hdsdp_retcode HKKTCreate(hdsdp_kkt **pHKKT);                                              /* hdsdp_schur.c:167 */
hdsdp_retcode HKKTInit(hdsdp_kkt *HKKT, int nRow, int nCones, hdsdp_cone **cones);        /* :181 */
hdsdp_retcode HKKTBuildUp(hdsdp_kkt *HKKT, int typeKKT);                                  /* :256 */
hdsdp_retcode HKKTBuildUpExtraCone(hdsdp_kkt *HKKT, hdsdp_cone *cone, int typeKKT);       /* :270 */
hdsdp_retcode HKKTBuildUpFixed(hdsdp_kkt *HKKT, int typeKKT, int kktStrategy);            /* :279 */
void HKKTExport(hdsdp_kkt *HKKT, double *dKKTASinvVec, double *dKKTASinvRdSinvVec, double *dKKTASinvCSinvVec,
                double *dCSinvCSinv, double *dCSinv, double *dCSinvRdCSinv, double *dTraceSinv); /* :293 */
hdsdp_retcode HKKTFactorize(hdsdp_kkt *HKKT);                                             /* :328 */
hdsdp_retcode HKKTSolve(hdsdp_kkt *HKKT, double *dRhsVec, double *dLhsVec);               /* :338 */
void HKKTRegularize(hdsdp_kkt *HKKT, double dKKTReg);                                     /* :348 */
void HKKTRegisterPSDP(hdsdp_kkt *HKKT, double **dPrimalScalX);                            /* :375 */
void HKKTClear(hdsdp_kkt *HKKT);                                                          /* :382 */
void HKKTDestroy(hdsdp_kkt **pHKKT);                                                      /* :408 */

/* ==================  linear-system operator: linalg/hdsdp_linsolver.h:16-28  ================
 * DENSE_DIRECT (the dual matrix S): blocked Cholesky on the device; "not positive definite" is a value of PsdCheck
 * and a failure of Numeric, as in lapackLinSolver* (hdsdp_linsolver.c:1082-1144).
 * SPARSE_DIRECT (a sparse dual matrix S; QDLDL in the reference, hdsdp_linsolver.c:509-809): Symbolic takes the lower-
 * triangular CSC pattern, Numeric / PsdCheck the values; the matrix is factored densely on the device.  Result-equivalent:
 * QDLDL's forward / backward solves carry the D^-1/2 scaling, GetDiag returns sqrt(D), Invert the full inverse.
 * DENSE_ITERATIVE (the Schur matrix M): the same direct Cholesky instead of the reference's PCG.  When M is not
 * numerically positive definite -- Numeric's factorisation fails, or Solve returns NaN -- the object switches itself to
 * DENSE_INDEFINITE like HFpLinsysSwitchToIndefinite (hdsdp_linsolver.c:1827-1857, called at :2034-2039 and :2098-2103)
 * and from then on factors with a pivoted device factorisation (csrc/lu.hip; same solutions as the reference's
 * dsytrf/dsytrs to rounding); LinType reads DENSE_INDEFINITE afterwards, PsdCheck / GetDiag fail and FSolve / BSolve /
 * Invert do nothing on such an object (:1729-1797). */
hdsdp_retcode HFpLinsysCreate(hdsdp_linsys_fp **pHLin, int nCol, linsys_type Ltype);      /* hdsdp_linsolver.c:1859 */
void HFpLinsysSetParam(hdsdp_linsys_fp *HLin, double relTol, double absTol, int nThreads, int maxIter,
                       int nRestartFreq);                                                 /* :2000 */
hdsdp_retcode HFpLinsysSymbolic(hdsdp_linsys_fp *HLin, int *colMatBeg, int *colMatIdx);   /* :2020 */
hdsdp_retcode HFpLinsysNumeric(hdsdp_linsys_fp *HLin, int *colMatBeg, int *colMatIdx, double *colMatElem); /* :2026 */
hdsdp_retcode HFpLinsysSwitchToBackUp(hdsdp_linsys_fp *HLin);                             /* :2047 */
hdsdp_retcode HFpLinsysPsdCheck(hdsdp_linsys_fp *HLin, int *colMatBeg, int *colMatIdx, double *colMatElem,
                                int *isPsd);                                              /* :2059 */
void HFpLinsysFSolve(hdsdp_linsys_fp *HLin, int nRhs, double *rhsVec, double *solVec);    /* :2065 */
void HFpLinsysBSolve(hdsdp_linsys_fp *HLin, int nRhs, double *rhsVec, double *solVec);    /* :2072 */
hdsdp_retcode HFpLinsysSolve(hdsdp_linsys_fp *HLin, int nRhs, double *rhsVec, double *solVec); /* :2079 */
hdsdp_retcode HFpLinsysGetDiag(hdsdp_linsys_fp *HLin, double *diagElem);                  /* :2113 */
void HFpLinsysInvert(hdsdp_linsys_fp *HLin, double *dFullMatrix, double *dAuxiMatrix);    /* :2119 */
void HFpLinsysClear(hdsdp_linsys_fp *HLin);                                               /* :2126 */
void HFpLinsysDestroy(hdsdp_linsys_fp **HLin);                                            /* :2138 */

/* ==========  MI355X SDP cone: the reference's cone slots that feed the Schur path  ==========
 * HMiConeCreateSDP    = HUserDataSetConeData + HConeCreate/SetData/ProcData/PresolveData for one
 *                       HDSDP_CONETYPE_DENSE_SDP block (tests/test_file_io.c:402-416): the CSC input is
 *                       the reference's user_data layout, shape n(n+1)/2 x (m+1), column 0 = C
 *                       (interface/def_hdsdp_user_data.h:16-32).  Coefficient data is uploaded to HBM once.
 * HMiConeCreateSynthetic builds the SURVEY.md 8(d) synthetic dense family directly in HBM (64-bit
 *                       offsets; the reference's int32 CSC cannot hold n=m=2000 fully dense).
 * rank/world shard the constraint rows (row i is owned by rank i % world); pass 0,1 for one GPU.
 * The remaining functions dispatch through the cone vtable like interface/hdsdp_conic.c:203-330.
 */
hdsdp_retcode HMiConeCreateSDP(hdsdp_cone **pCone, int iCone, int nRow, int nCol, const int *coneMatBeg,
                               const int *coneMatIdx, const double *coneMatElem, int rank, int world);
hdsdp_retcode HMiConeCreateSynthetic(hdsdp_cone **pCone, int iCone, int nCol, int nRow, int rank, int world);
/* 64-bit ingest (SURVEY finding 6, row f3).  The reference's user data is ONE CSC with `int` column pointers
 * (interface/def_hdsdp_user_data.h:22-32): a block whose columns hold more than 2^31 - 1 entries together -- fully dense
 * n = m = 2000 is 4.0e9 -- cannot be expressed in it.  Two ways in that can:
 *   HMiConeCreateSDP64   the same CSC with 64-bit column pointers (row indices stay `int`: a packed index is below
 *                        n(n+1)/2 < 2^31 for n <= 65535, the largest block dimension accepted);
 *   HMiConeBuilder*      column by column: Begin, then AddColumn once per non-zero column in any order (iCol 0 = the objective
 *                        C, iCol i = A_i; entries = packed lower index + value, any order, at most n(n+1)/2 of them), then
 *                        Finish, which returns the cone (columns never given are zero matrices).  A column is classified as it
 *                        arrives and the caller's arrays are not kept, so the caller may generate, hand over and free one
 *                        column at a time and never holds a 2^31-entry array; the library's own host copy goes back as soon as
 *                        the block's device layout is written (blocks on the congruence + Gram path).
 * Both give the device data HMiConeCreateSDP gives on the same entries (tests/test_gpu_ingest.py: bit-identical dual matrix and
 * Schur matrix on every CSC golden).  All three ways in refuse (RETCODE_FAILED) a column whose pointers run backwards, a packed
 * index outside [0, n(n+1)/2) and a packed index named twice in one column (tests/test_abi_cpu.py::
 * test_presolve_refuses_malformed_columns). */
hdsdp_retcode HMiConeCreateSDP64(hdsdp_cone **pCone, int iCone, int nRow, int nCol, const int64_t *coneMatBeg,
                                 const int *coneMatIdx, const double *coneMatElem, int rank, int world);
typedef struct HMiConeBuilder_s HMiConeBuilder;
hdsdp_retcode HMiConeBuilderBegin(HMiConeBuilder **pBuilder, int iCone, int nRow, int nCol, int rank, int world);
hdsdp_retcode HMiConeBuilderAddColumn(HMiConeBuilder *builder, int iCol, int64_t nnz, const int *packedIdx, const double *val);
int64_t HMiConeBuilderStored(const HMiConeBuilder *builder);          /* entries handed in so far */
hdsdp_retcode HMiConeBuilderFinish(HMiConeBuilder **pBuilder, hdsdp_cone **pCone);   /* frees the builder either way */
void HMiConeBuilderAbort(HMiConeBuilder **pBuilder);
void HMiConeDestroy(hdsdp_cone **pCone);
void HMiConeSetStart(hdsdp_cone *cone, double dConeStartVal);                   /* HConeSetStart  hdsdp_conic.c:222 */
void HMiConeUpdate(hdsdp_cone *cone, double barHsdTau, double *rowDual);        /* HConeUpdate    :228 */
hdsdp_retcode HMiConeCheckIsInterior(hdsdp_cone *cone, double barHsdTau, double *rowDual, int *isInterior);
hdsdp_retcode HMiConeGetLogBarrier(hdsdp_cone *cone, double barHsdTau, double *rowDual, int whichBuffer,
                                   double *logdet);
/* HConeRatioTest (hdsdp_conic.c:270; sdpDenseConeRatioTestImpl hdsdp_conic_sdp.c:1640-1686 + HLanczosSolve
 * linalg/hdsdp_lanczos.c:161-292) on the device: largest step with S + step * dS >= 0, where
 * dS = barHsdTauStep*C - sum rowDualStep_i A_i + dAdaRatio*Rd*I and S is the matrix of the chosen buffer: the current
 * dual matrix (BUFFER_DUALVAR, 0) or the trial point factored last in the checker (BUFFER_DUALCHECK, 1).  Also reachable
 * through the cone's coneRatioTest slot. */
hdsdp_retcode HMiConeRatioTest(hdsdp_cone *cone, double barHsdTauStep, double *rowDualStep, double dAdaRatio,
                               int whichBuffer, double *maxStep);
/* the rest of the dual line search, device resident (hdsdp_conic.c:350-387, :410-421; hdsdp_conic_sdp.c:2192-2241,
 * :2333-2361): an interior check of dCCoef*C + dACoefScal*sum dACoef_i A_i + dEyeCoef*I in either buffer, S + dStep*dS
 * with the dS of the last ratio test (in place for BUFFER_DUALVAR, into the checker for BUFFER_DUALCHECK), and the two
 * scalar setters.  HMiConeGetLogBarrier(whichBuffer = 1, rowDual = NULL) reads the checker's factor. */
hdsdp_retcode HMiConeCheckIsInteriorExpert(hdsdp_cone *cone, double dCCoef, double dACoefScal, double *dACoef, double dEyeCoef,
                                           int whichBuffer, int *isInterior);
hdsdp_retcode HMiConeAddStepToBufferAndCheck(hdsdp_cone *cone, double dStep, int whichBuffer, int *isInterior);
/* HConeDetectFeature (interface/hdsdp_conic.c:423-428): the cone's `getstat` slot -- class counts, "no primal interior", implied
 * trace bound, "very dense" as the reference's feature detection decides them (interface/hdsdp_conic_sdp.c:2651-2758), answered
 * from the engine's own presolve: a driver whose SDP blocks live here needs no CPU cone beside them.
 * Entries the cone does not decide are left as the caller set them. */
void HMiConeDetectFeature(hdsdp_cone *cone, double *rowRHS, int coneIntFeatures[20], double coneDblFeatures[20]);
void HMiConeReduceResi(hdsdp_cone *cone, double dResiReduction);
void HMiConeSetPerturb(hdsdp_cone *cone, double dDualPerturb);
/* the remaining HCone* utilities of a dense SDP block (hdsdp_conic.h:44-61; hdsdp_conic_sdp.c:1558-1614, :2470-2560):
 * data norms (whichNorm: ABS_NORM 1 = sum |a_ij|, FRO_NORM 2, over the full symmetric matrices; rows: sum resp. root of the
 * sum of squares over the constraints), scaling of the objective, and for a primal matrix X (host, n x n, symmetric)
 * y_i += <A_i, X>, <C, X>, <X, S>; HMiConeGetDual copies S (symmetrised) to the host. */
double HMiConeGetCoeffNorm(hdsdp_cone *cone, int whichNorm);
double HMiConeGetObjNorm(hdsdp_cone *cone, int whichNorm);
void HMiConeScalByConstant(hdsdp_cone *cone, double dScal);
void HMiConeComputeATimesXpy(hdsdp_cone *cone, double *dConePrimal, double *dATimesX);
double HMiConeComputeXDotS(hdsdp_cone *cone, double *dConePrimal);
double HMiConeComputeTraceCX(hdsdp_cone *cone, double *dConePrimal);
void HMiConeGetDual(hdsdp_cone *cone, double *dConeDual, double *dConeDual2);
/* HConeGetPrimal (hdsdp_conic.c:389; sdpDenseConeGetPrimal hdsdp_conic_sdp.c:2393-2446): primal recovery
 * X = mu * L^-T (sym(L^-1 dS L^-T) + I) L^-1 with S = C - sum rowDual_i A_i = L L^T and dS = sum rowDualStep_i A_i;
 * dConePrimal receives the n x n matrix (host, column-major), dConePrimal2 is unused scratch.  If S is not positive
 * definite a message is printed and dConePrimal is left untouched, like the reference.  Also the conePRecover slot. */
void HMiConeGetPrimal(hdsdp_cone *cone, double dBarrierMu, double *dRowDual, double *dRowDualStep, double *dConePrimal,
                      double *dConePrimal2);
/* the pseudo-random Lanczos start vector of the reference (HLanczosIPrepare, hdsdp_lanczos.c:33-42), host only */
void HMiLanczosStartVector(int n, double *v);
/* presolve results (classification hdsdp_sdpdata.c:2321-2458, ordering + plan hdsdp_conic_sdp.c:539-676);
 * each output may be NULL; arrays have nRow entries */
void HMiConeGetPresolve(hdsdp_cone *cone, int *coefType, int *coefRank, int *coefNnz, int *kktPerm,
                        int *kktStrategy, int *objType);
/* the same presolve without creating a cone (host only, touches no device): for CPU-side checks */
hdsdp_retcode HMiPresolveCSC(int nRow, int nCol, const int *coneMatBeg, const int *coneMatIdx,
                             const double *coneMatElem, int *coefType, int *coefRank, int *coefNnz, int *kktPerm,
                             int *kktStrategy, int *objType);
/* host copies of device state for parity checks: S (n x n, lower valid), b_i = tr(A_i) */
hdsdp_retcode HMiConeGetDualMatrix(hdsdp_cone *cone, double *S);
hdsdp_retcode HMiConeGetTraces(hdsdp_cone *cone, double *trA);
/* which device path the cone's builder uses: 0 = dense congruence + Gram (MFMA), 1 = rank-one */
int HMiConeGetPath(hdsdp_cone *cone);
/* the zero-suppressed copy of the constraint data the S / dS sweeps read (csrc/schur.h: HdmZs): returns 1 and the number of
 * stored values / of skyline positions it stands for when the cone has built one, 0 when its sweeps read the dense storage */
int HMiConeSweepInfo(hdsdp_cone *cone, int64_t *values, int64_t *positions);
/* Streamed constraint data (synthetic family, csrc/engine_cone.h: MiCone::streamed): when the A_L forms of all owned rows do
 * not fit next to what a build needs -- BASELINE configs[4], n = 2000, m = 8000, on ONE device: 136 GB of them beside 130 GB of
 * transformed rows -- HMiConeCreateSynthetic keeps none of them: every consumer (congruence batches, S / dS sweeps without a
 * sweep copy, corrector dots, norms, A X) regenerates the rows it needs into a batch buffer from the counter-based generator,
 * same bits every time.  Returns 1 and the batch size in rows for such a cone, 0 for resident data.  HDSDP_MI355X_STREAM_A
 * forces either way. */
int HMiConeGetStreaming(hdsdp_cone *cone, int *batchRows);
/* on = 1: build the copy now whatever the block's size and fill, and sweep from it; on = 0: sweep from the dense storage
 * (a copy that exists is kept).  For tests and A/B runs; the default is the rule of HDSDP_MI355X_ZS in the table below. */
int HMiConeUseSweepCopy(hdsdp_cone *cone, int on);

/* ===============================  device-resident fast path  ===============================
 * The reference boundary hands host buffers (S in, M out).  For benchmarking with inputs resident in
 * HBM, these keep M on the device between BuildUp / Factorize / Solve (no PCIe round trip of M).  */
void HMiKKTSetHostMirror(hdsdp_kkt *HKKT, int mirrorM);   /* default 1: kktMatElem is refreshed after BuildUp; with 0
                                                              HKKTRegularize / Factorize / Solve act on the device copy */
/* multi-GPU (world > 1): constraint rows are sharded (row i on rank i % world).  Each rank congruence-
 * transforms its own rows, a transpose (all-to-all) re-shards the transformed data from "by constraint"
 * to "by packed-index range", each rank forms the Gram partial sum over its range, and an all-reduce
 * (sum) assembles the augmented Gram matrix on every rank.  The two collectives are supplied by the host
 * program (torch.distributed over RCCL in hdsdp_amd/dist.py); both are called with the engine stream
 * idle and must return with the data in place.
 *   alltoall(ctx): send chunk r of the send buffer to rank r, receive chunk r of the recv buffer from rank r
 *   allreduce(ctx, buf, count): in-place sum of `count` doubles at device pointer `buf` over all ranks */
typedef int (*hmi_alltoall_fn)(void *ctx);
typedef int (*hmi_allreduce_fn)(void *ctx, void *buf, int64_t count);
void HMiConeSetExchange(hdsdp_cone *cone, hmi_alltoall_fn a2a, hmi_allreduce_fn ar, void *ctx);
/* optional: the all-to-all in `npieces` pieces along the packed index, so that the Gram product over a piece's index
 * range runs while the later pieces are still in flight.  start(ctx, offset, count, piece): begin exchanging the `count`
 * doubles at `offset` of EVERY chunk (send chunk r -> rank r, into recv chunk <source rank> at the same offset), may
 * return before the data has arrived; wait(ctx, piece): return once that piece is in place.  Pieces are started in
 * order 0..npieces-1 and waited for in that order.  start(.., piece) is called once the send ranges of that piece are
 * final; the engine stream may still be computing the ranges of LATER pieces (the congruence's second step runs by
 * packed-index range in piece order, so the exchange also overlaps the congruence; HDSDP_MI355X_STAGED_A2A=0 restores
 * "congruence complete, stream idle, then exchange"), so start() must only touch the ranges it is given. */
typedef int (*hmi_alltoall_piece_fn)(void *ctx, int64_t offset, int64_t count, int piece);
typedef int (*hmi_alltoall_wait_fn)(void *ctx, int piece);
void HMiConeSetExchangePieces(hdsdp_cone *cone, hmi_alltoall_piece_fn start, hmi_alltoall_wait_fn wait, int npieces);
/* HConeBuildPrimalXSXDirection (interface/hdsdp_conic.c:335-338; the cone's coneBuildPrimalDirection slot, used by the primal
 * refinement hdsdp_psdp.c:236,295): XSX += X^T D X, D = the resident dual matrix (iDualMat != 0) or the dual step of the last
 * ratio test; X and XSX are n x n column-major host matrices */
void HMiConeBuildPrimalXSXDirection(hdsdp_cone *cone, double *dPrimalScalMatrix, double *dPrimalXSXBuffer, int iDualMat);
/* what the last HKKTBuildUp did: pieces of the exchange (1 = one blocking all-to-all) and, when the congruence's second
 * step was staged by packed-index range, the number of launches it was cut into (0 = not staged) */
void HMiConeGetExchangeStats(hdsdp_cone *cone, int *pieces, int *stagedLaunches);
/* Where the last SHARDED Schur build (world > 1) of one shard spent its time, so that a multi-GPU bench line says where a step went
 * and not only how long it took (the cone loop being sharded: interface/hdsdp_schur.c:256-268).  `shard`: index inside an in-process
 * device group (0 for a process-per-GPU cone).  Milliseconds; device times are differences of HIP events on the shard's engine stream,
 * host times wall clock around the two blocking hooks.  Layout of `out` (8 + 6 * pieces doubles):
 *   [0] pieces of the exchange   [1] 1 = congruence step 2 ran by packed-index range (staged)   [2] triangular inverse of the factor
 *   [3] congruence step 1 (staged) or both congruence steps (not staged)   [4] slab reduction   [5] all-reduce of the Gram matrix (host)
 *   [6] extraction into M and the vectors   [7] world
 *   then per piece k:  [8+6k] step 2 of the tile columns piece k needs   [+1] engine stream idle before piece k's Gram splits (the
 *   piece had not arrived: the exchange WAIT)   [+2] host time inside the wait hook   [+3] Gram splits of piece k   [+4] bytes this shard
 *   sent for piece k   [+5] host time from handing piece k to the transport until its wait returned
 * Returns the doubles written, 0 when no sharded build has run, -(doubles needed) when `cap` is too small, -1 for a bad shard. */
int HMiConeGetBuildProfile(hdsdp_cone *cone, int shard, double *out, int cap);
/* exchange buffers (device pointers, `world` chunks of *chunkCount doubles each); the caller may instead
 * supply its own (e.g. torch-allocated) buffers before the first HKKTBuildUp: world * chunkCount doubles of payload
 * plus 8192 doubles of slack behind it (the Gram kernel stages whole 128-row tiles without a row mask) */
hdsdp_retcode HMiConeGetExchangeBuffers(hdsdp_cone *cone, void **sendBuf, void **recvBuf, int64_t *chunkCount);
hdsdp_retcode HMiConeSetExchangeBuffers(hdsdp_cone *cone, void *sendBuf, void *recvBuf);
void *HMiKKTDeviceMatrix(hdsdp_kkt *HKKT, int64_t *ld);   /* device pointer of M (m x m, lower valid) */
/* nRows full symmetric rows of the device copy of M after a build (out: nRows x m doubles, row by row): what a caller that
 * keeps M on the device (host mirror off, m = 8000: 512 MB) needs for a row-subset check against tests/golden/full8000_rows.npz */
hdsdp_retcode HMiKKTGetRows(hdsdp_kkt *HKKT, int nRows, const int *rows, double *out);

/* ======================  several GPUs behind the C ABI (one process, one caller thread)  ======================
 * The reference's driver is a single-threaded process (tests/sdpasolve.c; interface/hdsdp_algo.c:1082-1101 calls
 * HKKTBuildUp / Factorize / Solve in sequence), so for it the sharding has to happen below these calls.
 * HMiSetDevices(n, ids), or HDSDP_MI355X_GPUS=n in the environment of an UNCHANGED driver (devices 0..n-1), configures a
 * device group before the first cone is created.  From then on HMiConeCreateSDP / HMiConeCreateSynthetic (called with
 * rank 0, world 1) return a GROUP CONE for every dense block of dimension >= HMiSetShardMinDim (default 512, env
 * HDSDP_MI355X_SHARD_MIN_N) that takes the congruence + Gram path: its constraint rows are dealt cyclically over n shards,
 * one per device, each with its own stream and host worker thread; every cone slot fans out to the shards, the all-to-all
 * and the all-reduce of the sharded build run over RCCL inside the library (grouped ncclSend/ncclRecv per piece of the
 * exchange, ncclAllReduce), and only shard 0 writes into the caller's Schur operator, which lives on ids[0].  Blocks on
 * the rank-one / sparse paths and small blocks stay plain single-device cones on ids[0].
 * Device ids may repeat (HDSDP_MI355X_LOOPBACK=1 with the environment form: shard r on device r mod visible): the
 * shards then share a device and exchange by device-to-device copies -- a rehearsal of world = n on a 1-GPU box; the same
 * copy transport can be forced between distinct, peer-accessible devices with HDSDP_MI355X_TRANSPORT=copy.
 * The process-per-GPU mode above (HMiConeSetExchange*, torchrun) is unaffected.  All return 0 on success. */
int HMiSetDevices(int nDevices, const int *deviceIds);
/* the same with the transport stated by the caller: 0 = device copies (peer access), 1 = RCCL (one device per shard; with
 * shared devices the group falls back to copies and HMiGetDeviceGroup says so), -1 = HDSDP_MI355X_TRANSPORT, else copies.
 * tests/test_gpu_group.py::test_config5_at_size_on_eight_devices asks for each in turn; bench.py --gpus N asks for RCCL once
 * HMiRcclGroupSelfTest has passed over the same devices in a child process. */
int HMiSetDevicesEx(int nDevices, const int *deviceIds, int transport);
int HMiGetDeviceGroup(int *deviceIds, int maxIds, int *transport);   /* returns the shard count (1 = no group); transport
                                                                        0 = device copies, 1 = RCCL, -1 = none */
void HMiSetShardMinDim(int nMin);
int HMiConeGetShardCount(hdsdp_cone *cone);                          /* 1 for a plain cone */
void HMiConeGetGroupTraffic(hdsdp_cone *cone, int64_t *bytesAllToAll, int64_t *bytesAllReduce);   /* sent by shard 0 so far */
/* One launch for a whole Phase-A pass of a SMALL rank-one block (BASELINE configs 2-3; csrc/small.hip).  The sequence
 *     HConeCheckIsInterior(tau, y) -> HKKTBuildUp(KKT_TYPE_INFEASIBLE) -> HKKTFactorize -> HKKTSolve x 3
 * (interface/hdsdp_algo.c:1082-1101: right-hand sides rhs, ASinv, ASinvRdSinv) is forty launches and six host
 * synchronisations on the call-by-call path, which is what a 100 x 100 block costs there, not its arithmetic.  For an
 * operator with ONE engine cone whose constraints are all rank one (n, m <= 128, at most 4 dense factors) this entry does
 * the whole sequence in one single-workgroup kernel and one synchronisation, with the same results: *isInterior and the
 * log-determinant of S, the operator's dASinvVec / dASinvRdSinvVec / dTraceSinv (and kktMatElem with the host mirror on),
 * the three solutions; the Cholesky factors of S and M are left in the objects HFpLinsys* / HKKTSolve use, so any later call
 * on the same state works as after the separate calls.  HMiKKTPhaseAEligible says whether the operator qualifies; the call
 * itself returns HDSDP_RETCODE_FAILED if not (nothing has been touched then).  The reference's call sites stay as they are:
 * this is an additional entry for drivers that want the latency, used by bench.py's small-config lines. */
int HMiKKTPhaseAEligible(hdsdp_kkt *HKKT);
hdsdp_retcode HMiKKTPhaseA(hdsdp_kkt *HKKT, double barHsdTau, double *rowDual, double *rhs, double *d1, double *d2, double *d3,
                           int *isInterior, double *logdet);
/* wall time the caller's thread has spent below this C ABI, by category (outermost entry only, so the categories add up):
 * [0] HKKTBuildUp of the M-forming types, [1] HKKTBuildUp(KKT_TYPE_CORRECTOR), [2] HKKTFactorize, [3] HKKTSolve,
 * [4] cone slots that assemble and factor S (update, interior checks, barrier, line search), [5] ratio test, [6] primal
 * recovery and the cone utilities, [7] HFpLinsys* called directly (the reference's CPU cones).  Returns the number of
 * categories.  HDSDP_MI355X_CALL_STATS=1 in the environment prints the table on stderr when the process exits. */
int HMiGetCallStats(double *seconds, int64_t *calls, int n);
const char *HMiCallStatName(int k);
void HMiResetCallStats(void);
int HMiRcclSelfTest(int device);   /* one-rank communicator on `device` (-1: the engine's): all-reduce + grouped send/recv, checked */
/* the same over a whole group of distinct devices, one host thread per device as the group's workers drive it: communicators,
 * an all-reduce, and the grouped ncclSend/ncclRecv of one piece of the sharded build's exchange, values compared bit for bit;
 * every wait is bounded by timeoutMs (<= 0: 60 s) and a rank that fails aborts its communicator so that no peer blocks.
 * 0 = passed, else the first failing stage (csrc/group_impl.h: rccl_group_self_test).  Touches no engine state. */
int HMiRcclGroupSelfTest(int nDevices, const int *deviceIds, int timeoutMs);

/* ================================  ingest (host only)  ================================
 * SDPA sparse format reader with the reference's semantics (interface/hdsdp_file_io.c:34-381): one CSC per
 * SDP block, shape n(n+1)/2 x (m+1), column 0 = C = -F0; rhs = the c vector.  Pointers stay valid until
 * HMiSDPAFree. */
typedef struct HMiSDPA_s HMiSDPA;
hdsdp_retcode HMiReadSDPA(const char *fname, HMiSDPA **out);
void HMiSDPAGetDims(const HMiSDPA *p, int *nConstrs, int *nBlks, int *nLpCols);
hdsdp_retcode HMiSDPAGetBlock(const HMiSDPA *p, int iBlk, int *dim, const int **beg, const int **idx,
                              const double **val);
/* the same with the 64-bit column pointers the reader keeps (a block may hold more than 2^31 - 1 entries; HMiSDPAGetBlock
 * refuses such a block): feeds HMiConeCreateSDP64 */
hdsdp_retcode HMiSDPAGetBlock64(const HMiSDPA *p, int iBlk, int *dim, const int64_t **beg, const int **idx,
                                const double **val);
const double *HMiSDPAGetRHS(const HMiSDPA *p);
void HMiSDPAFree(HMiSDPA **pp);


/* =====================================  environment switches  =====================================
 * Every environment variable the shipped library reads (tests/test_abi_cpu.py holds this table to the sources, both ways).
 * "other position" tests: tests/test_gpu_switches.py runs every device path under the non-default setting in a child
 * process and compares with the default run.  A -DHDM_DIAGNOSTICS build (python -m hdsdp_amd.build --diagnostics) reads
 * three more -- HDM_VAR, HDM_CONG2_DIRECT, HDM_DBG_SYNC -- for stamped kernels and timing ablations; the product never does.
 *
 *  variable                       default   meaning                                                      covered by
 *  -- configuration ---------------------------------------------------------------------------------------------------
 *  LOCAL_RANK                     0         device of the caller's context (process-per-GPU mode)         test_gpu_dist.py
 *  HDSDP_MI355X_GPUS              1         shards of the in-process device group (unchanged driver)      test_gpu_group.py::test_unchanged_driver_shards_by_environment
 *  HDSDP_MI355X_LOOPBACK          0         shards may share devices (rehearsal)                          same
 *  HDSDP_MI355X_SHARD_MIN_N       512       smallest block dimension that is sharded                      same
 *  HDSDP_MI355X_TRANSPORT         copy      device group transport when the caller states none            test_gpu_group.py::test_config5_at_size_on_eight_devices[rccl|copy]
 *                                           (HMiSetDevices, HDSDP_MI355X_GPUS): copy | rccl                (needs 8 devices), test_transport_request_is_honoured
 *  HDSDP_MI355X_A2A_PIECES        8         pieces of the sharded build's exchange                        test_gpu_switches.py
 *  HDSDP_MI355X_STAGED_A2A        1         step 2 by packed-index range, pieces leave as they finish     test_gpu_switches.py
 *  HDSDP_MI355X_FORCE_GEMM        0         every block takes the congruence + Gram path                  test_gpu_switches.py, test_gpu_parity.py
 *  HDSDP_MI355X_FORCE_PATH        -         force device path 0 / 1 / 2 (tests)                           test_gpu_parity.py::test_every_device_path_gives_the_same_numbers
 *  HDSDP_MI355X_SPARSE_KKT        1         0: always the dense Schur matrix                              test_gpu_switches.py
 *  HDSDP_MI355X_KKT_ENVELOPE      1         block-envelope factorisation of a sparse operator             test_gpu_switches.py
 *  HDSDP_MI355X_KKT_RCM           1         reverse Cuthill-McKee order of a sparse operator              test_gpu_switches.py
 *  HDSDP_MI355X_KKT_TILES         1         tile form of a sparse operator when it pays (bsparse.h)       test_gpu_switches.py
 *  HDSDP_MI355X_AFFINE_S          by cost   0 / 1 / 2: dual matrix short-cuts (engine_cone.h)             test_gpu_switches.py
 *  HDSDP_MI355X_SMALL_CHECK       1         one-launch interior check of small blocks                     test_gpu_switches.py
 *  HDSDP_MI355X_ZS                by cost   0 / 1 / 2: zero-suppressed copy for the S / dS sweeps (schur.h) test_gpu_switches.py
 *  HDSDP_MI355X_STREAM_A          by memory 0 / 1: synthetic constraint data resident / regenerated per     test_gpu_streamed.py
 *                                           batch (MiCone::streamed)
 *  HDM_TCAP_GIB                   32        GiB of congruence intermediates per launch group              test_gpu_switches.py, test_gpu_group.py
 *  HDM_BC                         1024      constraints per congruence launch (upper bound)               test_gpu_switches.py
 *  HDM_NSPLIT                     by size   slabs of the Gram product's K splits                          test_gpu_switches.py
 *  HDM_GRAM_KSTAGES               by size   stages (16 k) per (split, tile) job of the Gram product on one    test_gpu_switches.py (with HDM_NSPLIT=8: more
 *                                           device; more splits than slabs run in groups (gram_all)           splits than slabs)
 *  HDM_GRAM_QUEUE                 1         Gram jobs from ONE queue in split order; 0: one queue per XCD      test_gpu_switches.py
 *  HDSDP_MI355X_HOST_THREADS      min(16,   host threads of the ingest (presolve per column, staging of the    test_gpu_switches.py; the threaded form: test_gpu_ingest.py (syn2000x32)
 *                                 cores)    upload)
 *  HDM_SHARE_T_SLABS              1         intermediates and Gram slabs share one buffer (one GPU)       test_gpu_switches.py
 *  -- fallbacks kept reachable (the default is the fast form) ----------------------------------------------------------
 *  HDM_PERSIST                    1         persistent GEMM workgroups; 0: one workgroup per tile         test_gpu_switches.py, test_gpu_kernels.py
 *  HDM_PERSIST_RESERVE_CUS        0 / 8     CUs a persistent launch leaves to the collectives             test_gpu_switches.py, test_gpu_kernels.py
 *  HDM_DIAG_SWEEP                 1         register-sweep diagonal block; 0: LDS-panel kernel            test_gpu_switches.py, test_gpu_kernels.py
 *  HDM_CHOL_K128                  1         Cholesky panel / update as 64-row tiles straight from global  test_gpu_switches.py, test_gpu_kernels.py
 *                                           memory; 0: the general GEMM kernel
 *  HDM_TRSV_FLOW                  1         single-launch substitution; 0: per-block launches             test_gpu_switches.py, test_gpu_kernels.py
 *  HDM_TRSV_FLOW_FAIL_ONCE        0         test hook: throw the first single-launch result away          test_gpu_kernels.py::test_fallback_chains_of_the_factor_and_solve_kernels
 *  HDM_GRAPHS                     0         DIAGNOSTIC ONLY, not a supported position: 1 / 2 replay the    (diagnostic; still run by test_gpu_switches.py and
 *                                           factorisation / substitution chains from a hipGraph.  No gain     test_gpu_kernels.py so that it keeps working)
 *                                            and two runtime hazards on record: a replayed
 *                                           MEMSET node once wrote bytes that were not its captured value
 *                                           (profiles/r04_c_poison.txt; not reproducible stand-alone:
 *                                           tools/probes/memset_node_probe.hip, profiles/r05_e_memset_node_
 *                                           probe.txt -- the captured chains hold kernel nodes only since),
 *                                           and rocprofv3 dies beneath hipGraphLaunch (profiles/r04_a_*)
 *  HDM_SYM_COMBINE_SKY            1         S assembly in storage order; 0: element-indexed kernel        test_gpu_switches.py
 *  HDM_LANCZOS_WHOLE              1         small blocks: whole ratio test in one launch                  test_gpu_switches.py
 *  HDM_LANCZOS_FUSED              1         small blocks: three Lanczos steps per launch                  test_gpu_switches.py
 *  HDM_LANCZOS_GROUP              1         large blocks: steps between Ritz checks queued back to back   test_gpu_switches.py
 *  HDM_LANCZOS_BIG                1         large blocks: those steps in one launch of co-resident groups test_gpu_switches.py
 *  HDSDP_MI355X_PRELOAD           1         code objects are loaded on a helper thread when the first    test_gpu_switches.py
 *                                           context opens; 0: each on its first launch
 *  -- output only (no code path changes) --------------------------------------------------------------------------------
 *  HDSDP_MI355X_CALL_STATS        0         table of wall time below the C ABI at exit                    tools/small_driver_stats.sh
 *  HDSDP_MI355X_TRACE             0         synchronise and report after every entry                      (diagnostic)
 *  HDSDP_MI355X_RATIO_DEBUG       0         one line per ratio test: Lanczos steps, time                  (diagnostic)
 *  HDSDP_MI355X_AFFINE_DEBUG      0         one line per dual-matrix request that missed the last ratio    (diagnostic)
 *                                           test's line (engine_cone.h: cone_assemble)
 *  HDM_POISON                     0         new device memory is filled with 0xFF (NaN): a read of         (diagnostic; tools/poison_run.sh)
 *                                           never-written memory turns the results into NaNs
 */

/* ==================================  utilities  ================================== */
int HMiDeviceInit(int device);         /* hipSetDevice + stream; returns 0 on success */
int HMiDeviceSynchronize(void);
void *HMiStream(void);                 /* hipStream_t all kernels are launched on */
const char *HMiVersion(void);
/* timing of the last HKKTBuildUp / HKKTFactorize / HKKTSolve stages measured with HIP events on the
 * engine stream, milliseconds: [0] factor inverse, [1] congruence, [2] gram, [3] reduce+extract,
 * [4] gram kernel launches, [5] congruence kernel launches */
void HMiGetStageTimes(double *ms, int n);

/* live per-kernel timing with HIP events on the engine stream (for bench.py's roofline block): roles are
 * [0] helper GEMMs (Cholesky/TRTRI), [1] congruence step 1 (T = Linv A), [2] congruence step 2 without its full diagonal
 * tiles, [3] Gram, [4] the full diagonal tiles of congruence step 2 (a kernel of their own: P + P^T from one product).
 * Each array has 5 entries: total ms, algorithmic flops and launch count since the last call. */
void HMiSetKernelTiming(int on);
int HMiGetKernelTiming(double *ms, double *flops, int64_t *launches);
/* the same with a fourth array: the flops the launches' MFMA instructions EXECUTED (2048 per v_mfma_f64_16x16x4_f64), counted on the
 * host from the tile lists and the kernels' stage sequences -- issued / algorithmic is the granularity loss of a role (sub-blocks that
 * straddle a diagonal, live ranges of triangular K blocks, padding rows); bench.py: roofline.issued_over_valid */
int HMiGetKernelTimingEx(double *ms, double *flops, double *issued, int64_t *launches);
/* diagnostic builds (HDM_VAR=32): per-workgroup s_memtime stamps of launches with the given role, 8 words each */
void HMiSetDebugBuffer(void *dev, int role);

/* raw kernels (device pointers) exported for unit tests and micro-benchmarks */
int HMiGemmNT(const double *A, int64_t lda, int aKMajor, const double *B, int64_t ldb, int bKMajor, double *C,
              int64_t ldc, int M, int N, int K, double alpha, double beta, int kLimit, int lowerOnly);
int HMiPotrf(double *A_dev, int n, int64_t lda, int *info);  /* in place, lower */
/* blocked Cholesky + one solve of a host matrix (lower triangle, column-major, ld = n) with a block envelope: first[i] = first
   128-block column with an entry in block row i, NULL = dense; L_host (n x n) receives the factor if not NULL */
int HMiCholEnvelopeSolve(const double *A_host, int n, const int *first, const double *b, double *x, double *L_host, int *info);
/* host only: the reverse Cuthill-McKee order HKKTInit considers for a sparse Schur pattern (lower-triangular CSC in, perm[old] = new out) */
int HMiRcmOrder(int m, const int *colBeg, const int *rowIdx, int *perm);
/* how the operator's matrix will be factored: *permuted = 1 if the factor object holds P M P' (reverse Cuthill-McKee order of the
   sparse pattern), *fraction = 128-blocks inside the pattern's block envelope / blocks of the dense lower triangle */
void HMiKKTEnvelopeInfo(hdsdp_kkt *HKKT, int *permuted, double *fraction);
/* A sparse Schur operator whose factor's block pattern fills at most half of the triangle is kept in TILE form on the device
   (csrc/bsparse.h): rows reordered (dense rows last, reverse Cuthill-McKee for the rest), matrix and factor stored as the
   128 x 128 tiles inside the block pattern of the Cholesky factor only, factored left-looking by levels of the block elimination
   tree -- the device counterpart of the reference's sparse direct solver behind the aggregated-pattern CSC operator
   (interface/hdsdp_schur.c:46-139, linalg/hdsdp_linsolver.c:509-809).  Returns 1 and fills the outputs when HKKT is in that
   form: tiles stored, tiles of the dense lower triangle, levels, bytes of device memory of the tile stores. */
int HMiKKTTileInfo(hdsdp_kkt *HKKT, int *tiles, int64_t *denseTiles, int *levels, int64_t *bytes);
/* The tile-form factorisation is an LDL' WITHOUT pivoting in signed-Cholesky form, M = L~ S L~' with S = diag(+-1) -- what the
   reference's sparse direct solver does (external/qdldl.c; linalg/hdsdp_linsolver.c:596-626): an indefinite matrix factors and
   solves, only an exactly zero pivot fails HKKTFactorize, and a positive definite matrix gets its Cholesky factor bit for bit.
   Negative pivots of the operator's last factorisation (= negative eigenvalues of M, by inertia); -1 if HKKT is not in tile form
   or not factored. */
int HMiKKTNegativePivots(hdsdp_kkt *HKKT);
/* the tile-form factorisation and solve on a host matrix (lower-triangular CSC, diagonal entry first in every column):
   *info = 0 or first zero pivot + 1; stats[0..3] = block rows, tiles stored, levels, negative pivots; ms (may be NULL) =
   factorisation time, best of three */
int HMiBspSolve(int m, const int *colBeg, const int *rowIdx, const double *val, const double *b, double *x, int *info, int *stats,
                double *ms);
int HMiCholEnvelopeProbe(int n, int band, int reps, double *ms_dense, double *ms_env);   /* factorisation time of a block-banded matrix, dense vs on its envelope */
double HMiDiagBlockProbe(int variant, int reps);   /* us per 128 x 128 diagonal-block kernel of the Cholesky (0: LDS panels, 1: register sweep) */
double HMiMfmaPeakProbe(int iters);    /* measured fp64 MFMA TFLOP/s of a register-only loop */
/* GEMM-shaped issue probe: mode 0 = 16 accumulators x (4+4) operand registers, 1 = one operand pair */
double HMiMfmaIssueProbe(int mode, int wgPerCu, int iters);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HDSDP_MI355X_H */
