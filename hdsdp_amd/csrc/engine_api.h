// engine_api.h -- exported C ABI, second part: the HMi* cone entries, the device group entries, the fused Phase-A pass, call statistics
// Implementation header of engine.hip: included exactly once, there, in this order (inside extern "C"); split out of a 3 300-line file in round 3, nothing else changed.

hdsdp_retcode HMiConeCreateSDP(hdsdp_cone **pCone, int iCone, int nRow, int nCol, const int *coneMatBeg,
                               const int *coneMatIdx, const double *coneMatElem, int rank, int world) {
    return cone_create_csc(pCone, iCone, nRow, nCol, coneMatBeg, coneMatIdx, coneMatElem, rank, world);
}
hdsdp_retcode HMiConeCreateSDP64(hdsdp_cone **pCone, int iCone, int nRow, int nCol, const int64_t *coneMatBeg,
                                 const int *coneMatIdx, const double *coneMatElem, int rank, int world) {
    return cone_create_csc(pCone, iCone, nRow, nCol, coneMatBeg, coneMatIdx, coneMatElem, rank, world);
}

// ---------------------------------------------------------------- column-by-column ingest (64-bit totals)
struct HMiConeBuilder_s : MiConeBuilder {};
hdsdp_retcode HMiConeBuilderBegin(HMiConeBuilder **pBuilder, int iCone, int nRow, int nCol, int rank, int world) {
    if (!pBuilder || nRow < 1 || nCol < 1 || nCol > 65535 || world < 1 || rank < 0 || rank >= world) return HDSDP_RETCODE_FAILED;
    HMiConeBuilder_s *b = new HMiConeBuilder_s();
    b->iCone = iCone; b->rank = rank; b->world = world;
    b->blk.n = nCol; b->blk.m = nRow;
    b->blk.rows.assign((size_t) nRow, MiCoeff());
    b->seen.assign((size_t) nRow + 1, 0);
    *pBuilder = b;
    return HDSDP_RETCODE_OK;
}
hdsdp_retcode HMiConeBuilderAddColumn(HMiConeBuilder *b, int iCol, int64_t nnz, const int *packedIdx, const double *val) {
    if (!b || iCol < 0 || iCol > b->blk.m || nnz < 0 || (nnz > 0 && (!packedIdx || !val))) return HDSDP_RETCODE_FAILED;
    if (b->seen[iCol]) { fprintf(stderr, "[hdsdp_mi355x] HMiConeBuilderAddColumn: column %d was given before\n", iCol); return HDSDP_RETCODE_FAILED; }
    MiCoeff &dst = (iCol == 0) ? b->blk.obj : b->blk.rows[iCol - 1];
    if (mi_coeff_build(dst, b->blk.n, (long) nnz, packedIdx, val)) {
        fprintf(stderr, "[hdsdp_mi355x] HMiConeBuilderAddColumn: column %d has a packed index outside [0, n(n+1)/2) or too many entries\n", iCol);
        return HDSDP_RETCODE_FAILED;
    }
    b->seen[iCol] = 1;
    return HDSDP_RETCODE_OK;
}
int64_t HMiConeBuilderStored(const HMiConeBuilder *b) {
    if (!b) return 0;
    int64_t t = b->blk.obj.stored;
    for (const MiCoeff &c : b->blk.rows) t += c.stored;
    return t;
}
hdsdp_retcode HMiConeBuilderFinish(HMiConeBuilder **pBuilder, hdsdp_cone **pCone) {
    if (!pBuilder || !*pBuilder || !pCone) return HDSDP_RETCODE_FAILED;
    HMiConeBuilder_s *b = *pBuilder;
    mi_block_plan(b->blk);                       // (columns never given are zero matrices)
    const hdsdp_retcode rc = cone_from_block(pCone, b->iCone, b->blk, b->rank, b->world);
    delete b;
    *pBuilder = nullptr;
    return rc;
}
void HMiConeBuilderAbort(HMiConeBuilder **pBuilder) {
    if (pBuilder && *pBuilder) { delete *pBuilder; *pBuilder = nullptr; }
}

hdsdp_retcode HMiConeCreateSynthetic(hdsdp_cone **pCone, int iCone, int nCol, int nRow, int rank, int world) {
    if (!pCone || nRow < 1 || nCol < 1 || world < 1 || rank < 0 || rank >= world) return HDSDP_RETCODE_FAILED;
    if (group_configure_from_env()) return HDSDP_RETCODE_FAILED;
    if (ensure_ctx()) return HDSDP_RETCODE_FAILED;
    if (rank == 0 && world == 1 && group_wants_synthetic(nRow, nCol))
        return group_create_cone(pCone, iCone, nRow, nCol, nullptr, true);
    MiCone *c = nullptr;
    hdsdp_retcode rc = make_synth_cone(&c, nCol, nRow, rank, world);
    if (rc != HDSDP_RETCODE_OK) return rc;
    *pCone = new_cone_shell(c, iCone);
    return HDSDP_RETCODE_OK;
}

void HMiConeDestroy(hdsdp_cone **pCone) {
    if (!pCone || !*pCone) return;
    if ((*pCone)->coneDestroyData) (*pCone)->coneDestroyData(&(*pCone)->coneData);
    free(*pCone);
    *pCone = nullptr;
}
void HMiConeSetStart(hdsdp_cone *cone, double v) { cone->coneSetStart(cone->coneData, v); }
void HMiConeUpdate(hdsdp_cone *cone, double tau, double *y) { cone->coneUpdate(cone->coneData, tau, y); }
hdsdp_retcode HMiConeCheckIsInterior(hdsdp_cone *cone, double tau, double *y, int *isInterior) {
    return cone->coneInteriorCheck(cone->coneData, tau, y, isInterior);
}
hdsdp_retcode HMiConeGetLogBarrier(hdsdp_cone *cone, double tau, double *y, int whichBuffer, double *logdet) {
    return cone->coneGetBarrier(cone->coneData, tau, y, whichBuffer, logdet);
}
hdsdp_retcode HMiConeRatioTest(hdsdp_cone *cone, double dTauStep, double *dy, double dAdaRatio, int whichBuffer,
                               double *maxStep) {
    return cone->coneRatioTest(cone->coneData, dTauStep, dy, dAdaRatio, whichBuffer, maxStep);
}
void HMiLanczosStartVector(int n, double *v) { hdm_lanczos_start_vector(n, v); }
hdsdp_retcode HMiConeCheckIsInteriorExpert(hdsdp_cone *cone, double dCCoef, double dACoefScal, double *dACoef, double dEyeCoef,
                                           int whichBuffer, int *isInterior) {
    return cone->coneInteriorCheckExpert(cone->coneData, dCCoef, dACoefScal, dACoef, dEyeCoef, whichBuffer, isInterior);
}
hdsdp_retcode HMiConeAddStepToBufferAndCheck(hdsdp_cone *cone, double dStep, int whichBuffer, int *isInterior) {
    return cone->coneAxpyBufferAndCheck(cone->coneData, dStep, whichBuffer, isInterior);
}
double HMiConeGetCoeffNorm(hdsdp_cone *cone, int whichNorm) { return cone->coneGetCoeffNorm(cone->coneData, whichNorm); }
double HMiConeGetObjNorm(hdsdp_cone *cone, int whichNorm) { return cone->coneGetObjNorm(cone->coneData, whichNorm); }
void HMiConeScalByConstant(hdsdp_cone *cone, double dScal) { cone->coneScal(cone->coneData, dScal); }
void HMiConeComputeATimesXpy(hdsdp_cone *cone, double *dConePrimal, double *dATimesX) {
    cone->coneATimesXpy(cone->coneData, dConePrimal, dATimesX);
}
double HMiConeComputeXDotS(hdsdp_cone *cone, double *dConePrimal) { return cone->coneXDotS(cone->coneData, dConePrimal); }
double HMiConeComputeTraceCX(hdsdp_cone *cone, double *dConePrimal) { return cone->coneTraceCX(cone->coneData, dConePrimal); }
void HMiConeGetDual(hdsdp_cone *cone, double *dConeDual, double *dConeDual2) {
    cone->coneDRecover(cone->coneData, dConeDual, dConeDual2);
}
void HMiConeDetectFeature(hdsdp_cone *cone, double *rowRHS, int coneIntFeatures[20], double coneDblFeatures[20]) {   // hdsdp_conic.c:423-428
    cone->getstat(cone->coneData, rowRHS, coneIntFeatures, coneDblFeatures);
}
void HMiConeReduceResi(hdsdp_cone *cone, double dResiReduction) { cone->coneReduceResi(cone->coneData, dResiReduction); }
void HMiConeSetPerturb(hdsdp_cone *cone, double dDualPerturb) { cone->coneSetPerturb(cone->coneData, dDualPerturb); }
void HMiConeGetPrimal(hdsdp_cone *cone, double dBarrierMu, double *dRowDual, double *dRowDualStep, double *dConePrimal,
                      double *dConePrimal2) {
    cone->conePRecover(cone->coneData, dBarrierMu, dRowDual, dRowDualStep, dConePrimal, dConePrimal2);
}
void HMiConeGetPresolve(hdsdp_cone *cone, int *coefType, int *coefRank, int *coefNnz, int *kktPerm, int *kktStrategy,
                        int *objType) {
    MiCone *c = cone_data(cone);
    for (int i = 0; i < c->m && !c->synthetic; ++i) {
        if (coefType) coefType[i] = c->blk.rows[i].type;
        if (coefRank) coefRank[i] = c->blk.rows[i].rank;
        if (coefNnz) coefNnz[i] = c->blk.rows[i].nnz;
        if (kktPerm) kktPerm[i] = c->blk.perm[i];
        if (kktStrategy) kktStrategy[i] = c->blk.strategy[i];
    }
    if (objType) *objType = c->synthetic ? MI_COEFF_DENSE : c->blk.obj.type;
}
hdsdp_retcode HMiPresolveCSC(int nRow, int nCol, const int *coneMatBeg, const int *coneMatIdx,
                             const double *coneMatElem, int *coefType, int *coefRank, int *coefNnz, int *kktPerm,
                             int *kktStrategy, int *objType) {
    if (nRow < 1 || nCol < 1 || !coneMatBeg) return HDSDP_RETCODE_FAILED;
    MiBlockData blk;
    if (mi_block_from_csc(blk, nRow, nCol, coneMatBeg, coneMatIdx, coneMatElem)) return HDSDP_RETCODE_FAILED;
    for (int i = 0; i < nRow; ++i) {
        if (coefType) coefType[i] = blk.rows[i].type;
        if (coefRank) coefRank[i] = blk.rows[i].rank;
        if (coefNnz) coefNnz[i] = blk.rows[i].nnz;
        if (kktPerm) kktPerm[i] = blk.perm[i];
        if (kktStrategy) kktStrategy[i] = blk.strategy[i];
    }
    if (objType) *objType = blk.obj.type;
    return HDSDP_RETCODE_OK;
}
hdsdp_retcode HMiConeGetDualMatrix(hdsdp_cone *cone, double *S) {
    MiCone *c = cone_data(cone);
    if (hipMemcpy2DAsync(S, sizeof(double) * c->n, c->S, sizeof(double) * c->n16, sizeof(double) * c->n, c->n,
                         hipMemcpyDeviceToHost, g.stream) != hipSuccess) return HDSDP_RETCODE_FAILED;
    return hipStreamSynchronize(g.stream) == hipSuccess ? HDSDP_RETCODE_OK : HDSDP_RETCODE_FAILED;
}
hdsdp_retcode HMiConeGetTraces(hdsdp_cone *cone, double *trA) {
    MiCone *c = cone_data(cone);
    if (!c->trA) return HDSDP_RETCODE_FAILED;
    memcpy(trA, c->trA, sizeof(double) * c->m);
    return HDSDP_RETCODE_OK;
}
int HMiConeGetPath(hdsdp_cone *cone) { return cone_data(cone)->path; }
int HMiConeUseSweepCopy(hdsdp_cone *cone, int on) {
    MiCone *c = cone_data(cone);
    if (ensure_ctx()) return 1;
    c->pS_ok = c->pD_ok = false;     // the next request is assembled, not short-cut
    if (!on) { c->zs_state = -1; return 0; }
    if (!c->zs.val && cone_has_rows(c) && c->mloc > 0 &&
        hdm_zs_build_from([&](int q0, int nb) { return cone_rows(c, q0, nb); }, cone_batch(c), c->astride, c->mloc, c->astride, 1.0,
                          &c->zs, g.stream)) return 1;
    c->zs_state = c->zs.val ? 1 : -1;
    return c->zs.val ? 0 : 1;
}
int HMiConeGetStreaming(hdsdp_cone *cone, int *batchRows) {
    const MiCone *c = cone_data(cone);
    if (batchRows) *batchRows = c->streamed ? c->Bs : 0;
    return c->streamed ? 1 : 0;
}
int HMiConeSweepInfo(hdsdp_cone *cone, int64_t *values, int64_t *positions) {
    const MiCone *c = cone_data(cone);
    const bool on = (c->zs_state == 1);
    if (values) *values = on ? (int64_t) c->zs.nnz : 0;
    if (positions) *positions = on ? (int64_t) c->zs.sky * c->zs.m : 0;
    return on ? 1 : 0;
}

// ---------------------------------------------------------------- single-process multi-device mode (group_impl.h)
int HMiSetDevicesEx(int nDevices, const int *deviceIds, int transport) {
    if (nDevices < 1 || !deviceIds || transport < -1 || transport > GRP_RCCL) return 1;
    g_group_env_done = true;     // an explicit call overrides HDSDP_MI355X_GPUS
    // -1: the environment's choice (HDSDP_MI355X_TRANSPORT), else the default -- group_setup reads it
    return group_setup(nDevices, deviceIds, transport);
}
int HMiSetDevices(int nDevices, const int *deviceIds) { return HMiSetDevicesEx(nDevices, deviceIds, -1); }
int HMiGetDeviceGroup(int *deviceIds, int maxIds, int *transport) {
    if (!g_group) { if (transport) *transport = -1; if (deviceIds && maxIds > 0 && g_main.init) deviceIds[0] = g_main.device; return g_main.init ? 1 : 0; }
    for (int r = 0; r < g_group->W && r < maxIds; ++r) if (deviceIds) deviceIds[r] = g_group->dev[r];
    if (transport) *transport = g_group->transport;
    return g_group->W;
}
void HMiSetShardMinDim(int nMin) { if (g_group) g_group->min_n = nMin; }
int HMiConeGetShardCount(hdsdp_cone *cone) {
    return (cone && cone->coneBuildSchur == gc_build_schur) ? ((MiConeGroup *) cone->coneData)->G->W : 1;
}
// the last sharded build's profile of shard `shard` (a plain cone is its own shard 0): see include/hdsdp_mi355x.h for the layout
int HMiConeGetBuildProfile(hdsdp_cone *cone, int shard, double *out, int cap) {
    MiCone *c = nullptr;
    if (cone->coneBuildSchur == gc_build_schur) {
        MiConeGroup *cg = (MiConeGroup *) cone->coneData;
        if (shard < 0 || shard >= (int) cg->shard.size()) return -1;
        c = cg->shard[shard];
    } else {
        if (shard != 0) return -1;
        c = (MiCone *) cone->coneData;
    }
    const MiCone::BuildProfile &pf = c->prof;
    if (!pf.valid) return 0;
    const int need = 8 + 6 * pf.pieces;
    if (!out || cap < need) return -need;
    out[0] = pf.pieces; out[1] = pf.staged; out[2] = pf.invert; out[3] = pf.staged ? pf.step1 : pf.cong;
    out[4] = pf.reduce; out[5] = pf.allreduce_host; out[6] = pf.extract; out[7] = c->world;
    for (int k = 0; k < pf.pieces; ++k) {
        double *o = out + 8 + 6 * k;
        o[0] = pf.step2[k]; o[1] = pf.wait_gpu[k]; o[2] = pf.wait_host[k]; o[3] = pf.gram[k]; o[4] = pf.bytes[k]; o[5] = pf.flight[k];
    }
    return need;
}

void HMiConeGetGroupTraffic(hdsdp_cone *cone, int64_t *bytesAllToAll, int64_t *bytesAllReduce) {
    int64_t a = 0, b = 0;
    if (cone && cone->coneBuildSchur == gc_build_schur) { MiConeGroup *cg = (MiConeGroup *) cone->coneData; a = cg->bytes_a2a; b = cg->bytes_ar; }
    if (bytesAllToAll) *bytesAllToAll = a;
    if (bytesAllReduce) *bytesAllReduce = b;
}
// ---------------------------------------------------------------- fused small-block Phase-A pass (small.hip)
static int small_plan(MiCone *c) {
    MiCone::SmallPlan &sp = c->small;
    if (sp.state) return sp.state;
    sp.state = -1;
    if (c->path != PATH_R1 || c->world != 1 || c->synthetic || c->n > SMALL_P || c->mloc != c->m || c->m > SMALL_P) return -1;
    const int n = c->n, m = c->m;
    std::vector<int> fp(m + 1, 0), fi, dense_of(m, -1), dense_rows;
    std::vector<double> fv, sg(m, 0.0);
    for (int q = 0; q < m; ++q) {
        const MiCoeff &co = c->blk.rows[c->own[q]];
        if (co.type != MI_COEFF_SPR1 && co.type != MI_COEFF_DSR1) return -1;
        int nz = 0;
        for (int r = 0; r < n; ++r) nz += (co.factor[r] != 0.0);
        sg[q] = co.sign;
        if (nz > SMALL_SPMAX) {                       // dense factor: all n entries, in order
            if ((int) dense_rows.size() >= SMALL_NDENSE) return -1;
            dense_of[q] = (int) dense_rows.size();
            dense_rows.push_back(q);
            for (int r = 0; r < n; ++r) { fi.push_back(r); fv.push_back(co.factor[r]); }
        } else {
            for (int r = 0; r < n; ++r) if (co.factor[r] != 0.0) { fi.push_back(r); fv.push_back(co.factor[r]); }
        }
        fp[q + 1] = (int) fi.size();
    }
    dense_rows.resize(SMALL_NDENSE, 0);
    const size_t nf = std::max<size_t>(1, fi.size());
    if (hipMalloc((void **) &sp.fp, sizeof(int) * (m + 1)) != hipSuccess || hipMalloc((void **) &sp.fi, sizeof(int) * nf) != hipSuccess ||
        hipMalloc((void **) &sp.fv, sizeof(double) * nf) != hipSuccess || hipMalloc((void **) &sp.sgn, sizeof(double) * m) != hipSuccess ||
        hipMalloc((void **) &sp.dense_of, sizeof(int) * m) != hipSuccess ||
        hipMalloc((void **) &sp.dense_rows, sizeof(int) * SMALL_NDENSE) != hipSuccess ||
        hipHostMalloc((void **) &sp.io_host, sizeof(double) * (7 * (size_t) m + 16), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **) &sp.io_dev, sp.io_host, 0) != hipSuccess)
        return -1;
    if (hdm_memcpy_h2d_sync(sp.fp, fp.data(), sizeof(int) * (m + 1)) != hipSuccess ||
        (fi.size() && (hdm_memcpy_h2d_sync(sp.fi, fi.data(), sizeof(int) * fi.size()) != hipSuccess ||
                       hdm_memcpy_h2d_sync(sp.fv, fv.data(), sizeof(double) * fv.size()) != hipSuccess)) ||
        hdm_memcpy_h2d_sync(sp.sgn, sg.data(), sizeof(double) * m) != hipSuccess ||
        hdm_memcpy_h2d_sync(sp.dense_of, dense_of.data(), sizeof(int) * m) != hipSuccess ||
        hdm_memcpy_h2d_sync(sp.dense_rows, dense_rows.data(), sizeof(int) * SMALL_NDENSE) != hipSuccess)
        return -1;
    sp.ndense = 0;
    for (int q = 0; q < m; ++q) sp.ndense += (dense_of[q] >= 0);
    sp.state = 1;
    return 1;
}

int HMiKKTPhaseAEligible(hdsdp_kkt *HKKT) {
    if (!HKKT || HKKT->nCones != 1 || HKKT->isKKTSparse) return 0;
    hdsdp_cone *hc = HKKT->cones[0];
    if (hc->coneBuildSchur != cone_build_schur) return 0;
    MiCone *c = (MiCone *) hc->coneData;
    return (c->m == HKKT->nRow && small_plan(c) == 1) ? 1 : 0;
}

hdsdp_retcode HMiKKTPhaseA(hdsdp_kkt *HKKT, double barHsdTau, double *rowDual, double *rhs, double *d1, double *d2, double *d3,
                           int *isInterior, double *logdet) {
    StatScope stat_(ST_BUILD_M, __func__);
    if (!HMiKKTPhaseAEligible(HKKT)) return HDSDP_RETCODE_FAILED;
    MiCone *c = (MiCone *) HKKT->cones[0]->coneData;
    MiCone::SmallPlan &sp = c->small;
    MiKKTPriv *pv = priv_of(HKKT);
    MiLin *ls = (MiLin *) c->dualFactor->chol, *lm = (MiLin *) HKKT->kktM->chol;
    const int m = c->m, n = c->n;
    HIP_RC(hipStreamSynchronize(g.stream));            // the mapped block is about to be rewritten
    double *yin = sp.io_host, *bin = sp.io_host + m, *out = sp.io_host + 2 * (size_t) m;
    for (int i = 0; i < m; ++i) { yin[i] = rowDual ? rowDual[i] : 0.0; bin[i] = rhs ? rhs[i] : 0.0; }
    out[0] = -1.0;
    HdmSmallArgs a = {};
    a.n = n; a.m = m; a.C = c->Cfull; a.ldc = c->n16;
    a.fp = sp.fp; a.fi = sp.fi; a.fv = sp.fv; a.sgn = sp.sgn; a.dense_of = sp.dense_of; a.ndense = sp.ndense; a.dense_rows = sp.dense_rows;
    a.y = sp.io_dev; a.b = sp.io_dev + m; a.out = sp.io_dev + 2 * (size_t) m;
    a.tau = barHsdTau; a.eye = -c->Rd + c->perturb; a.Rd = c->Rd;
    a.Sout = c->S; a.lds = c->n16;
    c->pS_ok = false;                                  // (the pass writes S itself)
    a.LS = ls->ch.L; a.WS = ls->ch.Dinv; a.M = lm->Mdev; a.ldm = lm->ch.npad; a.LM = lm->ch.L; a.WM = lm->ch.Dinv;
    if (ls->ch.npad != SMALL_P || lm->ch.npad != SMALL_P) return HDSDP_RETCODE_FAILED;
    // (the operator's accumulators as HKKTBuildUp(KKT_TYPE_INFEASIBLE) leaves them, hdsdp_schur.c:141-165, :256-268: the
    // kernel itself zeroes what it does not fill of the 128 x 128 device matrix)
    RC(hdm_small_phase_a(a, g.stream));
    if (pv->mirror)
        HIP_RC(hipMemcpy2DAsync(HKKT->kktMatElem, sizeof(double) * m, lm->Mdev, sizeof(double) * SMALL_P, sizeof(double) * m, m,
                                hipMemcpyDeviceToHost, g.stream));
    HIP_RC(hipStreamSynchronize(g.stream));
    const int infoS = (int) out[0], infoM = (int) out[1];
    ls->ch.factored = (infoS == 0); ls->ch.have_inv = false;
    c->dualFactor->nFactorizes += 1;
    if (isInterior) *isInterior = (infoS == 0);
    if (infoS != 0) return HDSDP_RETCODE_OK;           // "not positive definite" is a value, not an error
    if (logdet) *logdet = out[2];
    memset(HKKT->dASinvVec, 0, sizeof(double) * m); memset(HKKT->dASinvRdSinvVec, 0, sizeof(double) * m);
    for (int i = 0; i < m; ++i) { HKKT->dASinvVec[i] = out[4 + i]; HKKT->dASinvRdSinvVec[i] = out[4 + m + i]; }
    HKKT->dTraceSinv = (c->Rd != 0.0) ? out[3] : 0.0;
    pv->Mdev_valid = true;
    lm->ch.factored = (infoM == 0); lm->ch.have_inv = false;
    lm->srcHost = nullptr; lm->srcDev = lm->Mdev; lm->srcLd = SMALL_P;
    HKKT->kktM->nFactorizes += 1;
    if (infoM != 0) {
        // the Schur matrix is not numerically positive definite: the multi-launch path's way out (pivoted solver) takes over
        fprintf(stderr, "[hdsdp_mi355x] HMiKKTPhaseA: Schur matrix is not positive definite (pivot %d); solving through HKKTFactorize\n", infoM);
        if (HKKTFactorize(HKKT) != HDSDP_RETCODE_OK) return HDSDP_RETCODE_FAILED;
        if (d1 && HKKTSolve(HKKT, rhs, d1) != HDSDP_RETCODE_OK) return HDSDP_RETCODE_FAILED;
        if (d2 && HKKTSolve(HKKT, HKKT->dASinvVec, d2) != HDSDP_RETCODE_OK) return HDSDP_RETCODE_FAILED;
        if (d3 && HKKTSolve(HKKT, HKKT->dASinvRdSinvVec, d3) != HDSDP_RETCODE_OK) return HDSDP_RETCODE_FAILED;
        return HDSDP_RETCODE_OK;
    }
    HKKT->kktM->nSolves += 3;
    if (d1) memcpy(d1, out + 4 + 2 * (size_t) m, sizeof(double) * m);
    if (d2) memcpy(d2, out + 4 + 3 * (size_t) m, sizeof(double) * m);
    if (d3) memcpy(d3, out + 4 + 4 * (size_t) m, sizeof(double) * m);
    const double *stamp = out + 4 + 5 * (size_t) m;
    for (int i = 0; i < 6; ++i) g.stage_ms[i] = (stamp[i + 1] - stamp[i]) * 1e-5;   // HMiGetStageTimes: 100 MHz ticks -> ms
    g.stage_ms[6] = (stamp[6] > stamp[0]) ? stamp[7] / ((stamp[6] - stamp[0]) * 10.0) : 0.0;   // shader clock during the pass, GHz
    return HDSDP_RETCODE_OK;
}

// host only (no device call): the reverse Cuthill-McKee order HKKTInit looks at for a sparse Schur pattern; lower-triangular CSC in,
// perm[old] = new out
int HMiRcmOrder(int m, const int *colBeg, const int *rowIdx, int *perm) {
    if (m <= 0 || !colBeg || !rowIdx || !perm) return 1;
    std::vector<int> beg(colBeg, colBeg + m + 1), idx(rowIdx, rowIdx + colBeg[m]);
    const std::vector<int> p = rcm_order(m, beg, idx);
    for (int i = 0; i < m; ++i) perm[i] = p[i];
    return 0;
}

// how the operator's matrix will be factored: *permuted = 1 if the factor object holds P M P' (reverse Cuthill-McKee order of
// the sparse pattern), *fraction = blocks inside the pattern's block envelope / blocks of the dense lower triangle (1 = dense)
int HMiKKTTileInfo(hdsdp_kkt *HKKT, int *tiles, int64_t *denseTiles, int *levels, int64_t *bytes) {
    if (tiles) *tiles = 0;
    if (denseTiles) *denseTiles = 0;
    if (levels) *levels = 0;
    if (bytes) *bytes = 0;
    if (!HKKT || !HKKT->kktM) return 0;
    MiLin *l = (MiLin *) HKKT->kktM->chol;
    if (!l->bsp) return 0;
    if (tiles) *tiles = l->bsp->ntiles;
    if (denseTiles) *denseTiles = l->bsp->dense_tiles();
    if (levels) *levels = l->bsp->nlevels;
    if (bytes) *bytes = (int64_t) l->bsp->bytes();
    return 1;
}
int HMiKKTNegativePivots(hdsdp_kkt *HKKT) {
    if (!HKKT || !HKKT->kktM) return -1;
    MiLin *l = (MiLin *) HKKT->kktM->chol;
    return (l->bsp && l->bsp->factored) ? l->bsp->negative : -1;
}
int HMiBspSolve(int m, const int *colBeg, const int *rowIdx, const double *val, const double *b, double *x, int *info, int *stats,
                double *ms) {
    if (ensure_ctx()) return 1;
    HdmBsp bs;
    int rc = 1;
    int *rows_d = nullptr, *cols_d = nullptr;
    double *vals_d = nullptr;
    do {
        if (bs.init(m, colBeg, rowIdx, 1.0)) break;
        const long nnz = colBeg[m];
        std::vector<int> cols((size_t) nnz);
        for (int c = 0; c < m; ++c) for (int q = colBeg[c]; q < colBeg[c + 1]; ++q) cols[q] = c;
        if (hipMalloc((void **) &rows_d, sizeof(int) * nnz) != hipSuccess || hipMalloc((void **) &cols_d, sizeof(int) * nnz) != hipSuccess ||
            hipMalloc((void **) &vals_d, sizeof(double) * nnz) != hipSuccess) break;
        if (hipMemcpy(rows_d, rowIdx, sizeof(int) * nnz, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(cols_d, cols.data(), sizeof(int) * nnz, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(vals_d, val, sizeof(double) * nnz, hipMemcpyHostToDevice) != hipSuccess) break;
        hipLaunchKernelGGL(mi_csc_scatter_kernel, dim3((unsigned) ((nnz + 255) / 256)), dim3(256), 0, g.stream, bs.view_M(), rows_d, cols_d, nnz, vals_d);
        int inf = 0;
        const int reps = ms ? 3 : 1;
        float best = 1e30f;
        bool bad = false;
        for (int r = 0; r < reps && !bad; ++r) {
            if (bs.load_M(g.stream)) { bad = true; break; }
            (void) hipEventRecord(g.ev[6], g.stream);
            if (bs.factor(g.stream, &inf)) { bad = true; break; }
            (void) hipEventRecord(g.ev[7], g.stream);
            (void) hipEventSynchronize(g.ev[7]);
            float e = 0.f;
            (void) hipEventElapsedTime(&e, g.ev[6], g.ev[7]);
            best = std::min(best, e);
        }
        if (bad) break;
        if (info) *info = inf;
        if (ms) *ms = best;
        if (stats) { stats[0] = bs.nb; stats[1] = bs.ntiles; stats[2] = bs.nlevels; stats[3] = bs.negative; }
        if (inf == 0 && b && x && bs.solve_host(b, x, g.stream)) break;
        rc = 0;
    } while (0);
    if (rows_d) (void) hipFree(rows_d);
    if (cols_d) (void) hipFree(cols_d);
    if (vals_d) (void) hipFree(vals_d);
    bs.destroy();
    return rc;
}
void HMiKKTEnvelopeInfo(hdsdp_kkt *HKKT, int *permuted, double *fraction) {
    if (permuted) *permuted = 0;
    if (fraction) *fraction = 1.0;
    if (!HKKT || !HKKT->kktM) return;
    MiLin *l = (MiLin *) HKKT->kktM->chol;
    if (l->bsp) { if (permuted) *permuted = 1; if (fraction) *fraction = (double) l->bsp->ntiles / (double) l->bsp->dense_tiles(); return; }
    if (permuted) *permuted = l->perm.empty() ? 0 : 1;
    if (fraction && !l->ch.env_colh.empty()) {
        double in = 0.0, all = 0.0;
        for (int k = 0; k < l->ch.nblk; ++k) { in += l->ch.env_colh[k] - k + 1; all += l->ch.nblk - k; }
        *fraction = in / all;
    }
}

int HMiGetCallStats(double *seconds, int64_t *calls, int n) {
    for (int k = 0; k < n && k < ST_N; ++k) { if (seconds) seconds[k] = g_stat_sec[k]; if (calls) calls[k] = g_stat_calls[k]; }
    return ST_N;
}
const char *HMiCallStatName(int k) { return (k >= 0 && k < ST_N) ? g_stat_name[k] : ""; }
void HMiResetCallStats(void) { for (int k = 0; k < ST_N; ++k) { g_stat_sec[k] = 0.0; g_stat_calls[k] = 0; } g_stat_nfn = 0; }
int HMiRcclSelfTest(int device) {
    if (ensure_ctx()) return 1;
    const int d = device < 0 ? g_main.device : device;
    return rccl_group_self_test(1, &d, 60000);
}
int HMiRcclGroupSelfTest(int nDevices, const int *deviceIds, int timeoutMs) {
    if (nDevices < 1 || !deviceIds) return 1;
    return rccl_group_self_test(nDevices, deviceIds, timeoutMs);
}
