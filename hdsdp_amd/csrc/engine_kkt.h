// engine_kkt.h -- exported C ABI, first part: utilities, HFpLinsys* and HKKT* (dense and sparse Schur operator, RCM order, host mirror, row access)
// Implementation header of engine.hip: included exactly once, there, in this order (inside extern "C"); split out of a 3 300-line file in round 3, nothing else changed.

const char *HMiVersion(void) { return "hdsdp-mi355x 0.1 (gfx950, fp64 MFMA)"; }

int HMiDeviceInit(int device) {
    if (g.init) return 0;
    if (device >= 0) {
        char buf[16];
        snprintf(buf, sizeof(buf), "%d", device);
        setenv("LOCAL_RANK", buf, 0);
    }
    return ensure_ctx();
}
int HMiDeviceSynchronize(void) {
    if (ensure_ctx()) return 1;
    HDM_HIP_CHECK(hipStreamSynchronize(g.stream));
    return 0;
}
void *HMiStream(void) { return ensure_ctx() ? nullptr : (void *) g.stream; }
void HMiSetKernelTiming(int on) { hdm_timing_enable(on); }
void HMiSetDebugBuffer(void *dev, int role) { hdm_set_debug_buffer((unsigned long long *) dev, role); }
int HMiGetKernelTiming(double *ms, double *flops, int64_t *launches) {
    long l[HDM_NROLES];
    if (hdm_timing_collect(ms, flops, l)) return 1;
    for (int r = 0; r < HDM_NROLES; ++r) launches[r] = l[r];
    return 0;
}
int HMiGetKernelTimingEx(double *ms, double *flops, double *issued, int64_t *launches) {
    long l[HDM_NROLES];
    if (hdm_timing_collect(ms, flops, l, issued)) return 1;
    for (int r = 0; r < HDM_NROLES; ++r) launches[r] = l[r];
    return 0;
}
void HMiGetStageTimes(double *ms, int n) {
    for (int i = 0; i < n && i < 8; ++i) ms[i] = g.stage_ms[i];
}

// ---------------------------------------------------------------- HFpLinsys*
hdsdp_retcode HFpLinsysCreate(hdsdp_linsys_fp **pHLin, int nCol, linsys_type Ltype) {
    if (!pHLin) return HDSDP_RETCODE_FAILED;
    switch (Ltype) {
        case HDSDP_LINSYS_DENSE_DIRECT:
        case HDSDP_LINSYS_DENSE_ITERATIVE:  // Schur system: solved by a direct blocked Cholesky here (stricter
            break;                          // than the reference's PCG to 1e-12, hdsdp_linsolver.c:1446-1588)
        case HDSDP_LINSYS_SPARSE_DIRECT:    // sparse dual matrix: CSC in, dense factorisation on the device (see MiLin)
            break;
        default:
            fprintf(stderr, "[hdsdp_mi355x] HFpLinsysCreate: linsys_type %d is not on the accelerated path "
                            "(sparse indefinite / iterative backends stay with the CPU reference; DENSE_INDEFINITE is only reached by switching)\n", (int) Ltype);
            return HDSDP_RETCODE_FAILED;
    }
    hdsdp_linsys_fp *h = (hdsdp_linsys_fp *) calloc(1, sizeof(hdsdp_linsys_fp));
    if (!h) return HDSDP_RETCODE_MEMORY;
    h->nCol = nCol;
    h->LinType = Ltype;
    h->cholCreate = lin_create;
    h->cholSetParam = lin_setparam;
    h->cholSymbolic = lin_symbolic;
    h->cholNumeric = lin_numeric;
    h->cholPsdCheck = lin_psdcheck;
    h->cholFSolve = lin_fsolve;
    h->cholBSolve = lin_bsolve;
    h->cholSolve = lin_solve;
    h->cholGetDiag = lin_getdiag;
    h->cholInvert = lin_invert;
    h->cholDestroy = lin_destroy;
    hdsdp_retcode rc = h->cholCreate(&h->chol, nCol);
    if (rc != HDSDP_RETCODE_OK) { free(h); return rc; }
    ((MiLin *) h->chol)->type = Ltype;
    ((MiLin *) h->chol)->csc_in = (Ltype == HDSDP_LINSYS_SPARSE_DIRECT);
    *pHLin = h;
    return HDSDP_RETCODE_OK;
}
void HFpLinsysSetParam(hdsdp_linsys_fp *HLin, double relTol, double absTol, int nThreads, int maxIter, int nRestartFreq) {
    (void) nThreads; (void) nRestartFreq;
    MiLin *l = (MiLin *) HLin->chol;
    l->relTol = relTol; l->absTol = absTol; l->maxIter = maxIter;  // recorded; the direct solve needs none
}
hdsdp_retcode HFpLinsysSymbolic(hdsdp_linsys_fp *HLin, int *colMatBeg, int *colMatIdx) {
    return HLin->cholSymbolic(HLin->chol, colMatBeg, colMatIdx);
}
hdsdp_retcode HFpLinsysNumeric(hdsdp_linsys_fp *HLin, int *colMatBeg, int *colMatIdx, double *colMatElem) {
    StatScope stat_(ST_LINSYS, __func__);
    // hdsdp_linsolver.c:2029-2044: a failed factorisation of the Schur system switches to the indefinite solver
    HLin->nFactorizes += 1;
    hdsdp_retcode rc = HLin->cholNumeric(HLin->chol, colMatBeg, colMatIdx, colMatElem);
    if (rc == HDSDP_RETCODE_FAILED && HLin->LinType == HDSDP_LINSYS_DENSE_ITERATIVE) {
        fprintf(stderr, "[hdsdp_mi355x] KKT system is almost indefinite. Switch to the pivoted (LDL-equivalent) solver.\n");
        rc = lin_switch_indefinite(HLin);
    }
    return rc;
}
hdsdp_retcode HFpLinsysSwitchToBackUp(hdsdp_linsys_fp *HLin) { (void) HLin; return HDSDP_RETCODE_OK; }
hdsdp_retcode HFpLinsysPsdCheck(hdsdp_linsys_fp *HLin, int *colMatBeg, int *colMatIdx, double *colMatElem, int *isPsd) {
    StatScope stat_(ST_LINSYS, __func__);
    HLin->nFactorizes += 1;
    return HLin->cholPsdCheck(HLin->chol, colMatBeg, colMatIdx, colMatElem, isPsd);
}
void HFpLinsysFSolve(hdsdp_linsys_fp *HLin, int nRhs, double *rhsVec, double *solVec) {
    StatScope stat_(ST_LINSYS, __func__);
    HLin->nSolves += 1;
    HLin->cholFSolve(HLin->chol, nRhs, rhsVec, solVec);
}
void HFpLinsysBSolve(hdsdp_linsys_fp *HLin, int nRhs, double *rhsVec, double *solVec) {
    StatScope stat_(ST_LINSYS, __func__);
    HLin->nSolves += 1;
    HLin->cholBSolve(HLin->chol, nRhs, rhsVec, solVec);
}
hdsdp_retcode HFpLinsysSolve(hdsdp_linsys_fp *HLin, int nRhs, double *rhsVec, double *solVec) {
    StatScope stat_(ST_LINSYS, __func__);
    // hdsdp_linsolver.c:2085-2110: NaN in the solution (or the right-hand side) counts as a failure, and a failed solve
    // of the Schur system switches to the indefinite solver and solves again
    hdsdp_retcode rc = HLin->cholSolve(HLin->chol, nRhs, rhsVec, solVec);
    if (solVec && solVec[0] != solVec[0]) rc = HDSDP_RETCODE_FAILED;
    if (rhsVec[0] != rhsVec[0]) rc = HDSDP_RETCODE_FAILED;
    if (rc != HDSDP_RETCODE_OK && HLin->LinType == HDSDP_LINSYS_DENSE_ITERATIVE) {
        fprintf(stderr, "[hdsdp_mi355x] KKT system is unstable. Switch to the pivoted (LDL-equivalent) solver.\n");
        if (lin_switch_indefinite(HLin) != HDSDP_RETCODE_OK) return HDSDP_RETCODE_FAILED;
        return HFpLinsysSolve(HLin, nRhs, rhsVec, solVec);
    }
    HLin->nSolves += 1;
    return rc;
}
hdsdp_retcode HFpLinsysGetDiag(hdsdp_linsys_fp *HLin, double *diagElem) { return HLin->cholGetDiag(HLin->chol, diagElem); }
void HFpLinsysInvert(hdsdp_linsys_fp *HLin, double *dFullMatrix, double *dAuxiMatrix) {
    StatScope stat_(ST_LINSYS, __func__);
    HLin->cholInvert(HLin->chol, dFullMatrix, dAuxiMatrix);
}
void HFpLinsysClear(hdsdp_linsys_fp *HLin) {
    if (!HLin) return;
    if (HLin->cholDestroy) HLin->cholDestroy(&HLin->chol);
    memset(HLin, 0, sizeof(hdsdp_linsys_fp));
}
void HFpLinsysDestroy(hdsdp_linsys_fp **pHLin) {
    if (!pHLin || !*pHLin) return;
    HFpLinsysClear(*pHLin);
    free(*pHLin);
    *pHLin = nullptr;
}

// ---------------------------------------------------------------- HKKT*
hdsdp_retcode HKKTCreate(hdsdp_kkt **pHKKT) {
    if (!pHKKT) return HDSDP_RETCODE_FAILED;
    hdsdp_kkt *k = (hdsdp_kkt *) calloc(1, sizeof(hdsdp_kkt));
    if (!k) return HDSDP_RETCODE_MEMORY;
    *pHKKT = k;
    return HDSDP_RETCODE_OK;
}

// (reverse Cuthill-McKee: hdm_rcm_order, bsparse.hip)
static std::vector<int> rcm_order(int m, const std::vector<int> &beg, const std::vector<int> &idx) { return hdm_rcm_order(m, beg, idx); }

// a linear-system object for a sparse Schur operator in TILE form: the same vtable, no dense factor behind it
static hdsdp_retcode linsys_create_tiles(hdsdp_linsys_fp **pHLin, int nCol, HdmBsp *bsp) {
    hdsdp_linsys_fp *h = (hdsdp_linsys_fp *) calloc(1, sizeof(hdsdp_linsys_fp));
    if (!h) return HDSDP_RETCODE_MEMORY;
    h->nCol = nCol; h->LinType = HDSDP_LINSYS_SPARSE_DIRECT;
    h->cholCreate = lin_create; h->cholSetParam = lin_setparam; h->cholSymbolic = lin_symbolic; h->cholNumeric = lin_numeric;
    h->cholPsdCheck = lin_psdcheck; h->cholFSolve = lin_fsolve; h->cholBSolve = lin_bsolve; h->cholSolve = lin_solve;
    h->cholGetDiag = lin_getdiag; h->cholInvert = lin_invert; h->cholDestroy = lin_destroy;
    MiLin *l = new MiLin();
    l->n = nCol; l->type = HDSDP_LINSYS_SPARSE_DIRECT; l->csc_in = true; l->bsp = bsp;
    h->chol = l;
    *pHLin = h;
    return HDSDP_RETCODE_OK;
}

hdsdp_retcode HKKTInit(hdsdp_kkt *HKKT, int nRow, int nCones, hdsdp_cone **cones) {
    if (ensure_ctx()) return HDSDP_RETCODE_FAILED;
    HKKT->nRow = nRow;
    HKKT->nCones = nCones;
    HKKT->cones = cones;
    int maxDim = 0;
    for (int i = 0; i < nCones; ++i) maxDim = std::max(maxDim, cones[i]->coneGetDim(cones[i]->coneData));
    HKKT->maxConeDim = maxDim;
    const size_t nn = (size_t) maxDim * maxDim;
    HKKT->invBuffer = (double *) calloc(nn, sizeof(double));
    HKKT->kktBuffer = (double *) calloc(nn, sizeof(double));
    HKKT->kktBuffer2 = (double *) calloc(nn, sizeof(double));
    HKKT->dASinvVec = (double *) calloc(nRow, sizeof(double));
    HKKT->dASinvCSinvVec = (double *) calloc(nRow, sizeof(double));
    HKKT->dASinvRdSinvVec = (double *) calloc(nRow, sizeof(double));
    HKKT->kktDiag = (double **) calloc(nRow, sizeof(double *));
    if (!HKKT->invBuffer || !HKKT->kktBuffer || !HKKT->kktBuffer2 || !HKKT->dASinvVec || !HKKT->dASinvCSinvVec ||
        !HKKT->dASinvRdSinvVec || !HKKT->kktDiag)
        return HDSDP_RETCODE_MEMORY;
    // Dense Schur matrix (hdsdp_schur.c:11-44) or the aggregated-pattern CSC (:46-139): the reference's own rule.  A cone
    // whose share of M reaches 0.3 m^2 entries makes it dense at once (:229-238); otherwise the columns' patterns are
    // collected from the cones (coneAddSymNz / coneGetKKTMap) and the CSC is kept unless it grows to 0.3 m^2 (:104-108).
    // HDSDP_MI355X_SPARSE_KKT=0 forces the dense matrix.
    MiKKTPriv *pv0 = priv_of(HKKT);
    HKKT->isKKTSparse = 1;
    const int64_t nDense = (int64_t) (0.3 * (double) nRow * (double) nRow);     // HDSDP_SPARSE_SCHUR_THRESHOLD, hdsdp.h:29
    if (const char *e = getenv("HDSDP_MI355X_SPARSE_KKT")) if (atoi(e) == 0) HKKT->isKKTSparse = 0;
    for (int i = 0; i < nCones && HKKT->isKKTSparse; ++i) {
        if (!cones[i]->coneGetSymNnz || !cones[i]->coneAddSymNz || !cones[i]->coneGetKKTMap ||
            cones[i]->coneGetSymNnz(cones[i]->coneData) >= nDense) HKKT->isKKTSparse = 0;
    }
    if (HKKT->isKKTSparse) {
        for (int i = 0; i < nCones; ++i)      // an engine cone may serve a second operator: its pattern walk starts over
            if (cones[i]->coneBuildSchur == cone_build_schur) ((MiCone *) cones[i]->coneData)->kkt_counted = 0;
        std::vector<int> beg((size_t) nRow + 1, 0), idx, col((size_t) nRow);
        for (int iCol = 0; iCol < nRow && HKKT->isKKTSparse; ++iCol) {
            std::fill(col.begin(), col.end(), 0);
            for (int i = 0; i < nCones; ++i) cones[i]->coneAddSymNz(cones[i]->coneData, iCol, col.data());
            for (int iRow = iCol; iRow < nRow; ++iRow)
                if (col[iRow]) { col[iRow] = (int) idx.size(); idx.push_back(iRow); }
            for (int i = 0; i < nCones; ++i) cones[i]->coneGetKKTMap(cones[i]->coneData, iCol, col.data());
            beg[iCol + 1] = (int) idx.size();
            if ((int64_t) idx.size() >= nDense) HKKT->isKKTSparse = 0;      // aggregation made it dense after all
        }
        // a constraint no cone has data for leaves an empty column: the reference stops there ("KKT solver detects an
        // empty column", :116-121); the engine keeps such an operator usable on the dense matrix, where the row simply
        // stays zero until a CPU cone (the bound cone's diagonal) or the regularisation fills it
        for (int iCol = 0; iCol < nRow && HKKT->isKKTSparse; ++iCol)
            if (beg[iCol] == beg[iCol + 1] || idx[beg[iCol]] != iCol) HKKT->isKKTSparse = 0;
        if (HKKT->isKKTSparse) {
            const size_t nnz = idx.size();
            HKKT->kktMatBeg = (int *) malloc(sizeof(int) * ((size_t) nRow + 1));
            HKKT->kktMatIdx = (int *) malloc(sizeof(int) * std::max<size_t>(1, nnz));
            if (!HKKT->kktMatBeg || !HKKT->kktMatIdx) return HDSDP_RETCODE_MEMORY;
            memcpy(HKKT->kktMatBeg, beg.data(), sizeof(int) * ((size_t) nRow + 1));
            memcpy(HKKT->kktMatIdx, idx.data(), sizeof(int) * nnz);
            if (hipHostMalloc((void **) &HKKT->kktMatElem, sizeof(double) * std::max<size_t>(1, nnz), hipHostMallocDefault) != hipSuccess)
                return HDSDP_RETCODE_MEMORY;
            memset(HKKT->kktMatElem, 0, sizeof(double) * nnz);
            for (int iCol = 0; iCol < nRow; ++iCol) HKKT->kktDiag[iCol] = &HKKT->kktMatElem[beg[iCol]];
            // Device storage of a sparse operator.  TILE form (bsparse.h) when it pays: the rows are reordered (dense rows last,
            // reverse Cuthill-McKee for the rest), and matrix and factor exist only as the 128 x 128 tiles inside the block
            // pattern of the Cholesky factor -- O(tiles of L) memory, a level-scheduled left-looking factorisation.  Taken when
            // those tiles are at most half of the dense lower triangle's (HDSDP_MI355X_KKT_TILES=0: never); otherwise the dense
            // device matrix factored on its block envelope, as in round 2.
            HdmBsp *bsp = nullptr;
            {
                static const bool use_tiles = [] { const char *e = getenv("HDSDP_MI355X_KKT_TILES"); return !(e && atoi(e) == 0); }();
                if (use_tiles && nRow > 2 * 128) {
                    bsp = new HdmBsp();
                    if (bsp->init(nRow, beg.data(), idx.data(), 0.5)) { bsp->destroy(); delete bsp; bsp = nullptr; }
                }
            }
            hdsdp_retcode rcs = bsp ? linsys_create_tiles(&HKKT->kktM, nRow, bsp) : HFpLinsysCreate(&HKKT->kktM, nRow, HDSDP_LINSYS_SPARSE_DIRECT);
            if (rcs != HDSDP_RETCODE_OK) return rcs;
            rcs = HFpLinsysSymbolic(HKKT->kktM, HKKT->kktMatBeg, HKKT->kktMatIdx);
            if (rcs != HDSDP_RETCODE_OK) return rcs;
            // the pattern as (row, column) pairs on the device
            std::vector<int> cols(nnz);
            for (int iCol = 0; iCol < nRow; ++iCol)
                for (int q = beg[iCol]; q < beg[iCol + 1]; ++q) cols[q] = iCol;
            pv0->nnz = (long) nnz;
            if (hipMalloc((void **) &pv0->sp_rows, sizeof(int) * std::max<size_t>(1, nnz)) != hipSuccess ||
                hipMalloc((void **) &pv0->sp_cols, sizeof(int) * std::max<size_t>(1, nnz)) != hipSuccess ||
                hipMalloc((void **) &pv0->sp_vals, sizeof(double) * std::max<size_t>(1, nnz)) != hipSuccess)
                return HDSDP_RETCODE_MEMORY;
            if (hdm_memcpy_h2d_sync(pv0->sp_rows, idx.data(), sizeof(int) * nnz) != hipSuccess ||
                hdm_memcpy_h2d_sync(pv0->sp_cols, cols.data(), sizeof(int) * nnz) != hipSuccess)
                return HDSDP_RETCODE_FAILED;
            // the pattern's block envelope: the blocked Cholesky of the (dense, mostly zero) device matrix stops each block
            // column where the envelope ends and the substitutions skip the blocks outside (HdmChol::set_envelope).  The
            // factor of a matrix fills inside its row envelope only, so nothing is approximated.  If a reverse Cuthill-McKee
            // order of the pattern makes the envelope cheaper, the factor object holds P M P' instead (MiLin::perm): the
            // builders keep writing M at the driver's indices, the pattern's entries are scattered to their permuted places
            // when the matrix is loaded for factorisation, and right-hand sides / solutions are permuted on the host.
            {
                static const bool use_env = [] { const char *e = getenv("HDSDP_MI355X_KKT_ENVELOPE"); return !(e && atoi(e) == 0); }();
                static const bool use_rcm = [] { const char *e = getenv("HDSDP_MI355X_KKT_RCM"); return !(e && atoi(e) == 0); }();
                MiLin *lm = (MiLin *) HKKT->kktM->chol;
                if (use_env && lm && !lm->bsp && lm->ch.nblk > 1) {
                    const int nb = lm->ch.nblk;
                    auto envelope = [&](const std::vector<int> *perm, std::vector<int> &first) {   // returns the factorisation's cost in block products
                        first.resize(nb);
                        for (int b = 0; b < nb; ++b) first[b] = b;
                        for (size_t q = 0; q < nnz; ++q) {
                            int r = idx[q], c = cols[q];
                            if (perm) { r = (*perm)[r]; c = (*perm)[c]; if (r < c) std::swap(r, c); }
                            const int br = r / 128, bc = c / 128;
                            if (bc < first[br]) first[br] = bc;
                        }
                        std::vector<int> colh(nb);
                        for (int k = 0; k < nb; ++k) colh[k] = k;
                        for (int b = 0; b < nb; ++b) for (int k = first[b]; k <= b; ++k) colh[k] = std::max(colh[k], b);
                        double cost = 0.0;
                        for (int k = 0; k < nb; ++k) { const double h = colh[k] - k; cost += 1.0 + h + 0.5 * h * (h + 1.0); }
                        return cost;
                    };
                    std::vector<int> first_nat, first_rcm, perm;
                    const double cost_nat = envelope(nullptr, first_nat);
                    double cost_rcm = INFINITY;
                    // (the reordering is looked for where it can pay: patterns up to 5e7 entries -- its adjacency lists take 8 bytes per
                    // entry on the host -- that do not already fill most of the triangle)
                    if (use_rcm && nnz <= 50000000 && (double) nnz < 0.15 * (double) nRow * nRow) { perm = rcm_order(nRow, beg, idx); cost_rcm = envelope(&perm, first_rcm); }
                    if (cost_rcm < 0.8 * cost_nat) {
                        std::vector<int> prow(nnz), pcol(nnz);
                        for (size_t q = 0; q < nnz; ++q) {
                            int r = perm[idx[q]], c = perm[cols[q]];
                            if (r < c) std::swap(r, c);
                            prow[q] = r; pcol[q] = c;
                        }
                        if (hipMalloc((void **) &pv0->sp_prow, sizeof(int) * std::max<size_t>(1, nnz)) != hipSuccess ||
                            hipMalloc((void **) &pv0->sp_pcol, sizeof(int) * std::max<size_t>(1, nnz)) != hipSuccess)
                            return HDSDP_RETCODE_MEMORY;
                        if (hdm_memcpy_h2d_sync(pv0->sp_prow, prow.data(), sizeof(int) * nnz) != hipSuccess ||
                            hdm_memcpy_h2d_sync(pv0->sp_pcol, pcol.data(), sizeof(int) * nnz) != hipSuccess)
                            return HDSDP_RETCODE_FAILED;
                        lm->perm = perm;
                        if (lm->ch.set_envelope(first_rcm.data())) return HDSDP_RETCODE_FAILED;
                    } else if (lm->ch.set_envelope(first_nat.data())) return HDSDP_RETCODE_FAILED;
                }
            }
            printf("    Using sparse Schur complement (%d nnzs)\n", HKKT->kktMatBeg[nRow]);
            if (bsp)
                fprintf(stderr, "[hdsdp_mi355x] sparse Schur operator in tile form: %d of %ld tiles (%.2f GiB instead of %.2f), %d levels\n", bsp->ntiles,
                        bsp->dense_tiles(), (double) bsp->bytes() / (1 << 30), 2.0 * 8.0 * (double) bsp->nb * 128 * bsp->nb * 128 / (1 << 30), bsp->nlevels);
        }
    }
    if (!HKKT->isKKTSparse) {
        // pinned so the D2H/H2D of M after BuildUp / before Factorize runs at PCIe rate
        if (hipHostMalloc((void **) &HKKT->kktMatElem, sizeof(double) * (size_t) nRow * nRow, hipHostMallocDefault) != hipSuccess)
            return HDSDP_RETCODE_MEMORY;
        memset(HKKT->kktMatElem, 0, sizeof(double) * (size_t) nRow * nRow);
        hdsdp_retcode rc = HFpLinsysCreate(&HKKT->kktM, nRow, HDSDP_LINSYS_DENSE_ITERATIVE);
        if (rc != HDSDP_RETCODE_OK) return rc;
        double acc = 1e-12;  // KKT_ACCURACY (hdsdp.h:27); loosened for big systems exactly as hdsdp_schur.c:21-35
        int iters = -1;
        if (nRow > 20000) { acc *= 100.0; iters = 500; } else if (nRow > 15000) { acc *= 50.0; iters = 450; }
        else if (nRow > 5000) { acc *= 5.0; iters = 120; }
        HFpLinsysSetParam(HKKT->kktM, 5.0 * acc, acc, -1, iters, -1);
        for (int i = 0; i < nRow; ++i) HKKT->kktDiag[i] = &HKKT->kktMatElem[i + (size_t) i * nRow];
    }
    // On the device M stays a dense m x m matrix in either case: the cones' builders write it at global (row, column)
    // indices, the blocked Cholesky factors it densely.  What the sparse form changes is the host side -- the matrix the
    // driver, the CPU cones (through kktMapping / kktDiag) and HKKTRegularize see is the nnz-long CSC, not m^2 doubles.
    MiLin *l = (MiLin *) HKKT->kktM->chol;
    if (!l->bsp) {
        const size_t mm = sizeof(double) * (size_t) l->ch.npad * l->ch.npad;
        if (hipMalloc((void **) &l->Mdev, mm) != hipSuccess) return HDSDP_RETCODE_MEMORY;
        if (hdm_memset_sync(l->Mdev, 0, mm) != hipSuccess) return HDSDP_RETCODE_FAILED;
    }
    MiKKTPriv *pv = priv_of(HKKT);
    if (hipMalloc((void **) &pv->vecs, sizeof(double) * (3 * (size_t) nRow + 4)) != hipSuccess) return HDSDP_RETCODE_MEMORY;
    pv->n_engine = pv->n_foreign = 0;
    for (int i = 0; i < nCones; ++i) {
        if (cones[i]->coneBuildSchur == cone_build_schur || cones[i]->coneBuildSchur == gc_build_schur) pv->n_engine += 1;
        else pv->n_foreign += 1;
    }
    HKKT->dPrimalX = nullptr;
    return HDSDP_RETCODE_OK;
}

static hdsdp_retcode kkt_clean(hdsdp_kkt *HKKT, int typeKKT) {  // hdsdp_schur.c:141-165
    const int m = HKKT->nRow;
    MiKKTPriv *pv = priv_of(HKKT);
    memset(HKKT->dASinvVec, 0, sizeof(double) * m);
    memset(HKKT->dASinvRdSinvVec, 0, sizeof(double) * m);
    if (typeKKT == KKT_TYPE_HOMOGENEOUS) {
        memset(HKKT->dASinvCSinvVec, 0, sizeof(double) * m);
        HKKT->dCSinv = HKKT->dCSinvCSinv = HKKT->dCSinvRdSinv = 0.0;
    }
    HKKT->dTraceSinv = 0.0;
    if (hipMemsetAsync(pv->vecs, 0, sizeof(double) * (3 * (size_t) m + 4), g.stream) != hipSuccess) return HDSDP_RETCODE_FAILED;
    if (typeKKT != KKT_TYPE_CORRECTOR) {
        MiLin *l = (MiLin *) HKKT->kktM->chol;
        if (l->bsp) { if (l->bsp->zero_M(g.stream)) return HDSDP_RETCODE_FAILED; }
        else if (hipMemsetAsync(l->Mdev, 0, sizeof(double) * (size_t) l->ch.npad * l->ch.npad, g.stream) != hipSuccess)
            return HDSDP_RETCODE_FAILED;
        if (HKKT->isKKTSparse) memset(HKKT->kktMatElem, 0, sizeof(double) * (size_t) HKKT->kktMatBeg[m]);   // (CPU cones add into it)
        // (dense host matrix: CPU cones add into it, so it starts from zero -- but with engine cones only, kkt_pull's copy
        // of the whole m x m device matrix replaces every entry, and 8 m^2 bytes of host memset per call are saved: 4 ms at
        // m = 2000, twice per iteration of the reference's driver)
        else if (pv->mirror && !(pv->n_foreign == 0 && pv->n_engine > 0)) memset(HKKT->kktMatElem, 0, sizeof(double) * (size_t) m * m);
    }
    return HDSDP_RETCODE_OK;
}

static hdsdp_retcode kkt_pull(hdsdp_kkt *HKKT, int typeKKT) {
    // device accumulators -> the host fields the driver and the CPU cones read (def_hdsdp_schur.h:44-61)
    const int m = HKKT->nRow;
    MiKKTPriv *pv = priv_of(HKKT);
    std::vector<double> h(3 * (size_t) m + 4);
    if (hipMemcpyAsync(h.data(), pv->vecs, sizeof(double) * h.size(), hipMemcpyDeviceToHost, g.stream) != hipSuccess)
        return HDSDP_RETCODE_FAILED;
    // M: every cone ACCUMULATES (hdsdp_schur.c:256-268).  The engine's cones did so on the device, foreign (CPU) cones
    // straight into kktMatElem: with only engine cones the device matrix simply replaces the (zeroed) host one, with
    // only foreign cones there is nothing to bring back, and in the mixed case the device part is added to the host part.
    bool add_M = false;
    size_t mcount = 0;     // entries of the host matrix that came back through Mtmp
    if (typeKKT != KKT_TYPE_CORRECTOR && pv->mirror && pv->n_engine > 0) {
        long ld = 0;
        double *Mdev = kkt_Mdev(HKKT, &ld);
        double *dst = HKKT->kktMatElem;
        mcount = HKKT->isKKTSparse ? (size_t) pv->nnz : (size_t) m * m;
        if (pv->n_foreign > 0) {
            if (!pv->Mtmp && hipHostMalloc((void **) &pv->Mtmp, sizeof(double) * std::max<size_t>(1, mcount)) != hipSuccess) return HDSDP_RETCODE_MEMORY;
            dst = pv->Mtmp;
            add_M = true;
        }
        if (HKKT->isKKTSparse) {
            // the pattern's entries of the dense device matrix (an engine cone only writes inside the pattern it declared)
            if (pv->nnz > 0) {
                hipLaunchKernelGGL(mi_csc_gather_kernel, dim3((unsigned) ((pv->nnz + 255) / 256)), dim3(256), 0, g.stream, kkt_view(HKKT),
                                   pv->sp_rows, pv->sp_cols, pv->nnz, pv->sp_vals);
                if (hipMemcpyAsync(dst, pv->sp_vals, sizeof(double) * (size_t) pv->nnz, hipMemcpyDeviceToHost, g.stream) != hipSuccess)
                    return HDSDP_RETCODE_FAILED;
            }
        } else if (hipMemcpy2DAsync(dst, sizeof(double) * m, Mdev, sizeof(double) * ld, sizeof(double) * m, m,
                                    hipMemcpyDeviceToHost, g.stream) != hipSuccess)
            return HDSDP_RETCODE_FAILED;
    }
    if (hipStreamSynchronize(g.stream) != hipSuccess) return HDSDP_RETCODE_FAILED;
    if (add_M) {
        if (HKKT->isKKTSparse) for (size_t q = 0; q < mcount; ++q) HKKT->kktMatElem[q] += pv->Mtmp[q];
        else
            for (int j = 0; j < m; ++j)                    // lower triangle, column-major
                for (int i = j; i < m; ++i) HKKT->kktMatElem[i + (size_t) j * m] += pv->Mtmp[i + (size_t) j * m];
    }
    for (int i = 0; i < m; ++i) {
        HKKT->dASinvVec[i] += h[i];
        HKKT->dASinvRdSinvVec[i] += h[m + i];
        if (typeKKT == KKT_TYPE_HOMOGENEOUS) HKKT->dASinvCSinvVec[i] += h[2 * (size_t) m + i];
    }
    if (typeKKT != KKT_TYPE_CORRECTOR) HKKT->dTraceSinv += h[3 * (size_t) m];
    if (typeKKT == KKT_TYPE_HOMOGENEOUS) {
        HKKT->dCSinv += h[3 * (size_t) m + 1];
        HKKT->dCSinvCSinv += h[3 * (size_t) m + 2];
        HKKT->dCSinvRdSinv += h[3 * (size_t) m + 3];
    }
    pv->Mdev_valid = (typeKKT != KKT_TYPE_CORRECTOR) ? true : pv->Mdev_valid;
    return HDSDP_RETCODE_OK;
}

hdsdp_retcode HKKTBuildUp(hdsdp_kkt *HKKT, int typeKKT) {
    StatScope stat_(typeKKT == KKT_TYPE_CORRECTOR ? ST_BUILD_CORR : ST_BUILD_M, __func__);
    hdsdp_retcode rc = kkt_clean(HKKT, typeKKT);
    if (rc != HDSDP_RETCODE_OK) return rc;
    for (int i = 0; i < HKKT->nCones; ++i) {
        hdsdp_cone *c = HKKT->cones[i];
        rc = c->coneBuildSchur(c->coneData, c->iCone, HKKT, typeKKT);  // == HConeBuildSchurComplement
        if (stat_trace()) {
            const hipError_t e = hipDeviceSynchronize();
            fprintf(stderr, "[hdsdp_mi355x trace]   build type %d, cone %d of %d -> rc %d, %s\n", typeKKT, i, HKKT->nCones, (int) rc,
                    e == hipSuccess ? "ok" : hipGetErrorName(e));
        }
        if (rc != HDSDP_RETCODE_OK) return rc;
    }
    return kkt_pull(HKKT, typeKKT);
}

hdsdp_retcode HKKTBuildUpExtraCone(hdsdp_kkt *HKKT, hdsdp_cone *cone, int typeKKT) {
    StatScope stat_(typeKKT == KKT_TYPE_CORRECTOR ? ST_BUILD_CORR : ST_BUILD_M, __func__);
    // CPU cones (bound / LP, hdsdp_conic_bound.c:201-249) write straight into the host fields
    return cone->coneBuildSchur(cone->coneData, cone->iCone, HKKT, typeKKT);
}

hdsdp_retcode HKKTBuildUpFixed(hdsdp_kkt *HKKT, int typeKKT, int kktStrategy) {
    StatScope stat_(typeKKT == KKT_TYPE_CORRECTOR ? ST_BUILD_CORR : ST_BUILD_M, __func__);
    hdsdp_retcode rc = kkt_clean(HKKT, typeKKT);
    if (rc != HDSDP_RETCODE_OK) return rc;
    for (int i = 0; i < HKKT->nCones; ++i) {
        hdsdp_cone *c = HKKT->cones[i];
        rc = c->coneBuildSchurFixed(c->coneData, c->iCone, HKKT, typeKKT, kktStrategy);
        if (rc != HDSDP_RETCODE_OK) return rc;
    }
    return kkt_pull(HKKT, typeKKT);
}

void HKKTExport(hdsdp_kkt *HKKT, double *dKKTASinvVec, double *dKKTASinvRdSinvVec, double *dKKTASinvCSinvVec,
                double *dCSinvCSinv, double *dCSinv, double *dCSinvRdCSinv, double *dTraceSinv) {
    const size_t b = sizeof(double) * (size_t) HKKT->nRow;
    if (dKKTASinvVec) memcpy(dKKTASinvVec, HKKT->dASinvVec, b);
    if (dKKTASinvRdSinvVec) memcpy(dKKTASinvRdSinvVec, HKKT->dASinvRdSinvVec, b);
    if (dKKTASinvCSinvVec) memcpy(dKKTASinvCSinvVec, HKKT->dASinvCSinvVec, b);
    if (dCSinvCSinv) *dCSinvCSinv = HKKT->dCSinvCSinv;
    if (dCSinv) *dCSinv = HKKT->dCSinv;
    if (dCSinvRdCSinv) *dCSinvRdCSinv = HKKT->dCSinvRdSinv;
    if (dTraceSinv) *dTraceSinv = HKKT->dTraceSinv;
}

static HdmMatView dense_view(double *base, long ld) { HdmMatView v; v.base = base; v.ld = ld; return v; }

hdsdp_retcode HKKTFactorize(hdsdp_kkt *HKKT) {
    StatScope stat_(ST_FACTORIZE, __func__);
    // hdsdp_schur.c:328-336.  With the host mirror on, the host matrix is authoritative (the driver and
    // the CPU cones may have touched it through kktDiag / kktMatElem); otherwise factor the device copy.
    MiKKTPriv *pv = priv_of(HKKT);
    MiLin *l = (MiLin *) HKKT->kktM->chol;
    HKKT->kktM->nFactorizes += 1;
    int info = 0;
    if (l->bsp) {
        // tile form: the factor store is filled from the host CSC (host mirror on: the driver and the CPU cones may have touched
        // it) or from the accumulation store, then factored level by level (bsparse.hip)
        if (pv->mirror) {
            if (l->bsp->zero_L(g.stream)) return HDSDP_RETCODE_FAILED;
            if (pv->nnz > 0) {
                if (hipMemcpyAsync(pv->sp_vals, HKKT->kktMatElem, sizeof(double) * (size_t) pv->nnz, hipMemcpyHostToDevice, g.stream) != hipSuccess)
                    return HDSDP_RETCODE_FAILED;
                hipLaunchKernelGGL(mi_csc_scatter_kernel, dim3((unsigned) ((pv->nnz + 255) / 256)), dim3(256), 0, g.stream, l->bsp->view_L(),
                                   pv->sp_rows, pv->sp_cols, pv->nnz, pv->sp_vals);
            }
        } else {
            if (!pv->Mdev_valid || l->bsp->load_M(g.stream)) return HDSDP_RETCODE_FAILED;
        }
        // LDL' like the reference's sparse direct solver (linalg/hdsdp_linsolver.c:596-626 over external/qdldl.c): an indefinite
        // matrix factors and the solves go on with the signed factor; only a pivot that is exactly zero is a failure
        int nneg = 0;
        if (l->bsp->factor(g.stream, &info, &nneg)) return HDSDP_RETCODE_FAILED;
        if (info != 0) {
            fprintf(stderr, "[hdsdp_mi355x] HKKTFactorize: sparse Schur matrix (tile form): zero pivot at row %d of the reordered matrix\n", info);
            return HDSDP_RETCODE_FAILED;
        }
        return HDSDP_RETCODE_OK;
    }
    if (pv->mirror && HKKT->isKKTSparse) {
        // the host CSC is authoritative: its values go up (nnz doubles) and are scattered over the zeroed dense device
        // matrix, which is then factored like the dense operator's
        if (hipMemsetAsync(l->Mdev, 0, sizeof(double) * (size_t) l->ch.npad * l->ch.npad, g.stream) != hipSuccess) return HDSDP_RETCODE_FAILED;
        if (pv->nnz > 0) {
            if (hipMemcpyAsync(pv->sp_vals, HKKT->kktMatElem, sizeof(double) * (size_t) pv->nnz, hipMemcpyHostToDevice, g.stream) != hipSuccess)
                return HDSDP_RETCODE_FAILED;
            hipLaunchKernelGGL(mi_csc_scatter_kernel, dim3((unsigned) ((pv->nnz + 255) / 256)), dim3(256), 0, g.stream,
                               dense_view(l->Mdev, l->ch.npad), pv->sp_rows, pv->sp_cols, pv->nnz, pv->sp_vals);
        }
        pv->Mdev_valid = true;
        l->srcHost = nullptr; l->srcDev = l->Mdev; l->srcLd = l->ch.npad;
    } else if (pv->mirror) {
        l->srcHost = HKKT->kktMatElem; l->srcDev = nullptr; l->srcLd = HKKT->nRow;
    } else {
        if (!pv->Mdev_valid) return HDSDP_RETCODE_FAILED;
        l->srcHost = nullptr; l->srcDev = l->Mdev; l->srcLd = l->ch.npad;
    }
    if (l->indef) return lin_factor_indef(l);     // switched earlier: stays switched (hdsdp_linsolver.c:1838)
    if (pv->mirror && !HKKT->isKKTSparse) {
        if (l->ch.load_host(HKKT->kktMatElem, HKKT->nRow, g.stream)) return HDSDP_RETCODE_FAILED;
    } else if (HKKT->isKKTSparse && !l->perm.empty()) {
        // the factor object holds P M P': the pattern's entries (already in sp_vals when they came up from the host CSC,
        // gathered from the device matrix otherwise) go to their permuted places in a zeroed image
        if (!pv->mirror && pv->nnz > 0)
            hipLaunchKernelGGL(mi_csc_gather_kernel, dim3((unsigned) ((pv->nnz + 255) / 256)), dim3(256), 0, g.stream,
                               dense_view(l->Mdev, l->ch.npad), pv->sp_rows, pv->sp_cols, pv->nnz, pv->sp_vals);
        if (hipMemsetAsync(l->ch.L, 0, sizeof(double) * (size_t) l->ch.npad * l->ch.npad, g.stream) != hipSuccess) return HDSDP_RETCODE_FAILED;
        if (pv->nnz > 0)
            hipLaunchKernelGGL(mi_csc_scatter_kernel, dim3((unsigned) ((pv->nnz + 255) / 256)), dim3(256), 0, g.stream,
                               dense_view(l->ch.L, l->ch.npad), pv->sp_prow, pv->sp_pcol, pv->nnz, pv->sp_vals);
        if (l->ch.finish_load(g.stream)) return HDSDP_RETCODE_FAILED;
    } else {
        if (l->ch.load_device(l->Mdev, l->ch.npad, g.stream)) return HDSDP_RETCODE_FAILED;
    }
    if (l->ch.factor(g.stream, &info)) return HDSDP_RETCODE_FAILED;
    if (info != 0) {
        // hdsdp_linsolver.c:2034-2039: the Schur system falls back to the symmetric-indefinite solver
        fprintf(stderr, "[hdsdp_mi355x] HKKTFactorize: Schur matrix is not positive definite (pivot %d). "
                        "Switch to the pivoted (LDL-equivalent) solver.\n", info);
        return lin_switch_indefinite(HKKT->kktM);
    }
    return HDSDP_RETCODE_OK;
}

hdsdp_retcode HKKTSolve(hdsdp_kkt *HKKT, double *dRhsVec, double *dLhsVec) {
    StatScope stat_(ST_SOLVE, __func__);
    return HFpLinsysSolve(HKKT->kktM, 1, dRhsVec, dLhsVec);
}

void HKKTRegularize(hdsdp_kkt *HKKT, double dKKTReg) {  // hdsdp_schur.c:348-373
    MiKKTPriv *pv = priv_of(HKKT);
    if (!pv->mirror) {
        // device-resident M (HMiKKTSetHostMirror(.., 0)): same rule on the device copy; the diagonal (m doubles)
        // makes the round trip, the matrix does not
        MiLin *l = (MiLin *) HKKT->kktM->chol;
        if (l->bsp) {
            // tile form: the pattern's values make the round trip (nnz doubles), the rule runs on the diagonal entries (the first of
            // every column), and they go back into the accumulation store
            if (!pv->Mdev_valid || pv->nnz <= 0) return;
            const int m = HKKT->nRow;
            std::vector<double> v((size_t) pv->nnz);
            hipLaunchKernelGGL(mi_csc_gather_kernel, dim3((unsigned) ((pv->nnz + 255) / 256)), dim3(256), 0, g.stream, l->bsp->view_M(),
                               pv->sp_rows, pv->sp_cols, pv->nnz, pv->sp_vals);
            if (hipMemcpyAsync(v.data(), pv->sp_vals, sizeof(double) * v.size(), hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
                hipStreamSynchronize(g.stream) != hipSuccess) return;
            double mn = INFINITY;
            for (int i = 0; i < m; ++i) mn = std::min(mn, v[HKKT->kktMatBeg[i]]);
            const double reg = std::min(dKKTReg * mn, 1e-05);
            if (reg < 1e-14) return;
            for (int i = 0; i < m; ++i) v[HKKT->kktMatBeg[i]] += reg;
            if (hipMemcpyAsync(pv->sp_vals, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice, g.stream) != hipSuccess) return;
            hipLaunchKernelGGL(mi_csc_scatter_kernel, dim3((unsigned) ((pv->nnz + 255) / 256)), dim3(256), 0, g.stream, l->bsp->view_M(),
                               pv->sp_rows, pv->sp_cols, pv->nnz, pv->sp_vals);
            (void) hipStreamSynchronize(g.stream);      // v is read by an asynchronous copy
            return;
        }
        if (!pv->Mdev_valid || !l->Mdev) return;
        const int m = HKKT->nRow;
        const size_t pitch = sizeof(double) * ((size_t) l->ch.npad + 1);
        std::vector<double> d(m);
        if (hipStreamSynchronize(g.stream) != hipSuccess) return;
        if (hipMemcpy2D(d.data(), sizeof(double), l->Mdev, pitch, sizeof(double), m, hipMemcpyDeviceToHost) != hipSuccess) return;
        double mn = INFINITY;
        for (int i = 0; i < m; ++i) mn = std::min(mn, d[i]);
        double reg = std::min(dKKTReg * mn, 1e-05);
        if (reg < 1e-14) return;
        for (int i = 0; i < m; ++i) d[i] += reg;
        (void) hipMemcpy2D(l->Mdev, pitch, d.data(), sizeof(double), sizeof(double), m, hipMemcpyHostToDevice);
        return;
    }
    double mn = INFINITY;
    for (int i = 0; i < HKKT->nRow; ++i) mn = std::min(mn, *HKKT->kktDiag[i]);
    dKKTReg = std::min(dKKTReg * mn, 1e-05);
    if (dKKTReg < 1e-14) dKKTReg = 0.0;
    for (int i = 0; i < HKKT->nRow; ++i) *HKKT->kktDiag[i] += dKKTReg;
}

void HKKTRegisterPSDP(hdsdp_kkt *HKKT, double **dPrimalX) { HKKT->dPrimalX = dPrimalX; }

void HKKTClear(hdsdp_kkt *HKKT) {
    if (!HKKT) return;
    free(HKKT->dASinvVec); free(HKKT->dASinvCSinvVec); free(HKKT->dASinvRdSinvVec);
    free(HKKT->invBuffer); free(HKKT->kktBuffer); free(HKKT->kktBuffer2);
    free(HKKT->kktMatBeg); free(HKKT->kktMatIdx);
    if (HKKT->kktMatElem) (void) hipHostFree(HKKT->kktMatElem);
    free(HKKT->kktDiag);
    HFpLinsysDestroy(&HKKT->kktM);
    priv_drop(HKKT);
    memset(HKKT, 0, sizeof(hdsdp_kkt));
}

void HKKTDestroy(hdsdp_kkt **pHKKT) {
    if (!pHKKT || !*pHKKT) return;
    HKKTClear(*pHKKT);
    free(*pHKKT);
    *pHKKT = nullptr;
}

void HMiKKTSetHostMirror(hdsdp_kkt *HKKT, int mirrorM) {
    MiKKTPriv *pv = priv_of(HKKT);
    if (!mirrorM && pv->n_foreign > 0) {
        fprintf(stderr, "[hdsdp_mi355x] HMiKKTSetHostMirror(0) ignored: %d cone(s) of this operator accumulate on the host\n",
                pv->n_foreign);
        return;
    }
    pv->mirror = mirrorM;
}
void HMiConeSetExchangePieces(hdsdp_cone *cone, hmi_alltoall_piece_fn start, hmi_alltoall_wait_fn wait, int npieces) {
    MiCone *c = cone_data(cone);
    c->a2a_start = start; c->a2a_wait = wait; c->a2a_pieces = std::max(1, npieces);
}
void HMiConeBuildPrimalXSXDirection(hdsdp_cone *cone, double *dPrimalScalMatrix, double *dPrimalXSXBuffer, int iDualMat) {
    cone->coneBuildPrimalDirection(cone->coneData, nullptr, dPrimalScalMatrix, dPrimalXSXBuffer, iDualMat);
}
void HMiConeGetExchangeStats(hdsdp_cone *cone, int *pieces, int *stagedLaunches) {
    MiCone *c = cone_data(cone);
    if (pieces) *pieces = c->last_pieces;
    if (stagedLaunches) *stagedLaunches = c->last_staged;
}
void HMiConeSetExchange(hdsdp_cone *cone, hmi_alltoall_fn a2a, hmi_allreduce_fn ar, void *ctx) {
    MiCone *c = cone_data(cone);
    c->alltoall = a2a; c->allreduce = ar; c->xctx = ctx;
}
hdsdp_retcode HMiConeGetExchangeBuffers(hdsdp_cone *cone, void **sendBuf, void **recvBuf, int64_t *chunkCount) {
    MiCone *c = cone_data(cone);
    if (chunkCount) *chunkCount = (int64_t) c->npb_loc * c->Lr * 16;
    if (sendBuf) *sendBuf = c->AhatLoc;
    if (recvBuf) *recvBuf = c->AhatAll;
    return HDSDP_RETCODE_OK;
}
hdsdp_retcode HMiConeSetExchangeBuffers(hdsdp_cone *cone, void *sendBuf, void *recvBuf) {
    MiCone *c = cone_data(cone);
    if (c->work_ready || !sendBuf || !recvBuf) return HDSDP_RETCODE_FAILED;
    c->AhatLoc = (double *) sendBuf;
    c->AhatAll = (c->world == 1) ? c->AhatLoc : (double *) recvBuf;
    c->ext_ahat = true;
    const size_t ahat = sizeof(double) * (size_t) c->world * c->npb_loc * c->Lr * 16;
    if (hipMemsetAsync(c->AhatLoc, 0, ahat, g.stream) != hipSuccess) return HDSDP_RETCODE_FAILED;
    if (c->AhatAll != c->AhatLoc && hipMemsetAsync(c->AhatAll, 0, ahat, g.stream) != hipSuccess) return HDSDP_RETCODE_FAILED;
    return HDSDP_RETCODE_OK;
}
void *HMiKKTDeviceMatrix(hdsdp_kkt *HKKT, int64_t *ld) {
    long l = 0;
    double *p = kkt_Mdev(HKKT, &l);
    if (ld) *ld = l;
    return p;
}
hdsdp_retcode HMiKKTGetRows(hdsdp_kkt *HKKT, int nRows, const int *rows, double *out) {
    // full symmetric rows of the device copy of M (lower triangle stored: dense matrix or tile store)
    MiKKTPriv *pv = priv_of(HKKT);
    const int m = HKKT->nRow;
    if (!pv->Mdev_valid) return HDSDP_RETCODE_FAILED;
    double *tmp = nullptr;
    HIP_RC(hipMalloc((void **) &tmp, sizeof(double) * (size_t) std::max(1, m)));
    hdsdp_retcode rc = HDSDP_RETCODE_OK;
    for (int r = 0; r < nRows && rc == HDSDP_RETCODE_OK; ++r) {
        if (rows[r] < 0 || rows[r] >= m) { rc = HDSDP_RETCODE_FAILED; break; }
        hipLaunchKernelGGL(mi_get_row_kernel, dim3((m + 255) / 256), dim3(256), 0, g.stream, kkt_view(HKKT), rows[r], m, tmp);
        if (hipMemcpyAsync(out + (size_t) r * m, tmp, sizeof(double) * (size_t) m, hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
            hipStreamSynchronize(g.stream) != hipSuccess) rc = HDSDP_RETCODE_FAILED;
    }
    (void) hipFree(tmp);
    return rc;
}
