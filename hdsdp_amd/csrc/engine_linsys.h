// engine_linsys.h -- the linear-system objects behind HFpLinsys* (dense / CSC-in dual matrices, the Schur system with its pivoted way out)
// Implementation header of engine.hip: included exactly once, there, in this order (the pieces share the anonymous namespace
// and the engine's thread-local context `g`); split out of a 3 300-line file in round 3, nothing else changed.
// =============================================================================================
// linear-system objects
// =============================================================================================
struct MiLin {
    int n = 0;
    linsys_type type = HDSDP_LINSYS_DENSE_DIRECT;
    HdmChol ch;
    double *work = nullptr;  // npad x npad device scratch (Invert)
    double relTol = 0, absTol = 0;
    int maxIter = -1;
    // Schur systems: M lives here (device, ld = ch.npad) before factorisation
    double *Mdev = nullptr;
    // sparse Schur operator in tile form (bsparse.h): the matrix and its factor are 128 x 128 tiles inside the block pattern of
    // the Cholesky factor; no dense m x m array exists anywhere (ch stays uninitialised, Mdev null)
    HdmBsp *bsp = nullptr;
    // symmetric-indefinite fallback (HFpLinsysSwitchToIndefinite, hdsdp_linsolver.c:1827-1857): once switched, every
    // later factorisation goes through the pivoted solver, like the reference's replaced vtable
    HdmLu *lu = nullptr;
    bool indef = false;
    // sparse Schur operator: the factor object holds P M P' (perm[old] = new, a bandwidth-reducing order of the pattern);
    // right-hand sides go in permuted and solutions come back in the caller's order.  Empty = identity.
    std::vector<int> perm;
    std::vector<double> pbuf;
    const double *srcHost = nullptr, *srcDev = nullptr;   // where the last factorised matrix came from (lower valid)
    long srcLd = 0;
    // HDSDP_LINSYS_SPARSE_DIRECT (the reference's QDLDL backend for a sparse dual matrix, hdsdp_linsolver.c:509-809):
    // the matrix arrives as a lower-triangular CSC and is factored densely on the device.  Result-equivalent for every
    // caller: QDLDL's forward / backward solves carry the D^-1/2 scaling (:669-721), i.e. they ARE the Cholesky factor's,
    // GetDiag returns sqrt(D) (:746-756), Invert the full inverse (:758-772); a fill-reducing order only changes the
    // factor by an orthogonal similarity, which neither logdet nor the Lanczos spectrum sees.
    bool csc_in = false;
    std::vector<int> cscBeg, cscIdx;
    std::vector<double> dense;
};

hdsdp_retcode lin_create(void **pchol, int nCol) {
    if (ensure_ctx()) return HDSDP_RETCODE_FAILED;
    MiLin *l = new MiLin();
    l->n = nCol;
    if (l->ch.init(nCol)) { delete l; return HDSDP_RETCODE_MEMORY; }
    *pchol = l;
    return HDSDP_RETCODE_OK;
}
void lin_setparam(void *chol, void *param) { (void) chol; (void) param; }
hdsdp_retcode lin_symbolic(void *chol, int *colMatBeg, int *colMatIdx) {
    MiLin *l = (MiLin *) chol;
    if (l->csc_in && colMatBeg && colMatIdx) {   // keep the pattern: later calls may pass it again or not at all
        l->cscBeg.assign(colMatBeg, colMatBeg + l->n + 1);
        l->cscIdx.assign(colMatIdx, colMatIdx + colMatBeg[l->n]);
    }
    return HDSDP_RETCODE_OK;
}
// lower-triangular CSC -> dense n x n column-major (lower triangle valid), on the host: a format conversion of n^2 doubles
const double *lin_densify(MiLin *l, const int *colMatBeg, const int *colMatIdx, const double *colMatElem) {
    const int *beg = colMatBeg ? colMatBeg : (l->cscBeg.empty() ? nullptr : l->cscBeg.data());
    const int *idx = colMatIdx ? colMatIdx : (l->cscIdx.empty() ? nullptr : l->cscIdx.data());
    if (!beg || !idx || !colMatElem) return nullptr;
    const size_t n = (size_t) l->n;
    l->dense.assign(n * n, 0.0);
    for (size_t j = 0; j < n; ++j)
        for (int p = beg[j]; p < beg[j + 1]; ++p) {
            const size_t i = (size_t) idx[p];
            if (i >= j) l->dense[i + j * n] = colMatElem[p];
            else l->dense[j + i * n] = colMatElem[p];     // an upper entry, should a caller hand one over
        }
    return l->dense.data();
}

hdsdp_retcode lin_factor_host(MiLin *l, const double *A, int *info) {
    RC(l->ch.load_host(A, l->n, g.stream));
    RC(l->ch.factor(g.stream, info));
    return HDSDP_RETCODE_OK;
}
// lapackIndefiniteLinSolverNumeric (hdsdp_linsolver.c:1706-1727): copy + pivoted factorisation; a singular matrix fails
hdsdp_retcode lin_factor_indef(MiLin *l) {
    if (!l->lu) {
        l->lu = new HdmLu();
        if (l->lu->init(l->n)) { l->lu->destroy(); delete l->lu; l->lu = nullptr; return HDSDP_RETCODE_MEMORY; }
    }
    if (l->srcDev) RC(l->lu->load_device_lower(l->srcDev, l->srcLd, g.stream));
    else if (l->srcHost) RC(l->lu->load_host_lower(l->srcHost, l->srcLd, g.stream));
    else return HDSDP_RETCODE_FAILED;
    int info = 0;
    RC(l->lu->factor(g.stream, &info));
    return info == 0 ? HDSDP_RETCODE_OK : HDSDP_RETCODE_FAILED;
}
// linalg/hdsdp_linsolver.c:1082-1110 (copy + dpotrf; info != 0 is a failure here)
hdsdp_retcode lin_numeric(void *chol, int *colMatBeg, int *colMatIdx, double *colMatElem) {
    MiLin *l = (MiLin *) chol;
    if (l->bsp) return HDSDP_RETCODE_FAILED;      // (a tile-form Schur operator is factored by HKKTFactorize, nothing else owns one)
    if (l->csc_in) {
        colMatElem = const_cast<double *>(lin_densify(l, colMatBeg, colMatIdx, colMatElem));
        if (!colMatElem) return HDSDP_RETCODE_FAILED;
    }
    l->srcHost = colMatElem; l->srcDev = nullptr; l->srcLd = l->n;
    if (l->indef) return lin_factor_indef(l);
    int info = 0;
    if (lin_factor_host(l, colMatElem, &info) != HDSDP_RETCODE_OK) return HDSDP_RETCODE_FAILED;
    return info == 0 ? HDSDP_RETCODE_OK : HDSDP_RETCODE_FAILED;
}
// HFpLinsysSwitchToIndefinite (hdsdp_linsolver.c:1827-1857): only the Schur system (DENSE_ITERATIVE) has this way out;
// the matrix is re-read from where the failed factorisation took it
hdsdp_retcode lin_switch_indefinite(hdsdp_linsys_fp *HLin) {
    MiLin *l = (MiLin *) HLin->chol;
    if (l->bsp) return HDSDP_RETCODE_OK;      // the tile form's factorisation is an LDL' already (bsparse.h): nothing to switch to
    HLin->LinType = HDSDP_LINSYS_DENSE_INDEFINITE;
    l->indef = true;
    return lin_factor_indef(l);
}
// linalg/hdsdp_linsolver.c:1112-1144 (info > 0 => "not PSD" is a value, not an error)
hdsdp_retcode lin_psdcheck(void *chol, int *colMatBeg, int *colMatIdx, double *colMatElem, int *isPsd) {
    MiLin *l = (MiLin *) chol;
    if (l->bsp) return HDSDP_RETCODE_FAILED;
    if (l->csc_in) {
        colMatElem = const_cast<double *>(lin_densify(l, colMatBeg, colMatIdx, colMatElem));
        if (!colMatElem) return HDSDP_RETCODE_FAILED;
    }
    if (l->indef) return HDSDP_RETCODE_FAILED;   // :1729-1739, no PSD check on the pivoted factor
    int info = 0;
    if (lin_factor_host(l, colMatElem, &info) != HDSDP_RETCODE_OK) return HDSDP_RETCODE_FAILED;
    *isPsd = (info == 0) ? 1 : 0;
    return HDSDP_RETCODE_OK;
}
// :1146-1196 dtrsm with L / L^T ; solVec == NULL => in place
void lin_fsolve(void *chol, int nRhs, double *rhs, double *sol) {
    MiLin *l = (MiLin *) chol;
    if (l->indef || l->bsp) return;                        // :1741-1759, no half solves with the pivoted factor
    // the slot returns void (hdsdp_linsolver.h:22): a device failure can only be reported, and poisons the output so that
    // the caller's next NaN check (e.g. HFpLinsysSolve, :2085-2110) sees it
    if (l->ch.solve_host(rhs, sol ? sol : rhs, nRhs, 1, g.stream)) {
        fprintf(stderr, "[hdsdp_mi355x] forward substitution failed on the device\n");
        (sol ? sol : rhs)[0] = NAN;
    }
}
void lin_bsolve(void *chol, int nRhs, double *rhs, double *sol) {
    MiLin *l = (MiLin *) chol;
    if (l->indef || l->bsp) return;
    if (l->ch.solve_host(rhs, sol ? sol : rhs, nRhs, 2, g.stream)) {
        fprintf(stderr, "[hdsdp_mi355x] backward substitution failed on the device\n");
        (sol ? sol : rhs)[0] = NAN;
    }
}
// :1198-1225 dpotrs
hdsdp_retcode lin_solve(void *chol, int nRhs, double *rhs, double *sol) {
    MiLin *l = (MiLin *) chol;
    if (l->indef) {                              // :1761-1780 dsytrs
        if (!l->lu || !l->lu->factored) return HDSDP_RETCODE_FAILED;
        RC(l->lu->solve_host(rhs, sol ? sol : rhs, nRhs, g.stream));
        return HDSDP_RETCODE_OK;
    }
    if (l->bsp) {
        for (int r = 0; r < nRhs; ++r)
            RC(l->bsp->solve_host(rhs + (size_t) r * l->n, (sol ? sol : rhs) + (size_t) r * l->n, g.stream));
        return HDSDP_RETCODE_OK;
    }
    if (!l->ch.factored) return HDSDP_RETCODE_FAILED;
    if (!l->perm.empty()) {
        const int n = l->n;
        double *out = sol ? sol : rhs;
        l->pbuf.resize((size_t) n * nRhs);
        for (int r = 0; r < nRhs; ++r)
            for (int i = 0; i < n; ++i) l->pbuf[(size_t) r * n + l->perm[i]] = rhs[(size_t) r * n + i];
        RC(l->ch.solve_host(l->pbuf.data(), l->pbuf.data(), nRhs, 0, g.stream));
        for (int r = 0; r < nRhs; ++r)
            for (int i = 0; i < n; ++i) out[(size_t) r * n + i] = l->pbuf[(size_t) r * n + l->perm[i]];
        return HDSDP_RETCODE_OK;
    }
    RC(l->ch.solve_host(rhs, sol ? sol : rhs, nRhs, 0, g.stream));
    return HDSDP_RETCODE_OK;
}
// :1227-1236
hdsdp_retcode lin_getdiag(void *chol, double *diag) {
    MiLin *l = (MiLin *) chol;
    if (l->indef || l->bsp) return HDSDP_RETCODE_FAILED;   // :1782-1788
    RC(l->ch.get_diag(diag, g.stream));
    return HDSDP_RETCODE_OK;
}
// :1238-1260 dpotri + HUtilMatSymmetrize: full symmetric inverse into dFullMatrix (n x n)
void lin_invert(void *chol, double *dFull, double *) {
    MiLin *l = (MiLin *) chol;
    if (l->indef || l->bsp) return;              // :1790-1797
    HdmChol &c = l->ch;
    if (!l->work) {
        if (hipMalloc((void **) &l->work, sizeof(double) * (size_t) c.npad * c.npad) != hipSuccess) return;
    }
    if (c.inverse_full(l->work, c.npad, g.stream)) return;
    (void) hipMemcpy2DAsync(dFull, sizeof(double) * c.n, l->work, sizeof(double) * c.npad, sizeof(double) * c.n, c.n,
                            hipMemcpyDeviceToHost, g.stream);
    (void) hipStreamSynchronize(g.stream);
}
void lin_destroy(void **pchol) {
    if (!pchol || !*pchol) return;
    MiLin *l = (MiLin *) *pchol;
    l->ch.destroy();
    if (l->lu) { l->lu->destroy(); delete l->lu; }
    if (l->work) (void) hipFree(l->work);
    if (l->Mdev) (void) hipFree(l->Mdev);
    if (l->bsp) { l->bsp->destroy(); delete l->bsp; }
    delete l;
    *pchol = nullptr;
}
