// engine_build_small.h -- the rank-one and the sparse-gather Schur builders, cone destruction and the cone shell (vtable)
// Implementation header of engine.hip: included exactly once, there, in this order (the pieces share the anonymous namespace
// and the engine's thread-local context `g`); split out of a 3 300-line file in round 3, nothing else changed.

#define TRACE_STEP(msg)                                                                                          \
    do {                                                                                                         \
        if (stat_trace()) {                                                                                      \
            const hipError_t e_ = hipDeviceSynchronize();                                                        \
            fprintf(stderr, "[hdsdp_mi355x trace]     %s (n %d, rows %d, m %d) -> %s\n", msg, c->n, c->mloc, m, \
                    e_ == hipSuccess ? "ok" : hipGetErrorName(e_));                                              \
        }                                                                                                        \
    } while (0)
hdsdp_retcode build_r1_path(MiCone *c, hdsdp_kkt *kkt, MiKKTPriv *pv, int typeKKT) {
    // all (non-zero) constraints are rank one: A_i = s_i a_i a_i'  (reference strategy M2)
    MiLin *l = (MiLin *) c->dualFactor->chol;
    HdmChol &ch = l->ch;
    const int m = kkt->nRow;
    const int n16 = c->n16, m16 = c->mloc16;
    RC(ch.invert_factor(g.stream));
    TRACE_STEP("r1 step 1");
    HdmGemmArgs u = {};  // U = Linv * Avec
    u.A = ch.Linv; u.lda = ch.npad; u.B = c->Avec; u.ldb = n16; u.b_kmajor = 1; u.C = c->U; u.ldc = n16;
    u.M = n16; u.N = m16; u.K = n16; u.batch = 1; u.alpha = 1.0; u.klimit = HDM_KLIM_BY_M; u.epilogue = HDM_EPI_STORE;
    RC(hdm_launch_gemm(u, g.stream));
    TRACE_STEP("r1 step 2");
    HdmGemmArgs v = {};  // V = Linv^T * U = S^-1 * Avec
    v.A = ch.Linv; v.lda = ch.npad; v.a_kmajor = 1; v.B = c->U; v.ldb = n16; v.b_kmajor = 1; v.C = c->V; v.ldc = n16;
    v.M = n16; v.N = m16; v.K = n16; v.batch = 1; v.alpha = 1.0; v.epilogue = HDM_EPI_STORE;
    RC(hdm_launch_gemm(v, g.stream));
    TRACE_STEP("r1 step 3");
    if (typeKKT == KKT_TYPE_CORRECTOR) {
        // ASinv_i = s_i a_i' S^-1 a_i = s_i <u_i,u_i>; ASinvRdSinv_i = Rd s_i |v_i|^2
        hipLaunchKernelGGL(mi_col_dot_kernel, dim3((c->mloc + 3) / 4), dim3(256), 0, g.stream, c->U, c->U, (long) n16,
                           n16, c->sgn, c->rows_own, c->mloc, pv->vecs);
    TRACE_STEP("r1 step 4");
        if (c->Rd != 0.0) RC(hdm_r1_colnorm(c->V, n16, n16, c->sgn, c->rows_own, c->mloc, c->Rd, pv->vecs + m, g.stream));
        return HDSDP_RETCODE_OK;
    }
    HdmGemmArgs gq = {};  // Gr1 = U^T U
    gq.A = c->U; gq.lda = n16; gq.a_kmajor = 1; gq.B = c->U; gq.ldb = n16; gq.b_kmajor = 1; gq.C = c->Gr1; gq.ldc = m16;
    gq.M = m16; gq.N = m16; gq.K = n16; gq.batch = 1; gq.alpha = 1.0; gq.lower_only = 1; gq.epilogue = HDM_EPI_STORE;
    RC(hdm_launch_gemm(gq, g.stream));
    TRACE_STEP("r1 step 5");
    RC(hdm_r1_hadamard(c->Gr1, m16, c->sgn, c->rows_own, c->mloc, kkt_view(kkt), pv->vecs, g.stream));
    TRACE_STEP("r1 step 6");
    if (c->Rd != 0.0) {
        RC(hdm_r1_colnorm(c->V, n16, n16, c->sgn, c->rows_own, c->mloc, c->Rd, pv->vecs + m, g.stream));
    TRACE_STEP("r1 step 7");
        // TraceSinv = |Linv|_F^2
        hipLaunchKernelGGL(mi_mat_dot_kernel, dim3(1), dim3(256), 0, g.stream, ch.Linv, (long) ch.npad, ch.Linv,
                           (long) ch.npad, c->n, 0, 1.0, pv->vecs + 3 * m);
    TRACE_STEP("r1 step 8");
    }
    if (typeKKT == KKT_TYPE_HOMOGENEOUS && c->rank == 0) {
        // Ct = Linv C Linv^T (full);  ASinvCSinv_i = s_i u_i' Ct u_i;  CSinv = tr Ct; CSinvCSinv = |Ct|_F^2;
        // CSinvRdSinv = Rd <Ct, Linv Linv^T>
        HdmGemmArgs k1 = {};
        k1.A = ch.Linv; k1.lda = ch.npad; k1.B = c->Cfull; k1.ldb = n16; k1.C = c->W; k1.ldc = n16;
        k1.M = n16; k1.N = n16; k1.K = n16; k1.batch = 1; k1.alpha = 1.0; k1.klimit = HDM_KLIM_BY_M;
        RC(hdm_launch_gemm(k1, g.stream));
    TRACE_STEP("r1 step 9");
        HdmGemmArgs k2 = {};
        k2.A = c->W; k2.lda = n16; k2.B = ch.Linv; k2.ldb = ch.npad; k2.C = c->Ct; k2.ldc = n16;
        k2.M = n16; k2.N = n16; k2.K = n16; k2.batch = 1; k2.alpha = 1.0; k2.klimit = HDM_KLIM_BY_N;
        RC(hdm_launch_gemm(k2, g.stream));
    TRACE_STEP("r1 step 10");
        HdmGemmArgs w = {};  // W = Ct * U
        w.A = c->Ct; w.lda = n16; w.B = c->U; w.ldb = n16; w.b_kmajor = 1; w.C = c->W; w.ldc = n16;
        w.M = n16; w.N = m16; w.K = n16; w.batch = 1; w.alpha = 1.0;
        RC(hdm_launch_gemm(w, g.stream));
    TRACE_STEP("r1 step 11");
        hipLaunchKernelGGL(mi_col_dot_kernel, dim3((c->mloc + 3) / 4), dim3(256), 0, g.stream, c->U, c->W, (long) n16,
                           n16, c->sgn, c->rows_own, c->mloc, pv->vecs + 2 * m);
    TRACE_STEP("r1 step 12");
        hipLaunchKernelGGL(mi_mat_dot_kernel, dim3(1), dim3(256), 0, g.stream, c->Ct, (long) n16, nullptr, 0L, c->n, 1,
                           1.0, pv->vecs + 3 * m + 1);
    TRACE_STEP("r1 step 13");
        hipLaunchKernelGGL(mi_mat_dot_kernel, dim3(1), dim3(256), 0, g.stream, c->Ct, (long) n16, c->Ct, (long) n16,
                           c->n, 0, 1.0, pv->vecs + 3 * m + 2);
    TRACE_STEP("r1 step 14");
        if (c->Rd != 0.0) {
            HdmGemmArgs q = {};  // Xinv := Linv Linv^T
            q.A = ch.Linv; q.lda = ch.npad; q.B = ch.Linv; q.ldb = ch.npad; q.C = c->Xinv; q.ldc = n16;
            q.M = n16; q.N = n16; q.K = n16; q.batch = 1; q.alpha = 1.0;
            RC(hdm_launch_gemm(q, g.stream));
    TRACE_STEP("r1 step 15");
            hipLaunchKernelGGL(mi_mat_dot_kernel, dim3(1), dim3(256), 0, g.stream, c->Ct, (long) n16, c->Xinv,
                               (long) n16, c->n, 0, c->Rd, pv->vecs + 3 * m + 3);
    TRACE_STEP("r1 step 16");
        }
    }
    HIP_RC(hipGetLastError());
    return HDSDP_RETCODE_OK;
}

hdsdp_retcode build_sparse_path(MiCone *c, hdsdp_kkt *kkt, MiKKTPriv *pv, int typeKKT) {
    // every constraint is a short triplet list: gather from X = S^-1 (reference strategy M5, and the corrector /
    // HSD components that the reference evaluates with the same gathers, hdsdp_conic_sdp.c:923-1056)
    MiLin *l = (MiLin *) c->dualFactor->chol;
    HdmChol &ch = l->ch;
    const int m = kkt->nRow;
    const long ldx = ch.npad;
    RC(ch.inverse_full(c->Xinv, ldx, g.stream));
    RC(hdm_sparse_dot(c->sp_rp, c->sp_ti, c->sp_tj, c->sp_tv, c->Xinv, ldx, c->mloc, c->rows_own, 1.0, pv->vecs, g.stream));
    if (c->Rd != 0.0) {
        HdmGemmArgs q = {};  // Y = X X^T = S^-2
        q.A = c->Xinv; q.lda = ldx; q.B = c->Xinv; q.ldb = ldx; q.C = c->Yinv; q.ldc = ldx;
        q.M = c->n16; q.N = c->n16; q.K = c->n16; q.batch = 1; q.alpha = 1.0; q.epilogue = HDM_EPI_STORE;
        RC(hdm_launch_gemm(q, g.stream));
        RC(hdm_sparse_dot(c->sp_rp, c->sp_ti, c->sp_tj, c->sp_tv, c->Yinv, ldx, c->mloc, c->rows_own, c->Rd,
                          pv->vecs + m, g.stream));
    }
    if (typeKKT == KKT_TYPE_CORRECTOR) return HDSDP_RETCODE_OK;
    if (c->Rd != 0.0)
        hipLaunchKernelGGL(mi_mat_dot_kernel, dim3(1), dim3(256), 0, g.stream, c->Xinv, ldx, nullptr, 0L, c->n, 1, 1.0,
                           pv->vecs + 3 * m);
    RC(hdm_sparse_pairs(c->sp_rp, c->sp_ti, c->sp_tj, c->sp_tv, c->Xinv, ldx, c->mloc, c->rows_own, kkt_view(kkt), g.stream));
    if (typeKKT == KKT_TYPE_HOMOGENEOUS) {
        HdmGemmArgs w = {};  // W = X C,  Ct = W X = X C X
        w.A = c->Xinv; w.lda = ldx; w.B = c->Cfull; w.ldb = c->n16; w.C = c->W; w.ldc = ldx;
        w.M = c->n16; w.N = c->n16; w.K = c->n16; w.batch = 1; w.alpha = 1.0; w.epilogue = HDM_EPI_STORE;
        RC(hdm_launch_gemm(w, g.stream));
        HdmGemmArgs x = {};
        x.A = c->W; x.lda = ldx; x.B = c->Xinv; x.ldb = ldx; x.C = c->Ct; x.ldc = ldx;
        x.M = c->n16; x.N = c->n16; x.K = c->n16; x.batch = 1; x.alpha = 1.0; x.epilogue = HDM_EPI_STORE;
        RC(hdm_launch_gemm(x, g.stream));
        RC(hdm_sparse_dot(c->sp_rp, c->sp_ti, c->sp_tj, c->sp_tv, c->Ct, ldx, c->mloc, c->rows_own, 1.0,
                          pv->vecs + 2 * m, g.stream));
        hipLaunchKernelGGL(mi_mat_dot_kernel, dim3(1), dim3(256), 0, g.stream, c->Cfull, (long) c->n16, c->Xinv, ldx, c->n,
                           0, 1.0, pv->vecs + 3 * m + 1);
        hipLaunchKernelGGL(mi_mat_dot_kernel, dim3(1), dim3(256), 0, g.stream, c->Cfull, (long) c->n16, c->Ct, ldx, c->n, 0,
                           1.0, pv->vecs + 3 * m + 2);
        if (c->Rd != 0.0)
            hipLaunchKernelGGL(mi_mat_dot_kernel, dim3(1), dim3(256), 0, g.stream, c->Cfull, (long) c->n16, c->Yinv, ldx,
                               c->n, 0, c->Rd, pv->vecs + 3 * m + 3);
    }
    HIP_RC(hipGetLastError());
    return HDSDP_RETCODE_OK;
}

void cone_destroy_data(void **pcd) {
    if (!pcd || !*pcd) return;
    MiCone *c = (MiCone *) *pcd;
    if (c->shared_ts) c->slabs = nullptr;      // one buffer, freed as T
    double *bufs[] = {c->Afull, c->Abatch, c->Cfull, c->CL, c->Avec, c->sgn, c->S, c->Scheck, c->ydev, c->T, c->slabs, c->Gm,
                      c->U, c->V, c->Gr1, c->Ct, c->W, c->Xinv, c->Yinv};
    for (double *b : bufs)
        if (b) (void) hipFree(b);
    hdm_zs_free(&c->zs);
    if (!c->ext_ahat) {
        if (c->AhatAll && c->AhatAll != c->AhatLoc) (void) hipFree(c->AhatAll);
        if (c->AhatLoc) (void) hipFree(c->AhatLoc);
    }
    if (c->sp_rp) (void) hipFree(c->sp_rp);
    if (c->sp_ti) (void) hipFree(c->sp_ti);
    if (c->sp_tj) (void) hipFree(c->sp_tj);
    if (c->sp_tv) (void) hipFree(c->sp_tv);
    if (c->rows_seg) (void) hipFree(c->rows_seg);
    if (c->rows_own) (void) hipFree(c->rows_own);
    if (c->trA) free(c->trA);
    if (c->yhost) (void) hipHostFree(c->yhost);
    { int *ip[] = {c->small.fp, c->small.fi, c->small.dense_of, c->small.dense_rows}; for (int *q : ip) if (q) (void) hipFree(q); }
    if (c->small.fv) (void) hipFree(c->small.fv);
    if (c->small.sgn) (void) hipFree(c->small.sgn);
    if (c->small.io_host) (void) hipHostFree(c->small.io_host);
    if (c->corr) (void) hipFree(c->corr);
    HFpLinsysDestroy(&c->dualFactor);
    if (c->primal) { c->primal->destroy(); delete c->primal; }
    if (c->lanczos) { c->lanczos->destroy(); delete c->lanczos; }
    if (c->chk_host) (void) hipHostFree(c->chk_host);
    if (c->checker) { c->checker->destroy(); delete c->checker; }
    for (hipEvent_t e : c->piece_ev) if (e) (void) hipEventDestroy(e);
    for (hipEvent_t e : c->pe_s2) if (e) (void) hipEventDestroy(e);
    for (hipEvent_t e : c->pe_ga) if (e) (void) hipEventDestroy(e);
    for (hipEvent_t e : c->pe_gb) if (e) (void) hipEventDestroy(e);
    if (c->pe_s1) (void) hipEventDestroy(c->pe_s1);
    if (c->dS) (void) hipFree(c->dS);
    if (c->Xup) (void) hipFree(c->Xup);
    if (c->Pr1) (void) hipFree(c->Pr1);
    if (c->Pr2) (void) hipFree(c->Pr2);
    delete c;
    *pcd = nullptr;
}

hdsdp_cone *new_cone_shell(MiCone *c, int iCone) {
    hdsdp_cone *h = (hdsdp_cone *) calloc(1, sizeof(hdsdp_cone));
    h->iCone = iCone;
    h->cone = HDSDP_CONETYPE_DENSE_SDP;
    h->coneData = c;
    h->coneDestroyData = cone_destroy_data;
    h->coneSetStart = cone_setstart;
    h->coneUpdate = cone_update;
    h->coneGetSymNnz = cone_getsymnnz;
    h->coneAddSymNz = cone_add_sym_nz;
    h->coneGetKKTMap = cone_get_kkt_map;
    h->coneGetDim = cone_getdim;
    h->coneBuildSchur = cone_build_schur;
    h->coneBuildSchurFixed = cone_build_schur_fixed;
    h->coneBuildPrimalDirection = cone_build_primal_dir;
    h->coneInteriorCheck = cone_interior;
    h->coneRatioTest = cone_ratio_test;
    h->conePRecover = cone_precover;
    h->coneInteriorCheckExpert = cone_interior_expert;
    h->coneAxpyBufferAndCheck = cone_axpy_check;
    h->coneReduceResi = cone_reduce_resi;
    h->coneSetPerturb = cone_set_perturb;
    h->getstat = cone_getstat;
    h->coneView = cone_view;
    h->coneGetCoeffNorm = cone_coeff_norm;
    h->coneGetObjNorm = cone_obj_norm;
    h->coneScal = cone_scal;
    h->coneATimesXpy = cone_a_times_x;
    h->coneTraceCX = cone_trace_cx;
    h->coneXDotS = cone_x_dot_s;
    h->coneDRecover = cone_get_dual;
    h->coneGetBarrier = cone_barrier;
    return h;
}
