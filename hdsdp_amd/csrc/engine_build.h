// engine_build.h -- the GPU Schur builders: congruence + Gram (all build types, the exchange of a sharded build, KKT_TYPE_PRIMAL), corrector components
// Implementation header of engine.hip: included exactly once, there, in this order (the pieces share the anonymous namespace
// and the engine's thread-local context `g`); split out of a 3 300-line file in round 3, nothing else changed.
// --- the GPU Schur builder ---------------------------------------------------------------------
// share of step 2's work that falls into the tile columns of `mask` (tile (tm, tn), tm >= tn, runs tn + 1 K blocks)
double cong2_mask_share(int NT, unsigned long long mask) {
    if (!mask || NT > 64) return 1.0;
    double all = 0.0, sel = 0.0;
    for (int tn = 0; tn < NT; ++tn) {
        const double w = (double) (NT - tn) * (tn + 1);
        all += w;
        if ((mask >> tn) & 1ULL) sel += w;
    }
    return all > 0.0 ? sel / all : 1.0;
}

// phase 0: both steps; 1: step 1 only; 2: step 2 only (count <= Bc, T still holds step 1's output), optionally only the
// output tiles of the tile columns in `colmask` -- the multi-GPU build runs step 2 by packed-index range so that the
// finished ranges can leave for the other ranks while the rest is still being computed
// `asrc_span`: elements readable from Asrc (the buffer's operand slack included), for the launcher's check of the
// unmasked tile loads
int congruence_rows(MiCone *c, HdmChol &ch, const double *Asrc, long astride, long asrc_span, int count, long row0,
                    int phase = 0, unsigned long long colmask = 0) {
    // rows row0 .. row0+count-1 of AhatLoc  <-  blocked( Linv * A * Linv^T ),  A = A_L + A_L^T given in A_L form:
    //   step 1  U  = Linv * A_L                 (lower x lower = lower triangular: k in [col tile, row tile], n^3/3)
    //   step 2  At = U * Linv^T + Linv * U^T    (SYR2K form, lower tiles, k <= col tile, 2n^3/3)
    // i.e. n^3 flops per constraint instead of the 4/3 n^3 of (Linv A) Linv^T, and half the intermediate traffic.
    const long nn = (long) c->n16 * c->n16;
    const double n3 = (double) c->n * c->n * c->n;
    for (int b0 = 0; b0 < count; b0 += c->Bc) {
        const int nb = std::min(c->Bc, count - b0);
        HdmGemmArgs k1 = {};
        k1.A = ch.Linv; k1.lda = ch.npad; k1.strideA = 0;
        k1.B = Asrc + (long) b0 * astride; k1.ldb = c->n16; k1.strideB = astride; k1.b_kmajor = 1; k1.b_sky = 1;
        k1.C = c->T; k1.ldc = c->n16; k1.strideC = nn;
        k1.M = c->n16; k1.N = c->n16; k1.K = c->n16; k1.batch = nb; k1.alpha = 1.0;
        k1.klimit = HDM_KLIM_BAND; k1.lower_only = 1; k1.epilogue = HDM_EPI_STORE; k1.role = HDM_ROLE_CONG1;
        k1.flops = (double) nb * n3 / 3.0;
        const long linv_span = (long) ch.npad * ch.npad;
        const long t_span = nn * c->Bc + (long) (hdm_operand_pad(c->n16) / sizeof(double));
        k1.spanA = linv_span; k1.spanB = asrc_span - (long) b0 * astride;
        if (phase != 2 && c->shared_ts && hdm_zero_diag_upper(c->T, nn, c->n16, nb, g.stream)) return 1;
        if (phase != 2 && hdm_launch_gemm(k1, g.stream)) return 1;
        if (phase == 1) continue;
        HdmGemmArgs k2 = {};
        k2.A = c->T; k2.lda = c->n16; k2.strideA = nn;
        k2.B = ch.Linv; k2.ldb = ch.npad; k2.strideB = 0;
        k2.A2 = ch.Linv; k2.lda2 = ch.npad; k2.strideA2 = 0;
        k2.B2 = c->T; k2.ldb2 = c->n16; k2.strideB2 = nn;
        k2.C = c->AhatLoc; k2.M = c->n16; k2.N = c->n16; k2.K = c->n16; k2.batch = nb; k2.alpha = 1.0;
        k2.klimit = HDM_KLIM_BY_N; k2.lower_only = 1; k2.epilogue = HDM_EPI_BLOCKED;
        k2.blk_row_stride = c->Lr; k2.blk_row0 = row0 + b0; k2.nblk = c->nblk; k2.role = HDM_ROLE_CONG2;
        k2.tile_col_mask = colmask;
        k2.spanA = t_span; k2.spanB = linv_span; k2.spanA2 = linv_span; k2.spanB2 = t_span;
        k2.flops = (double) nb * n3 * 2.0 / 3.0 * cong2_mask_share((c->n16 + HDM_TILE - 1) / HDM_TILE, colmask);
        if (hdm_launch_gemm(k2, g.stream)) return 1;
    }
    return 0;
}

// Gram partial sums of the K splits [z0, z0 + nz): slabs slab0.. <- (or +=, `accumulate`) Ahat * Ahat^T over their share of this
// rank's p-range.  slab0 < 0: slab z0 (one slab per split)
int gram_splits(MiCone *c, int z0, int nz, int slab0 = -1, bool accumulate = false) {
    HdmGemmArgs gq = {};
    gq.A = c->AhatAll; gq.B = c->AhatAll; gq.a_kmajor = 1; gq.b_kmajor = 1;
    gq.lda = 16; gq.ldb = 16; gq.a_kblk = (long) c->Lr * 16; gq.b_kblk = (long) c->Lr * 16;
    if (c->world > 1) { gq.seg_rows = c->Lr; gq.seg_extra = c->npb_loc * c->Lr * 16 - (long) c->Lr * 16; }
    gq.ldc = c->R; gq.M = (int) c->R; gq.N = (int) c->R; gq.K = (int) (c->npb_loc * 16);
    gq.lower_only = 1; gq.epilogue = HDM_EPI_SLAB; gq.batch = nz;
    const long chunk = (c->npb_loc + c->nsplit - 1) / c->nsplit;
    gq.k_chunk = chunk * 16; gq.slab_stride = c->R * c->R; gq.alpha = 1.0; gq.role = HDM_ROLE_GRAM;
    gq.k_base = (long) z0 * gq.k_chunk;
    gq.spanA = gq.spanB = (long) c->world * c->npb_loc * c->Lr * 16 + HDM_OPERAND_PAD_DOUBLES;
    gq.C = c->slabs + (long) (slab0 < 0 ? z0 : slab0) * gq.slab_stride;
    gq.beta = accumulate ? 1.0 : 0.0;
    gq.queue_global = c->gram_queue_global ? 1 : 0;
    {   // (m+3)(m+4)/2 inner products of length n(n+1)/2 (this rank's share), 2 flops each
        const double rows = (double) c->m + 3.0;
        gq.flops = rows * (rows + 1.0) * 0.5 * ((double) c->n * (c->n + 1) * 0.5) * 2.0 / c->world * ((double) nz / c->nsplit);
    }
    return hdm_launch_gemm(gq, g.stream);
}

// the K splits [z0, z0 + nz) in groups of nslab, launch after launch on the engine stream, every group accumulating into the slabs
// the one before left (`fresh`: the first group of a build overwrites them).  Fixed order: bitwise reproducible.
// The first group of a build is its largest (a range's groups are nslab, nslab, ..., remainder; the pieces of an exchange are equal
// ranges), so it initialises every slab a later group adds to; slabs_used = its size is what the reduction sums.
int gram_range(MiCone *c, int z0, int nz, bool fresh) {
    for (int z = z0; z < z0 + nz; z += c->nslab) {
        const int k = std::min(c->nslab, z0 + nz - z);
        if (fresh) c->slabs_used = k;
        else if (k > c->slabs_used) { fprintf(stderr, "[hdsdp_mi355x] Gram groups out of order\n"); return 1; }
        if (gram_splits(c, z, k, 0, !fresh)) return 1;
        fresh = false;
    }
    return 0;
}
int gram_reduce(MiCone *c) {
    return hdm_slab_reduce(c->slabs, c->R * c->R, c->slabs_used, c->Gm, c->R * c->R, c->R, g.stream);
}
int gram_all(MiCone *c) {
    // Gm(lower) = sum over this rank's p-range of Ahat * Ahat^T, rows in segment order (more splits than slabs:
    // cone_alloc_gemm_work says why)
    if (gram_range(c, 0, c->nsplit, true)) return 1;
    return gram_reduce(c);
}

// world > 1: the all-to-all that re-shards Ahat from "by constraint" to "by packed-index range", and the Gram product.
// With the piecewise hooks registered the exchange runs in pieces along the packed index and the Gram splits of a piece
// start as soon as it has arrived, while the later pieces are still on the links.
// number of pieces of the piecewise exchange (whole groups of Gram K splits)
int exchange_pieces(const MiCone *c) {
    int P = (c->a2a_start && c->a2a_wait) ? c->a2a_pieces : 1;
    if (const char *e = getenv("HDSDP_MI355X_A2A_PIECES")) P = std::max(1, atoi(e));
    if (!(c->a2a_start && c->a2a_wait)) P = 1;
    while (P > 1 && (c->nsplit % P)) --P;
    return P;
}
// p-blocks [lo, hi) of every destination's chunk that piece k of P carries
void piece_range(const MiCone *c, int k, int P, long *lo, long *hi) {
    const long chunk = (c->npb_loc + c->nsplit - 1) / c->nsplit;   // p-blocks per split
    const int zper = c->nsplit / P;
    *lo = std::min<long>(c->npb_loc, (long) k * zper * chunk);
    *hi = (k == P - 1) ? c->npb_loc : std::min<long>(c->npb_loc, (long) (k + 1) * zper * chunk);
}
// Tile columns of congruence step 2 whose output piece k needs.  P-block q belongs to the 16 x 16 sub-block q / 16 of the
// blocked lower triangle, sub-blocks are numbered column by column (column bj starts at bj*nblk - bj(bj-1)/2), and tile
// column tn produces the sub-block columns 8 tn .. 8 tn + 7: a range of p-blocks is a range of tile columns.
unsigned long long piece_tile_cols(const MiCone *c, int k, int P) {
    long lo, hi;
    piece_range(c, k, P, &lo, &hi);
    auto col_of = [&](long sub) {
        int bj = 0;
        while (bj + 1 < c->nblk && (long) (bj + 1) * c->nblk - (long) (bj + 1) * bj / 2 <= sub) ++bj;
        return bj;
    };
    unsigned long long mask = 0;
    for (int d = 0; d < c->world; ++d) {
        const long g0 = (long) d * c->npb_loc + lo, g1 = std::min<long>(c->npb, (long) d * c->npb_loc + hi);
        if (g0 >= g1) continue;
        for (int tn = col_of(g0 / 16) / 8; tn <= col_of((g1 - 1) / 16) / 8; ++tn) mask |= 1ULL << tn;
    }
    return mask;
}

static inline double host_now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline int prof_record(hipEvent_t &ev, hipStream_t s) {
    if (!ev && hipEventCreate(&ev) != hipSuccess) return 1;
    return hipEventRecord(ev, s) != hipSuccess;
}

// `staged`: congruence step 2 was launched piece by piece and c->piece_ev[k] marks the point where piece k's p-blocks
// are final, so piece k can leave while the later tile columns are still being computed; otherwise the whole stream
// is drained first.
hdsdp_retcode exchange_and_gram(MiCone *c, bool staged = false) {
    if (!c->alltoall && !(c->a2a_start && c->a2a_wait)) {
        fprintf(stderr, "[hdsdp_mi355x] world > 1 but no exchange hook registered\n");
        return HDSDP_RETCODE_FAILED;
    }
    if (!staged) HIP_RC(hipStreamSynchronize(g.stream));
    const int P = exchange_pieces(c);
    const double sent_share = (double) (c->world - 1) * 8.0;   // bytes this shard sends per double of a piece
    if (P <= 1) {
        if (staged) HIP_RC(hipStreamSynchronize(g.stream));
        const double t0 = host_now();
        if (c->alltoall) { if (c->alltoall(c->xctx)) return HDSDP_RETCODE_FAILED; }
        else {
            if (c->a2a_start(c->xctx, 0, (int64_t) c->npb_loc * c->Lr * 16, 0) || c->a2a_wait(c->xctx, 0)) return HDSDP_RETCODE_FAILED;
        }
        c->prof.wait_host[0] = c->prof.flight[0] = (host_now() - t0) * 1e3;
        c->prof.bytes[0] = sent_share * (double) c->npb_loc * c->Lr * 16;
        if (prof_record(c->pe_ga[0], g.stream) || gram_range(c, 0, c->nsplit, true) || prof_record(c->pe_gb[0], g.stream)) return HDSDP_RETCODE_FAILED;
        return gram_reduce(c) ? HDSDP_RETCODE_FAILED : HDSDP_RETCODE_OK;
    }
    const int zper = c->nsplit / P;
    for (int k = 0; k < P; ++k) {
        long lo, hi;
        piece_range(c, k, P, &lo, &hi);
        const int64_t off = (int64_t) lo * c->Lr * 16, end = (int64_t) hi * c->Lr * 16;   // doubles inside a chunk
        if (staged) HIP_RC(hipEventSynchronize(c->piece_ev[k]));
        c->pt_start[k] = host_now();
        c->prof.bytes[k] = sent_share * (double) (end - off);
        if (c->a2a_start(c->xctx, off, end - off, k)) {
            if (k == 0 && c->alltoall) {
                // the piecewise flavour is not available in this process group: one blocking exchange from now on
                fprintf(stderr, "[hdsdp_mi355x] piecewise all-to-all failed to start; using the blocking exchange\n");
                c->a2a_pieces = 1; c->a2a_start = nullptr; c->a2a_wait = nullptr;
                HIP_RC(hipStreamSynchronize(g.stream));
                if (c->alltoall(c->xctx)) return HDSDP_RETCODE_FAILED;
                return gram_all(c) ? HDSDP_RETCODE_FAILED : HDSDP_RETCODE_OK;
            }
            return HDSDP_RETCODE_FAILED;
        }
    }
    for (int k = 0; k < P; ++k) {
        const double t0 = host_now();
        if (c->a2a_wait(c->xctx, k)) return HDSDP_RETCODE_FAILED;
        const double t1 = host_now();
        c->prof.wait_host[k] = (t1 - t0) * 1e3;
        c->prof.flight[k] = (t1 - c->pt_start[k]) * 1e3;
        // (piece after piece into the same slabs: they are launched in order on one stream)
        if (prof_record(c->pe_ga[k], g.stream) || gram_range(c, k * zper, zper, k == 0) || prof_record(c->pe_gb[k], g.stream))
            return HDSDP_RETCODE_FAILED;
    }
    return gram_reduce(c) ? HDSDP_RETCODE_FAILED : HDSDP_RETCODE_OK;
}

hdsdp_retcode build_gemm_path(MiCone *c, hdsdp_kkt *kkt, MiKKTPriv *pv, int typeKKT, HdmChol *chOverride = nullptr);
double *kkt_Mdev(hdsdp_kkt *kkt, long *ld);
HdmMatView kkt_view(hdsdp_kkt *kkt);
hdsdp_retcode build_r1_path(MiCone *c, hdsdp_kkt *kkt, MiKKTPriv *pv, int typeKKT);
hdsdp_retcode build_sparse_path(MiCone *c, hdsdp_kkt *kkt, MiKKTPriv *pv, int typeKKT);

// KKT_TYPE_PRIMAL (hdsdp_conic_sdp.c:1745-1753; driver hdsdp_psdp.c:156,203,420): the builder runs on the registered
// primal matrix X in place of S^-1.  S^-1 = Linv^T Linv enters every path only through the lower-triangular Linv, so
// X is brought to the same form: factor the index-reversed matrix J X J = F F^T on the device, then W = J F^T J is
// lower triangular with W^T W = X and takes Linv's place in the GEMM path (all strategies give the same numbers, and
// the reference itself re-routes M2 columns for this type, :1782-1788).  X must be positive definite, which a
// primal interior point is; an indefinite X is reported like a failed dpotrf.
// KKT_TYPE_PRIMAL with a registered matrix that is NOT positive definite (the primal refinement does hand such iterates
// over, hdsdp_psdp.c:203,420; the reference's trace formulas do not care): no triangular factor exists, so the product is
// formed the way the reference's M3 column does it, one owned row at a time:  B_i = X A_i X  (three plain MFMA GEMMs on
// the A_L form: X A = X A_L + X A_L^T),  then  M_ij = <A_j, B_i>  for all j in one pass over the resident constraint data.
// 4 n^3 + m n^2 flops per row instead of the congruence path's n^3 + m n^2 / 2: a fallback, used only on this condition.
hdsdp_retcode build_primal_general(MiCone *c, hdsdp_kkt *kkt, MiKKTPriv *pv, const double *X) {
    if (c->world > 1) {
        fprintf(stderr, "[hdsdp_mi355x] KKT_TYPE_PRIMAL: an indefinite primal matrix is not supported on a sharded block\n");
        return HDSDP_RETCODE_FAILED;
    }
    const int n = c->n, m = kkt->nRow;
    long ldx = 0, ldm = 0;
    RC(cone_upload_X(c, X, &ldx));
    const size_t np2 = sizeof(double) * (size_t) ldx * ldx;
    if (!c->Pr1) HIP_RC(hipMalloc((void **) &c->Pr1, np2));
    if (!c->Pr2) HIP_RC(hipMalloc((void **) &c->Pr2, np2));
    (void) ldm;
    const HdmMatView Mview = kkt_view(kkt);
    double *row = nullptr, *ALsq = nullptr;
    HIP_RC(hipMalloc((void **) &row, sizeof(double) * (size_t) m));
    HIP_RC(hipMalloc((void **) &ALsq, sizeof(double) * (size_t) c->n16 * c->n16 + hdm_operand_pad(c->n16)));
    HdmGemmArgs q = {};
    q.M = c->n16; q.N = c->n16; q.K = c->n16; q.batch = 1; q.alpha = 1.0; q.epilogue = HDM_EPI_STORE; q.ldc = ldx;
    // vectors: ASinv_i = <A_i, X>, ASinvRdSinv_i = Rd <A_i, X^2>   (Pr2 <- X X^T)
    q.A = c->Xup; q.lda = ldx; q.B = c->Xup; q.ldb = ldx; q.C = c->Pr2;
    hdsdp_retcode rc = HDSDP_RETCODE_OK;
    if (hdm_launch_gemm(q, g.stream) ||
        cone_sym_dot2(c, c->Xup, c->Pr2, ldx, pv->vecs, pv->vecs + m, 2.0, 2.0 * c->Rd))
        rc = HDSDP_RETCODE_FAILED;
    for (int qi = 0; qi < c->mloc && rc == HDSDP_RETCODE_OK; ++qi) {
        // this fallback multiplies with A_L as a generic operand in both orientations: unpack the row's skyline storage
        // into a square scratch matrix first
        const double *Arow = cone_rows(c, qi, 1);
        if (!Arow || hdm_sky_to_square(Arow, ALsq, c->n16, g.stream)) { rc = HDSDP_RETCODE_FAILED; break; }
        const double *AL = ALsq;
        // Pr1 = X A_L            (B operand element (j, k) = A_L(k, j): K-major)
        q.A = c->Xup; q.lda = ldx; q.a_kmajor = 0; q.B = AL; q.ldb = c->n16; q.b_kmajor = 1; q.C = c->Pr1; q.beta = 0.0;
        if (hdm_launch_gemm(q, g.stream)) { rc = HDSDP_RETCODE_FAILED; break; }
        // Pr1 += X A_L^T         (B operand element (j, k) = A_L(j, k): M-major)
        q.b_kmajor = 0; q.beta = 1.0;
        if (hdm_launch_gemm(q, g.stream)) { rc = HDSDP_RETCODE_FAILED; break; }
        // Pr2 = Pr1 X            (B operand element (j, k) = X(k, j): K-major)
        q.A = c->Pr1; q.lda = ldx; q.B = c->Xup; q.ldb = ldx; q.b_kmajor = 1; q.C = c->Pr2; q.beta = 0.0;
        if (hdm_launch_gemm(q, g.stream)) { rc = HDSDP_RETCODE_FAILED; break; }
        if (hipMemsetAsync(row, 0, sizeof(double) * (size_t) m, g.stream) != hipSuccess ||
            cone_sym_dot2(c, c->Pr2, nullptr, ldx, row, row, 2.0, 0.0)) { rc = HDSDP_RETCODE_FAILED; break; }
        hipLaunchKernelGGL(mi_put_row_kernel, dim3((m + 255) / 256), dim3(256), 0, g.stream, Mview, c->own[qi], row, m);
    }
    if (hipStreamSynchronize(g.stream) != hipSuccess) rc = HDSDP_RETCODE_FAILED;
    (void) hipFree(row);
    (void) hipFree(ALsq);
    if (rc != HDSDP_RETCODE_OK) return rc;
    if (c->Rd != 0.0) {                      // dTraceSinv += tr X (hdsdp_conic_sdp.c:1767-1769)
        double tr = 0.0;
        for (int i = 0; i < n; ++i) tr += X[(size_t) i * (n + 1)];
        kkt->dTraceSinv += tr;
    }
    return HDSDP_RETCODE_OK;
}

hdsdp_retcode build_primal(MiCone *c, int iCone, hdsdp_kkt *kkt, MiKKTPriv *pv) {
    if (!kkt->dPrimalX || !kkt->dPrimalX[iCone]) return HDSDP_RETCODE_FAILED;   // :1747-1750
    const double *X = kkt->dPrimalX[iCone];
    const int n = c->n;
    if (!c->primal) {
        c->primal = new HdmChol();
        if (c->primal->init(n)) return HDSDP_RETCODE_MEMORY;
    }
    std::vector<double> Xr((size_t) n * n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Xr[(size_t) i + (size_t) j * n] = X[(size_t) (n - 1 - i) + (size_t) (n - 1 - j) * n];
    HdmChol &ch = *c->primal;
    int info = 0;
    if (ch.load_host(Xr.data(), n, g.stream)) return HDSDP_RETCODE_FAILED;
    HIP_RC(hipStreamSynchronize(g.stream));   // Xr is pageable host memory going out of scope
    if (ch.factor(g.stream, &info)) return HDSDP_RETCODE_FAILED;
    if (info != 0) return build_primal_general(c, kkt, pv, X);   // X is not positive definite: no factor to lean on
    if (ch.set_reverse_inverse(g.stream)) return HDSDP_RETCODE_FAILED;
    return build_gemm_path(c, kkt, pv, KKT_TYPE_PRIMAL, &ch);
}

hdsdp_retcode cone_build_schur(void *cd, int iCone, void *kktv, int typeKKT) {
    (void) iCone;
    MiCone *c = (MiCone *) cd;
    hdsdp_kkt *kkt = (hdsdp_kkt *) kktv;
    MiKKTPriv *pv = priv_of(kkt);
    if (typeKKT == KKT_TYPE_PRIMAL) return build_primal(c, iCone, kkt, pv);
    MiLin *l = (MiLin *) c->dualFactor->chol;
    if (!l->ch.factored) {
        fprintf(stderr, "[hdsdp_mi355x] BuildSchur: the dual matrix has no valid Cholesky factor\n");
        return HDSDP_RETCODE_FAILED;
    }
    if (c->path == PATH_R1) return build_r1_path(c, kkt, pv, typeKKT);
    if (c->path == PATH_SPARSE) return build_sparse_path(c, kkt, pv, typeKKT);
    return build_gemm_path(c, kkt, pv, typeKKT);
}
hdsdp_retcode cone_build_schur_fixed(void *cd, int iCone, void *kktv, int typeKKT, int strategy) {
    (void) strategy;  // all strategies are the same numbers (reference invariant, hdsdp_utils.c:536-707)
    return cone_build_schur(cd, iCone, kktv, typeKKT);
}

// where the builders put M: the dense device matrix, or the tile store of a sparse operator in tile form
HdmMatView kkt_view(hdsdp_kkt *kkt) {
    MiLin *l = (MiLin *) kkt->kktM->chol;
    if (l->bsp) return l->bsp->view_M();
    HdmMatView v;
    v.base = l->Mdev; v.ld = l->ch.npad;
    return v;
}
double *kkt_Mdev(hdsdp_kkt *kkt, long *ld) {
    MiLin *l = (MiLin *) kkt->kktM->chol;
    if (ld) *ld = l->ch.npad;
    return l->Mdev;
}

hdsdp_retcode corrector_components(MiCone *c, HdmChol &ch, MiKKTPriv *pv, int m) {
    // ASinv_i = <A_i, S^-1>, ASinvRdSinv_i = Rd <A_i, S^-2>   (hdsdp_conic_sdp.c:1035-1056)
    const size_t nn = sizeof(double) * (size_t) ch.npad * ch.npad;
    if (!c->Xinv) { if (hipMalloc((void **) &c->Xinv, nn) != hipSuccess) return HDSDP_RETCODE_MEMORY; }
    if (!c->Yinv) { if (hipMalloc((void **) &c->Yinv, nn) != hipSuccess) return HDSDP_RETCODE_MEMORY; }
    RC(ch.inverse_full(c->Xinv, ch.npad, g.stream));
    const double *Y = nullptr;
    if (c->Rd != 0.0) {
        HdmGemmArgs q = {};  // Y = X * X^T = S^-2
        q.A = c->Xinv; q.lda = ch.npad; q.B = c->Xinv; q.ldb = ch.npad; q.C = c->Yinv; q.ldc = ch.npad;
        q.M = c->n16; q.N = c->n16; q.K = c->n16; q.batch = 1; q.alpha = 1.0; q.epilogue = HDM_EPI_STORE;
        RC(hdm_launch_gemm(q, g.stream));
        Y = c->Yinv;
    }
    // A is stored in A_L form: <A, X> = 2 <A_L, X>
    if (c->world == 1) {
        RC(cone_sym_dot2(c, c->Xinv, Y, ch.npad, pv->vecs, pv->vecs + m, 2.0, 2.0 * c->Rd));
        return HDSDP_RETCODE_OK;
    }
    // Sharded block: pv->vecs is the accumulator of the whole operator (every engine cone adds into it), so the sum
    // over the ranks runs on this cone's own contribution only and is added afterwards; reducing pv->vecs itself would
    // multiply what the cones before this one have put there by the number of ranks.
    if (!c->corr) HIP_RC(hipMalloc((void **) &c->corr, sizeof(double) * 2 * (size_t) m));
    HIP_RC(hipMemsetAsync(c->corr, 0, sizeof(double) * 2 * (size_t) m, g.stream));
    RC(cone_sym_dot2(c, c->Xinv, Y, ch.npad, c->corr, c->corr + m, 2.0, 2.0 * c->Rd));
    HIP_RC(hipStreamSynchronize(g.stream));
    if (!c->allreduce || c->allreduce(c->xctx, c->corr, (int64_t) 2 * m)) return HDSDP_RETCODE_FAILED;
    if (c->kkt_owner) RC(hdm_axpy_mat(pv->vecs, pv->vecs, c->corr, 1.0, 2L * m, g.stream));
    HIP_RC(hipStreamSynchronize(g.stream));
    return HDSDP_RETCODE_OK;
}

hdsdp_retcode build_gemm_path(MiCone *c, hdsdp_kkt *kkt, MiKKTPriv *pv, int typeKKT, HdmChol *chOverride) {
    MiLin *l = (MiLin *) c->dualFactor->chol;
    HdmChol &ch = chOverride ? *chOverride : l->ch;
    const int m = kkt->nRow;
    if (typeKKT == KKT_TYPE_CORRECTOR) return corrector_components(c, ch, pv, m);
    if (!c->work_ready) {
        if (cone_alloc_gemm_work(c)) return HDSDP_RETCODE_MEMORY;
        c->work_ready = true;
    }
    HIP_RC(hipEventRecord(g.ev[0], g.stream));
    RC(ch.invert_factor(g.stream));
    HIP_RC(hipEventRecord(g.ev[1], g.stream));
    const long opad = (long) (hdm_operand_pad(c->n16) / sizeof(double));       // slack behind Afull / CL / T (allocation sites)
    const long afull_span = c->astride * std::max(1, c->mloc) + opad;
    // Multi-GPU: run step 2 of the owned rows by packed-index range, in the order of the exchange pieces, so that a piece
    // crosses the links while the later ranges are still being computed (at two ranks the all-to-all moves 8 GB per
    // rank over a single link, more than the Gram product alone can hide).  Needs the piecewise exchange hooks, all
    // owned rows in one launch group and at most 64 tile columns.  HDSDP_MI355X_STAGED_A2A=0: drain, then exchange.
    const int NT = (c->n16 + HDM_TILE - 1) / HDM_TILE;
    int P = (c->world > 1) ? exchange_pieces(c) : 1;
    bool staged = c->world > 1 && P > 1 && P <= 64 && c->mloc <= c->Bc && NT <= 64;
    if (const char *e = getenv("HDSDP_MI355X_STAGED_A2A")) staged = staged && atoi(e) != 0;
    c->last_pieces = P; c->last_staged = 0;
    if (c->streamed) {
        // constraint data not resident: a batch is regenerated, transformed, and its buffer reused (stream order keeps the
        // generator of batch k + 1 behind step 1 of batch k, the only reader)
        staged = false;
        for (int q0 = 0; q0 < c->mloc; q0 += c->Bs) {
            const int nb = std::min(c->Bs, c->mloc - q0);
            const double *A = cone_rows(c, q0, nb);
            if (!A) return HDSDP_RETCODE_FAILED;
            RC(congruence_rows(c, ch, A, c->astride, c->astride * (long) c->Bs + opad, nb, q0));
        }
    } else if (!staged) RC(congruence_rows(c, ch, c->Afull, c->astride, afull_span, c->mloc, 0));
    if (c->rank == 0) {
        // "I row": A = I => T = Linv, At = Linv Linv^T.  Reuse step 2 with T := Linv.
        HdmGemmArgs k2 = {};
        k2.A = ch.Linv; k2.lda = ch.npad; k2.B = ch.Linv; k2.ldb = ch.npad; k2.C = c->AhatLoc;
        k2.M = c->n16; k2.N = c->n16; k2.K = c->n16; k2.batch = 1; k2.alpha = 1.0;
        k2.klimit = HDM_KLIM_BY_N; k2.lower_only = 1; k2.epilogue = HDM_EPI_BLOCKED;
        k2.blk_row_stride = c->Lr; k2.blk_row0 = c->mloc; k2.nblk = c->nblk;
        RC(hdm_launch_gemm(k2, g.stream));
        if (typeKKT == KKT_TYPE_HOMOGENEOUS) {
            if (!c->CL) {
                HIP_RC(hipMalloc((void **) &c->CL, sizeof(double) * (size_t) c->astride + hdm_operand_pad(c->n16)));
                HIP_RC(hipMemsetAsync(c->CL, 0, sizeof(double) * (size_t) c->astride, g.stream));
                RC(hdm_lower_half(c->Cfull, c->CL, c->n, c->n16, g.stream));
            }
            RC(congruence_rows(c, ch, c->CL, c->astride, c->astride + opad, 1, c->mloc + 2));
        }
    }
    if (staged) {
        RC(congruence_rows(c, ch, c->Afull, c->astride, afull_span, c->mloc, 0, 1));
        if (prof_record(c->pe_s1, g.stream)) return HDSDP_RETCODE_FAILED;
        const unsigned long long all = (NT >= 64) ? ~0ULL : ((1ULL << NT) - 1);
        unsigned long long done = 0;
        for (int k = 0; k < P; ++k) {
            unsigned long long mk = (k == P - 1 ? all : piece_tile_cols(c, k, P)) & all & ~done;
            if (mk) { RC(congruence_rows(c, ch, c->Afull, c->astride, afull_span, c->mloc, 0, 2, mk)); c->last_staged += 1; }
            done |= mk;
            if (!c->piece_ev[k]) HIP_RC(hipEventCreateWithFlags(&c->piece_ev[k], hipEventDisableTiming));
            HIP_RC(hipEventRecord(c->piece_ev[k], g.stream));
            if (prof_record(c->pe_s2[k], g.stream)) return HDSDP_RETCODE_FAILED;
        }
    }
    HIP_RC(hipEventRecord(g.ev[2], g.stream));
    c->prof.valid = false;
    if (c->world > 1) { RC(exchange_and_gram(c, staged)); }
    else { RC(gram_all(c)); }
    HIP_RC(hipEventRecord(g.ev[3], g.stream));
    if (c->world > 1) {
        HIP_RC(hipStreamSynchronize(g.stream));
        const double t0 = host_now();
        if (!c->allreduce || c->allreduce(c->xctx, c->Gm, (int64_t) c->R * c->R)) return HDSDP_RETCODE_FAILED;
        c->prof.allreduce_host = (host_now() - t0) * 1e3;
    }
    const int hsd = (typeKKT == KKT_TYPE_HOMOGENEOUS);
    const long pI = (c->world == 1) ? c->mloc : (c->m + c->world - 1) / c->world;  // rows owned by rank 0 = position of the "I row"
    if (c->kkt_owner)
        RC(hdm_extract(c->Gm, c->R, c->R, pI, c->rows_seg, kkt_view(kkt), pv->vecs, pv->vecs + m, pv->vecs + 2 * m,
                       pv->vecs + 3 * m, c->Rd, hsd, g.stream));
    HIP_RC(hipEventRecord(g.ev[4], g.stream));
    HIP_RC(hipEventSynchronize(g.ev[4]));
    float ms = 0;
    for (int i = 0; i < 4; ++i) {
        (void) hipEventElapsedTime(&ms, g.ev[i], g.ev[i + 1]);
        g.stage_ms[i] = ms;
    }
    if (c->world > 1) {
        // the sharded build's profile: every event above has completed (ev[4] is the last on the stream)
        auto el = [](hipEvent_t a, hipEvent_t b) { float t = 0.f; return (a && b && hipEventElapsedTime(&t, a, b) == hipSuccess) ? (double) t : 0.0; };
        MiCone::BuildProfile &pf = c->prof;
        const int Pp = std::max(1, std::min(c->last_pieces, MiCone::BuildProfile::MAXP));
        pf.pieces = Pp; pf.staged = staged ? 1 : 0;
        pf.invert = g.stage_ms[0]; pf.cong = g.stage_ms[1]; pf.extract = g.stage_ms[3];
        pf.step1 = staged ? el(g.ev[1], c->pe_s1) : 0.0;
        for (int k = 0; k < Pp; ++k) {
            pf.step2[k] = staged ? el(k == 0 ? c->pe_s1 : c->pe_s2[k - 1], c->pe_s2[k]) : 0.0;
            pf.wait_gpu[k] = el(k == 0 ? g.ev[2] : c->pe_gb[k - 1], c->pe_ga[k]);
            pf.gram[k] = el(c->pe_ga[k], c->pe_gb[k]);
        }
        pf.reduce = el(c->pe_gb[Pp - 1], g.ev[3]);
        pf.valid = true;
    }
    return HDSDP_RETCODE_OK;
}
