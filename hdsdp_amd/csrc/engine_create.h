// engine_create.h -- cone construction: upload of the constraint data, choice of the device path, the synthetic family
// Implementation header of engine.hip: included exactly once, there, in this order (the pieces share the anonymous namespace
// and the engine's thread-local context `g`); split out of a 3 300-line file in round 3, nothing else changed.
int grp_sum_launch(const double *const *ptrs, int W, long lo, long cnt, double *out, hipStream_t s) {
    MiGrpPtrs pl = {};
    for (int q = 0; q < W && q < 16; ++q) pl.p[q] = ptrs[q];
    hipLaunchKernelGGL(mi_grp_sum_kernel, dim3((unsigned) ((cnt + 255) / 256)), dim3(256), 0, s, pl, W, lo, cnt, out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
// ---------------------------------------------------------------- cone construction
// Rows of a presolved block -> A_L forms in skyline storage on the device.  The entries travel as they are -- (packed index,
// value), 12 bytes each -- and are scattered on the device into a zeroed target (hdm_scatter_low): host threads copy a group
// of rows into one of two pinned staging buffers while the other one crosses PCIe.  (Up to round 4 the host densified every row
// into a packed image first -- a 16 MB memset and an 800 K-entry scatter per row on one core, one synchronisation per 16 rows:
// most of the 14 s the reference's driver spent in its pre-solver at n = m = 2000.)
struct RowUploader {
    struct Stage { int *hi = nullptr, *di = nullptr; double *hv = nullptr, *dv = nullptr; long *hb = nullptr, *db = nullptr; hipEvent_t ev = nullptr; bool used = false; } st[2];
    long cap = 0;
    static constexpr int rowcap = 1024;
    int which = 0;
    int init(const MiCone *c) {
        long maxrow = 1, total = 0;
        for (int q = 0; q < c->mloc; ++q) {
            const long k = (long) c->blk.rows[c->own[q]].idx.size();
            maxrow = std::max(maxrow, k);
            total += k;
        }
        // entries per staging buffer: 192 MiB worth, or one row if that is more -- and never more than the block holds (a problem
        // of many small blocks creates many cones: each would otherwise pin 384 MiB of host memory for a few KB of entries)
        cap = std::max(maxrow, std::min((long) ((192L << 20) / 12), std::max(1L, total)));
        for (auto &b : st) {
            if (hipHostMalloc((void **) &b.hi, sizeof(int) * (size_t) cap, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void **) &b.hv, sizeof(double) * (size_t) cap, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void **) &b.hb, sizeof(long) * (rowcap + 1), hipHostMallocDefault) != hipSuccess ||
                hipMalloc((void **) &b.di, sizeof(int) * (size_t) cap) != hipSuccess || hipMalloc((void **) &b.dv, sizeof(double) * (size_t) cap) != hipSuccess ||
                hipMalloc((void **) &b.db, sizeof(long) * (rowcap + 1)) != hipSuccess || hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess)
                return 1;
        }
        return 0;
    }
    // owned rows q0 .. q0 + count - 1 into the ZEROED storage at dst (stride c->astride), queued on the engine stream
    int rows(const MiCone *c, int q0, int count, double *dst) {
        for (int r0 = q0; r0 < q0 + count;) {
            Stage &b = st[which];
            which ^= 1;
            if (b.used && hipEventSynchronize(b.ev) != hipSuccess) return 1;   // its previous group has left the buffer
            int nc = 0;
            long tot = 0, mx = 0;
            b.hb[0] = 0;
            while (r0 + nc < q0 + count && nc < rowcap) {
                const long k = (long) c->blk.rows[c->own[r0 + nc]].idx.size();
                if (nc > 0 && tot + k > cap) break;
                tot += k; mx = std::max(mx, k); nc += 1;
                b.hb[nc] = tot;
            }
            std::atomic<int> next{0};
            mi_parallel(tot >= (1L << 20) ? 0 : 1, [&](int) {
                for (int q = next.fetch_add(1); q < nc; q = next.fetch_add(1)) {
                    const MiCoeff &co = c->blk.rows[c->own[r0 + q]];
                    if (co.idx.empty()) continue;
                    memcpy(b.hi + b.hb[q], co.idx.data(), sizeof(int) * co.idx.size());
                    memcpy(b.hv + b.hb[q], co.val.data(), sizeof(double) * co.val.size());
                }
            });
            if (tot > 0) {
                if (hipMemcpyAsync(b.di, b.hi, sizeof(int) * (size_t) tot, hipMemcpyHostToDevice, g.stream) != hipSuccess ||
                    hipMemcpyAsync(b.dv, b.hv, sizeof(double) * (size_t) tot, hipMemcpyHostToDevice, g.stream) != hipSuccess ||
                    hipMemcpyAsync(b.db, b.hb, sizeof(long) * (size_t) (nc + 1), hipMemcpyHostToDevice, g.stream) != hipSuccess ||
                    hdm_scatter_low(b.di, b.dv, b.db, mx, dst + (long) (r0 - q0) * c->astride, c->astride, c->n, c->n16, nc, g.stream))
                    return 1;
            }
            if (hipEventRecord(b.ev, g.stream) != hipSuccess) return 1;
            b.used = true;
            r0 += nc;
        }
        return 0;
    }
    ~RowUploader() {
        for (auto &b : st) {
            if (b.ev) (void) hipEventSynchronize(b.ev);
            if (b.hi) (void) hipHostFree(b.hi);
            if (b.hv) (void) hipHostFree(b.hv);
            if (b.hb) (void) hipHostFree(b.hb);
            if (b.di) (void) hipFree(b.di);
            if (b.dv) (void) hipFree(b.dv);
            if (b.db) (void) hipFree(b.db);
            if (b.ev) (void) hipEventDestroy(b.ev);
        }
    }
};

static int upload_objective(MiCone *c) {   // full symmetric (it feeds the S assembly), through a packed image
    const long P = (long) c->n * (c->n + 1) / 2;
    const long nn = (long) c->n16 * c->n16;
    double *stage_dev = nullptr, *stage_host = nullptr;
    HDM_HIP_CHECK(hipMalloc((void **) &stage_dev, sizeof(double) * (size_t) P));
    HDM_HIP_CHECK(hipHostMalloc((void **) &stage_host, sizeof(double) * (size_t) P, hipHostMallocDefault));
    memset(stage_host, 0, sizeof(double) * (size_t) P);
    for (size_t e = 0; e < c->blk.obj.idx.size(); ++e) stage_host[c->blk.obj.idx[e]] = c->blk.obj.val[e];
    HDM_HIP_CHECK(hipMemcpyAsync(stage_dev, stage_host, sizeof(double) * (size_t) P, hipMemcpyHostToDevice, g.stream));
    if (hdm_unpack_sym(stage_dev, P, c->Cfull, nn, c->n, c->n16, 1, g.stream)) return 1;
    HDM_HIP_CHECK(hipStreamSynchronize(g.stream));
    (void) hipFree(stage_dev);
    (void) hipHostFree(stage_host);
    return 0;
}

// every owned constraint resident as an A_L form in skyline storage
static int upload_dense_rows(MiCone *c) {
    HDM_HIP_CHECK(hipMalloc((void **) &c->Afull, sizeof(double) * (size_t) c->astride * std::max(1, c->mloc) + hdm_operand_pad(c->n16)));
    HDM_HIP_CHECK(hdm_memset_sync(c->Afull, 0, sizeof(double) * (size_t) c->astride * std::max(1, c->mloc)));
    {
        RowUploader up;
        if (up.init(c) || up.rows(c, 0, c->mloc, c->Afull) || hipStreamSynchronize(g.stream) != hipSuccess) return 1;
    }
    return upload_objective(c);
}

// Resident or streamed constraint data (MiCone::streamed)?  Resident needs the skyline storage of all owned rows NEXT TO what a
// build takes -- the transformed rows, intermediates / Gram slabs, the Schur matrix with its factor and some slack; if the device
// does not have that, the rows are regenerated per batch instead.  HDSDP_MI355X_STREAM_A=1 forces streaming (tests run small
// blocks both ways), 0 forbids it.
static bool cone_wants_streaming(const MiCone *c) {
    const double rows = (double) std::max(1, c->mloc);
    const double afull = 8.0 * (double) c->astride * rows;
    const double ahat = 8.0 * (double) c->world * c->npb_loc * c->Lr * 16 * (c->world == 1 ? 1.0 : 2.0);
    const double work = std::min(41.0 * (1L << 30), std::max(8.0 * (double) c->n16 * c->n16 * std::min(rows, 1024.0),
                                                             8.0 * (double) c->R * c->R * 8.0));
    // (the Schur matrix, its factor and its inverse blocks, plus 16 GiB of slack: at n = 2000, m = 8000 the resident form comes
    // to 285 of the 287 GiB a fresh MI355X shows -- it has run, but nothing else may be on the device then)
    const double schur = 3.0 * 8.0 * (double) c->m * c->m + (16.0 * (1L << 30));
    size_t fr = 0, tot = 0;
    bool stream = false;
    // one device: against what is FREE now (whatever else lives on the device counts).  A shard of a sharded block
    // (process per GPU or device group): against the device's CAPACITY, so that every rank of a job on identical devices
    // takes the same decision -- streaming changes the launch staging, and ranks that disagreed would run differently
    // labelled, differently timed steps (equal results); a resident allocation that then fails is an error with a message.
    if (hipMemGetInfo(&fr, &tot) == hipSuccess)
        stream = (afull + ahat + work + schur > (c->world > 1 ? 0.97 * (double) tot : (double) fr));
    (void) hipGetLastError();
    if (const char *e = getenv("HDSDP_MI355X_STREAM_A")) stream = (atoi(e) != 0);
    return stream && c->mloc > 0;
}
static int cone_alloc_batch(MiCone *c) {
    // batch = what one congruence launch takes (cone_alloc_gemm_work evens its launches out the same way)
    long bmax = 1024;
    if (const char *e = getenv("HDM_BC")) bmax = std::max(1L, atol(e));
    const long launches = (c->mloc + bmax - 1) / bmax;
    c->Bs = (int) ((c->mloc + launches - 1) / launches);
    if (hipMalloc((void **) &c->Abatch, sizeof(double) * (size_t) c->astride * c->Bs + hdm_operand_pad(c->n16)) != hipSuccess) {
        fprintf(stderr, "[hdsdp_mi355x] cannot allocate %.1f GiB for a batch of constraint matrices\n", (double) c->astride * c->Bs * 8 / (1 << 30));
        return 1;
    }
    return hdm_memset_sync(c->Abatch, 0, sizeof(double) * (size_t) c->astride * c->Bs) != hipSuccess;
}

// INGESTED rows that are too many to stay resident (round 5): batch after batch they are scattered into the batch buffer and
// compressed into the zero-suppressed copy (two passes: counts, then values -- every batch crosses PCIe twice), which from then
// on is the ONLY image of the constraint data: the sweeps read it as they always do, and whoever needs the dense A_L forms
// (the congruence, norms, A X) has a batch expanded from it (cone_rows -> hdm_zs_expand) -- what the counter-based generator
// does for the synthetic family.  n = 2000, m = 8000 at 40 % fill: 57 GB instead of 136.
static int upload_streamed_rows(MiCone *c) {
    if (cone_alloc_batch(c)) return 1;
    {
        RowUploader up;
        if (up.init(c)) return 1;
        bool bad = false;
        auto source = [&](int q0, int nb) -> const double * {
            if (hipMemsetAsync(c->Abatch, 0, sizeof(double) * (size_t) c->astride * nb, g.stream) != hipSuccess || up.rows(c, q0, nb, c->Abatch)) { bad = true; return nullptr; }
            return c->Abatch;
        };
        if (hdm_zs_build_from(source, c->Bs, c->astride, c->mloc, c->astride, 1.0, &c->zs, g.stream) || bad) return 1;
        if (hipStreamSynchronize(g.stream) != hipSuccess) return 1;
    }
    if (!c->zs.val) {
        fprintf(stderr, "[hdsdp_mi355x] block n = %d, m = %d: no room for the compressed copy its streamed rows live in\n", c->n, c->mloc);
        return 1;
    }
    c->zs_state = 1;
    c->rows_from_zs = true;
    c->streamed = true;
    fprintf(stderr, "[hdsdp_mi355x] block n = %d, m = %d: %.1f GiB of constraint data are kept as a %.1f GiB zero-suppressed copy and expanded %d matrices at a time, not resident\n",
            c->n, c->mloc, (double) c->astride * c->mloc * 8 / (1 << 30),
            (8.0 * (double) c->zs.nnz + 192.0 * (double) c->zs.nchunk * c->mloc) / (1 << 30), c->Bs);
    return upload_objective(c);
}

// device path a block takes by itself: the rank-one fast path iff every non-zero constraint is rank one (the reference's
// all-M2 plans), the sparse gather path iff all rows are short triplet lists, else congruence + Gram; sharded blocks
// (world > 1) always take the latter
static int natural_path(const MiBlockData &blk, int nRow, int nCol, int world) {
    int nz = 0, r1 = 0;
    for (int i = 0; i < nRow; ++i) {
        int t = blk.rows[i].type;
        if (t != MI_COEFF_ZERO) nz++;
        if (t == MI_COEFF_SPR1 || t == MI_COEFF_DSR1) r1++;
    }
    int path = (nz > 0 && r1 == nz && world == 1) ? PATH_R1 : PATH_GEMM;
    if (path == PATH_GEMM && world == 1 && nz > 0) {
        // sparse gather path: only triplet-class rows, and the pair products are far cheaper than m congruences
        bool all_sparse = true;
        double tot = 0.0;
        for (int i = 0; i < nRow; ++i) {
            int t = blk.rows[i].type;
            if (t == MI_COEFF_DENSE || t == MI_COEFF_DSR1) all_sparse = false;
            tot += (double) blk.rows[i].stored;
        }
        if (all_sparse && tot * tot < 0.05 * (double) nRow * nCol * (double) nCol * nCol) path = PATH_SPARSE;
    }
    return path;
}

// the device data of one SDP block for rank `rank` of `world` on the calling thread's context, from the block's presolved
// host data.  A shard of a sharded block (world > 1) copies the entries of its own rows only (the others keep class, counts
// and trace): W shards of one block together hold one more copy of the data, not W.  With `take` the source is moved from.
static hdsdp_retcode make_sdp_cone_from_block(MiCone **out, MiBlockData &src, bool take, int rank, int world) {
    if (ensure_ctx()) return HDSDP_RETCODE_FAILED;
    const int nRow = src.m, nCol = src.n;
    MiCone *c = new MiCone();
    c->n = nCol; c->m = nRow; c->rank = rank; c->world = world;
    if (world > 1) hdm_gemm_reserve_cus(8);   // the exchange's collectives run beside the persistent GEMM launches
    if (take) c->blk = std::move(src);
    else if (world == 1) c->blk = src;
    else {
        c->blk.n = src.n; c->blk.m = src.m; c->blk.perm = src.perm; c->blk.strategy = src.strategy; c->blk.stored = src.stored;
        for (int t = 0; t < 5; ++t) c->blk.counts[t] = src.counts[t];
        c->blk.obj = src.obj;
        c->blk.rows.resize((size_t) nRow);
        for (int i = 0; i < nRow; ++i) {
            if (i % world == rank) { c->blk.rows[i] = src.rows[i]; continue; }
            MiCoeff &d = c->blk.rows[i];
            const MiCoeff &o = src.rows[i];
            d.type = o.type; d.nnz = o.nnz; d.rank = o.rank; d.stored = o.stored; d.trace = o.trace; d.sign = o.sign; d.factor_nnz = o.factor_nnz;
            d.is_eye = o.is_eye; d.eye_val = o.eye_val; d.unit_col = o.unit_col;
        }
    }
    if (cone_alloc_common(c)) return HDSDP_RETCODE_MEMORY;
    c->trA = (double *) calloc(nRow, sizeof(double));
    for (int i = 0; i < nRow; ++i) c->trA[i] = c->blk.rows[i].trace;
    c->path = natural_path(c->blk, nRow, nCol, world);
    const char *force = getenv("HDSDP_MI355X_FORCE_GEMM");
    if (force && atoi(force)) c->path = PATH_GEMM;
    const char *forcesp = getenv("HDSDP_MI355X_FORCE_PATH");
    if (forcesp && world == 1) c->path = atoi(forcesp);
    if (c->mloc == 0) c->path = PATH_GEMM;   // no constraint touches this block: only the objective's scalars remain
    {
        CreateTimer t_(1);
        // (rows too many to stay resident beside the work buffers -- congruence + Gram path only -- live in the compressed copy)
        const bool stream = (c->path == PATH_GEMM) && cone_wants_streaming(c);
        if (stream ? upload_streamed_rows(c) : upload_dense_rows(c)) return HDSDP_RETCODE_MEMORY;   // the dense forms also feed the S assembly
    }
    if (c->path == PATH_R1) {
        c->mloc16 = (int) hdm_roundup(std::max(1, c->mloc), 16);
        const size_t av = sizeof(double) * (size_t) c->n16 * c->mloc16;
        std::vector<double> hA((size_t) c->n16 * c->mloc16, 0.0), hs(c->mloc16, 0.0);
        for (int q = 0; q < c->mloc; ++q) {
            const MiCoeff &co = c->blk.rows[c->own[q]];
            if (co.type == MI_COEFF_ZERO) continue;
            for (int r = 0; r < nCol; ++r) hA[(size_t) q * c->n16 + r] = co.factor[r];
            hs[q] = co.sign;
        }
        if (hipMalloc((void **) &c->Avec, av) != hipSuccess || hipMalloc((void **) &c->U, av) != hipSuccess ||
            hipMalloc((void **) &c->V, av) != hipSuccess || hipMalloc((void **) &c->W, std::max(av, sizeof(double) * (size_t) c->n16 * c->n16)) != hipSuccess ||
            hipMalloc((void **) &c->sgn, sizeof(double) * c->mloc16) != hipSuccess ||
            hipMalloc((void **) &c->Gr1, sizeof(double) * (size_t) c->mloc16 * c->mloc16) != hipSuccess ||
            hipMalloc((void **) &c->Ct, sizeof(double) * (size_t) c->n16 * c->n16) != hipSuccess ||
            hipMalloc((void **) &c->Xinv, sizeof(double) * (size_t) c->n16 * c->n16) != hipSuccess)
            return HDSDP_RETCODE_MEMORY;
        if (hdm_memcpy_h2d_sync(c->Avec, hA.data(), av) != hipSuccess ||
            hdm_memcpy_h2d_sync(c->sgn, hs.data(), sizeof(double) * c->mloc16) != hipSuccess)
            return HDSDP_RETCODE_FAILED;
    }
    if (c->path == PATH_SPARSE) {
        std::vector<int> rp(c->mloc + 1, 0), ti, tj;
        std::vector<double> tv;
        for (int q = 0; q < c->mloc; ++q) {
            const MiCoeff &co = c->blk.rows[c->own[q]];
            for (size_t e = 0; e < co.idx.size(); ++e) {
                long pk = co.idx[e];
                int col = 0; long start = 0;
                while (pk >= start + (nCol - col)) { start += nCol - col; ++col; }
                ti.push_back(col + (int) (pk - start)); tj.push_back(col); tv.push_back(co.val[e]);
            }
            rp[q + 1] = (int) ti.size();
        }
        const size_t nt = std::max<size_t>(1, ti.size());
        const size_t nn2 = sizeof(double) * (size_t) hdm_roundup(nCol, 128) * hdm_roundup(nCol, 128);
        if (hipMalloc((void **) &c->sp_rp, sizeof(int) * rp.size()) != hipSuccess ||
            hipMalloc((void **) &c->sp_ti, sizeof(int) * nt) != hipSuccess ||
            hipMalloc((void **) &c->sp_tj, sizeof(int) * nt) != hipSuccess ||
            hipMalloc((void **) &c->sp_tv, sizeof(double) * nt) != hipSuccess ||
            hipMalloc((void **) &c->Xinv, nn2) != hipSuccess || hipMalloc((void **) &c->Yinv, nn2) != hipSuccess ||
            hipMalloc((void **) &c->W, nn2) != hipSuccess || hipMalloc((void **) &c->Ct, nn2) != hipSuccess)
            return HDSDP_RETCODE_MEMORY;
        if (hdm_memcpy_h2d_sync(c->sp_rp, rp.data(), sizeof(int) * rp.size()) != hipSuccess ||
            (ti.size() && (hdm_memcpy_h2d_sync(c->sp_ti, ti.data(), sizeof(int) * ti.size()) != hipSuccess ||
                           hdm_memcpy_h2d_sync(c->sp_tj, tj.data(), sizeof(int) * tj.size()) != hipSuccess ||
                           hdm_memcpy_h2d_sync(c->sp_tv, tv.data(), sizeof(double) * tv.size()) != hipSuccess)))
            return HDSDP_RETCODE_FAILED;
    }
    if (hipStreamSynchronize(g.stream) != hipSuccess) return HDSDP_RETCODE_FAILED;
    // On the congruence + Gram path everything downstream reads the device copy (A_L forms, the sweep copy): the host copy of
    // the entries -- 19 GB at n = m = 2000, 40 %-filled -- goes back now.  The rank-one and gather paths keep theirs (small,
    // and their host-side norms and plans read them).
    if (c->path == PATH_GEMM) { for (MiCoeff &co : c->blk.rows) co.release(); c->blk.obj.release(); }
    {
        CreateTimer t_(2);
        if (cone_build_zs(c)) return HDSDP_RETCODE_FAILED;
    }
    *out = c;
    return HDSDP_RETCODE_OK;
}

static hdsdp_retcode make_synth_cone(MiCone **out, int nCol, int nRow, int rank, int world) {
    if (ensure_ctx()) return HDSDP_RETCODE_FAILED;
    MiCone *c = new MiCone();
    c->n = nCol; c->m = nRow; c->rank = rank; c->world = world; c->synthetic = true; c->path = PATH_GEMM;
    if (world > 1) hdm_gemm_reserve_cus(8);
    if (cone_alloc_common(c)) return HDSDP_RETCODE_MEMORY;
    c->streamed = cone_wants_streaming(c);
    if (c->streamed) {
        if (cone_alloc_batch(c)) return HDSDP_RETCODE_MEMORY;
        fprintf(stderr, "[hdsdp_mi355x] block n = %d, m = %d: %.1f GiB of constraint data are streamed (regenerated %d matrices at a time), not resident\n",
                c->n, c->mloc, (double) c->astride * c->mloc * 8 / (1 << 30), c->Bs);
    } else {
        if (hipMalloc((void **) &c->Afull, sizeof(double) * (size_t) c->astride * std::max(1, c->mloc) + hdm_operand_pad(c->n16)) != hipSuccess) {
            fprintf(stderr, "[hdsdp_mi355x] cannot allocate %.1f GiB for the constraint matrices\n",
                    (double) c->astride * c->mloc * 8 / (1 << 30));
            return HDSDP_RETCODE_MEMORY;
        }
        if (hdm_memset_sync(c->Afull, 0, sizeof(double) * (size_t) c->astride * std::max(1, c->mloc)) != hipSuccess) return HDSDP_RETCODE_FAILED;
        for (int q = 0; q < c->mloc; ++q)  // owned rows are strided in the global numbering
            if (hdm_synth_fill_low(c->Afull + (long) q * c->astride, c->astride, c->n, c->n16, c->own[q], 1, g.stream)) return HDSDP_RETCODE_FAILED;
    }
    if (hdm_synth_obj(c->Cfull, c->n, c->n16, c->m, g.stream)) return HDSDP_RETCODE_FAILED;
    // b_i = tr(A_i): diagonal draws only (host, m*n splitmix evaluations)
    c->trA = (double *) calloc(nRow, sizeof(double));
    const uint64_t P = (uint64_t) nCol * (nCol + 1) / 2, gam = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < nRow; ++i) {
        double tr = 0.0;
        uint64_t k = 0;
        for (int j = 0; j < nCol; ++j) {
            uint64_t z = gam + (2 * ((uint64_t) i * P + k) + 1) * gam;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z = z ^ (z >> 31);
            tr += 2.0 * ((double) (z >> 11) / 9007199254740992.0) - 1.0;
            k += nCol - j;
        }
        c->trA[i] = tr;
    }
    if (hipStreamSynchronize(g.stream) != hipSuccess) return HDSDP_RETCODE_FAILED;
    if (cone_build_zs(c)) return HDSDP_RETCODE_FAILED;
    *out = c;
    return HDSDP_RETCODE_OK;
}
