// gemm_f64.hip -- fp64 MFMA (v_mfma_f64_16x16x4_f64) tiled GEMM family for gfx950 / MI355X.
//
// One kernel template serves every dense contraction on the Schur hot path:
//   * congruence step 1   T_i  = Linv * A_i                     (K loop cut by the row tile: Linv lower)
//   * congruence step 2   At_i = lower(T_i * Linv^T)            (K loop cut by the col tile, lower tiles,
//                                                                16x16-blocked sqrt(2)-weighted output)
//   * Gram / SYRK         M    = Ahat * Ahat^T  (K ~ n(n+1)/2)  (split-K slabs, lower tiles)
//   * blocked Cholesky    panel TRSM via inverted diagonal block, trailing SYRK update, TRTRI products
//
// These replace the reference's level-1/2 BLAS loops: dspmv x n + ddot x n(n+1)/2 in
// kkt3sinvAsinv (linalg/hdsdp_sdpdata.c:1184-1223), the packed dot kkt3AdotB (:1304-1328) and
// dpotrf/dpotri/dtrsm (linalg/hdsdp_linsolver.c:1096,1250,1158).
//
// Geometry: 128x128 workgroup tile, 4 waves (2x2) of 64x64, 16 accumulators of 16x16 per wave
// (128 VGPRs), BK=16 per LDS stage, 2 stages (72 KiB LDS => 2 workgroups/CU).
// LDS images are padded so every ds_read_b64 fragment read is bank-conflict free:
//   M-major tile  s[k][i], row stride 144 doubles (1152 B == 128 mod 256)
//   K-major tile  s[i][k], row stride  18 doubles ( 144 B : 16 lanes hit 16 distinct bank pairs)
// MFMA operand map (f64 16x16x4): lane l supplies X[x0 + (l&15)][k0 + (l>>4)] for both operands;
// result lane l, reg r holds D[row=(l>>4)+4r][col=l&15].  We feed the C-row operand (A) as the
// MFMA "B" input so that D's lane-contiguous index is C's row index => 128-byte contiguous stores.
#include "gemm_tile.h"
#include "gemm_persist.h"
// ---------------------------------------------------------------------------------------------
// host side: tile lists (heaviest first) cached in device memory
// ---------------------------------------------------------------------------------------------
namespace {
struct TimedLaunch { hipEvent_t e0, e1; int role; double flops, issued; };
bool g_timing = false;
thread_local bool g_capturing = false;   // per host thread: each shard of a device group captures its own chains
std::vector<TimedLaunch> g_timed;
unsigned long long *g_dbg = nullptr;
int g_dbg_role = -1;
struct TileList {
    int2 *dev = nullptr;
    int n = 0;
};
std::mutex g_tl_mutex;
std::map<std::tuple<int, int, int, int, int, unsigned long long>, TileList> g_tl_cache;

// counters of the persistent launches: a ring of 8-int slots per device (a launch zeroes its slot on its own stream;
// launches on different streams of one device -- the loopback rehearsal of a device group -- never share a slot), and the
// number of workgroups the device holds at two per CU
// CUs a persistent launch leaves free (hdm_gemm_reserve_cus): a sharded build overlaps its exchange -- RCCL kernels on a
// side stream -- with these launches, and a grid that fills every slot of the chip would hold the collective's workgroups
// back until the whole launch has drained
int g_reserved_cus = 0;
struct PersistDev { int *ring = nullptr; unsigned seq = 0; int slots = 0; };
std::map<int, PersistDev> g_persist_dev;
constexpr int PERSIST_RING = 256;
int persist_counters(int **cnt, int *slots) {
    int dev = 0;
    HDM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_tl_mutex);
    PersistDev &pd = g_persist_dev[dev];
    if (!pd.ring) {
        int cus = 0;
        HDM_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        HDM_HIP_CHECK(hipMalloc((void **) &pd.ring, sizeof(int) * 8 * PERSIST_RING));
        pd.slots = 2 * std::max(1, cus);
    }
    static const int env_reserve = [] { const char *e = getenv("HDM_PERSIST_RESERVE_CUS"); return e ? atoi(e) : -1; }();
    const int reserve = env_reserve >= 0 ? env_reserve : g_reserved_cus;
    *cnt = pd.ring + 8 * (pd.seq++ % PERSIST_RING);
    *slots = std::max(2, pd.slots - 2 * reserve);
    return 0;
}

// subset: 0 = all tiles; 1 = the full diagonal tiles only (tm == tn with 8 valid sub-tile rows; rows = the M dimension);
// 2 = all the others
int get_tiles(int MT, int NT, int klimit, int lower_only, unsigned long long colmask, TileList &out, int subset = 0, int rows = 0) {
    int dev = 0;
    HDM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_tl_mutex);
    if (NT > 64) colmask = 0;
    auto key = std::make_tuple(dev, MT, NT, klimit + 16 * subset + 64 * (subset ? rows : 0), lower_only, colmask);
    auto it = g_tl_cache.find(key);
    if (it != g_tl_cache.end()) {
        out = it->second;
        return 0;
    }
    std::vector<std::pair<long, int2>> v;
    for (int tm = 0; tm < MT; ++tm)
        for (int tn = 0; tn < NT; ++tn) {
            if (lower_only && tm < tn) continue;
            if (colmask && !((colmask >> tn) & 1ULL)) continue;
            if (subset) {
                const bool full_diag = (tm == tn) && (((rows - tm * HDM_TILE + 15) >> 4) >= 8);
                if ((subset == 1) != full_diag) continue;
            }
            long w = 1;
            if (klimit == HDM_KLIM_BY_M) w = tm + 1;
            if (klimit == HDM_KLIM_BY_N) w = tn + 1;
            if (klimit == HDM_KLIM_BAND) { if (tm < tn) continue; w = tm - tn + 1; }
            v.push_back({w, make_int2(tm, tn)});
        }
    // (tiles of equal weight stay in row-major order: a Z-order curve over (tm, tn), meant to let neighbours in the list
    // share a column panel as well as a row panel in L2, RAISED the Gram kernel's fabric traffic from 391 to 425 GB per
    // launch and changed no time -- same box, round 2)
    std::stable_sort(v.begin(), v.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
    std::vector<int2> h(v.size());
    for (size_t i = 0; i < v.size(); ++i) h[i] = v[i].second;
    TileList tl;
    tl.n = (int) h.size();
    HDM_HIP_CHECK(hipMalloc((void **) &tl.dev, sizeof(int2) * std::max<size_t>(1, h.size())));
    HDM_HIP_CHECK(hdm_memcpy_h2d_sync(tl.dev, h.data(), sizeof(int2) * h.size()));
    g_tl_cache[key] = tl;
    out = tl;
    return 0;
}
}  // namespace

static int launch_tiles(const HdmGemmArgs &args, int subset, hipStream_t stream);

// Flops the MFMA instructions of one launch EXECUTE (2048 per v_mfma_f64_16x16x4_f64), counted on the host from the tile
// list and the kernel's own stage sequences (gemm_tile.h: hdm_gemm_tile) -- next to HdmGemmArgs.flops, the algorithmic count
// on valid data.  The difference is granularity: 16 x 16 sub-blocks that straddle a diagonal, the live ranges of the
// triangular K blocks (stage-level skipping keeps 1280 of a diagonal block's 2048 MFMAs per tile where 1152 would do), the
// cell-dealt tiles' diagonal blocks run in full, rows past the matrix edge, the zero stage that closes an odd stage count.
// Held to the counters: SQ_VALU_MFMA_BUSY_CYCLES / 64 of profiles/r05_a_8000_* is this count to 2e-5 for the four kernels.
static double issued_mfma_flops(const HdmGemmArgs &a, int subset) {
    const int MT = (a.M + HDM_TILE - 1) / HDM_TILE, NT = (a.N + HDM_TILE - 1) / HDM_TILE;
    const unsigned long long colmask = (NT > 64) ? 0ULL : a.tile_col_mask;
    const int npass = (a.role == HDM_ROLE_CONG2 && a.A2) ? 2 : ((a.role == HDM_ROLE_GENERIC && a.A2) ? 2 : 1);
    auto pairs = [](long nst) { return nst <= 0 ? 0L : ((nst + 1) / 2) * 2; };
    double mf = 0.0;   // MFMA instructions of ONE batch entry (SLAB: summed over the splits below)
    const int nz = (a.epilogue == HDM_EPI_SLAB) ? a.batch : 1;
    for (int z = 0; z < nz; ++z)
        for (int tm = 0; tm < MT; ++tm)
            for (int tn = 0; tn < NT; ++tn) {
                if (a.lower_only && tm < tn) continue;
                if (colmask && !((colmask >> tn) & 1ULL)) continue;
                if (a.klimit == HDM_KLIM_BAND && tm < tn) continue;
                const int m0 = tm * HDM_TILE, n0 = tn * HDM_TILE;
                const int rvd = std::min(8, (a.M - m0 + 15) >> 4);
                if (subset) {
                    const bool full_diag = (tm == tn) && rvd >= 8;
                    if ((subset == 1) != full_diag) continue;
                }
                long kbeg = 0, kend = a.K;
                if (a.klimit == HDM_KLIM_BY_M) kend = std::min<long>(a.K, (long) (tm + 1) * HDM_TILE);
                if (a.klimit == HDM_KLIM_BY_N) kend = std::min<long>(a.K, (long) (tn + 1) * HDM_TILE);
                if (a.klimit == HDM_KLIM_BAND) { kbeg = (long) tn * HDM_TILE; kend = std::min<long>(a.K, (long) (tm + 1) * HDM_TILE); }
                if (a.epilogue == HDM_EPI_SLAB) { kbeg = a.k_base + (long) z * a.k_chunk; kend = std::min(kend, kbeg + a.k_chunk); }
                const long nst = std::max(0L, kend / HDM_BK - kbeg / HDM_BK) * npass;
                if (a.role == HDM_ROLE_GENERIC) { mf += 256.0 * nst; continue; }
                const int rvc = std::max(4, rvd);   // the cell lists exist for 4..7 valid sub-tile rows
                const bool symdiag = (a.role == HDM_ROLE_CONG2D);
                if (!symdiag && a.lower_only && tm == tn) {                              // diagonal tile, cell-dealt
                    const int cells = (rvd < 8) ? rvc * (rvc + 1) / 2 : 36;
                    mf += 4.0 * cells * pairs(nst);
                } else if (!symdiag && tm != tn && m0 + HDM_TILE > a.M && n0 + HDM_TILE <= a.N) {   // bottom-edge tile
                    mf += 4.0 * (8 * rvc) * pairs(nst);
                } else if (a.role == HDM_ROLE_CONG2) {      // both products: full stages, then the diagonal block's live ranges 16+12+8+4
                    mf += 2.0 * ((double) tn * 8 * 256 + 1280);
                } else if (a.role == HDM_ROLE_CONG2D) {     // one product, both operands triangular in the last block: 16+9+4+1
                    mf += (double) tn * 8 * 256 + 960;
                } else if (a.role == HDM_ROLE_CONG1) {      // first and last K block triangular on one side each
                    mf += 2.0 * 1280 + (double) (tm - tn - 1) * 2048;
                } else {
                    mf += 256.0 * pairs(nst);
                }
            }
    return mf * 2048.0 * ((a.epilogue == HDM_EPI_SLAB) ? 1.0 : (double) a.batch);
}

int hdm_launch_gemm(const HdmGemmArgs &args, hipStream_t stream) {
    if (args.M <= 0 || args.N <= 0 || args.batch <= 0) return 0;
    if ((args.M % 8) || (args.N % 8) || (args.K % HDM_BK)) {
        fprintf(stderr, "[hdsdp_mi355x] gemm: M,N must be multiples of 8 and K of 16 (M=%d N=%d K=%d)\n",
                args.M, args.N, args.K);
        return 1;
    }
    if (args.epilogue == HDM_EPI_SLAB && (args.k_chunk <= 0 || args.k_chunk % HDM_BK)) {
        fprintf(stderr, "[hdsdp_mi355x] gemm: bad k_chunk\n");
        return 1;
    }
    const int NT = (args.N + HDM_TILE - 1) / HDM_TILE;
    if (args.role == HDM_ROLE_CONG2) {
        // the role IS the SYR2K form lower(U W^T + W U^T) into the blocked layout: its full diagonal tiles compute one
        // product and add the transpose (gemm_tile.h: symdiag), which is only that tile if the second pair mirrors the first
        const bool mirrored = args.A2 == args.B && args.B2 == args.A && args.lda2 == args.ldb && args.ldb2 == args.lda &&
                              args.strideA2 == args.strideB && args.strideB2 == args.strideA;
        if (!mirrored || !args.lower_only || args.epilogue != HDM_EPI_BLOCKED || args.klimit != HDM_KLIM_BY_N ||
            args.M != args.N || args.a_kmajor || args.b_kmajor) {
            fprintf(stderr, "[hdsdp_mi355x] gemm role 2 is lower(A B^T + B A^T), blocked output, K cut by the column tile: launch refused\n");
            return 1;
        }
    }
    if (args.role != HDM_ROLE_GENERIC) {
        // Roles 1-3 fetch their tiles with buffer loads whose offsets -- lane part + chunk part, 32 bits -- must stay below the
        // 2 GiB the buffer spans from the k block's scalar base (gemm_tile.h: SStager): 127 rows of a K-major operand, 12 k rows
        // of an M-major one
        const long lds_[4] = {args.lda, args.ldb, args.A2 ? args.lda2 : 0, args.B2 ? args.ldb2 : 0};
        for (long ld : lds_)
            if (ld < 0 || ld > (1L << 20)) {
                fprintf(stderr, "[hdsdp_mi355x] gemm role %d: leading dimension %ld is beyond what the scalar stager addresses: launch refused\n", args.role, ld);
                return 1;
            }
        // Roles 1-3 stage whole 128-row tiles WITHOUT a row mask (SStager::load_nomask): rows past the matrix edge are
        // read and thrown away.  Every such launch therefore states how many elements are readable from each operand
        // pointer, and the launch is refused unless the farthest element an unmasked load can touch lies inside.
        auto farthest = [&](bool kmajor, long ld, long kblk, long stride, int rows, bool seg) {
            const long maxrow = (long) ((rows + HDM_TILE - 1) / HDM_TILE) * HDM_TILE - 1;
            long kfirst = 0, klast = args.K - 1;
            if (args.epilogue == HDM_EPI_SLAB) {
                kfirst = args.k_base;
                klast = std::min<long>(args.K, args.k_base + (long) args.batch * args.k_chunk) - 1;
            }
            (void) kfirst;
            const long nb = (args.epilogue == HDM_EPI_SLAB) ? 1 : args.batch;
            long off = (nb - 1) * stride;
            if (kmajor) {
                off += (klast / HDM_BK) * (kblk ? kblk : HDM_BK) + maxrow * ld + (HDM_BK - 1);
                if (seg && args.seg_rows) off += (maxrow / args.seg_rows) * args.seg_extra;
            } else {
                off += klast * ld + maxrow;
            }
            return off + 1;
        };
        struct { const double *p; long span; bool km; long ld, kblk, stride; int rows; const char *nm; } ops[4] = {
            {args.A, args.spanA, args.a_kmajor != 0, args.lda, args.a_kblk, args.strideA, args.M, "A"},
            {args.B, args.spanB, args.b_kmajor != 0, args.ldb, args.b_kblk, args.strideB, args.N, "B"},
            {args.A2, args.spanA2, args.a_kmajor != 0, args.lda2, (long) HDM_BK, args.strideA2, args.M, "A2"},
            {args.B2, args.spanB2, args.b_kmajor != 0, args.ldb2, (long) HDM_BK, args.strideB2, args.N, "B2"}};
        for (auto &o : ops) {
            if (!o.p) continue;
            long need = farthest(o.km, o.ld, o.kblk, o.stride, o.rows, true);
            if (o.p == args.B && args.b_sky) {
                // skyline operand: the last panel (width w < 128 columns, height w) is read as 128 "rows" of its leading
                // dimension, i.e. up to (128 - w) columns' worth past the matrix
                const int t = (args.N + HDM_TILE - 1) / HDM_TILE - 1;
                const long h = args.N - 128L * t;
                need = (long) (args.batch - 1) * args.strideB + hdm_sky_panel(t, args.N) + 127 * h + h;
            }
            if (o.span < need) {
                fprintf(stderr, "[hdsdp_mi355x] gemm role %d: operand %s needs %ld readable elements for its unmasked tile "
                                "loads, the caller vouches for %ld: launch refused\n", args.role, o.nm, need, o.span);
                return 1;
            }
        }
    }
    if (args.role == HDM_ROLE_CONG2) {
        // two kernels: the full diagonal tiles as P + P^T (role HDM_ROLE_CONG2D), then everything else.  The launch's
        // algorithmic flops are split by what the tiles hold: output element (i, j), i >= j, is 2 products x (j + 1) terms
        double all = 0.0, diag = 0.0;
        for (int tn = 0; tn < NT; ++tn) {
            if (args.tile_col_mask && NT <= 64 && !((args.tile_col_mask >> tn) & 1ULL)) continue;
            const int j0 = tn * HDM_TILE, j1 = std::min(args.N, j0 + HDM_TILE);
            const bool full = (((args.M - j0 + 15) >> 4) >= 8);
            for (int j = j0; j < j1; ++j) {
                all += (double) (args.M - j) * (j + 1);
                if (full) diag += (double) (j1 - j) * (j + 1);
            }
        }
        const double share = all > 0.0 ? diag / all : 0.0;
        HdmGemmArgs dg = args;
        dg.role = HDM_ROLE_CONG2D; dg.A2 = nullptr; dg.B2 = nullptr; dg.flops = args.flops * share;
        if (launch_tiles(dg, 1, stream)) return 1;
        HdmGemmArgs mn = args;
        mn.flops = args.flops * (1.0 - share);
        return launch_tiles(mn, 2, stream);
    }
    return launch_tiles(args, 0, stream);
}

// one kernel launch over a subset of the tile list (get_tiles); args.role selects the kernel
static int launch_tiles(const HdmGemmArgs &args, int subset, hipStream_t stream) {
    const int MT = (args.M + HDM_TILE - 1) / HDM_TILE, NT = (args.N + HDM_TILE - 1) / HDM_TILE;
    TileList tl;
    if (get_tiles(MT, NT, args.klimit, args.lower_only, args.tile_col_mask, tl, subset, args.M)) return 1;
    if (tl.n == 0) return 0;
    HdmGemmDev d;
    d.a = args;
    if (d.a.a_kblk == 0) d.a.a_kblk = HDM_BK;
    if (d.a.b_kblk == 0) d.a.b_kblk = HDM_BK;
    d.tiles = tl.dev;
    d.ntiles = tl.n;
    d.dbg = (g_dbg && args.role == g_dbg_role) ? g_dbg : nullptr;
#ifdef HDM_DIAGNOSTICS
    if (d.dbg && !g_capturing && getenv("HDM_DBG_SYNC")) HDM_HIP_CHECK(hipDeviceSynchronize());   // diagnostic: isolate the stamped launch from its neighbours
#endif
    // Kernel variants.  The shipped library carries ONE loop for the three Schur roles (variant 64: rotated, explicitly
    // interleaved K loop, persistent and one-workgroup-per-tile forms) and the masked loop of generic launches (variant 0).
    // A -DHDM_DIAGNOSTICS build (python -m hdsdp_amd.build --diagnostics) adds what the measurement tools select with HDM_VAR /
    // HDM_CONG2_DIRECT -- and nothing else can: 0 = the earlier loop for the roles (same-box A/B); +32 = per-workgroup
    // s_memtime stamps (tools/wg_timeline*.py); 192 = timing-only ablation in which every tile stages rows 0..127 (all
    // operand traffic L2-resident, WRONG RESULTS); 576 = step 2 without its main tiles' stores (WRONG RESULTS, persistent form
    // only); 320 = the LDS-free, barrier-free step-2 body (cong2_direct_body; bit-identical results; measured twice on one box:
    // same time with one workgroup per tile, 1.4 % slower than the LDS loop with persistent workgroups).
    int var = 64;
#ifdef HDM_DIAGNOSTICS
    static const int g_env_var = [] { const char *e = getenv("HDM_VAR"); return e ? atoi(e) : -1; }();
    if (g_env_var >= 0) var = g_env_var;
    if (var == 576 && args.role != HDM_ROLE_CONG2) var = 64;
    if (args.role == HDM_ROLE_CONG2) {
        static const int g_direct = [] { const char *e = getenv("HDM_CONG2_DIRECT"); return e ? atoi(e) : 0; }();
        const bool direct = g_direct && var == 64 && args.A2 && !args.a_kmajor && !args.b_kmajor &&
                            args.epilogue == HDM_EPI_BLOCKED && args.lower_only && args.klimit == HDM_KLIM_BY_N &&
                            args.A2 == args.B && args.B2 == args.A;
        if (direct) var = 320;
    }
#endif
    long nwg = (long) tl.n * (args.batch >= 8 ? ((args.batch + 7) & ~7) : args.batch);
    // roles 1-3 with a batch: persistent workgroups drawing tiles from per-XCD counters (hdm_gemm_persist_kernel), where
    // the variant has a persistent form (gemm_persist.hip); the grid is then what the chip holds
    static const int g_persist = [] { const char *e = getenv("HDM_PERSIST"); return e ? atoi(e) : 1; }();
    int *cnt = nullptr;
    const bool persist = g_persist && args.role != HDM_ROLE_GENERIC && args.batch >= 8 && !g_capturing &&
                         hdm_persist_supported(args.a_kmajor != 0, args.b_kmajor != 0, args.role, var);
    if (!persist && (var == 320 || var == 576)) var = 64;       // persistent-only variants
    if (persist) {
        int slots = 0;
        if (persist_counters(&cnt, &slots)) return 1;
        HDM_HIP_CHECK(hipMemsetAsync(cnt, 0, 8 * sizeof(int), stream));
        nwg = std::min<long>(nwg, slots);
    }
    dim3 grid((unsigned) nwg), block(256);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = g_timing && !g_capturing;
    if (timed) {
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
        HDM_HIP_CHECK(hipEventRecord(e0, stream));
    }
    const int lay = (args.a_kmajor ? 2 : 0) | (args.b_kmajor ? 1 : 0);
    if (persist) {
        if (hdm_launch_persist(args.a_kmajor != 0, args.b_kmajor != 0, args.role, var, grid, block, stream, d, cnt)) return 1;
    } else {
#define HDM_LAUNCH(AK, BK, R, V) hipLaunchKernelGGL((hdm_gemm_kernel<AK, BK, R, V>), grid, block, 0, stream, d)
#ifdef HDM_DIAGNOSTICS
#define HDM_LAUNCH_V(AK, BK, R)                                   \
    switch (var) {                                                \
        case 32: HDM_LAUNCH(AK, BK, R, 32); break;                \
        case 96: HDM_LAUNCH(AK, BK, R, 96); break;                \
        case 192: HDM_LAUNCH(AK, BK, R, 192); break;              \
        case 0: HDM_LAUNCH(AK, BK, R, 0); break;                  \
        default: HDM_LAUNCH(AK, BK, R, 64);                       \
    }
#else
#define HDM_LAUNCH_V(AK, BK, R) HDM_LAUNCH(AK, BK, R, 64)
#endif
        switch (args.role) {
            case HDM_ROLE_CONG1: HDM_LAUNCH_V(false, true, HDM_ROLE_CONG1); break;   // U = Linv * A_L (A_L k-contiguous)
            case HDM_ROLE_CONG2: HDM_LAUNCH_V(false, false, HDM_ROLE_CONG2); break;  // At = U Linv^T + Linv U^T
            case HDM_ROLE_GRAM: HDM_LAUNCH_V(true, true, HDM_ROLE_GRAM); break;      // M = Ahat Ahat^T
            case HDM_ROLE_CONG2D: HDM_LAUNCH_V(false, false, HDM_ROLE_CONG2D); break; // step 2's full diagonal tiles: P + P^T
            default:
                if (lay == 3) HDM_LAUNCH(true, true, HDM_ROLE_GENERIC, 0);
                else if (lay == 2) HDM_LAUNCH(true, false, HDM_ROLE_GENERIC, 0);
                else if (lay == 1) HDM_LAUNCH(false, true, HDM_ROLE_GENERIC, 0);
#ifdef HDM_DIAGNOSTICS
                else if (var == 32) HDM_LAUNCH(false, false, HDM_ROLE_GENERIC, 32);   // diagnostic stamps, tools/wg_timeline_gemm.py
#endif
                else HDM_LAUNCH(false, false, HDM_ROLE_GENERIC, 0);
        }
    }
#undef HDM_LAUNCH_V
#undef HDM_LAUNCH
    if (timed) {
        HDM_HIP_CHECK(hipEventRecord(e1, stream));
        std::lock_guard<std::mutex> lk(g_tl_mutex);
        g_timed.push_back({e0, e1, args.role, args.flops, issued_mfma_flops(args, subset)});
    }
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

void hdm_timing_enable(int on) { g_timing = (on != 0); }
void hdm_gemm_capture_mode(int on) { g_capturing = (on != 0); }
void hdm_gemm_reserve_cus(int cus) { g_reserved_cus = std::max(0, cus); }

int hdm_timing_collect(double *ms, double *flops, long *launches, double *issued) {
    for (int r = 0; r < HDM_NROLES; ++r) { ms[r] = 0.0; flops[r] = 0.0; launches[r] = 0; if (issued) issued[r] = 0.0; }
    std::lock_guard<std::mutex> lk(g_tl_mutex);
    for (auto &t : g_timed) {
        float e = 0.f;
        HDM_HIP_CHECK(hipEventSynchronize(t.e1));
        HDM_HIP_CHECK(hipEventElapsedTime(&e, t.e0, t.e1));
        const int r = (t.role >= 0 && t.role < HDM_NROLES) ? t.role : 0;
        ms[r] += e; flops[r] += t.flops; launches[r] += 1;
        if (issued) issued[r] += t.issued;
        (void) hipEventDestroy(t.e0);
        (void) hipEventDestroy(t.e1);
    }
    g_timed.clear();
    return 0;
}

// diagnostic: per-workgroup timestamps of the next launches with the given role (HDM_VAR=32 builds the stamps in)
void hdm_set_debug_buffer(unsigned long long *dev, int role) { g_dbg = dev; g_dbg_role = role; }

// one kernel of this translation unit (= one code object): what the preload thread asks the runtime about (engine.hip: preload_modules)
const void *hdm_module_handle_gemm_f64() { return (const void *) hdm_gemm_kernel<false, false, HDM_ROLE_GENERIC, 0>; }
