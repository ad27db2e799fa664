// group_impl.h -- single-process multi-device mode of the engine.  Implementation header: included exactly once, by
// engine.hip, inside its anonymous namespace (it needs MiCone and the cone slots defined there).
//
// Why: the reference's driver is ONE single-threaded process (tests/sdpasolve.c, interface/hdsdp_algo.c:1082-1101), so
// "constraints sharded over up to 8 GPUs" has to happen behind the C ABI, below HKKTBuildUp.  With a device group
// configured (HMiSetDevices, or HDSDP_MI355X_GPUS=N in the environment of an unchanged driver) a dense SDP block that
// takes the congruence + Gram path is created as a GROUP CONE: W shards (MiCone with rank r of world W, the same sharded
// code path the process-per-GPU mode runs under torchrun), each on its own device with its own context and a persistent
// host worker thread.  Every slot of the group cone runs the plain cone's slot on all shards at once; the replicated
// state (S, its factor, dS, the checker) evolves identically on every shard, results are taken from shard 0, and only
// shard 0 writes into the caller's Schur operator, whose M lives on device ids[0].
//
// The two collectives of the sharded build are implemented here behind the same hooks
// (hmi_alltoall_piece_fn / hmi_alltoall_wait_fn / hmi_allreduce_fn) the process-per-GPU mode fills from Python:
//   transport COPIES : device-to-device copies (one per destination) and a fixed-order sum kernel.  The DEFAULT: between
//                      peer-accessible devices (xGMI: the copies are direct device-to-device transfers) and the only
//                      choice when several shards share one device (HDSDP_MI355X_LOOPBACK=1: a rehearsal of world = N
//                      on a 1-GPU box, which is where every line of it has been run and checked).
//   transport RCCL   : ncclCommInitAll over the group's devices; the piecewise all-to-all is a grouped ncclSend/ncclRecv
//                      per piece on a side stream, the all-reduce ncclAllReduce on the engine stream.  OPT-IN
//                      (HDSDP_MI355X_TRANSPORT=rccl), or chosen by itself when the devices are not peer-accessible: this
//                      pool has no multi-GPU box, so between two devices it has only ever run as a one-rank self-test
//                      (HMiRcclSelfTest) -- UNVERIFIED ON HARDWARE until tests/test_gpu_group.py::
//                      test_config5_at_size_on_eight_devices[rccl] has passed somewhere.  Its waits poll and watch the
//                      group's failure flag; a failed shard aborts the communicators so that nobody blocks in a collective
//                      a peer will never join, and the group refuses further work.
// (engine.hip includes <atomic>, <condition_variable>, <functional>, <mutex>, <thread> and <rccl/rccl.h> at global scope)

enum { GRP_COPIES = 0, GRP_RCCL = 1 };
#define GRP_MAX_SHARDS 16
#define GRP_MAX_PIECES 64

struct MiGroup {
    int W = 0;
    std::vector<int> dev;
    std::vector<Ctx> ctx;            // one context per shard (device, engine stream, events)
    std::vector<hipStream_t> cstream;  // exchange stream per shard
    bool shared_device = false;
    int transport = GRP_COPIES;
    std::vector<ncclComm_t> comm;
    int min_n = 512;                 // blocks smaller than this stay on one device
    int n_cones = 0;
    // worker pool: thread r is bound to device dev[r] and context ctx[r] for its whole life
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int(int)> job;
    long gen = 0;
    int done = 0;
    std::vector<int> rc;
    bool quit = false;
    std::atomic<bool> failed{false};
    bool dead = false;               // a collective was aborted (RCCL communicators gone): the group takes no more jobs
    // barrier among the workers
    std::mutex bmu;
    std::condition_variable bcv;
    int bcount = 0;
    long bgen = 0;
    // all-reduce of the copy transport
    std::vector<double *> ar_ptr, ar_tmp;
    std::vector<size_t> ar_cap;

    int barrier() {
        std::unique_lock<std::mutex> lk(bmu);
        const long my = bgen;
        if (++bcount == W) { bcount = 0; ++bgen; bcv.notify_all(); return failed.load() ? 1 : 0; }
        bcv.wait(lk, [&] { return bgen != my || failed.load(); });
        return failed.load() ? 1 : 0;
    }
    void fail() { failed.store(true); std::lock_guard<std::mutex> lk(bmu); bcv.notify_all(); }
    // wait for an event / a stream's work without blocking in the runtime: a peer that failed will never enqueue its half
    // of a collective, so the wait watches the failure flag; with RCCL the communicators are aborted then (that is what
    // releases a kernel already spinning on the peer) and the group is dead
    int wait_event(hipEvent_t ev, int r) {
        for (long spins = 0;; ++spins) {
            const hipError_t e = hipEventQuery(ev);
            if (e == hipSuccess) return 0;
            if (e != hipErrorNotReady) { (void) hipGetLastError(); fail(); return 1; }
            if (failed.load()) {
                if (transport == 1 /* GRP_RCCL */ && r >= 0 && r < (int) comm.size() && comm[r]) {
                    (void) ncclCommAbort(comm[r]);
                    comm[r] = nullptr;
                    dead = true;
                }
                return 1;
            }
            if (spins < 2000) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }

    void worker(int r) {
        if (hipSetDevice(dev[r]) != hipSuccess) fprintf(stderr, "[hdsdp_mi355x] shard %d: cannot select device %d\n", r, dev[r]);
        t_ctx = &ctx[r];
        long seen = 0;
        for (;;) {
            std::function<int(int)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
                f = job;
            }
            const int res = f(r);
            if (res) fail();
            {
                std::lock_guard<std::mutex> lk(mu);
                rc[r] = res;
                ++done;
            }
            cv.notify_all();
        }
    }
    // run f(r) on every worker and wait; 0 iff all returned 0
    int run(std::function<int(int)> f) {
        if (dead) { fprintf(stderr, "[hdsdp_mi355x] the device group was shut down by an earlier failure\n"); return 1; }
        failed.store(false);
        // a failed job may have left workers counted at the barrier: every job starts from a clean one
        { std::lock_guard<std::mutex> lk(bmu); bcount = 0; ++bgen; }
        std::unique_lock<std::mutex> lk(mu);
        job = std::move(f);
        done = 0;
        ++gen;
        cv.notify_all();
        cv.wait(lk, [&] { return done == W; });
        int any = 0;
        for (int r = 0; r < W; ++r) any |= rc[r];
        return any;
    }
};

MiGroup *g_group = nullptr;
bool g_group_env_done = false;

struct MiConeGroup;
struct ShardX { MiConeGroup *cg; int r; };
struct MiConeGroup {
    MiGroup *G = nullptr;
    int n = 0, m = 0;
    std::vector<MiCone *> shard;
    std::vector<ShardX> x;
    // exchange bookkeeping
    std::vector<double *> sendp, recvp;
    std::vector<char> pub;
    std::vector<long> xgen;                              // exchanges started so far, per shard
    hipEvent_t pev[GRP_MAX_SHARDS][GRP_MAX_PIECES] = {};   // [source][piece]: the source's copies of that piece are queued
    std::atomic<long> posted[GRP_MAX_SHARDS][GRP_MAX_PIECES];
    long bytes_a2a = 0, bytes_ar = 0;                    // moved by shard 0 (statistics)
};

// ------------------------------------------------------------------------------------------------ group set-up
int group_setup(int n, const int *ids, int transport_request) {
    if (g_group) {
        if (g_group->n_cones > 0) { fprintf(stderr, "[hdsdp_mi355x] HMiSetDevices: cones of the previous device group are still alive\n"); return 1; }
        MiGroup *o = g_group;
        { std::lock_guard<std::mutex> lk(o->mu); o->quit = true; }
        o->cv.notify_all();
        for (auto &t : o->th) t.join();
        for (auto c : o->comm) if (c) (void) ncclCommDestroy(c);
        delete o;
        g_group = nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "[hdsdp_mi355x] no HIP device visible\n"); return 1; }
    if (n < 1 || n > GRP_MAX_SHARDS) { fprintf(stderr, "[hdsdp_mi355x] HMiSetDevices: 1..%d shards\n", GRP_MAX_SHARDS); return 1; }
    for (int r = 0; r < n; ++r)
        if (ids[r] < 0 || ids[r] >= ndev) { fprintf(stderr, "[hdsdp_mi355x] HMiSetDevices: device %d is not visible (%d device(s))\n", ids[r], ndev); return 1; }
    if (g_main.init && g_main.device != ids[0]) {
        fprintf(stderr, "[hdsdp_mi355x] HMiSetDevices: the engine already runs on device %d; the group's first device must be that one\n", g_main.device);
        return 1;
    }
    g_main_device_request = ids[0];
    if (ensure_ctx()) return 1;
    if (n == 1) return 0;          // one device: the plain single-device engine
    MiGroup *G = new MiGroup();
    G->W = n;
    G->dev.assign(ids, ids + n);
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if (ids[a] == ids[b]) G->shared_device = true;
    // every error below leaves through `bail`: what was created is destroyed and the caller's thread is back on its device
    auto bail = [&](const char *what) {
        if (what) fprintf(stderr, "[hdsdp_mi355x] HMiSetDevices: %s\n", what);
        for (auto c : G->comm) if (c) (void) ncclCommDestroy(c);
        for (int r = 0; r < (int) G->ctx.size(); ++r) {
            if (!G->ctx[r].init) continue;
            (void) hipSetDevice(ids[r]);
            if (G->cstream[r]) (void) hipStreamDestroy(G->cstream[r]);
            if (G->ctx[r].stream) (void) hipStreamDestroy(G->ctx[r].stream);
            for (int i = 0; i < 8; ++i) (void) hipEventDestroy(G->ctx[r].ev[i]);
        }
        (void) hipGetLastError();
        delete G;
        (void) hipSetDevice(g_main.device);
        return 1;
    };
    // transport: device copies unless RCCL is asked for (transport_request, or HDSDP_MI355X_TRANSPORT=rccl) -- see the header
    if (transport_request < 0) if (const char *t = getenv("HDSDP_MI355X_TRANSPORT")) transport_request = (strcmp(t, "rccl") == 0) ? GRP_RCCL : GRP_COPIES;
    G->transport = (transport_request == GRP_RCCL && !G->shared_device) ? GRP_RCCL : GRP_COPIES;
    if (transport_request == GRP_RCCL && G->shared_device)
        fprintf(stderr, "[hdsdp_mi355x] RCCL needs one device per shard; shards share a device here: using device copies\n");
    G->ctx.resize(n);
    G->cstream.assign(n, nullptr);
    G->rc.assign(n, 0);
    G->ar_ptr.assign(n, nullptr); G->ar_tmp.assign(n, nullptr); G->ar_cap.assign(n, 0);
    for (int r = 0; r < n; ++r) {
        if (ctx_open(G->ctx[r], ids[r])) return bail("cannot open a context on one of the devices");
        if (hipStreamCreateWithFlags(&G->cstream[r], hipStreamNonBlocking) != hipSuccess) return bail("cannot create the exchange stream");
    }
    if (!G->shared_device && G->transport == GRP_COPIES) {
        // the copies go device to device and the sum kernel reads the other shards' buffers in place: peer access both ways
        bool peers = true;
        for (int a = 0; a < n && peers; ++a)
            for (int b = 0; b < n && peers; ++b) {
                if (a == b) continue;
                int can = 0;
                (void) hipDeviceCanAccessPeer(&can, ids[a], ids[b]);
                if (!can) peers = false;
            }
        if (!peers) {
            if (transport_request == GRP_COPIES) return bail("the devices are not peer-accessible: the copy transport cannot be used");
            fprintf(stderr, "[hdsdp_mi355x] devices are not peer-accessible: using the RCCL transport (unverified between devices on this pool)\n");
            G->transport = GRP_RCCL;
        } else {
            for (int a = 0; a < n; ++a)
                for (int b = 0; b < n; ++b) {
                    if (a == b) continue;
                    (void) hipSetDevice(ids[a]);
                    hipError_t e = hipDeviceEnablePeerAccess(ids[b], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bail("hipDeviceEnablePeerAccess failed");
                    (void) hipGetLastError();
                }
        }
    }
    if (G->transport == GRP_RCCL) {
        G->comm.assign(n, nullptr);
        ncclResult_t e = ncclCommInitAll(G->comm.data(), n, G->dev.data());
        if (e != ncclSuccess) { fprintf(stderr, "[hdsdp_mi355x] ncclCommInitAll failed: %s\n", ncclGetErrorString(e)); return bail(nullptr); }
    }
    (void) hipSetDevice(ids[0]);
    hdm_flow_set_shared_device(G->shared_device ? 1 : 0);
    if (const char *e = getenv("HDSDP_MI355X_SHARD_MIN_N")) G->min_n = atoi(e);
    for (int r = 0; r < n; ++r) G->th.emplace_back([G, r] { G->worker(r); });
    g_group = G;
    fprintf(stderr, "[hdsdp_mi355x] device group: %d shards on device(s)", n);
    for (int r = 0; r < n; ++r) fprintf(stderr, " %d", ids[r]);
    fprintf(stderr, ", transport %s\n", G->transport == GRP_RCCL ? "RCCL" : "device copies");
    return 0;
}

// HDSDP_MI355X_GPUS=N in the environment of an unchanged driver: devices 0..N-1.  With fewer devices visible this is an
// error unless HDSDP_MI355X_LOOPBACK=1 asks for a rehearsal (shard r on device r mod visible).
int group_configure_from_env() {
    if (g_group_env_done || g_group) return 0;
    g_group_env_done = true;
    const char *e = getenv("HDSDP_MI355X_GPUS");
    if (!e || atoi(e) <= 1) return 0;
    const int n = atoi(e);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return 1;
    const char *lb = getenv("HDSDP_MI355X_LOOPBACK");
    const bool loop = lb && atoi(lb) != 0;
    if (n > ndev && !loop) {
        fprintf(stderr, "[hdsdp_mi355x] HDSDP_MI355X_GPUS=%d but %d device(s) visible (HDSDP_MI355X_LOOPBACK=1 rehearses the "
                        "sharding on the devices there are)\n", n, ndev);
        return 1;
    }
    int ids[GRP_MAX_SHARDS];
    if (n > GRP_MAX_SHARDS) return 1;
    for (int r = 0; r < n; ++r) ids[r] = loop ? (r % ndev) : r;
    return group_setup(n, ids, -1);
}

bool group_wants_synthetic(int nRow, int nCol) {
    MiGroup *G = g_group;
    return G && G->W >= 2 && nCol >= G->min_n && nRow >= G->W;
}
bool group_wants_block(const MiBlockData &blk) {
    if (!group_wants_synthetic(blk.m, blk.n)) return false;
    if (const char *f = getenv("HDSDP_MI355X_FORCE_GEMM")) if (atoi(f)) return true;
    return natural_path(blk, blk.m, blk.n, 1) == PATH_GEMM;
}

// ------------------------------------------------------------------------------------------------ collectives
int grp_sum_launch(const double *const *ptrs, int W, long lo, long cnt, double *out, hipStream_t s);   // kernel at global scope

int grp_a2a_start(void *vx, int64_t off, int64_t cnt, int piece) {
    ShardX *x = (ShardX *) vx;
    MiConeGroup *cg = x->cg;
    MiGroup *G = cg->G;
    const int r = x->r, W = G->W;
    MiCone *c = cg->shard[r];
    if (piece < 0 || piece >= GRP_MAX_PIECES) return 1;
    if (piece == 0) cg->xgen[r] += 1;
    const int64_t chunk = (int64_t) c->npb_loc * c->Lr * 16;
    if (!cg->pub[r]) {   // first exchange of this block: every shard learns where the others receive
        cg->sendp[r] = c->AhatLoc; cg->recvp[r] = c->AhatAll;
        cg->pub[r] = 1;
        if (G->barrier()) return 1;
    }
    if (!cg->pev[r][piece] && hipEventCreateWithFlags(&cg->pev[r][piece], hipEventDisableTiming) != hipSuccess) return 1;
    if (G->transport == GRP_RCCL) {
        if (!G->comm[r] || ncclGroupStart() != ncclSuccess) return 1;
        for (int d = 0; d < W; ++d) {
            if (ncclSend(c->AhatLoc + d * chunk + off, (size_t) cnt, ncclDouble, d, G->comm[r], G->cstream[r]) != ncclSuccess) return 1;
            if (ncclRecv(c->AhatAll + d * chunk + off, (size_t) cnt, ncclDouble, d, G->comm[r], G->cstream[r]) != ncclSuccess) return 1;
        }
        if (ncclGroupEnd() != ncclSuccess) return 1;
    } else {
        for (int d = 0; d < W; ++d)
            if (hipMemcpyAsync(cg->recvp[d] + r * chunk + off, c->AhatLoc + d * chunk + off, sizeof(double) * (size_t) cnt,
                               hipMemcpyDefault, G->cstream[r]) != hipSuccess) return 1;
    }
    if (hipEventRecord(cg->pev[r][piece], G->cstream[r]) != hipSuccess) return 1;
    cg->posted[r][piece].store(cg->xgen[r], std::memory_order_release);
    if (r == 0) cg->bytes_a2a += (long) cnt * 8 * (W - 1);
    return 0;
}

int grp_a2a_wait(void *vx, int piece) {
    ShardX *x = (ShardX *) vx;
    MiConeGroup *cg = x->cg;
    MiGroup *G = cg->G;
    const int r = x->r, W = G->W;
    if (G->transport == GRP_RCCL)   // my receives are part of my own grouped call
        return G->wait_event(cg->pev[r][piece], r);
    for (int q = 0; q < W; ++q) {
        while (cg->posted[q][piece].load(std::memory_order_acquire) < cg->xgen[r]) {
            if (G->failed.load()) return 1;
            std::this_thread::yield();
        }
        if (G->wait_event(cg->pev[q][piece], r)) return 1;
    }
    return 0;
}

int grp_a2a(void *vx) {   // the whole exchange as one piece
    ShardX *x = (ShardX *) vx;
    MiCone *c = x->cg->shard[x->r];
    if (grp_a2a_start(vx, 0, (int64_t) c->npb_loc * c->Lr * 16, 0)) return 1;
    return grp_a2a_wait(vx, 0);
}

// in-place sum of `count` doubles at `buf` over the shards; the engine stream of every shard is idle on entry and the
// data is in place on return.  Copy transport: shard r sums slice r of all buffers in shard order (so every shard ends up
// with bit-identical numbers), then everybody collects the slices.
int grp_allreduce(void *vx, void *buf, int64_t count) {
    ShardX *x = (ShardX *) vx;
    MiConeGroup *cg = x->cg;
    MiGroup *G = cg->G;
    const int r = x->r, W = G->W;
    hipStream_t s = G->ctx[r].stream;
    if (r == 0) cg->bytes_ar += (long) count * 8;
    if (G->transport == GRP_RCCL) {
        if (!G->comm[r] || ncclAllReduce(buf, buf, (size_t) count, ncclDouble, ncclSum, G->comm[r], s) != ncclSuccess) return 1;
        hipEvent_t ev = G->ctx[r].ev[7];
        if (hipEventRecord(ev, s) != hipSuccess) return 1;
        return G->wait_event(ev, r);
    }
    const int64_t per = (count + W - 1) / W;
    const int64_t lo = std::min<int64_t>(count, (int64_t) r * per), hi = std::min<int64_t>(count, lo + per);
    G->ar_ptr[r] = (double *) buf;
    if ((size_t) per > G->ar_cap[r]) {
        if (G->ar_tmp[r]) (void) hipFree(G->ar_tmp[r]);
        if (hipMalloc((void **) &G->ar_tmp[r], sizeof(double) * (size_t) per) != hipSuccess) { G->ar_tmp[r] = nullptr; G->ar_cap[r] = 0; return 1; }
        G->ar_cap[r] = (size_t) per;
    }
    if (G->barrier()) return 1;
    if (hi > lo) {
        if (grp_sum_launch(G->ar_ptr.data(), W, lo, hi - lo, G->ar_tmp[r], s)) return 1;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (G->barrier()) return 1;
    for (int q = 0; q < W; ++q) {
        const int64_t qlo = std::min<int64_t>(count, (int64_t) q * per), qhi = std::min<int64_t>(count, qlo + per);
        if (qhi > qlo && hipMemcpyAsync((double *) buf + qlo, G->ar_tmp[q], sizeof(double) * (size_t) (qhi - qlo),
                                        hipMemcpyDefault, s) != hipSuccess) return 1;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    return G->barrier();   // nobody may reuse its slice buffer before everyone has collected it
}

// ------------------------------------------------------------------------------------------------ group cone slots
template <class F> hdsdp_retcode grun(MiConeGroup *cg, F f) {
    if (g_main.init) (void) hipStreamSynchronize(g_main.stream);   // whatever the caller's thread queued (e.g. HKKTClean) is done
    const int rc = cg->G->run([&](int r) { return (int) f(r, cg->shard[r]); });
    return rc ? HDSDP_RETCODE_FAILED : HDSDP_RETCODE_OK;
}

void gc_setstart(void *cd, double v) { for (MiCone *c : ((MiConeGroup *) cd)->shard) cone_setstart(c, v); }
void gc_reduce_resi(void *cd, double v) { for (MiCone *c : ((MiConeGroup *) cd)->shard) cone_reduce_resi(c, v); }
void gc_set_perturb(void *cd, double v) { for (MiCone *c : ((MiConeGroup *) cd)->shard) cone_set_perturb(c, v); }
// feature detection and view: shard 0 holds class, rank, sign and factor of EVERY row (make_sdp_cone_from_block keeps those for
// the rows a shard does not own), which is all cone_getstat reads
void gc_getstat(void *cd, double *rowRHS, int intF[20], double dblF[20]) { cone_getstat(((MiConeGroup *) cd)->shard[0], rowRHS, intF, dblF); }
void gc_view(void *cd) { cone_view(((MiConeGroup *) cd)->shard[0]); }
int gc_getdim(void *cd) { return ((MiConeGroup *) cd)->n; }
int64_t gc_getsymnnz(void *cd) { MiConeGroup *cg = (MiConeGroup *) cd; return (int64_t) cg->m * cg->m; }

void gc_update(void *cd, double tau, double *y) {
    StatScope stat_(ST_ASSEMBLE_FACTOR, __func__);
    (void) grun((MiConeGroup *) cd, [&](int, MiCone *c) { cone_update(c, tau, y); return 0; });
}
hdsdp_retcode gc_interior(void *cd, double tau, double *y, int *isInterior) {
    StatScope stat_(ST_ASSEMBLE_FACTOR, __func__);
    MiConeGroup *cg = (MiConeGroup *) cd;
    std::vector<int> res(cg->G->W, 0);
    hdsdp_retcode rc = grun(cg, [&](int r, MiCone *c) { return cone_interior(c, tau, y, &res[r]); });
    if (isInterior) *isInterior = res[0];
    return rc;
}
hdsdp_retcode gc_interior_expert(void *cd, double dCCoef, double dACoefScal, double *dACoef, double dEyeCoef, int whichBuffer,
                                 int *isInterior) {
    StatScope stat_(ST_ASSEMBLE_FACTOR, __func__);
    MiConeGroup *cg = (MiConeGroup *) cd;
    std::vector<int> res(cg->G->W, 0);
    hdsdp_retcode rc = grun(cg, [&](int r, MiCone *c) {
        return cone_interior_expert(c, dCCoef, dACoefScal, dACoef, dEyeCoef, whichBuffer, &res[r]); });
    if (isInterior) *isInterior = res[0];
    return rc;
}
hdsdp_retcode gc_axpy_check(void *cd, double dStep, int whichBuffer, int *isInterior) {
    StatScope stat_(ST_ASSEMBLE_FACTOR, __func__);
    MiConeGroup *cg = (MiConeGroup *) cd;
    std::vector<int> res(cg->G->W, 0);
    hdsdp_retcode rc = grun(cg, [&](int r, MiCone *c) { return cone_axpy_check(c, dStep, whichBuffer, &res[r]); });
    if (isInterior) *isInterior = res[0];
    return rc;
}
hdsdp_retcode gc_barrier(void *cd, double tau, double *y, int whichBuffer, double *logdet) {
    StatScope stat_(ST_ASSEMBLE_FACTOR, __func__);
    MiConeGroup *cg = (MiConeGroup *) cd;
    std::vector<double> res(cg->G->W, 0.0);
    hdsdp_retcode rc = grun(cg, [&](int r, MiCone *c) { return cone_barrier(c, tau, y, whichBuffer, &res[r]); });
    if (logdet && rc == HDSDP_RETCODE_OK) *logdet = res[0];
    return rc;
}
hdsdp_retcode gc_ratio_test(void *cd, double dTauStep, double *dy, double dAdaRatio, int whichBuffer, double *maxStep) {
    StatScope stat_(ST_RATIO, __func__);
    MiConeGroup *cg = (MiConeGroup *) cd;
    std::vector<double> res(cg->G->W, 0.0);
    hdsdp_retcode rc = grun(cg, [&](int r, MiCone *c) { return cone_ratio_test(c, dTauStep, dy, dAdaRatio, whichBuffer, &res[r]); });
    if (maxStep && rc == HDSDP_RETCODE_OK) *maxStep = res[0];
    return rc;
}
hdsdp_retcode gc_build_schur(void *cd, int iCone, void *kktv, int typeKKT) {
    MiConeGroup *cg = (MiConeGroup *) cd;
    hdsdp_retcode rc = grun(cg, [&](int, MiCone *c) { return cone_build_schur(c, iCone, kktv, typeKKT); });
    for (int i = 0; i < 8; ++i) g_main.stage_ms[i] = cg->G->ctx[0].stage_ms[i];
    return rc;
}
hdsdp_retcode gc_build_schur_fixed(void *cd, int iCone, void *kktv, int typeKKT, int strategy) {
    (void) strategy;
    return gc_build_schur(cd, iCone, kktv, typeKKT);
}
// the slots below change no replicated state and have no collective inside: shard 0 alone answers
void gc_build_primal_dir(void *cd, void *kktv, double *X, double *XSX, int iDualMat) {
    StatScope stat_(ST_PRIMAL_UTIL, __func__);
    (void) grun((MiConeGroup *) cd, [&](int r, MiCone *c) { if (r == 0) cone_build_primal_dir(c, kktv, X, XSX, iDualMat); return 0; });
}
double gc_trace_cx(void *cd, double *X) {
    StatScope stat_(ST_PRIMAL_UTIL, __func__);
    double v = NAN;
    (void) grun((MiConeGroup *) cd, [&](int r, MiCone *c) { if (r == 0) v = cone_trace_cx(c, X); return 0; });
    return v;
}
double gc_x_dot_s(void *cd, double *X) {
    StatScope stat_(ST_PRIMAL_UTIL, __func__);
    double v = NAN;
    (void) grun((MiConeGroup *) cd, [&](int r, MiCone *c) { if (r == 0) v = cone_x_dot_s(c, X); return 0; });
    return v;
}
void gc_get_dual(void *cd, double *S, double *aux) {
    StatScope stat_(ST_PRIMAL_UTIL, __func__);
    (void) grun((MiConeGroup *) cd, [&](int r, MiCone *c) { if (r == 0) cone_get_dual(c, S, aux); return 0; });
}
// these have a collective inside (rows are sharded) or move replicated state: every shard runs them, shard 0's output counts
void gc_a_times_x(void *cd, double *X, double *ATimesX) {
    StatScope stat_(ST_PRIMAL_UTIL, __func__);
    MiConeGroup *cg = (MiConeGroup *) cd;
    (void) grun(cg, [&](int r, MiCone *c) {
        if (r == 0) { cone_a_times_x(c, X, ATimesX); return 0; }
        std::vector<double> scratch(cg->m, 0.0);
        cone_a_times_x(c, X, scratch.data());
        return 0; });
}
void gc_precover(void *cd, double mu, double *y, double *dy, double *X, double *aux) {
    StatScope stat_(ST_PRIMAL_UTIL, __func__);
    MiConeGroup *cg = (MiConeGroup *) cd;
    (void) grun(cg, [&](int r, MiCone *c) {
        if (r == 0) { cone_precover(c, mu, y, dy, X, aux); return 0; }
        std::vector<double> scratch((size_t) cg->n * cg->n);
        cone_precover(c, mu, y, dy, scratch.data(), nullptr);
        return 0; });
}
double gc_coeff_norm(void *cd, int whichNorm) {
    StatScope stat_(ST_PRIMAL_UTIL, __func__);
    MiConeGroup *cg = (MiConeGroup *) cd;
    std::vector<double> res(cg->G->W, NAN);
    (void) grun(cg, [&](int r, MiCone *c) { res[r] = cone_coeff_norm(c, whichNorm); return 0; });
    return res[0];
}
double gc_obj_norm(void *cd, int whichNorm) {
    StatScope stat_(ST_PRIMAL_UTIL, __func__);
    MiConeGroup *cg = (MiConeGroup *) cd;
    std::vector<double> res(cg->G->W, NAN);
    (void) grun(cg, [&](int r, MiCone *c) { res[r] = cone_obj_norm(c, whichNorm); return 0; });
    return res[0];
}
void gc_scal(void *cd, double dScal) { StatScope stat_(ST_PRIMAL_UTIL, __func__); (void) grun((MiConeGroup *) cd, [&](int, MiCone *c) { cone_scal(c, dScal); return 0; }); }

void gc_destroy_data(void **pcd) {
    if (!pcd || !*pcd) return;
    MiConeGroup *cg = (MiConeGroup *) *pcd;
    (void) grun(cg, [&](int r, MiCone *) { void *p = cg->shard[r]; cone_destroy_data(&p); cg->shard[r] = nullptr; return 0; });
    for (auto &row : cg->pev) for (hipEvent_t e : row) if (e) (void) hipEventDestroy(e);
    cg->G->n_cones -= 1;
    delete cg;
    *pcd = nullptr;
}

MiCone *cone_data(hdsdp_cone *cone) {
    if (cone->coneBuildSchur == gc_build_schur) return ((MiConeGroup *) cone->coneData)->shard[0];
    return (MiCone *) cone->coneData;
}

hdsdp_retcode group_create_cone(hdsdp_cone **pCone, int iCone, int nRow, int nCol, MiBlockData *blk, bool synthetic) {
    MiGroup *G = g_group;
    MiConeGroup *cg = new MiConeGroup();
    cg->G = G; cg->n = nCol; cg->m = nRow;
    cg->shard.assign(G->W, nullptr);
    cg->x.resize(G->W);
    cg->sendp.assign(G->W, nullptr); cg->recvp.assign(G->W, nullptr);
    cg->pub.assign(G->W, 0); cg->xgen.assign(G->W, 0);
    for (auto &row : cg->posted) for (auto &a : row) a.store(0);
    std::vector<hdsdp_retcode> rcs(G->W, HDSDP_RETCODE_OK);
    (void) grun(cg, [&](int r, MiCone *) {
        rcs[r] = synthetic ? make_synth_cone(&cg->shard[r], nCol, nRow, r, G->W)
                           : make_sdp_cone_from_block(&cg->shard[r], *blk, false, r, G->W);   // (each shard copies its own rows)
        return 0; });
    for (int r = 0; r < G->W; ++r)
        if (rcs[r] != HDSDP_RETCODE_OK || !cg->shard[r]) {
            fprintf(stderr, "[hdsdp_mi355x] shard %d of the block could not be created\n", r);
            const hdsdp_retcode bad = rcs[r] != HDSDP_RETCODE_OK ? rcs[r] : HDSDP_RETCODE_FAILED;
            // give back what the other shards took (tens of GB per device at n = 2000), each on its own device's thread
            (void) grun(cg, [&](int q, MiCone *) { if (cg->shard[q]) { void *p = cg->shard[q]; cone_destroy_data(&p); cg->shard[q] = nullptr; } return 0; });
            delete cg;
            return bad;
        }
    int pieces = 8;
    if (const char *e = getenv("HDSDP_MI355X_A2A_PIECES")) pieces = std::max(1, std::min(GRP_MAX_PIECES, atoi(e)));
    for (int r = 0; r < G->W; ++r) {
        MiCone *c = cg->shard[r];
        cg->x[r].cg = cg; cg->x[r].r = r;
        c->alltoall = grp_a2a; c->allreduce = grp_allreduce; c->xctx = &cg->x[r];
        c->a2a_start = grp_a2a_start; c->a2a_wait = grp_a2a_wait; c->a2a_pieces = pieces;
        c->kkt_owner = (r == 0);
    }
    G->n_cones += 1;
    hdsdp_cone *h = (hdsdp_cone *) calloc(1, sizeof(hdsdp_cone));
    h->iCone = iCone;
    h->cone = HDSDP_CONETYPE_DENSE_SDP;
    h->coneData = cg;
    h->coneDestroyData = gc_destroy_data;
    h->coneSetStart = gc_setstart;
    h->coneUpdate = gc_update;
    h->coneGetSymNnz = gc_getsymnnz;
    h->coneGetDim = gc_getdim;
    h->coneBuildSchur = gc_build_schur;
    h->coneBuildSchurFixed = gc_build_schur_fixed;
    h->coneBuildPrimalDirection = gc_build_primal_dir;
    h->coneInteriorCheck = gc_interior;
    h->coneRatioTest = gc_ratio_test;
    h->conePRecover = gc_precover;
    h->coneInteriorCheckExpert = gc_interior_expert;
    h->coneAxpyBufferAndCheck = gc_axpy_check;
    h->coneReduceResi = gc_reduce_resi;
    h->coneSetPerturb = gc_set_perturb;
    h->getstat = gc_getstat;
    h->coneView = gc_view;
    h->coneGetCoeffNorm = gc_coeff_norm;
    h->coneGetObjNorm = gc_obj_norm;
    h->coneScal = gc_scal;
    h->coneATimesXpy = gc_a_times_x;
    h->coneTraceCX = gc_trace_cx;
    h->coneXDotS = gc_x_dot_s;
    h->coneDRecover = gc_get_dual;
    h->coneGetBarrier = gc_barrier;
    *pCone = h;
    return HDSDP_RETCODE_OK;
}

// a presolved block becomes a cone: a group cone if a device group is configured and wants it, else a plain one
static hdsdp_retcode cone_from_block(hdsdp_cone **pCone, int iCone, MiBlockData &blk, int rank, int world) {
    if (group_configure_from_env()) return HDSDP_RETCODE_FAILED;
    if (ensure_ctx()) return HDSDP_RETCODE_FAILED;
    if (rank == 0 && world == 1 && group_wants_block(blk)) return group_create_cone(pCone, iCone, blk.m, blk.n, &blk, false);
    MiCone *c = nullptr;
    hdsdp_retcode rc = make_sdp_cone_from_block(&c, blk, true, rank, world);
    if (rc != HDSDP_RETCODE_OK) return rc;
    *pCone = new_cone_shell(c, iCone);
    return HDSDP_RETCODE_OK;
}
template <class Off> static hdsdp_retcode cone_create_csc(hdsdp_cone **pCone, int iCone, int nRow, int nCol, const Off *beg,
                                                          const int *idx, const double *val, int rank, int world) {
    if (!pCone || nRow < 1 || nCol < 1 || nCol > 65535 || world < 1 || rank < 0 || rank >= world || !beg) return HDSDP_RETCODE_FAILED;
    MiBlockData blk;
    {
        CreateTimer t_(0);
        if (mi_block_from_csc(blk, nRow, nCol, beg, idx, val)) {
            fprintf(stderr, "[hdsdp_mi355x] cone data: a column's pointers run backwards, or a packed index lies outside [0, n(n+1)/2) or is given twice\n");
            return HDSDP_RETCODE_FAILED;
        }
    }
    g_create_bytes += 12.0 * (double) blk.stored;
    return cone_from_block(pCone, iCone, blk, rank, world);
}
// column-by-column ingest (engine_api.h: HMiConeBuilder*): the block's presolved columns as they arrive
struct MiConeBuilder {
    int iCone = 0, rank = 0, world = 1;
    MiBlockData blk;
    std::vector<char> seen;
};

// RCCL self-test over the WHOLE group `ids` (n distinct devices; n = 1 is the one-rank form a 1-GPU box can run):
// communicators (ncclCommInitAll), then from one host thread per device -- the way the group's workers drive it -- an
// all-reduce and the grouped ncclSend/ncclRecv one piece of the sharded build's exchange is made of (grp_a2a_start), each
// checked for its values.  Every wait polls with a deadline; a rank that fails or runs out of time raises the shared flag
// and aborts its communicator, the others see the flag and abort theirs, so that nobody blocks in a collective a peer
// will never join.  (ncclCommInitAll itself cannot be bounded from inside: bench.py runs this test in a child process
// with a timeout before it lets RCCL carry the exchange.)
// 0 = passed; 1 bad arguments / device; 2 communicators; 3 allocation / copies; 4 all-reduce did not complete;
// 5 all-reduce values; 6 send/receive did not complete; 7 send/receive values; 8 shards share a device
int rccl_group_self_test(int n, const int *ids, int timeout_ms) {
    if (n < 1 || n > GRP_MAX_SHARDS || !ids) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return 1;
    for (int a = 0; a < n; ++a) {
        if (ids[a] < 0 || ids[a] >= ndev) return 1;
        for (int b = a + 1; b < n; ++b) if (ids[a] == ids[b]) return 8;
    }
    if (timeout_ms <= 0) timeout_ms = 60000;
    std::vector<ncclComm_t> comm(n, nullptr);
    std::vector<int> dev(ids, ids + n);
    if (ncclCommInitAll(comm.data(), n, dev.data()) != ncclSuccess) {
        for (auto c : comm) if (c) (void) ncclCommAbort(c);
        if (g_main.init) (void) hipSetDevice(g_main.device);
        return 2;
    }
    const int cnt = 1 << 18;                 // doubles per peer and piece: 2 MiB, enough to leave the eager protocol
    std::atomic<bool> failed{false};
    std::vector<int> rcs(n, 0);
    // the values are small multiples of 1/4 (sums exact in any order) resp. plain copies: both are compared bit for bit
    auto ar_val = [](int r, int i) { return (double) (r + 1) * (0.25 * (double) (i % 4096) - 7.0); };
    auto sr_val = [](int src, int dst, int i) { return 1000.0 * src + dst + 0.0009765625 * (double) (i % 1024); };
    auto rank = [&](int r) {
        int rc = 0;
        hipStream_t s = nullptr;
        hipEvent_t ev = nullptr;
        double *a = nullptr, *snd = nullptr, *rcv = nullptr;
        std::vector<double> h((size_t) cnt * n), out((size_t) cnt * n);
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
        auto bounded_wait = [&]() {          // 0: the event has completed
            for (;;) {
                const hipError_t e = hipEventQuery(ev);
                if (e == hipSuccess) return 0;
                if (e != hipErrorNotReady) { (void) hipGetLastError(); return 1; }
                if (failed.load() || std::chrono::steady_clock::now() > deadline) return 1;
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        };
        do {
            if (hipSetDevice(dev[r]) != hipSuccess) { rc = 1; break; }
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
                hipMalloc((void **) &a, sizeof(double) * cnt) != hipSuccess ||
                hipMalloc((void **) &snd, sizeof(double) * (size_t) cnt * n) != hipSuccess ||
                hipMalloc((void **) &rcv, sizeof(double) * (size_t) cnt * n) != hipSuccess) { rc = 3; break; }
            for (int i = 0; i < cnt; ++i) h[i] = ar_val(r, i);
            if (hipMemcpy(a, h.data(), sizeof(double) * cnt, hipMemcpyHostToDevice) != hipSuccess) { rc = 3; break; }
            for (int d = 0; d < n; ++d) for (int i = 0; i < cnt; ++i) h[(size_t) d * cnt + i] = sr_val(r, d, i);
            if (hipMemcpy(snd, h.data(), sizeof(double) * (size_t) cnt * n, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemset(rcv, 0, sizeof(double) * (size_t) cnt * n) != hipSuccess) { rc = 3; break; }
            // all-reduce (grp_allreduce)
            if (ncclAllReduce(a, a, (size_t) cnt, ncclDouble, ncclSum, comm[r], s) != ncclSuccess ||
                hipEventRecord(ev, s) != hipSuccess || bounded_wait()) { rc = 4; break; }
            if (hipMemcpy(out.data(), a, sizeof(double) * cnt, hipMemcpyDeviceToHost) != hipSuccess) { rc = 3; break; }
            const double tri = 0.5 * n * (n + 1);
            for (int i = 0; i < cnt && !rc; ++i) if (out[i] != tri * (0.25 * (double) (i % 4096) - 7.0)) rc = 5;
            if (rc) break;
            // one piece of the exchange (grp_a2a_start): chunk d of my send buffer to rank d, chunk d of my receive buffer from rank d
            bool okq = (ncclGroupStart() == ncclSuccess);
            for (int d = 0; d < n && okq; ++d)
                okq = ncclSend(snd + (size_t) d * cnt, (size_t) cnt, ncclDouble, d, comm[r], s) == ncclSuccess &&
                      ncclRecv(rcv + (size_t) d * cnt, (size_t) cnt, ncclDouble, d, comm[r], s) == ncclSuccess;
            if (!okq || ncclGroupEnd() != ncclSuccess || hipEventRecord(ev, s) != hipSuccess || bounded_wait()) { rc = 6; break; }
            if (hipMemcpy(out.data(), rcv, sizeof(double) * (size_t) cnt * n, hipMemcpyDeviceToHost) != hipSuccess) { rc = 3; break; }
            for (int d = 0; d < n && !rc; ++d)
                for (int i = 0; i < cnt && !rc; ++i) if (out[(size_t) d * cnt + i] != sr_val(d, r, i)) rc = 7;
        } while (0);
        if (rc) {                            // release whoever waits for this rank, then leave without joining anything else
            failed.store(true);
            if (comm[r]) { (void) ncclCommAbort(comm[r]); comm[r] = nullptr; }
        } else if (failed.load() && comm[r]) {
            (void) ncclCommAbort(comm[r]); comm[r] = nullptr;
        }
        if (a) (void) hipFree(a);
        if (snd) (void) hipFree(snd);
        if (rcv) (void) hipFree(rcv);
        if (ev) (void) hipEventDestroy(ev);
        if (s) (void) hipStreamDestroy(s);
        (void) hipGetLastError();
        rcs[r] = rc;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n; ++r) th.emplace_back(rank, r);
    rank(0);
    for (auto &t : th) t.join();
    for (int r = 0; r < n; ++r) if (comm[r]) (void) (failed.load() ? ncclCommAbort(comm[r]) : ncclCommDestroy(comm[r]));
    if (g_main.init) (void) hipSetDevice(g_main.device);
    int first = 0;
    for (int r = 0; r < n; ++r) if (rcs[r] && !first) first = rcs[r];
    if (first) for (int r = 0; r < n; ++r) if (rcs[r]) fprintf(stderr, "[hdsdp_mi355x] RCCL self-test: rank %d (device %d) failed at stage %d\n", r, dev[r], rcs[r]);
    return first;
}
