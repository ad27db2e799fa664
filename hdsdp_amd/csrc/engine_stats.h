// engine_stats.h -- where a caller's wall time goes below the C ABI (HMiGetCallStats, HDSDP_MI355X_CALL_STATS / _TRACE) and the error macros
// Implementation header of engine.hip: included exactly once, there, in this order (the pieces share the anonymous namespace
// and the engine's thread-local context `g`); split out of a 3 300-line file in round 3, nothing else changed.
// ---- where a caller's wall time goes inside the library (HMiGetCallStats; HDSDP_MI355X_CALL_STATS=1 prints the table at
// exit): only the caller's thread counts, and only the outermost entry (HKKTBuildUp calls the cones' slots, the cones
// call HFpLinsys*), so the categories add up to the time the driver spent below the C ABI
enum { ST_BUILD_M = 0, ST_BUILD_CORR, ST_FACTORIZE, ST_SOLVE, ST_ASSEMBLE_FACTOR, ST_RATIO, ST_PRIMAL_UTIL, ST_LINSYS, ST_N };
const char *g_stat_name[ST_N] = {"HKKTBuildUp (M-forming types)", "HKKTBuildUp (corrector)", "HKKTFactorize", "HKKTSolve",
                                 "cone: S assembly + factor (update, interior checks, barrier, line search)",
                                 "cone: ratio test", "cone: primal recovery + utilities", "HFpLinsys* called by CPU cones"};
double g_stat_sec[ST_N];
long g_stat_calls[ST_N];
// the same time by entry point (the outermost entry's function name), printed under the categories
struct StatFn { const char *name; int k; double sec; long calls; double mx; };
StatFn g_stat_fn[64];
int g_stat_nfn = 0;
thread_local int t_stat_depth = 0;
static bool stat_trace() { static int t = -1; if (t < 0) { const char *e = getenv("HDSDP_MI355X_TRACE"); t = (e && atoi(e)) ? 1 : 0; } return t == 1; }
struct StatScope {
    int k;
    bool on = false;
    std::chrono::steady_clock::time_point t0;
    const char *name;
    StatScope(int k_, const char *name_) : k(k_), name(name_) {
        if (t_ctx) return;                       // worker threads of a device group run below an entry that is already timed
        on = (t_stat_depth++ == 0);
        if (on) t0 = std::chrono::steady_clock::now();
    }
    ~StatScope() {
        if (t_ctx) return;
        --t_stat_depth;
        if (on && stat_trace()) {                // HDSDP_MI355X_TRACE=1: drain the device after every entry and say which
            const hipError_t e = hipDeviceSynchronize();
            fprintf(stderr, "[hdsdp_mi355x trace] %s -> %s\n", name, e == hipSuccess ? "ok" : hipGetErrorName(e));
        }
        if (on) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            g_stat_sec[k] += dt;
            g_stat_calls[k] += 1;
            int f = 0;
            while (f < g_stat_nfn && g_stat_fn[f].name != name) ++f;
            if (f == g_stat_nfn && g_stat_nfn < 64) g_stat_fn[g_stat_nfn++] = StatFn{name, k, 0.0, 0, 0.0};
            if (f < 64) { g_stat_fn[f].sec += dt; g_stat_fn[f].calls += 1; if (dt > g_stat_fn[f].mx) g_stat_fn[f].mx = dt; }
        }
    }
};
// cone creation (ingest), outside the categories above: seconds in the presolve of the caller's columns (class, rank-one probe,
// copy), in the upload + device layout of the rows, in the sweep copy; bytes of entries (12 per entry) that came in
double g_create_sec[3] = {0.0, 0.0, 0.0};
double g_create_bytes = 0.0;
struct CreateTimer {
    int k; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit CreateTimer(int k_) : k(k_) {}
    ~CreateTimer() { if (!t_ctx) g_create_sec[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
// how the requests for a dual / step matrix of single-device blocks were answered (engine_cone.h: cone_assemble): [0] the buffer
// already held the point, [1] a copy of S, [2] S + alpha dS on the last ratio test's line, [3] a sweep for a dual matrix because
// the point was not on that line (or no line was known), [4] a sweep because 16 affine updates in a row are refreshed, [5] a
// sweep for a step matrix (every ratio test), [6] sweeps of blocks that do not track points (small, sharded)
long g_asm_counts[7] = {0, 0, 0, 0, 0, 0, 0};
void stats_print_at_exit() {
    if (g_asm_counts[0] + g_asm_counts[1] + g_asm_counts[2] + g_asm_counts[3] + g_asm_counts[4] + g_asm_counts[5] + g_asm_counts[6] > 0)
        fprintf(stderr, "[hdsdp_mi355x] dual / step matrices asked for: %ld already in the buffer, %ld copies, %ld on the last ratio test's line (S + alpha dS), "
                        "%ld dual-matrix sweeps off that line, %ld refresh sweeps, %ld step-matrix sweeps, %ld sweeps of untracked blocks\n",
                g_asm_counts[0], g_asm_counts[1], g_asm_counts[2], g_asm_counts[3], g_asm_counts[4], g_asm_counts[5], g_asm_counts[6]);
    if (g_create_bytes > 0.0) {
        const double tc = g_create_sec[0] + g_create_sec[1] + g_create_sec[2];
        fprintf(stderr, "[hdsdp_mi355x] cone creation from caller data: %.3f s for %.2f GB of entries (%.2f GB/s): presolve of the columns %.3f s, "
                        "upload + device layout %.3f s, sweep copy %.3f s\n", tc, g_create_bytes * 1e-9, g_create_bytes * 1e-9 / std::max(tc, 1e-9),
                g_create_sec[0], g_create_sec[1], g_create_sec[2]);
    }
    double tot = 0.0;
    for (int k = 0; k < ST_N; ++k) tot += g_stat_sec[k];
    fprintf(stderr, "[hdsdp_mi355x] wall time below the C ABI: %.3f s\n", tot);
    for (int k = 0; k < ST_N; ++k) {
        if (!g_stat_calls[k]) continue;
        fprintf(stderr, "[hdsdp_mi355x]   %-78s %8ld calls %10.3f s\n", g_stat_name[k], g_stat_calls[k], g_stat_sec[k]);
        for (int f = 0; f < g_stat_nfn; ++f)
            if (g_stat_fn[f].k == k)
                fprintf(stderr, "[hdsdp_mi355x]       %-74s %8ld calls %10.3f s   (longest call %.1f ms)\n", g_stat_fn[f].name, g_stat_fn[f].calls,
                        g_stat_fn[f].sec, 1e3 * g_stat_fn[f].mx);
    }
}

#define HIP_RC(expr)                                                                             \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            fprintf(stderr, "[hdsdp_mi355x] HIP error %s at %s:%d\n", hipGetErrorName(_e), __FILE__, __LINE__); \
            return HDSDP_RETCODE_FAILED;                                                         \
        }                                                                                        \
    } while (0)

#define RC(x)                                    \
    do {                                         \
        if ((x) != 0) return HDSDP_RETCODE_FAILED; \
    } while (0)
