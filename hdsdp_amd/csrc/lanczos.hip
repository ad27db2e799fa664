// lanczos.hip -- device-resident ratio test  max{ alpha : S + alpha dS >= 0 }  (SURVEY.md 8(f) row 2).
//
// Reference: sdpDenseConeRatioTestImpl (interface/hdsdp_conic_sdp.c:1640-1686) forms dS and hands the operator
//     x -> L^-1 (-dS) L^-T x      (sdpDenseConeILanczosMultiply, :462-505: dtrsv, dsymv, dtrsv)
// to HLanczosSolve (linalg/hdsdp_lanczos.c:161-292): <= 30 Lanczos steps from a fixed pseudo-random start, a
// Ritz check every third step, and the conservative step 1 / (gamma + lambda_max) with gamma built from two Ritz
// residuals.  Here the recurrence and the acceptance logic are the same, step for step; what changes is where the
// work runs:
//   * the operator is three dense matrix-vector products with the explicit lower-triangular Linv that the Schur
//     build needs anyway (HBM-bound: 2 x 16 + 32 MB at n = 2000) instead of two latency-bound triangular solves,
//   * small blocks (n16 <= 256): the whole test is ONE single-workgroup launch (hdm_lanczos_whole_kernel), Ritz problems included;
//   * large blocks: the steps between two Ritz checks are one launch of co-resident workgroups (hdm_lanczos_group_kernel) that
//     returns the (alpha_k, beta_k) pairs through mapped pinned memory; the tridiagonal (k+1) x (k+1) Ritz problem is solved on
//     the host by implicit QL (tridiag_eig; no LAPACK dependency), cyclic Jacobi being the way out if it does not converge;
//   * the launch-per-product forms behind HDM_LANCZOS_WHOLE / _FUSED / _BIG / _GROUP = 0 remain as tested fallbacks.
// The start vector reproduces glibc's srand()/rand() stream (TYPE_3 additive feedback generator) without touching
// the process-wide libc state, so the device run starts from the reference's own vector.
#include "lanczos.h"
#include "chol.h"
#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

// ---- glibc random_r TYPE_3 (r[i] = r[i-3] + r[i-31]), as srand(seed) / rand() use it -------------------------------
struct GlibcRand {
    int f, b;   // front / rear indices of the 31-word state
    int32_t st[31];
    void seed(unsigned int s) {
        if (s == 0) s = 1;
        st[0] = (int32_t) s;
        long word = (int32_t) s;
        for (int i = 1; i < 31; ++i) {
            long hi = word / 127773, lo = word % 127773;
            word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            st[i] = (int32_t) word;
        }
        f = 3; b = 0;
        for (int i = 0; i < 310; ++i) (void) next();
    }
    int next() {
        uint32_t v = (uint32_t) st[f] + (uint32_t) st[b];
        st[f] = (int32_t) v;
        int out = (int) (v >> 1);
        if (++f >= 31) f = 0;
        if (++b >= 31) b = 0;
        return out;
    }
};

// single workgroup: three-term recurrence + normalisation of one Lanczos step (hdsdp_lanczos.c:199-218)
//   w -= hprev * Vprev ;  alp = -<w, Vk> ;  w += alp * Vk ;  nrm = |w| ;  Vnext = vnext = w / nrm  (if nrm > 0)
// w arrives as the `nchunk` partial sums of the operator's last product (hdm_gemv_n_kernel), summed here in chunk order;
// hprev is read from DEVICE memory (the previous step's norm), so the steps between two Ritz checks are queued back to
// back and the host reads their (alpha, beta) pairs with one synchronisation per group instead of one per step.
__global__ __launch_bounds__(1024) void hdm_lanczos_step_kernel(const double *__restrict__ part, int nchunk, double *__restrict__ w,
                                                                const double *__restrict__ Vprev, const double *__restrict__ hprev_dev,
                                                                const double *__restrict__ Vk, double *__restrict__ Vnext,
                                                                double *__restrict__ vnext, int n, double *__restrict__ out) {
    __shared__ double red[16];
    __shared__ double bc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto reduce = [&](double s) {
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += red[q]; bc = t; }
        __syncthreads();
        double r = bc;
        __syncthreads();
        return r;
    };
    const double hprev = Vprev ? *hprev_dev : 0.0;
    double s = 0.0;
    for (int i = tid; i < n; i += 1024) {
        double x = 0.0;
        for (int c = 0; c < nchunk; ++c) x += part[(long) c * n + i];
        if (Vprev) x -= hprev * Vprev[i];
        w[i] = x;
        s += x * Vk[i];
    }
    const double alp = -reduce(s);
    s = 0.0;
    for (int i = tid; i < n; i += 1024) {
        double x = w[i] + alp * Vk[i];
        w[i] = x;
        s += x * x;
    }
    const double nrm = sqrt(reduce(s));
    if (nrm > 0.0) {
        const double inv = 1.0 / nrm;
        for (int i = tid; i < n; i += 1024) { double x = w[i] * inv; Vnext[i] = x; vnext[i] = x; }
    }
    if (tid == 0) { out[0] = alp; out[1] = nrm; }
}

// Small blocks (n16 <= 256): up to three whole Lanczos steps -- operator application and recurrence -- in ONE single-workgroup
// launch.  The reference checks its Ritz values every third step (checkFreq, hdsdp_lanczos.c:187-189) and the host needs
// nothing but (alpha_k, beta_k) in between, so a group of steps is one launch and one synchronisation instead of five
// launches and a synchronisation per step: a ratio test on a 100 x 100 block went from 0.74 to 0.3 ms, and it is more than
// half of what the reference's driver spends below the C ABI on mcp100 / gpp100.  The matrices (<= 512 KB each) are read
// from L2; vectors live in LDS.  Sums are taken in a fixed order.  out: (alpha, beta) per step, then the number of steps done
// (a zero norm ends the group early, as it ends the reference's loop).
#define LZ_FUSED_MAX 256
#define LZ_NCHUNK 32       // column chunks of the operator's last product (partial sums, reduced in chunk order)
__global__ __launch_bounds__(1024) void hdm_lanczos_fused_kernel(const double *__restrict__ Linv, long ldl,
                                                                 const double *__restrict__ dS, long ldd, int n,
                                                                 double *__restrict__ V, long ldv, int k0, int nsteps, double hprev,
                                                                 double *__restrict__ blk, double *__restrict__ out) {
    __shared__ double sv[LZ_FUSED_MAX], st1[LZ_FUSED_MAX], st2[LZ_FUSED_MAX], part[4][LZ_FUSED_MAX], red[16];
    __shared__ double bc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto reduce = [&](double s) {
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += red[q]; bc = t; }
        __syncthreads();
        const double r = bc;
        __syncthreads();
        return r;
    };
    int done = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int k = k0 + s;
        if (tid < n) sv[tid] = V[tid + (long) k * ldv];
        __syncthreads();
        // t1 = Linv^T v: one wavefront per column (contiguous reads), rows above the diagonal skipped
        for (int j = wave; j < n; j += 16) {
            const double *col = Linv + (long) j * ldl;
            double a = 0.0;
            for (int i = (j & ~63) + lane; i < n; i += 64) a += ((i < j) ? 0.0 : col[i]) * sv[i];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) st1[j] = a;
        }
        __syncthreads();
        // t2 = -dS t1 (dS symmetric: column dots again)
        for (int j = wave; j < n; j += 16) {
            const double *col = dS + (long) j * ldd;
            double a = 0.0;
            for (int i = lane; i < n; i += 64) a += col[i] * st1[i];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) st2[j] = -a;
        }
        __syncthreads();
        // w = Linv t2: rows across lanes (coalesced), four column chunks summed in order
        {
            const int i = tid & (LZ_FUSED_MAX - 1), c = tid >> 8;
            const int cw = (n + 3) / 4, j0 = c * cw, j1 = min(n, j0 + cw), jend = min(j1, i + 1);
            double a0 = 0.0, a1 = 0.0;
            if (i < n) {
                int j = j0;
                for (; j + 1 < jend; j += 2) {
                    a0 += Linv[i + (long) j * ldl] * st2[j];
                    a1 += Linv[i + (long) (j + 1) * ldl] * st2[j + 1];
                }
                if (j < jend) a0 += Linv[i + (long) j * ldl] * st2[j];
            }
            part[c][i] = a0 + a1;
        }
        __syncthreads();
        // the three-term recurrence and the normalisation (hdsdp_lanczos.c:199-218): thread i keeps element i
        double x = 0.0, vk = 0.0;
        if (tid < n) {
            x = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
            vk = sv[tid];
            if (k > 0) x -= hprev * V[tid + (long) (k - 1) * ldv];
        }
        const double alp = -reduce((tid < n) ? x * vk : 0.0);
        x += alp * vk;
        const double nrm = sqrt(reduce((tid < n) ? x * x : 0.0));
        if (tid == 0) { out[2 * s] = alp; out[2 * s + 1] = nrm; }
        done = s + 1;
        if (!(nrm > 0.0)) break;                 // (uniform)
        if (tid < n) {
            const double xn = x * (1.0 / nrm);
            V[tid + (long) (k + 1) * ldv] = xn;
            blk[tid] = xn;
        }
        hprev = nrm;
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) out[2 * nsteps] = (double) done;
}

// ---------------------------------------------------------------------------------------------------------------------
// Large blocks (256 < n16 <= 4096): the Lanczos steps between two Ritz checks in ONE launch of LZG_WG co-resident workgroups.
//
// At n = 2000 a step was four dependent launches (three matrix-vector products, the recurrence) of 5-10 us of work each behind
// 10 us of launch gap each: 0.11 ms per step, and 186 of the 243 ratio tests of a headline solve run all 30 steps.  Here a
// step is three grid-wide barriers apart instead (five up to round 4): workgroup b owns the columns b, b + G, ... of each product -- all three are
// column dots (t1 = Linv^T v down the columns of Linv, t2 = -dS t1 down the columns of the symmetric dS, w = Linv t2 down the
// columns of a transposed copy of Linv made once per test) with the operand vector in LDS and 32-64 loads in flight per
// thread -- and of the recurrence; the two inner products of the recurrence are per-workgroup partial sums that every
// workgroup adds up in the same fixed order.  The barrier is a monotone counter (release / acquire at agent scope, bounded
// spin); the host launches this form only where the grid is co-resident by construction (one workgroup per CU, nothing else
// on the device) and falls back to the launch-per-product form for the rest of the object's life if a wait ever runs out.
// ---------------------------------------------------------------------------------------------------------------------
#define LZG_WG 256
struct LzgArgs {
    const double *Linv; long ldl; const double *LinvT; long ldt; const double *dS; long ldd;
    int n; double *V; long ldv; int k0, nsteps; double hprev;
    double *blk, *t1, *t2, *xw, *out; unsigned *sync;        // xw: x = w - beta v_{k-1}, whole (n); sync: 64 + LZG_WG words, see lzg_barrier
    unsigned epoch0;           // barrier epochs of this launch: epoch0 + 1, + 2, ... (the words are never reset)
    double *dbg;               // diagnostic (HDSDP_MI355X_RATIO_DEBUG=2): workgroup 0 adds up the 100 MHz ticks of its 9 phases here
};

// What workgroups hand to each other inside the launch -- the product vectors, the partial sums, the new basis vector: a few
// KB per step -- is written and read with agent-scope atomic accesses (they go past the caches that are not coherent between
// XCDs), so the barrier needs no cache maintenance: an agent-scope fence writes back / invalidates the XCD's whole L2, and a
// thousand waves doing that five times per Lanczos step made the first version of this kernel slower than the launches it
// replaced (0.17 against 0.12 ms per step).  The matrices are only read and stay cached.
__device__ __forceinline__ double lzg_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lzg_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// barrier number `epoch` (1, 2, ...) of a launch of G <= 256 workgroups, without read-modify-write atomics (an atomic addition
// that every XCD must see is performed at the memory side, one after the other per address): every workgroup stores the epoch
// into its OWN word once its own stores have completed, wave 0 of workgroup 0 polls the G words (four per lane) and then stores
// the epoch into the go word, which everybody else polls.
// sync[0]: go word; sync[1]: a wait ran out; sync[64 + b]: workgroup b's word.
__device__ __forceinline__ bool lzg_barrier(unsigned *sync, unsigned epoch, int b, int G, int *s_ok) {
    // Every wave waits for its OWN hand-over stores before the workgroup barrier, so that the epoch word below cannot
    // overtake them: the stores are sc1 (coherent by themselves, no write-back needed), but two stores of one wave to
    // different addresses travel to different L2 channels and may complete out of order, and a workgroup-scope release
    // fence emits no vector-memory wait outside tgsplit mode.  (Checked in the ISA: s_waitcnt vmcnt(0) precedes s_barrier.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x < 64) {                       // wave 0 (the other waves wait at the __syncthreads below)
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(sync + 64 + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        if (b == 0) {
            for (int it = 0;; ++it) {
                bool all = true;
                for (int q = lane; q < G; q += 64)
                    all = all && (__hip_atomic_load(sync + 64 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
                if (__all(all)) break;
                if (it > (1 << 21)) { ok = 0; break; }            // (wave-uniform)
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) __hip_atomic_store(ok ? sync : sync + 1, ok ? epoch : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == 0) {
            for (int it = 0; __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; ++it) {
                if (it > (1 << 22) || __hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) *s_ok = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return *s_ok != 0;
}

// res[c] = column (b + c G) of A . xs   for this workgroup's columns; MODE 0: all rows, 1: rows i >= j, 2: rows i <= j.
// Four columns x TRIPS rows per thread are loaded before anything is added; sums in a fixed order.
template <int TRIPS, int MODE>
__device__ __forceinline__ void lzg_col_dots(const double *__restrict__ A, long ld, int n, const double *xs, int b, int G,
                                             double *res, double (*red)[4]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c0 = 0; b + c0 * G < n; c0 += 4) {
        double v[4][TRIPS];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int j = b + (c0 + cc) * G;
#pragma unroll
            for (int it = 0; it < TRIPS; ++it) {
                const int i = tid + 256 * it;
                const bool ok = j < n && i < n && (MODE == 0 || (MODE == 1 ? i >= j : i <= j));
                v[cc][it] = ok ? A[i + (long) j * ld] : 0.0;
            }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            double a = 0.0;
#pragma unroll
            for (int it = 0; it < TRIPS; ++it) a += v[cc][it] * xs[tid + 256 * it];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) red[cc][wave] = a;
        }
        __syncthreads();
        if (tid < 4) res[c0 + tid] = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
        __syncthreads();
    }
}

template <int TRIPS>
__global__ __launch_bounds__(256) void hdm_lanczos_group_kernel(LzgArgs a) {
    __shared__ double xs[TRIPS * 256];
    __shared__ double res[16], red[4][4], pown[16];
    __shared__ double bc;
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, G = gridDim.x, n = a.n;
    unsigned nbar = 0;
    auto load_vec = [&](const double *x) {
#pragma unroll
        for (int it = 0; it < TRIPS; ++it) { const int i = tid + 256 * it; xs[i] = (i < n) ? lzg_ld(x + i) : 0.0; }
        __syncthreads();
    };
    // sum of the G per-workgroup partial sums, the same order in every workgroup
    auto total = [&](const double *p) {       // p: the G partial sums in LDS
        double t = 0.0;
        if (wave == 0) {
            for (int q = lane; q < G; q += 64) t += p[q];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0) bc = t;
        }
        __syncthreads();
        const double r = bc;
        __syncthreads();
        return r;
    };
    double hprev = a.hprev;
    int done = 0;
    unsigned long long tk = (a.dbg && b == 0 && tid == 0) ? __builtin_amdgcn_s_memrealtime() : 0ULL;
    auto stamp = [&](int slot) {
        if (a.dbg && b == 0 && tid == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            a.dbg[slot] += (double) (now - tk);
            tk = now;
        }
    };
    // THREE grid-wide barriers per step (five up to round 4).  After the third product every workgroup publishes its owned
    // elements of x = w - beta_{k-1} v_{k-1}; behind ONE barrier everybody then has the whole x and the whole v_k within reach
    // (a few KB, L2-resident) and finishes the step by itself: the per-workgroup partial sums of alpha = <x, v_k> and of
    // |x - alpha v_k|^2 -- the very sums the owners used to publish, each over its elements in the same order, added up in the
    // same fixed order (the five-barrier form's numbers up to how the compiler contracts the multiply-adds; every workgroup
    // gets the same ones) -- and the new basis vector, which stays in LDS as the next step's operand.  The owners still store their elements of v_{k+1} for the Ritz steps and the next launch.
    __shared__ double psum[LZG_WG];
    for (int s = 0; s < a.nsteps; ++s) {
        const int k = a.k0 + s;
        const double *vk = a.V + (long) k * a.ldv;
        if (tid < 16) {                          // the owned elements of v_{k-1}, for the recurrence (stored three barriers ago or more)
            const int j = b + tid * G;
            pown[tid] = (j < n && k > 0) ? lzg_ld(a.V + j + (long) (k - 1) * a.ldv) : 0.0;
        }
        if (s == 0) load_vec(vk);                // (later steps: xs holds v_k, made below)
        lzg_col_dots<TRIPS, 1>(a.Linv, a.ldl, n, xs, b, G, res, red);                 // t1 = Linv^T v
        if (tid < 16 && b + tid * G < n) lzg_st(a.t1 + b + tid * G, res[tid]);
        stamp(0);
        if (!lzg_barrier(a.sync, a.epoch0 + (++nbar), b, G, &s_ok)) { if (b == 0 && tid == 0) a.out[2 * a.nsteps + 1] = 1.0; return; }
        stamp(1);
        load_vec(a.t1);
        lzg_col_dots<TRIPS, 0>(a.dS, a.ldd, n, xs, b, G, res, red);                   // t2 = -dS t1
        if (tid < 16 && b + tid * G < n) lzg_st(a.t2 + b + tid * G, -res[tid]);
        stamp(2);
        if (!lzg_barrier(a.sync, a.epoch0 + (++nbar), b, G, &s_ok)) { if (b == 0 && tid == 0) a.out[2 * a.nsteps + 1] = 1.0; return; }
        stamp(3);
        load_vec(a.t2);
        lzg_col_dots<TRIPS, 2>(a.LinvT, a.ldt, n, xs, b, G, res, red);                // w = Linv t2
        // the three-term recurrence on the owned elements (hdsdp_lanczos.c:199-218): x = w - beta_{k-1} v_{k-1}, for everybody
        if (tid < 16 && b + tid * G < n) {
            double x = res[tid];
            if (k > 0) x -= hprev * pown[tid];
            lzg_st(a.xw + b + tid * G, x);
        }
        stamp(4);
        if (!lzg_barrier(a.sync, a.epoch0 + (++nbar), b, G, &s_ok)) { if (b == 0 && tid == 0) a.out[2 * a.nsteps + 1] = 1.0; return; }
        stamp(5);
        // thread q < G redoes workgroup q's partial sum of alpha over that workgroup's elements q, q + G, ... in order.  With a
        // full grid (G = 256 = the threads of a workgroup) those are the thread's own elements of the LDS vector too: one pass of
        // loads serves both sums and the new vector.
        double xr[TRIPS], vr[TRIPS];
        const bool own_deal = (G == 256);
        if (own_deal) {
#pragma unroll
            for (int it = 0; it < TRIPS; ++it) {
                const int i = tid + 256 * it;
                xr[it] = (i < n) ? lzg_ld(a.xw + i) : 0.0;
                vr[it] = (i < n) ? lzg_ld(vk + i) : 0.0;
            }
            double p = 0.0;
#pragma unroll
            for (int it = 0; it < TRIPS; ++it) if (tid + 256 * it < n) p += xr[it] * vr[it];
            psum[tid] = p;
        } else if (tid < G) {
            double p = 0.0;
            for (int j = tid; j < n; j += G) p += lzg_ld(a.xw + j) * lzg_ld(vk + j);
            psum[tid] = p;
        }
        __syncthreads();
        const double alp = -total(psum);
        if (own_deal) {
            double p = 0.0;
#pragma unroll
            for (int it = 0; it < TRIPS; ++it) {
                xr[it] = xr[it] + alp * vr[it];
                if (tid + 256 * it < n) p += xr[it] * xr[it];
            }
            psum[tid] = p;
        } else if (tid < G) {
            double p = 0.0;
            for (int j = tid; j < n; j += G) {
                const double x = lzg_ld(a.xw + j) + alp * lzg_ld(vk + j);
                p += x * x;
            }
            psum[tid] = p;
        }
        __syncthreads();
        const double nrm = sqrt(total(psum));
        if (b == 0 && tid == 0) { a.out[2 * s] = alp; a.out[2 * s + 1] = nrm; }
        done = s + 1;
        if (!(nrm > 0.0)) break;                 // (the same number in every workgroup)
        const double rn = 1.0 / nrm;
#pragma unroll
        for (int it = 0; it < TRIPS; ++it) {     // v_{k+1}, whole, into LDS: the next step's operand
            const int i = tid + 256 * it;
            if (own_deal) xs[i] = (i < n) ? xr[it] * rn : 0.0;
            else xs[i] = (i < n) ? (lzg_ld(a.xw + i) + alp * lzg_ld(vk + i)) * rn : 0.0;
        }
        __syncthreads();
        if (tid < 16 && b + tid * G < n) {       // the owners' elements of it, for the Ritz steps and the next launch
            const int j = b + tid * G;
            lzg_st(a.V + j + (long) (k + 1) * a.ldv, xs[j]);
            a.blk[j] = xs[j];
        }
        hprev = nrm;
        stamp(6);
    }
    if (b == 0 && tid == 0) a.out[2 * a.nsteps] = (double) done;
}

// B = A^T for an n x n column-major matrix (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void hdm_transpose_kernel(const double *__restrict__ A, long lda, double *__restrict__ B, long ldb, int n) {
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + tx, j = j0 + r;
        t[r][tx] = (i < n && j < n) ? A[i + (long) j * lda] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = j0 + tx, j = i0 + r;      // B(i, j) = A(j, i)
        if (i < n && j < n) B[i + (long) j * ldb] = t[tx][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small blocks (n16 <= 256): the WHOLE ratio test in ONE single-workgroup launch and one synchronisation.
//
// HLanczosSolve (hdsdp_lanczos.c:161-292) step for step -- start vector (fresh, or warm start + 1e-3 x the pseudo-random
// vector), up to 30 Lanczos steps, a Ritz check every third step, the two residual applications, the acceptance rule --
// with the small symmetric eigenproblem solved in the kernel too: the (k+1) x (k+1) matrix is TRIDIAGONAL (alpha on the
// diagonal, the norms beside it), so wave 0 runs the implicit QL iteration with the rotations' scalar recurrence computed
// redundantly by every lane and lane r keeping row r of the eigenvector matrix (the reference calls dsyevr on the same
// matrix stored densely; the multi-launch path uses a host QL iteration: eigenvalues agree to rounding).  The multi-launch form
// (a group of three steps per launch, Ritz checks on the host) costs 6-12 launches and as many synchronisations per ratio
// test: 0.53 ms on a 100 x 100 block, half of what the reference's driver spends below the C ABI on mcp100 / gpp100 and
// 40 % on truss1 (2247 ratio tests).
// out: [0] step, [1] Lanczos steps done, [2] status (0 ok, 1 failed: zero norm without convergence).
// RESIDENT = true (n16 <= 128): the two matrices are first packed into LDS (lower triangles, column by column: 2 x 66 KB at
// n16 = 128) and the 30-50 operator applications of a test read them there; from global memory every application is three
// dependent passes of L2 round trips -- 12 us each at n = 100, 0.42 ms per ratio test on mcp100.
#define LZ_MD 30
#define LZ_RESIDENT_MAX 128
template <bool RESIDENT>
__global__ __launch_bounds__(1024) void hdm_lanczos_whole_kernel(const double *__restrict__ Linv, long ldl, const double *__restrict__ dS,
                                                                 long ldd, int n, double *__restrict__ V, long ldv,
                                                                 const double *__restrict__ start, double *__restrict__ warm, int fresh,
                                                                 double *__restrict__ out) {
    __shared__ double sv[LZ_FUSED_MAX], st1[LZ_FUSED_MAX], st2[LZ_FUSED_MAX], part[4][LZ_FUSED_MAX], red[16];
    __shared__ double hd[LZ_MD + 2], he[LZ_MD + 2], td[LZ_MD + 2], te[LZ_MD + 2], Z[32 * 32], y1[32], y2[32];
    __shared__ double bc, sh_eig1, sh_eig2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    extern __shared__ __attribute__((aligned(16))) double lz_dyn[];
    double *Lp = lz_dyn, *Dp = lz_dyn + (RESIDENT ? (long) n * (n + 1) / 2 : 0);   // packed lower triangles: (i, j) at j n - j (j - 1) / 2 + i - j
    if (RESIDENT) {
        for (int j = wave; j < n; j += 16) {
            const long cj = (long) j * n - (long) j * (j - 1) / 2;
            for (int i = j + lane; i < n; i += 64) { Lp[cj + i - j] = Linv[i + (long) j * ldl]; Dp[cj + i - j] = dS[i + (long) j * ldd]; }
        }
        __syncthreads();
    }
    auto reduce = [&](double s) {
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += red[q]; bc = t; }
        __syncthreads();
        const double r = bc;
        __syncthreads();
        return r;
    };
    // w = Linv ( -dS ( Linv^T sv ) ): element tid of the result (0 beyond n); sv must be in place and visible
    auto apply = [&]() -> double {
        if (RESIDENT) {
            // n <= 128: thread (r8, c8) = row or column r8, eighth c8 of the summation range; no shuffles anywhere (a wave per
            // column with a shuffle tree per column made a product 3 us of serialised cross-lane latency, 8 us per Lanczos step)
            double *part8 = &part[0][0];                     // 8 x 128
            const int r8 = tid & 127, c8 = tid >> 7, cw8 = (n + 7) / 8, q0 = c8 * cw8, q1 = min(n, q0 + cw8);
            const int cr = r8 * n - r8 * (r8 - 1) / 2;       // start of column r8 in the packed triangles
            {                                                // t1 = Linv^T v: column r8, rows of the chunk below the diagonal
                double a0 = 0.0, a1 = 0.0;
                if (r8 < n) {
                    const double *col = Lp + cr - r8;
                    int i = max(r8, q0);
                    for (; i + 1 < q1; i += 2) { a0 += col[i] * sv[i]; a1 += col[i + 1] * sv[i + 1]; }
                    if (i < q1) a0 += col[i] * sv[i];
                }
                part8[c8 * 128 + r8] = a0 + a1;
            }
            __syncthreads();
            if (tid < n) {
                double t = 0.0;
                for (int c = 0; c < 8; ++c) t += part8[c * 128 + tid];
                st1[tid] = t;
            }
            __syncthreads();
            {                                                // t2 = -dS t1: row r8 of the symmetric matrix (row part left of the diagonal, column part below)
                double a0 = 0.0;
                if (r8 < n) {
                    const double *col = Dp + cr - r8;
                    int ci = q0 * n - q0 * (q0 - 1) / 2;       // start of column i, advanced with i
                    for (int i = q0; i < q1; ++i) {
                        a0 += ((i >= r8) ? col[i] : Dp[ci + (r8 - i)]) * st1[i];
                        ci += n - i;
                    }
                }
                part8[c8 * 128 + r8] = a0;
            }
            __syncthreads();
            if (tid < n) {
                double t = 0.0;
                for (int c = 0; c < 8; ++c) t += part8[c * 128 + tid];
                st2[tid] = -t;
            }
            __syncthreads();
            {                                                // w = Linv t2: row r8, columns of the chunk up to the diagonal
                double a0 = 0.0;
                if (r8 < n) {
                    const int j1 = min(q1, r8 + 1);
                    int cj = q0 * n - q0 * (q0 - 1) / 2;
                    for (int j = q0; j < j1; ++j) { a0 += Lp[cj + (r8 - j)] * st2[j]; cj += n - j; }
                }
                part8[c8 * 128 + r8] = a0;
            }
            __syncthreads();
            double xr = 0.0;
            if (tid < n) for (int c = 0; c < 8; ++c) xr += part8[c * 128 + tid];
            __syncthreads();
            return xr;
        }
        for (int j = wave; j < n; j += 16) {
            const double *col = Linv + (long) j * ldl;
            double a = 0.0;
            for (int i = (j & ~63) + lane; i < n; i += 64) a += ((i < j) ? 0.0 : col[i]) * sv[i];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) st1[j] = a;
        }
        __syncthreads();
        for (int j = wave; j < n; j += 16) {
            const double *col = dS + (long) j * ldd;
            double a = 0.0;
            for (int i = lane; i < n; i += 64) a += col[i] * st1[i];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) st2[j] = -a;
        }
        __syncthreads();
        {
            const int i = tid & (LZ_FUSED_MAX - 1), c = tid >> 8;
            const int cw = (n + 3) / 4, j0 = c * cw, j1 = min(n, j0 + cw), jend = min(j1, i + 1);
            double a0 = 0.0, a1 = 0.0;
            if (i < n) {
                int j = j0;
                for (; j + 1 < jend; j += 2) {
                    a0 += Linv[i + (long) j * ldl] * st2[j];
                    a1 += Linv[i + (long) (j + 1) * ldl] * st2[j + 1];
                }
                if (j < jend) a0 += Linv[i + (long) j * ldl] * st2[j];
            }
            part[c][i] = a0 + a1;
        }
        __syncthreads();
        double x = 0.0;
        if (tid < n) x = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
        __syncthreads();
        return x;
    };
    // ---- start vector (:166-181), normalised into V[:, 0]
    double v0 = 0.0;
    if (tid < n) v0 = fresh ? start[tid] : warm[tid] + 1e-03 * start[tid];
    {
        const double nr = sqrt(reduce(v0 * v0));
        const double inv = nr > 0.0 ? 1.0 / nr : 0.0;
        if (tid < n) V[tid] = v0 * inv;
    }
    if (tid < LZ_MD + 2) { hd[tid] = 0.0; he[tid] = 0.0; }
    __threadfence_block();
    __syncthreads();
    const int md = LZ_MD, checkFreq = 3;
    double step = 0.0, hprev = 0.0;
    int k = 0, status = 0;
    for (k = 0; k < md; ++k) {
        if (tid < n) sv[tid] = V[tid + (long) k * ldv];
        __syncthreads();
        double x = apply();
        double vk = 0.0;
        if (tid < n) {
            vk = sv[tid];
            if (k > 0) x -= hprev * V[tid + (long) (k - 1) * ldv];
        }
        const double alp = -reduce((tid < n) ? x * vk : 0.0);
        x += alp * vk;
        const double nrm = sqrt(reduce((tid < n) ? x * x : 0.0));
        if (nrm > 0.0 && tid < n) V[tid + (long) (k + 1) * ldv] = x * (1.0 / nrm);
        if (tid == 0) { hd[k] = -alp; he[k] = (nrm > 0.0) ? nrm : 0.0; }
        hprev = (nrm > 0.0) ? nrm : 0.0;
        __threadfence_block();
        __syncthreads();
        if (!((k + 1) % checkFreq == 0 || k > md - 1 || nrm == 0.0)) continue;
        // ---- Ritz values of the leading kp x kp tridiagonal matrix: implicit QL with eigenvectors, wave 0
        const int kp = k + 1;
        if (wave == 0) {
            // The tridiagonal matrix lives in REGISTERS, entry i in lane i; the scalar recurrence of a rotation fetches what it
            // needs with v_readlane (a few cycles) instead of an LDS round trip per operand, and takes one reciprocal square
            // root per rotation instead of a square root and a division: the chain of ~1100 dependent rotations of a 30 x 30
            // problem was 0.3 ms of LDS latency and long-latency arithmetic, and the checks of a 30-step test 1 ms altogether.
            if (lane < 32)
                for (int c = 0; c < kp; ++c) Z[lane + 32 * c] = (lane == c) ? 1.0 : 0.0;
            double dreg = (lane < kp) ? hd[lane] : 0.0, ereg = (lane < kp - 1) ? he[lane] : 0.0;
            auto rl = [](double x, int i) {
                int lo = __double2loint(x), hi = __double2hiint(x);
                lo = __builtin_amdgcn_readlane(lo, i); hi = __builtin_amdgcn_readlane(hi, i);
                return __hiloint2double(hi, lo);
            };
            __builtin_amdgcn_wave_barrier();
            for (int l = 0; l < kp; ++l) {
                for (int iter = 0; iter < 64; ++iter) {
                    int mm = l;
                    for (; mm < kp - 1; ++mm) {
                        const double dd = fabs(rl(dreg, mm)) + fabs(rl(dreg, mm + 1));
                        if (fabs(rl(ereg, mm)) <= 2.220446049250313e-16 * dd) break;
                    }
                    if (mm == l) break;
                    const double dl = rl(dreg, l), el = rl(ereg, l);
                    double g = (rl(dreg, l + 1) - dl) / (2.0 * el);
                    double r = sqrt(fma(g, g, 1.0));
                    g = rl(dreg, mm) - dl + el / (g + copysign(r, g));
                    double s = 1.0, c = 1.0, p = 0.0;
                    int i = mm - 1;
                    bool under = false;
                    for (; i >= l; --i) {
                        const double tei = rl(ereg, i);
                        double f = s * tei;
                        const double b = c * tei;
                        const double h2 = fma(f, f, g * g);
                        if (h2 == 0.0) {
                            if (lane == i + 1) { dreg -= p; ereg = 0.0; }
                            if (lane == mm) ereg = 0.0;
                            under = true;
                            break;
                        }
                        double rinv = __builtin_amdgcn_rsq(h2);
                        rinv = rinv * (1.5 - 0.5 * h2 * rinv * rinv);
                        rinv = rinv * (1.5 - 0.5 * h2 * rinv * rinv);
                        r = h2 * rinv;
                        if (lane == i + 1) ereg = r;
                        s = f * rinv; c = g * rinv;
                        g = rl(dreg, i + 1) - p;
                        r = (rl(dreg, i) - g) * s + 2.0 * c * b;
                        p = s * r;
                        if (lane == i + 1) dreg = g + p;
                        g = c * r - b;
                        if (lane < kp) {                       // row `lane` of the eigenvector matrix
                            f = Z[lane + 32 * (i + 1)];
                            Z[lane + 32 * (i + 1)] = s * Z[lane + 32 * i] + c * f;
                            Z[lane + 32 * i] = c * Z[lane + 32 * i] - s * f;
                        }
                    }
                    if (under) continue;
                    if (lane == l) { dreg -= p; ereg = g; }
                    if (lane == mm) ereg = 0.0;
                }
            }
            if (lane < kp) td[lane] = dreg;
            __builtin_amdgcn_wave_barrier();
            // the two largest eigenvalues and their vectors
            int i1 = 0;
            for (int c = 1; c < kp; ++c) if (td[c] > td[i1]) i1 = c;
            int i2 = (kp > 1) ? (i1 == 0 ? 1 : 0) : i1;
            for (int c = 0; c < kp; ++c) if (c != i1 && td[c] > td[i2]) i2 = c;
            // sign convention: the component of largest magnitude is positive -- what LAPACK's tridiagonal eigenvector routines
            // return (dstein scales that way, dstemr's twisted factorisation puts a positive 1 at the twist index) and what the
            // multi-launch path's host eigenvectors have always shown on the reference's goldens.  The sign of y1 matters: the
            // NEXT ratio test is warm-started from Op (V y1) + 1e-3 x the pseudo-random vector (:166-181).
            double sg1 = 1.0, sg2 = 1.0;
            {
                double b1 = 0.0, b2 = 0.0;
                for (int r = 0; r < kp; ++r) {
                    const double u = Z[r + 32 * i1], v = Z[r + 32 * i2];
                    if (fabs(u) > fabs(b1)) b1 = u;
                    if (fabs(v) > fabs(b2)) b2 = v;
                }
                if (b1 < 0.0) sg1 = -1.0;
                if (b2 < 0.0) sg2 = -1.0;
            }
            if (lane < kp) { y1[lane] = sg1 * Z[lane + 32 * i1]; y2[lane] = sg2 * Z[lane + 32 * i2]; }
            if (lane == 0) { sh_eig1 = td[i1]; sh_eig2 = td[i2]; }
        }
        __threadfence_block();
        __syncthreads();
        const double eig1 = sh_eig1, eig2 = sh_eig2;
        const double resiVal = fabs(he[k] * y1[k]);
        if (resiVal < 1e-04 || k >= md - 1) {
            // z1 = V y1 ; z2 = Op z1 ; warm start <- z2 ; resiVal1 = | z2 - eig1 z1 |
            double z = 0.0;
            if (tid < n) for (int c = 0; c < kp; ++c) z += V[tid + (long) c * ldv] * y1[c];
            if (tid < n) sv[tid] = z;
            __syncthreads();
            double w = apply();
            if (tid < n) warm[tid] = w;
            double d = (tid < n) ? w - eig1 * z : 0.0;
            const double r1 = sqrt(reduce(d * d));
            // z2' = V y2 ; resiVal2 = | Op z2' - eig1 z2' |   (the reference uses eig1 here too, :262-266)
            z = 0.0;
            if (tid < n) for (int c = 0; c < kp; ++c) z += V[tid + (long) c * ldv] * y2[c];
            if (tid < n) sv[tid] = z;
            __syncthreads();
            w = apply();
            d = (tid < n) ? w - eig1 * z : 0.0;
            const double r2 = sqrt(reduce(d * d));
            const double resiDiff = eig1 - eig2 - r2;
            double gam = (resiDiff > 0) ? resiDiff : 1e-16;
            const double sq = r1 * r1 / gam;
            gam = r1 < sq ? r1 : sq;
            if (gam < 1e-03 || gam + eig1 <= 0.5) {
                step = (gam + eig1 <= 0.0) ? INFINITY : 1.0 / (gam + eig1);
                break;
            } else {
                if (nrm == 0.0) { status = 1; break; }
                step = 1.0 / (gam + eig1);
            }
        }
    }
    if (tid == 0) { out[0] = step; out[1] = (double) k; out[2] = (double) status; }
}

// z = V[:, 0..kc) * coef   (single workgroup; V column stride ldv)
__global__ __launch_bounds__(1024) void hdm_lincomb_kernel(const double *__restrict__ V, long ldv, int kc,
                                                           const double *__restrict__ coef, double *__restrict__ z, int n) {
    __shared__ double cf[64];              // (the coefficients come from mapped host memory: one trip each)
    if (threadIdx.x < kc && threadIdx.x < 64) cf[threadIdx.x] = coef[threadIdx.x];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        double s = 0.0;
        for (int c = 0; c < kc; ++c) s += V[i + (long) c * ldv] * cf[c];
        z[i] = s;
    }
}

// out[0] = | a - lam * b |_2   (single workgroup)
__global__ __launch_bounds__(1024) void hdm_resnorm_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                           double lam, int n, double *__restrict__ out) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (int i = tid; i < n; i += 1024) { double x = a[i] - lam * b[i]; s += x * x; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += red[q]; out[0] = sqrt(t); }
}

// start vector of a test: the pseudo-random vector, or the previous test's Ritz image + 1e-3 x that vector (hdsdp_lanczos.c:166-181);
// (product and sum rounded separately, as the host loop this replaces did)
__global__ __launch_bounds__(1024) void hdm_warm_start_kernel(double *__restrict__ out, const double *__restrict__ warm,
                                                              const double *__restrict__ startd, int n, int n16, int fresh) {
    for (int i = threadIdx.x; i < n16; i += 1024)
        out[i] = (i >= n) ? 0.0 : (fresh ? startd[i] : __dadd_rn(warm[i], __dmul_rn(1e-03, startd[i])));
}

// v <- v / |v| into the block's column 0 and into V[:, 0]   (single workgroup)
__global__ __launch_bounds__(1024) void hdm_normalize_kernel(const double *__restrict__ v, double *__restrict__ V0,
                                                             double *__restrict__ blk, int n) {
    __shared__ double red[16];
    __shared__ double bc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (int i = tid; i < n; i += 1024) s += v[i] * v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += red[q]; bc = sqrt(t); }
    __syncthreads();
    const double inv = bc > 0.0 ? 1.0 / bc : 0.0;
    for (int i = tid; i < n; i += 1024) { double x = v[i] * inv; V0[i] = x; blk[i] = x; }
}

// y[j] = alpha * sum_i A[i + j*ld] x[i]  (transposed product: one wavefront per column, contiguous reads).
// lower != 0: A is lower triangular, rows i < j are skipped (half the traffic).
__global__ __launch_bounds__(256) void hdm_gemv_t_kernel(const double *__restrict__ A, long ld, int n, int lower, double alpha,
                                                         const double *__restrict__ x, double *__restrict__ y) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const double *col = A + (long) j * ld;
    double s0 = 0.0, s1 = 0.0;
    int i = (lower ? (j & ~63) : 0) + lane;
    for (; i + 64 < n; i += 128) {
        const double a0 = col[i], a1 = col[i + 64];
        s0 += ((lower && i < j) ? 0.0 : a0) * x[i];
        s1 += ((lower && i + 64 < j) ? 0.0 : a1) * x[i + 64];
    }
    if (i < n) s0 += ((lower && i < j) ? 0.0 : col[i]) * x[i];
    double s = s0 + s1;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[j] = alpha * s;
}

// part[c][i] = sum over the columns j of chunk c of A[i + j*ld] x[j]  (plain product, rows across lanes: coalesced);
// lower != 0: A is lower triangular, column j only reaches rows i >= j.  Reduced in chunk order by the kernel below.
__global__ __launch_bounds__(256) void hdm_gemv_n_kernel(const double *__restrict__ A, long ld, int n, int lower, int nchunk,
                                                         const double *__restrict__ x, double *__restrict__ part) {
    const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    const int cw = (n + nchunk - 1) / nchunk, j0 = c * cw, j1 = min(n, j0 + cw);
    if (i >= n) return;
    double s0 = 0.0, s1 = 0.0;
    const int jend = lower ? min(j1, i + 1) : j1;
    int j = j0;
    for (; j + 1 < jend; j += 2) {
        s0 += A[i + (long) j * ld] * x[j];
        s1 += A[i + (long) (j + 1) * ld] * x[j + 1];
    }
    if (j < jend) s0 += A[i + (long) j * ld] * x[j];
    part[(long) c * n + i] = s0 + s1;
}
__global__ void hdm_gemv_n_reduce_kernel(const double *__restrict__ part, int n, int nchunk, double alpha, double *__restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[(long) c * n + i];
    y[i] = alpha * s;
}

// upper triangle <- lower triangle of an n x n column-major matrix
__global__ void hdm_mirror_lower_kernel(double *A, long ld, int n) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long) n * n) return;
    int i = (int) (e % n), j = (int) (e / n);
    if (i > j) A[j + (long) i * ld] = A[i + (long) j * ld];
}

// cyclic Jacobi for a small dense symmetric matrix (column-major k x k); eigenvalues ascending in d, vectors in Y
void jacobi_eig(int k, std::vector<double> A, std::vector<double> &d, std::vector<double> &Y) {
    Y.assign((size_t) k * k, 0.0);
    for (int i = 0; i < k; ++i) Y[(size_t) i * k + i] = 1.0;
    auto a = [&](int i, int j) -> double & { return A[(size_t) j * k + i]; };
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < k; ++p)
            for (int q = p + 1; q < k; ++q) off += a(p, q) * a(p, q);
        if (off < 1e-300) break;
        for (int p = 0; p < k; ++p)
            for (int q = p + 1; q < k; ++q) {
                const double apq = a(p, q);
                if (fabs(apq) < 1e-300) continue;
                const double theta = (a(q, q) - a(p, p)) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int r = 0; r < k; ++r) {
                    const double arp = a(r, p), arq = a(r, q);
                    a(r, p) = c * arp - s * arq;
                    a(r, q) = s * arp + c * arq;
                }
                for (int r = 0; r < k; ++r) {
                    const double apr = a(p, r), aqr = a(q, r);
                    a(p, r) = c * apr - s * aqr;
                    a(q, r) = s * apr + c * aqr;
                }
                for (int r = 0; r < k; ++r) {
                    const double yrp = Y[(size_t) p * k + r], yrq = Y[(size_t) q * k + r];
                    Y[(size_t) p * k + r] = c * yrp - s * yrq;
                    Y[(size_t) q * k + r] = s * yrp + c * yrq;
                }
            }
    }
    d.resize(k);
    for (int i = 0; i < k; ++i) d[i] = a(i, i);
    // ascending selection sort of (value, vector)
    for (int i = 0; i < k; ++i) {
        int mn = i;
        for (int j = i + 1; j < k; ++j) if (d[j] < d[mn]) mn = j;
        if (mn != i) {
            std::swap(d[i], d[mn]);
            for (int r = 0; r < k; ++r) std::swap(Y[(size_t) i * k + r], Y[(size_t) mn * k + r]);
        }
    }
}

// The Ritz matrix of a Lanczos run is TRIDIAGONAL: implicit QL with Wilkinson shifts (the EISPACK tql2 recurrence) instead of the
// cyclic Jacobi above, which took 1.1 ms at k = 30 and 3.7 ms over the ten checks of a 30-step test on a host core -- three
// times the device time of those steps once they ran in one launch per group.  U: the symmetric k x k matrix (column-major; only
// its diagonal and first subdiagonal are read).  Eigenvalues ascending in d, vectors in the columns of Y.  Returns false if an
// eigenvalue does not converge in 60 sweeps (the caller then takes the Jacobi route).
bool tridiag_eig(int k, const std::vector<double> &U, std::vector<double> &d, std::vector<double> &Y) {
    std::vector<double> e(k, 0.0);
    d.resize(k);
    for (int i = 0; i < k; ++i) d[i] = U[(size_t) i * k + i];
    for (int i = 0; i + 1 < k; ++i) e[i] = U[(size_t) i * k + i + 1];       // T(i + 1, i)
    Y.assign((size_t) k * k, 0.0);
    for (int i = 0; i < k; ++i) Y[(size_t) i * k + i] = 1.0;
    for (int l = 0; l < k; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < k - 1; ++m) {
                const double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) + dd == dd) break;
            }
            if (m != l) {
                if (iter++ == 60) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + copysign(r, g));
                double sn = 1.0, cs = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = sn * e[i];
                    const double b = cs * e[i];
                    e[i + 1] = (r = hypot(f, g));
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    sn = f / r; cs = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * sn + 2.0 * cs * b;
                    d[i + 1] = g + (p = sn * r);
                    g = cs * r - b;
                    double *yi = &Y[(size_t) i * k], *yi1 = &Y[(size_t) (i + 1) * k];
                    for (int q = 0; q < k; ++q) {
                        f = yi1[q];
                        yi1[q] = sn * yi[q] + cs * f;
                        yi[q] = cs * yi[q] - sn * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    for (int i = 0; i < k; ++i) {                 // ascending selection sort of (value, vector)
        int mn = i;
        for (int j = i + 1; j < k; ++j) if (d[j] < d[mn]) mn = j;
        if (mn != i) {
            std::swap(d[i], d[mn]);
            for (int r = 0; r < k; ++r) std::swap(Y[(size_t) i * k + r], Y[(size_t) mn * k + r]);
        }
    }
    return true;
}

}  // namespace

void hdm_lanczos_start_vector(int n, double *p) {
    // HLanczosIPrepare (hdsdp_lanczos.c:33-42): srand(n); per entry srand(rand()); sqrt(sqrt(rand() % 1627)) * (rand() % 2 - 0.5)
    // (the reference re-seeds the one libc generator inside the loop, so a single generator object follows it)
    GlibcRand g;
    g.seed((unsigned int) n);
    for (int i = 0; i < n; ++i) {
        g.seed((unsigned int) g.next());
        const int a = g.next() % 1627;
        const int b = g.next() % 2;
        p[i] = sqrt(sqrt((double) a)) * ((double) b - 0.5);
    }
}

// A <- scale * ( sym(A) + diag_add * I )   (n x n, column-major): the two symmetrisation passes of the primal recovery
__global__ void hdm_sym_scale_kernel(double *A, long ld, int n, double diag_add, double scale) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long) n * n) return;
    int i = (int) (e % n), j = (int) (e / n);
    if (i < j) return;
    if (i == j) { A[i + (long) i * ld] = scale * (A[i + (long) i * ld] + diag_add); return; }
    const double v = scale * 0.5 * (A[i + (long) j * ld] + A[j + (long) i * ld]);
    A[i + (long) j * ld] = v;
    A[j + (long) i * ld] = v;
}

int hdm_sym_scale(double *A, long ld, int n, double diag_add, double scale, hipStream_t s) {
    long tot = (long) n * n;
    hipLaunchKernelGGL(hdm_sym_scale_kernel, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, s, A, ld, n, diag_add, scale);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// out = S + step * dS over `count` doubles (out may alias S)
__global__ void hdm_axpy_mat_kernel(double *out, const double *S, const double *dS, double step, long count) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) out[e] = S[e] + step * dS[e];
}

// out = S + step * dS + eye * I for an n x n matrix of leading dimension ld (out may alias S)
__global__ void hdm_axpy_mat_eye_kernel(double *out, const double *S, const double *dS, double step, double eye, long ld, int n) {
    long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ld * ld) return;
    const long i = e % ld, j = e / ld;
    double v = S[e] + step * dS[e];
    if (i == j && i < n) v += eye;
    out[e] = v;
}
int hdm_axpy_mat_eye(double *out, const double *S, const double *dS, double step, double eye, long ld, int n, hipStream_t s) {
    const long count = ld * ld;
    hipLaunchKernelGGL(hdm_axpy_mat_eye_kernel, dim3((unsigned) ((count + 255) / 256)), dim3(256), 0, s, out, S, dS, step, eye, ld, n);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int hdm_axpy_mat(double *out, const double *S, const double *dS, double step, long count, hipStream_t s) {
    hipLaunchKernelGGL(hdm_axpy_mat_kernel, dim3((unsigned) ((count + 255) / 256)), dim3(256), 0, s, out, S, dS, step, count);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int hdm_mirror_lower(double *A, long ld, int n, hipStream_t s) {
    long tot = (long) n * n;
    hipLaunchKernelGGL(hdm_mirror_lower_kernel, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, s, A, ld, n);
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int HdmLanczos::init(int n_) {
    n = n_;
    n16 = (n + 15) / 16 * 16;
    const size_t blk = sizeof(double) * (size_t) n16 * 8;
    HDM_HIP_CHECK(hipMalloc((void **) &V, sizeof(double) * (size_t) n16 * (maxdim + 1)));
    HDM_HIP_CHECK(hipMalloc((void **) &bv, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &b1, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &b2, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &bw, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &bz, blk));
    HDM_HIP_CHECK(hipMalloc((void **) &warm, sizeof(double) * (size_t) n16));
    HDM_HIP_CHECK(hipMalloc((void **) &tmp, sizeof(double) * (size_t) n16));
    // the scalars that travel between host and device -- (alpha, beta) pairs, Ritz coefficients, residual norms -- live in one
    // block of mapped pinned host memory: the kernels write their results straight into it and the host reads them after its
    // synchronisation (a copy into pageable memory per group of steps cost more than the group's kernels at n = 2000)
    HDM_HIP_CHECK(hipHostMalloc((void **) &scal_h, sizeof(double) * 128, hipHostMallocMapped));
    memset(scal_h, 0, sizeof(double) * 128);
    HDM_HIP_CHECK(hipHostGetDevicePointer((void **) &scal, scal_h, 0));
    HDM_HIP_CHECK(hipMalloc((void **) &part, sizeof(double) * 32 * (size_t) n16));
    for (double *b : {bv, b1, b2, bw, bz}) HDM_HIP_CHECK(hdm_memset_sync(b, 0, blk));
    HDM_HIP_CHECK(hdm_memset_sync(warm, 0, sizeof(double) * (size_t) n16));
    HDM_HIP_CHECK(hdm_memset_sync(tmp, 0, sizeof(double) * (size_t) n16));
    start.resize(n);
    hdm_lanczos_start_vector(n, start.data());
    {   // device copy of the start vector, zero padded (the single-launch form builds its own first vector)
        std::vector<double> sp(n16, 0.0);
        for (int i = 0; i < n; ++i) sp[i] = start[i];
        HDM_HIP_CHECK(hipMalloc((void **) &startd, sizeof(double) * (size_t) n16));
        HDM_HIP_CHECK(hipMemcpy(startd, sp.data(), sizeof(double) * (size_t) n16, hipMemcpyHostToDevice));
    }
    nComputed = 0;
    return 0;
}

void HdmLanczos::destroy() {
    for (double *b : {V, bv, b1, b2, bw, bz, warm, tmp, part, startd, LT})
        if (b) (void) hipFree(b);
    if (gsync) (void) hipFree(gsync);
    if (scal_h) (void) hipHostFree(scal_h);
    gsync = nullptr; scal_h = nullptr;
    V = bv = b1 = b2 = bw = bz = warm = tmp = scal = part = startd = LT = nullptr;
}

// out (column 0 of a vector block) = Linv * ( -dS * ( Linv^T * in ) ): three HBM-bound matrix-vector products with
// dedicated kernels (the first version went through the 128 x 128-tile GEMM with an 8-column block: 0.7 ms per
// application at n = 2000, launch- and tile-latency bound)
int HdmLanczos::apply(const double *Linv, long ldl, const double *dS, long ldd, const double *in, double *out, hipStream_t s) {
    const int nchunk = LZ_NCHUNK;
    // t1 = Linv^T v            (column dots of the lower-triangular Linv)
    hipLaunchKernelGGL(hdm_gemv_t_kernel, dim3((n16 + 3) / 4), dim3(256), 0, s, Linv, ldl, n16, 1, 1.0, in, b1);
    // t2 = -dS t1              (dS is symmetric: column dots again)
    hipLaunchKernelGGL(hdm_gemv_t_kernel, dim3((n16 + 3) / 4), dim3(256), 0, s, dS, ldd, n16, 0, -1.0, b1, b2);
    // w = Linv t2              (rows across lanes, 32 column chunks, deterministic two-level sum)
    hipLaunchKernelGGL(hdm_gemv_n_kernel, dim3((n16 + 255) / 256, nchunk), dim3(256), 0, s, Linv, ldl, n16, 1, nchunk, b2, part);
    if (out) hipLaunchKernelGGL(hdm_gemv_n_reduce_kernel, dim3((n16 + 255) / 256), dim3(256), 0, s, part, n16, nchunk, 1.0, out);   // (out == nullptr: the caller sums the partials itself)
    HDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int HdmLanczos::solve(const double *Linv, long ldl, const double *dS, long ldd, hipStream_t s, double *maxStep, int *steps) {
    static const bool whole_env = [] { const char *e = getenv("HDM_LANCZOS_WHOLE"); return !(e && atoi(e) == 0); }();   // 0: the multi-launch forms (A/B, tests)
    if (whole_env && n16 <= LZ_FUSED_MAX && maxdim == LZ_MD) {
        // small block: the whole test in one launch (hdm_lanczos_whole_kernel); three doubles come back
        if (n16 <= LZ_RESIDENT_MAX) {
            const size_t dyn = sizeof(double) * (size_t) n16 * (n16 + 1);      // two packed triangles
            static thread_local int configured_dev = -1;
            int dev = 0;
            HDM_HIP_CHECK(hipGetDevice(&dev));
            if (configured_dev != dev) {
                HDM_HIP_CHECK(hipFuncSetAttribute((const void *) hdm_lanczos_whole_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int) (sizeof(double) * LZ_RESIDENT_MAX * (LZ_RESIDENT_MAX + 1))));
                configured_dev = dev;
            }
            hipLaunchKernelGGL(hdm_lanczos_whole_kernel<true>, dim3(1), dim3(1024), dyn, s, Linv, ldl, dS, ldd, n16, V, (long) n16, startd,
                               warm, nComputed == 0 ? 1 : 0, scal + 60);
        } else {
            hipLaunchKernelGGL(hdm_lanczos_whole_kernel<false>, dim3(1), dim3(1024), 0, s, Linv, ldl, dS, ldd, n16, V, (long) n16, startd,
                               warm, nComputed == 0 ? 1 : 0, scal + 60);
        }
        HDM_HIP_CHECK(hipGetLastError());
        HDM_HIP_CHECK(hipStreamSynchronize(s));
        const double r[3] = {scal_h[60], scal_h[61], scal_h[62]};
        if (r[2] != 0.0) return 1;
        nComputed += 1;
        if (maxStep) *maxStep = r[0];
        if (steps) *steps = (int) r[1];
        return 0;
    }
    const int md = maxdim, nh = md + 1;
    std::vector<double> H((size_t) nh * nh, 0.0);
    auto Hm = [&](int i, int j) -> double & { return H[(size_t) j * nh + i]; };
    // starting vector: fresh, or the previous Ritz image + 1e-3 * the same pseudo-random vector (:166-181), made on the device
    hipLaunchKernelGGL(hdm_warm_start_kernel, dim3(1), dim3(1024), 0, s, tmp, warm, startd, n, n16, nComputed == 0 ? 1 : 0);
    HDM_HIP_CHECK(hipMemsetAsync(V, 0, sizeof(double) * (size_t) n16 * (md + 1), s));
    hipLaunchKernelGGL(hdm_normalize_kernel, dim3(1), dim3(1024), 0, s, tmp, V, bv, n16);
    HDM_HIP_CHECK(hipGetLastError());

    int checkFreq = md / 5;
    if (checkFreq > 3) checkFreq = 3;
    double step = 0.0;
    int k = 0;
    double hs[2] = {0.0, 0.0};
    std::vector<double> d, Y;
    static const bool fuse_env = [] { const char *e = getenv("HDM_LANCZOS_FUSED"); return !(e && atoi(e) == 0); }();
    const bool fused = fuse_env && n16 <= LZ_FUSED_MAX && checkFreq >= 1;
    static const bool group_env = [] { const char *e = getenv("HDM_LANCZOS_GROUP"); return !(e && atoi(e) == 0); }();   // 0: one synchronisation per step (A/B)
    // large blocks: the steps of a group in one launch of co-resident workgroups (hdm_lanczos_group_kernel); 0: a launch per product
    static const bool big_env = [] { const char *e = getenv("HDM_LANCZOS_BIG"); return !(e && atoi(e) == 0); }();
    bool big = big_env && big_ok && !fused && n16 > LZ_FUSED_MAX && n16 <= 4096 && checkFreq >= 1 && !hdm_flow_shared_device();
    if (big) {
        if (big_wg == 0) {
            int dev = 0, cus = 0;
            big_wg = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
                         ? std::min(LZG_WG, cus) : -1;
            (void) hipGetLastError();
        }
        if (big_wg <= 0 || (n16 + big_wg - 1) / big_wg > 16) big = false;
    }
    static const bool dbg2 = [] { const char *e = getenv("HDSDP_MI355X_RATIO_DEBUG"); return e && atoi(e) >= 2; }();
    if (big) {
        if (dbg2) HDM_HIP_CHECK(hipMemsetAsync(bz + 4 * (size_t) n16, 0, sizeof(double) * 16, s));
        if (!LT) HDM_HIP_CHECK(hipMalloc((void **) &LT, sizeof(double) * (size_t) n16 * n16));
        if (!gsync) {
            HDM_HIP_CHECK(hipMalloc((void **) &gsync, sizeof(unsigned) * (64 + LZG_WG)));
            HDM_HIP_CHECK(hipMemsetAsync(gsync, 0, sizeof(unsigned) * (64 + LZG_WG), s));
            sync_epoch = 0;
        }
        hipLaunchKernelGGL(hdm_transpose_kernel, dim3((n16 + 31) / 32, (n16 + 31) / 32), dim3(256), 0, s, Linv, ldl, LT, (long) n16, n16);
        HDM_HIP_CHECK(hipGetLastError());
    }
    double grp[2 * 8 + 1] = {0.0};               // (alpha, beta) of the current group of steps, fused form
    int grp_k0 = -1, grp_n = 0;
    for (k = 0; k < md; ++k) {
        const double hprev = (k > 0) ? Hm(k, k - 1) : 0.0;
        if (fused) {
            if (grp_k0 < 0 || k >= grp_k0 + grp_n) {       // next group: as many steps as lie before the next Ritz check
                grp_k0 = k;
                grp_n = std::min(std::min(checkFreq - (k % checkFreq), md - k), 8);
                hipLaunchKernelGGL(hdm_lanczos_fused_kernel, dim3(1), dim3(1024), 0, s, Linv, ldl, dS, ldd, n16, V, (long) n16, k, grp_n,
                                   hprev, bv, scal + 44);
                HDM_HIP_CHECK(hipGetLastError());
                HDM_HIP_CHECK(hipStreamSynchronize(s));
                for (int q = 0; q < 2 * grp_n + 1; ++q) grp[q] = scal_h[44 + q];   // (scal + 8 .. + 39, + 64 .. + 95: Ritz coefficients)
            }
            hs[0] = grp[2 * (k - grp_k0)]; hs[1] = grp[2 * (k - grp_k0) + 1];
        } else if (big) {
            if (grp_k0 < 0 || k >= grp_k0 + grp_n) {
                grp_k0 = k;
                grp_n = std::min(std::min(checkFreq - (k % checkFreq), md - k), 8);
                LzgArgs a = {};
                a.epoch0 = sync_epoch; sync_epoch += 3u * (unsigned) grp_n;     // (the barrier words are never reset: epochs only grow)
                scal_h[44 + 2 * grp_n + 1] = 0.0;                               // give-up word of this launch
                a.Linv = Linv; a.ldl = ldl; a.LinvT = LT; a.ldt = n16; a.dS = dS; a.ldd = ldd; a.n = n16; a.V = V; a.ldv = n16;
                a.k0 = k; a.nsteps = grp_n; a.hprev = hprev; a.blk = bv; a.t1 = b1; a.t2 = b2; a.xw = b1 + n16;     // (the vector blocks are n16 x 8: column 1 of b1 is free)
                a.out = scal + 44; a.sync = gsync; a.dbg = dbg2 ? bz + 4 * (size_t) n16 : nullptr;
                if (n16 <= 2048) hipLaunchKernelGGL(hdm_lanczos_group_kernel<8>, dim3(big_wg), dim3(256), 0, s, a);
                else hipLaunchKernelGGL(hdm_lanczos_group_kernel<16>, dim3(big_wg), dim3(256), 0, s, a);
                HDM_HIP_CHECK(hipGetLastError());
                HDM_HIP_CHECK(hipStreamSynchronize(s));
                for (int q = 0; q < 2 * grp_n + 1; ++q) grp[q] = scal_h[44 + q];
                if (scal_h[44 + 2 * grp_n + 1] != 0.0) {
                    HDM_HIP_CHECK(hipMemsetAsync(gsync, 0, sizeof(unsigned) * (64 + LZG_WG), s));
                    // a grid-wide wait ran out (the workgroups were not all resident): this object goes back to a launch per
                    // product for good, and this test starts over -- nothing of it has left the object yet
                    fprintf(stderr, "[hdsdp_mi355x] lanczos: grid-wide wait timed out, falling back to one launch per product\n");
                    big_ok = false;
                    return solve(Linv, ldl, dS, ldd, s, maxStep, steps);
                }
            }
            hs[0] = grp[2 * (k - grp_k0)]; hs[1] = grp[2 * (k - grp_k0) + 1];
        } else {
            // large blocks: the steps up to the next Ritz check are queued back to back (4 launches each: three products and
            // the recurrence, which takes the previous step's norm from device memory); one copy and one synchronisation
            // per group.  A zero norm inside a group ends the reference's loop at that step: the host stops there too and
            // what the later steps of the group computed is never looked at.
            if (grp_k0 < 0 || k >= grp_k0 + grp_n) {
                grp_k0 = k;
                grp_n = group_env ? std::min(std::min(checkFreq - (k % checkFreq), md - k), 8) : 1;
                for (int q = 0; q < grp_n; ++q) {
                    const int kk = k + q;
                    if (apply(Linv, ldl, dS, ldd, bv, nullptr, s)) return 1;
                    hipLaunchKernelGGL(hdm_lanczos_step_kernel, dim3(1), dim3(1024), 0, s, part, LZ_NCHUNK, bw,
                                       kk > 0 ? V + (size_t) (kk - 1) * n16 : nullptr,
                                       q > 0 ? scal + 44 + 2 * (q - 1) + 1 : scal + 43,
                                       V + (size_t) kk * n16, V + (size_t) (kk + 1) * n16, bv, n16, scal + 44 + 2 * q);
                }
                HDM_HIP_CHECK(hipGetLastError());
                HDM_HIP_CHECK(hipStreamSynchronize(s));
                for (int q = 0; q < 2 * grp_n; ++q) grp[q] = scal_h[44 + q];
                // the last norm of this group is the next group's hprev: keep it where the next group's first step reads it
                scal_h[43] = scal_h[44 + 2 * (grp_n - 1) + 1];
            }
            hs[0] = grp[2 * (k - grp_k0)]; hs[1] = grp[2 * (k - grp_k0) + 1];
        }
        const double vAlp = hs[0], normPres = hs[1];
        Hm(k, k) = -vAlp;
        if (normPres > 0.0) Hm(k + 1, k) = Hm(k, k + 1) = normPres;

        if ((k + 1) % checkFreq == 0 || k > md - 1 || normPres == 0.0) {
            const int kp = k + 1;
            std::vector<double> U((size_t) kp * kp);
            for (int j = 0; j < kp; ++j)
                for (int i = 0; i < kp; ++i) U[(size_t) j * kp + i] = 0.5 * (Hm(i, j) + Hm(j, i));
            if (!tridiag_eig(kp, U, d, Y)) jacobi_eig(kp, U, d, Y);
            const double eig1 = d[kp - 1], eig2 = kp > 1 ? d[kp - 2] : d[kp - 1];
            // (sign convention as in the single-launch form: largest-magnitude component positive)
            for (int col : {kp - 1, kp > 1 ? kp - 2 : kp - 1}) {
                double *yc = &Y[(size_t) col * kp], big = 0.0;
                for (int r = 0; r < kp; ++r) if (fabs(yc[r]) > fabs(big)) big = yc[r];
                if (big < 0.0) for (int r = 0; r < kp; ++r) yc[r] = -yc[r];
            }
            const double *y1 = &Y[(size_t) (kp - 1) * kp], *y2 = kp > 1 ? &Y[(size_t) (kp - 2) * kp] : y1;
            const double resiVal = fabs(Hm(kp, k) * y1[k]);
            if (resiVal < 1e-04 || k >= md - 1) {
                // z1 = V y1 ; z2 = Op z1 ; warm start <- z2 ; resiVal1 = | z2 - eig1 z1 |
                // z2' = V y2 ; resiVal2 = | Op z2' - eig1 z2' |   (the reference uses eig1 here too, :262-266)
                // (both coefficient vectors go into the mapped block, everything is queued, one synchronisation)
                for (int r = 0; r < kp; ++r) { scal_h[8 + r] = y1[r]; scal_h[64 + r] = y2[r]; }
                hipLaunchKernelGGL(hdm_lincomb_kernel, dim3(1), dim3(1024), 0, s, V, (long) n16, kp, scal + 8, bz, n16);
                if (apply(Linv, ldl, dS, ldd, bz, bw, s)) return 1;
                HDM_HIP_CHECK(hipMemcpyAsync(warm, bw, sizeof(double) * n16, hipMemcpyDeviceToDevice, s));
                hipLaunchKernelGGL(hdm_resnorm_kernel, dim3(1), dim3(1024), 0, s, bw, bz, eig1, n16, scal + 2);
                hipLaunchKernelGGL(hdm_lincomb_kernel, dim3(1), dim3(1024), 0, s, V, (long) n16, kp, scal + 64, bz, n16);
                if (apply(Linv, ldl, dS, ldd, bz, bw, s)) return 1;
                hipLaunchKernelGGL(hdm_resnorm_kernel, dim3(1), dim3(1024), 0, s, bw, bz, eig1, n16, scal + 3);
                HDM_HIP_CHECK(hipGetLastError());
                HDM_HIP_CHECK(hipStreamSynchronize(s));
                const double r12[2] = {scal_h[2], scal_h[3]};
                // after the second apply() the vector block bv still holds v_{k+1}: the recurrence can continue
                const double resiVal1 = r12[0], resiVal2 = r12[1];
                const double resiDiff = eig1 - eig2 - resiVal2;
                double gam = (resiDiff > 0) ? resiDiff : 1e-16;
                const double sq = resiVal1 * resiVal1 / gam;
                gam = resiVal1 < sq ? resiVal1 : sq;
                if (gam < 1e-03 || gam + eig1 <= 0.5) {
                    step = (gam + eig1 <= 0.0) ? INFINITY : 1.0 / (gam + eig1);
                    break;
                } else {
                    if (normPres == 0.0) return 1;
                    step = 1.0 / (gam + eig1);
                }
            }
        }
    }
    if (big && dbg2) {
        double t[10];
        HDM_HIP_CHECK(hipMemcpy(t, bz + 4 * (size_t) n16, sizeof(t), hipMemcpyDeviceToHost));
        fprintf(stderr, "[hdsdp_mi355x ratio] group kernel, workgroup 0, us: Linv^T v %.0f | wait %.0f | dS t1 %.0f | wait %.0f | Linv t2 + x %.0f | wait %.0f | "
                        "alpha, norm, v_{k+1} %.0f\n", t[0] / 100, t[1] / 100, t[2] / 100, t[3] / 100, t[4] / 100, t[5] / 100, t[6] / 100);
    }
    nComputed += 1;
    if (maxStep) *maxStep = step;
    if (steps) *steps = k;
    return 0;
}

// one kernel of this translation unit (= one code object): what the preload thread asks the runtime about (engine.hip: preload_modules)
const void *hdm_module_handle_lanczos() { return (const void *) hdm_lanczos_step_kernel; }
