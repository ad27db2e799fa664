// diagnostic: how many workgroups of a given shape does the dispatcher keep resident?  (not part of the library)
// Each workgroup spins for a fixed real-time interval and stamps start/end (100 MHz s_memrealtime) + HW ids.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));
// MODE 0: sleep-spin, 1: fp64 MFMA back to back (16 independent accumulators), 2: stream global memory, 3: both
template <int LDSB, int BIGV, int MODE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void spin(unsigned long long *out, unsigned long long ticks, const double *src = nullptr, size_t nsrc = 0) {
    __shared__ char lds[LDSB > 0 ? LDSB : 1];
    if (BIGV) asm volatile("v_mov_b32 v220, 0" ::: "v220");
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) lds[0] = 1;
    if (MODE == 0) {
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    } else {
        d4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
        double a = 1.0 + threadIdx.x, b = 0.5;
        size_t pos = ((size_t) blockIdx.x * 256 + threadIdx.x) * 2;
        double s = 0.0;
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
            if (MODE & 2) {
                for (int r = 0; r < 8; ++r) {
                    const double2 v = *reinterpret_cast<const double2 *>(src + (pos % nsrc));
                    s += v.x + v.y; pos += (size_t) 512 * 2048;
                }
            }
            if (MODE & 1) {
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
        }
        double t = s;
        for (int i = 0; i < 16; ++i) t += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        if (t == 12345.678) lds[1] = 2;
    }
    if (threadIdx.x == 0) {
        unsigned long long *d = out + (size_t) blockIdx.x * 4;
        d[0] = t0; d[1] = __builtin_amdgcn_s_memrealtime();
        d[2] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
        d[3] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) + lds[0] - 1;
    }
}

template <int LDSB, int BIGV, int MODE = 0> void run(const char *name, int nwg, unsigned long long ticks) {
    static double *src = nullptr; const size_t nsrc = (size_t) 1 << 29;  // 4 GiB
    if (!src) { hipMalloc((void **) &src, nsrc * 8); hipMemset(src, 0, nsrc * 8); hipDeviceSynchronize(); }
    unsigned long long *dev; std::vector<unsigned long long> h((size_t) nwg * 4);
    hipMalloc((void **) &dev, h.size() * 8); hipMemset(dev, 0, h.size() * 8); hipDeviceSynchronize();
    spin<LDSB, BIGV, MODE><<<nwg, 256>>>(dev, ticks, src, nsrc);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 1]); }
    int first = 0; int perx[8] = {0};
    for (int i = 0; i < nwg; ++i) if (h[4 * i] - t0 < ticks / 2) { first++; perx[h[4 * i + 3] & 7]++; }
    // time-averaged residency
    double busy = 0; for (int i = 0; i < nwg; ++i) busy += (double) (h[4 * i + 1] - h[4 * i]);
    printf("%-28s nwg %5d: resident in first round %4d (per XCD %d %d %d %d %d %d %d %d), rounds %.2f, avg resident %.1f\n", name, nwg, first,
           perx[0], perx[1], perx[2], perx[3], perx[4], perx[5], perx[6], perx[7], (double) (t1 - t0) / ticks, busy / (double) (t1 - t0));
    hipFree(dev);
}

int main() {
    const unsigned long long T = 200000;  // 2 ms
    run<0, 0>("lds 0, few vgprs", 8192, T / 4);
    run<65536, 0>("lds 64K, few vgprs", 2048, T);
    run<73728, 0>("lds 72K, few vgprs", 2048, T);
    run<81920, 0>("lds 80K, few vgprs", 2048, T);
    run<32768, 1>("lds 32K, 221 vgprs", 2048, T);
    run<65536, 1>("lds 64K, 221 vgprs", 2048, T);
    run<73728, 1>("lds 72K, 221 vgprs", 2048, T);
    run<73728, 1>("lds 72K, 221 vgprs, 512 wg", 512, T);
    run<81920, 1>("lds 80K, 221 vgprs", 2048, T);
    run<73728, 0, 1>("72K, 221 vgprs, MFMA", 2048, T);
    run<73728, 0, 1>("72K, 221 vgprs, MFMA long", 2048, 4 * T);
    run<73728, 0, 2>("72K, 221 vgprs, loads", 2048, T);
    run<73728, 0, 3>("72K, 221 vgprs, MFMA+loads", 2048, T);
    run<73728, 0, 3>("72K, 221, MFMA+loads long", 2048, 4 * T);
    return 0;
}
