// diagnostic (not part of the library): how long does a freed workgroup slot wait for its successor?
// A grid of workgroups shaped like the GEMM kernel's (256 threads, 72 KB of LDS, 2 per CU) in which workgroup i spins for
// dur[i] microseconds and stamps start / end (100 MHz s_memrealtime) and where it ran.  Three duration patterns: all equal;
// the mix of congruence step 2 (a tile list whose weights run from 1 to 16, heaviest first, repeated per batch entry); and
// the same mix in random order.  Reported: wall time against the ideal sum(dur) / slots, and per CU the share of its active
// span with two workgroups resident -- what tools/wg_timeline.py reports for the real kernels.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/dispatch_probe.hip -o tools/probes/dispatch_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

__global__ __launch_bounds__(256, 2) void spin(const unsigned *dur_ticks, unsigned long long *out) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) lds[0] = 1;
    const unsigned long long ticks = dur_ticks[blockIdx.x];
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        unsigned long long *d = out + (size_t) blockIdx.x * 4;
        d[0] = t0; d[1] = __builtin_amdgcn_s_memrealtime();
        d[2] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);     // HW_ID[15:0]
        d[3] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID
    }
}

int main() {
    const int entries = 400, tiles = 136, nwg = entries * tiles;
    std::vector<int> weight;                        // the lower tile list of a 16 x 16 tile grid, heaviest first (weight = tn + 1)
    for (int w = 16; w >= 1; --w) for (int tm = w - 1; tm < 16; ++tm) weight.push_back(w);
    (void) hipFuncSetAttribute((const void *) spin, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    unsigned *dd = nullptr; unsigned long long *od = nullptr;
    hipMalloc((void **) &dd, sizeof(unsigned) * nwg); hipMalloc((void **) &od, sizeof(unsigned long long) * 4 * nwg);
    std::mt19937 rng(1);
    for (int pattern = 0; pattern < 3; ++pattern) {
        std::vector<unsigned> dur(nwg);
        // workgroup ids are dealt round-robin over 8 XCDs; XCD x walks entries x, x + 8, ... and their tiles in order (the kernel's decode)
        for (int wg = 0; wg < nwg; ++wg) {
            const int idx = wg >> 3, t = idx % tiles;
            const double us = pattern == 0 ? 250.0 : 250.0 * weight[t] / 6.0;     // mean weight of the list is 6
            dur[wg] = (unsigned) (us * 100.0);
        }
        if (pattern == 2) std::shuffle(dur.begin(), dur.end(), rng);
        hipMemcpy(dd, dur.data(), sizeof(unsigned) * nwg, hipMemcpyHostToDevice);
        hipMemset(od, 0, sizeof(unsigned long long) * 4 * nwg);
        hipLaunchKernelGGL(spin, dim3(nwg), dim3(256), 72 * 1024, 0, dd, od);
        hipDeviceSynchronize();
        std::vector<unsigned long long> o((size_t) 4 * nwg);
        hipMemcpy(o.data(), od, sizeof(unsigned long long) * 4 * nwg, hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ULL, tmax = 0; double sum = 0.0;
        std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
        for (int i = 0; i < nwg; ++i) {
            tmin = std::min(tmin, o[4 * i]); tmax = std::max(tmax, o[4 * i + 1]); sum += (double) dur[i];
            const unsigned long long hw = o[4 * i + 2], cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | ((o[4 * i + 3] & 0xF) << 8);
            ev[cu].push_back({o[4 * i], +1}); ev[cu].push_back({o[4 * i + 1], -1});
        }
        double res[3] = {0, 0, 0}; double gap = 0.0; long ngap = 0;
        for (auto &kv : ev) {
            auto &v = kv.second; std::sort(v.begin(), v.end());
            int cur = 0; unsigned long long last = v[0].first, lastend = 0;
            for (auto &e : v) {
                res[std::min(cur, 2)] += (double) (e.first - last); last = e.first;
                if (e.second < 0) lastend = e.first;
                else if (lastend) { gap += (double) (e.first - lastend); ++ngap; lastend = 0; }
                cur += e.second;
            }
        }
        const double tot = res[0] + res[1] + res[2];
        printf("pattern %d (%s): %d workgroups on %zu CUs, wall %.2f ms, ideal %.2f ms (sum / %zu slots), efficiency %.3f; CU time with 0 / 1 / 2 resident %.3f %.3f %.3f; end -> next start on the CU %.1f us\n",
               pattern, pattern == 0 ? "all 250 us" : pattern == 1 ? "step-2 mix, list order" : "step-2 mix, shuffled", nwg, ev.size(), (tmax - tmin) / 1e5,
               sum / (2.0 * ev.size()) / 1e5, 2 * ev.size(), sum / (2.0 * ev.size()) / (double) (tmax - tmin), res[0] / tot, res[1] / tot, res[2] / tot, gap / std::max(1L, ngap) / 100.0);
    }
    return 0;
}
