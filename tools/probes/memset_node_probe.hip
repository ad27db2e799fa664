// memset_node_probe.hip -- diagnostic, not product (nothing of this repository's library is linked).
//
// Question (profiles/r04_c_poison.txt item 1): under HDM_POISON a REPLAYED hipGraph MEMSET node -- hipMemsetAsync(info, 0, 4),
// captured as the first node of a factorisation chain -- left 0xFFFFFFFF in its word on the fourth run of the exec, the value
// of the poison fills (plain hipMemset(p, 0xFF, bytes) of other, newly allocated buffers) that had run in between.  Is that a
// property of the runtime that a stand-alone program shows?  Each scenario captures  [memset(word, 0, 4); K no-op kernels],
// replays it R times, sets the word to a sentinel with a kernel before every replay, and between replays runs one kind of
// unrelated traffic; a replay after which the word is not 0 is reported with the value found.
//   hipcc --offload-arch=gfx950 -O2 -o memset_node_probe memset_node_probe.hip && ./memset_node_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorName(e_), __LINE__); exit(2); } } while (0)

__global__ void set_word(unsigned *p, unsigned v) { *p = v; }
__global__ void noop(const unsigned *p, unsigned *sink) { if (*p == 0xDEADBEEFu && sink) *sink = 1; }

enum Traffic { NONE = 0, MEMSET_FF_SMALL, MEMSET_FF_BIG, ALLOC_FILL_FREE, MEMSET_ASYNC_FF_SAME_STREAM, MEMSET_D32_OTHER_STREAM };
static const char *names[] = {"no traffic between replays", "hipMemset(other 256 B, 0xFF) + device sync", "hipMemset(other 64 MiB, 0xFF) + device sync",
                              "hipMalloc + hipMemset(0xFF) + sync + hipFree, sizes 4 B .. 32 MiB (what HDM_POISON does)",
                              "hipMemsetAsync(other, 0xFF, 4) on the replay's own stream", "hipMemsetD32Async(other, 0xA5A5A5A5) on another stream"};

static int scenario(Traffic tr, int replays, int kernels) {
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned *word = nullptr, *sink = nullptr, *other = nullptr, *big = nullptr;
    CK(hipMalloc((void **) &word, 4)); CK(hipMalloc((void **) &sink, 4)); CK(hipMalloc((void **) &other, 256)); CK(hipMalloc((void **) &big, 64 << 20));
    CK(hipMemset(word, 0xFF, 4)); CK(hipMemset(sink, 0, 4)); CK(hipDeviceSynchronize());
    // first run eager, as the engine's chains do
    CK(hipMemsetAsync(word, 0, 4, s));
    for (int k = 0; k < kernels; ++k) hipLaunchKernelGGL(noop, dim3(1), dim3(64), 0, s, word, sink);
    CK(hipStreamSynchronize(s));
    hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(word, 0, 4, s));
    for (int k = 0; k < kernels; ++k) hipLaunchKernelGGL(noop, dim3(1), dim3(64), 0, s, word, sink);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    int bad = 0; unsigned first_val = 0; int first_at = -1;
    std::vector<void *> keep;
    for (int r = 0; r < replays; ++r) {
        switch (tr) {
            case MEMSET_FF_SMALL: CK(hipMemset(other, 0xFF, 256)); CK(hipDeviceSynchronize()); break;
            case MEMSET_FF_BIG: CK(hipMemset(big, 0xFF, 64 << 20)); CK(hipDeviceSynchronize()); break;
            case ALLOC_FILL_FREE: {
                const size_t sizes[6] = {4, 2048, 33554432, 8, 1 << 20, 160000};
                for (int q = 0; q < 6; ++q) {
                    void *p = nullptr;
                    CK(hipMalloc(&p, sizes[(q + r) % 6])); CK(hipMemset(p, 0xFF, sizes[(q + r) % 6])); CK(hipDeviceSynchronize());
                    keep.push_back(p);
                }
                while (keep.size() > 4) { CK(hipFree(keep.front())); keep.erase(keep.begin()); }
                break;
            }
            case MEMSET_ASYNC_FF_SAME_STREAM: CK(hipMemsetAsync(other, 0xFF, 4, s)); break;
            case MEMSET_D32_OTHER_STREAM: CK(hipMemsetD32Async((hipDeviceptr_t) other, 0xA5A5A5A5, 16, s2)); CK(hipStreamSynchronize(s2)); break;
            default: break;
        }
        hipLaunchKernelGGL(set_word, dim3(1), dim3(1), 0, s, word, 0x12345678u);
        CK(hipGraphLaunch(ex, s));
        unsigned h = 7;
        CK(hipMemcpyAsync(&h, word, 4, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (h != 0) { bad += 1; if (first_at < 0) { first_at = r; first_val = h; } }
    }
    printf("%-100s %5d replays: %s", names[tr], replays, bad ? "MISBEHAVED" : "every replay wrote 0");
    if (bad) printf("  (%d replays left a non-zero word; first at replay %d: 0x%08X)", bad, first_at, first_val);
    printf("\n");
    fflush(stdout);
    for (void *p : keep) (void) hipFree(p);
    (void) hipGraphExecDestroy(ex);
    (void) hipFree(word); (void) hipFree(sink); (void) hipFree(other); (void) hipFree(big);
    (void) hipStreamDestroy(s); (void) hipStreamDestroy(s2);
    return bad;
}

int main(int argc, char **argv) {
    const int replays = argc > 1 ? atoi(argv[1]) : 200, kernels = argc > 2 ? atoi(argv[2]) : 48;
    int tot = 0;
    for (int t = NONE; t <= MEMSET_D32_OTHER_STREAM; ++t) tot += scenario((Traffic) t, replays, kernels) ? 1 : 0;
    printf("memset_node_probe: %d of 6 scenarios misbehaved\n", tot);
    return 0;
}
