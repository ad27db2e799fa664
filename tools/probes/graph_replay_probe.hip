// graph_replay_probe.hip -- diagnostic, not product: does `rocprofv3 --kernel-trace` survive many replays of ONE captured
// chain of short kernels?  (profiles/r04_a_headline_segv.txt: the headline solve under the profiler died inside
// librocprofiler-sdk's queue write interceptor beneath a hipGraphLaunch of a long-lived exec, after some hundreds of
// replays; nothing of this repository's library is linked here.)
//   hipcc --offload-arch=gfx950 -O2 -o graph_replay_probe graph_replay_probe.hip
//   rocprofv3 --kernel-trace --stats -d <dir> -- ./graph_replay_probe [replays=4000] [nodes=48]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void tick(int *p, int k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += k; }

int main(int argc, char **argv) {
    const int replays = argc > 1 ? atoi(argv[1]) : 4000, nodes = argc > 2 ? atoi(argv[2]) : 48;
    hipStream_t s;
    int *d = nullptr, h = 0;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **) &d, sizeof(int)) != hipSuccess) return 2;
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return 3;
    (void) hipMemsetAsync(d, 0, sizeof(int), s);
    for (int k = 0; k < nodes; ++k) hipLaunchKernelGGL(tick, dim3(1), dim3(64), 0, s, d, 1);
    if (hipStreamEndCapture(s, &g) != hipSuccess || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) return 4;
    (void) hipGraphDestroy(g);
    for (int r = 0; r < replays; ++r) {
        if (hipGraphLaunch(ex, s) != hipSuccess) { fprintf(stderr, "replay %d: launch failed\n", r); return 5; }
        if (hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 6;
        if (h != nodes) { fprintf(stderr, "replay %d: %d != %d\n", r, h, nodes); return 7; }
        if ((r + 1) % 500 == 0) { printf("replay %d ok\n", r + 1); fflush(stdout); }
    }
    printf("graph_replay_probe: %d replays of a %d-kernel graph completed\n", replays, nodes);
    (void) hipGraphExecDestroy(ex);
    return 0;
}
