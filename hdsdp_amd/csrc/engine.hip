// engine.hip -- host side of the MI355X Schur engine: the HKKT* / HFpLinsys* operator surface of the
// reference (interface/hdsdp_schur.c, linalg/hdsdp_linsolver.c) re-implemented over device-resident
// state, plus the MI355X SDP cone whose `coneBuildSchur` slot is the GPU builder that replaces
// sdpDenseConeGetKKT (interface/hdsdp_conic_sdp.c:1726-1812).
//
// Mathematical formulation.  With S = L L^T and At_i = L^-1 A_i L^-T:
//     M_ij               = tr(A_i S^-1 A_j S^-1)        = <At_i, At_j>
//     ASinv_i            = tr(A_i S^-1)                 = <At_i, I>        ("S row":  A = S -> At = I)
//     ASinvRdSinv_i / Rd = tr(S^-1 A_i S^-1)            = <At_i, L^-1 L^-T>("I row":  A = I)
//     ASinvCSinv_i       = tr(A_i S^-1 C S^-1)          = <At_i, Ct>       ("C row")
// so one augmented Gram matrix of the congruence-transformed constraints yields M, all three
// vectors and the four scalars (TraceSinv, CSinv, CSinvCSinv, CSinvRdSinv) of hdsdp_kkt.
// Every reference strategy M2..M5 (hdsdp_conic_sdp.c:687-985) evaluates these same quantities; the
// strategy plan is a CPU cost model and does not change the result (HUtilKKTCheck, hdsdp_utils.c:536-707).
#include "../../include/hdsdp_mi355x.h"
#include "chol.h"
#include "coeff.h"
#include "hdm_common.h"
#include "schur.h"
#include "lanczos.h"
#include "lu.h"
#include "small.h"
#include "bsparse.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <rccl/rccl.h>

// kernels defined further down in this file (global scope)
__global__ void mi_low_norms_kernel(const double *__restrict__ A, long astride, int n, long ld, int count, int a_l_form,
                                    double *__restrict__ out);
__global__ void mi_scale_kernel(double *__restrict__ A, long count, double s);
__global__ void mi_put_row_kernel(HdmMatView Mv, int i, const double *__restrict__ v, int m);
__global__ void mi_lower_dot_kernel(const double *__restrict__ S, long lds_, const double *__restrict__ X, long ldx, int n,
                                    double *__restrict__ out);
__global__ void mi_mat_dot_kernel(const double *__restrict__ X, long ldx, const double *__restrict__ Y, long ldy, int n,
                                  int diag_only, double scale, double *__restrict__ out);
__global__ void mi_csc_gather_kernel(HdmMatView Mv, const int *__restrict__ rows, const int *__restrict__ cols, long nnz,
                                     double *__restrict__ vals);
__global__ void mi_csc_scatter_kernel(HdmMatView Mv, const int *__restrict__ rows, const int *__restrict__ cols, long nnz,
                                      const double *__restrict__ vals);
__global__ void mi_get_row_kernel(HdmMatView Mv, int i, int m, double *__restrict__ out);

namespace {

struct Ctx {
    bool init = false;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8];
    double stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
// One context (device + stream + events) per host thread that drives a device: the caller's thread owns g_main; in the
// single-process multi-device mode (group_impl.h) every shard has a worker thread whose t_ctx points at the shard's own
// context, and all the code below reaches "its" stream through `g`.
Ctx g_main;
thread_local Ctx *t_ctx = nullptr;
inline Ctx &cur_ctx() { return t_ctx ? *t_ctx : g_main; }
#define g cur_ctx()

int g_main_device_request = -1;   // HMiSetDevices / HDSDP_MI355X_GPUS: device of the caller's context (= shard 0's)

void stats_print_at_exit();

// Code objects come up beside the caller's own start-up work, not inside its first HKKTBuildUp (hdm_common.h:
// hdm_module_handle_*).  One helper thread per process, started when the first context opens, joined at exit (the atexit
// handler is registered after the HIP runtime's own, so it runs before the runtime is taken down).
std::thread g_preload_thread;
std::once_flag g_preload_once;
void preload_join() { if (g_preload_thread.joinable()) g_preload_thread.join(); }
void preload_modules(int dev) {
    if (const char *e = getenv("HDSDP_MI355X_PRELOAD")) if (!atoi(e)) return;
    std::call_once(g_preload_once, [dev] {
        g_preload_thread = std::thread([dev] {
            if (hipSetDevice(dev) != hipSuccess) return;
            const void *handles[] = {hdm_module_handle_small(), hdm_module_handle_chol(), hdm_module_handle_schur(),
                                     hdm_module_handle_lanczos(), hdm_module_handle_gemm_f64(), hdm_module_handle_lu(),
                                     hdm_module_handle_bsparse(), hdm_module_handle_gemm_persist(), (const void *) mi_scale_kernel};
            for (const void *h : handles) { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, h); }
        });
        atexit(preload_join);
    });
}

int ctx_open(Ctx &c, int dev) {
    HDM_HIP_CHECK(hipSetDevice(dev));
    c.device = dev;
    HDM_HIP_CHECK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    for (int i = 0; i < 8; ++i) HDM_HIP_CHECK(hipEventCreate(&c.ev[i]));
    c.init = true;
    preload_modules(dev);
    return 0;
}

int ensure_ctx() {
    if (g.init) return 0;
    int dev = 0;
    const char *lr = getenv("LOCAL_RANK");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "[hdsdp_mi355x] no HIP device visible: this library has no CPU fallback\n");
        return 1;
    }
    if (lr) dev = atoi(lr) % ndev;
    if (g_main_device_request >= 0) dev = g_main_device_request % ndev;
    if (const char *e = getenv("HDSDP_MI355X_CALL_STATS")) if (atoi(e)) atexit(stats_print_at_exit);
    return ctx_open(g, dev);
}


#include "engine_stats.h"
#include "engine_linsys.h"
#include "engine_cone.h"
#include "engine_build.h"

}  // namespace

#include "engine_kernels.h"

namespace {
#include "engine_build_small.h"
}  // namespace

// =============================================================================================
// exported C ABI
// =============================================================================================
extern "C" {
#include "engine_kkt.h"
}  // extern "C"

namespace {
#include "engine_create.h"
// single-process multi-device mode: device groups, worker threads, the two transports (device copies / RCCL) and the
// group cone whose slots fan out to the shards
#include "group_impl.h"
}  // namespace

extern "C" {
#include "engine_api.h"
}  // extern "C"
