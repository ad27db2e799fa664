// gemm_persist.hip -- instantiations of the persistent-workgroup GEMM kernels (gemm_tile.h: hdm_gemm_persist_kernel).
// Persistent forms exist for the three Schur roles in their fixed operand layouts: the default loop (variant 64) -- the only
// one the shipped library carries.  A -DHDM_DIAGNOSTICS build (python -m hdsdp_amd.build --diagnostics ->
// libhdsdp_mi355x_diag.so, never loaded by the product or the tests) adds the stamped build (96), the timing-only ablations
// whose RESULTS ARE WRONG (192: every tile stages the same rows; 576: step 2 without its stores) and the LDS-free step-2
// body (320), for tools/wg_timeline.py, tools/build_timing.py and tools/ab_env.sh.
#include "gemm_persist.h"

template <bool AK, bool BK, int R, int V>
static int launch_p(dim3 grid, dim3 block, hipStream_t stream, const HdmGemmDev &d, int *cnt) {
    hipLaunchKernelGGL((hdm_gemm_persist_kernel<AK, BK, R, V>), grid, block, 0, stream, d, cnt);
    return 0;
}

bool hdm_persist_supported(bool a_kmajor, bool b_kmajor, int role, int variant) {
#ifdef HDM_DIAGNOSTICS
    const bool common = (variant == 64 || variant == 96 || variant == 192);
    const bool cong2_extra = (variant == 320 || variant == 576);
#else
    const bool common = (variant == 64), cong2_extra = false;
#endif
    if (role == HDM_ROLE_CONG1) return !a_kmajor && b_kmajor && common;
    if (role == HDM_ROLE_CONG2) return !a_kmajor && !b_kmajor && (common || cong2_extra);
    if (role == HDM_ROLE_GRAM) return a_kmajor && b_kmajor && common;
    if (role == HDM_ROLE_CONG2D) return !a_kmajor && !b_kmajor && common;
    return false;
}

int hdm_launch_persist(bool a_kmajor, bool b_kmajor, int role, int variant, dim3 grid, dim3 block, hipStream_t stream,
                       const HdmGemmDev &d, int *cnt) {
    if (role == HDM_ROLE_CONG1 && !a_kmajor && b_kmajor) {
        if (variant == 64) return launch_p<false, true, HDM_ROLE_CONG1, 64>(grid, block, stream, d, cnt);
#ifdef HDM_DIAGNOSTICS
        if (variant == 96) return launch_p<false, true, HDM_ROLE_CONG1, 96>(grid, block, stream, d, cnt);
        if (variant == 192) return launch_p<false, true, HDM_ROLE_CONG1, 192>(grid, block, stream, d, cnt);
#endif
    } else if (role == HDM_ROLE_CONG2 && !a_kmajor && !b_kmajor) {
        if (variant == 64) return launch_p<false, false, HDM_ROLE_CONG2, 64>(grid, block, stream, d, cnt);
#ifdef HDM_DIAGNOSTICS
        if (variant == 96) return launch_p<false, false, HDM_ROLE_CONG2, 96>(grid, block, stream, d, cnt);
        if (variant == 192) return launch_p<false, false, HDM_ROLE_CONG2, 192>(grid, block, stream, d, cnt);
        if (variant == 320) return launch_p<false, false, HDM_ROLE_CONG2, 320>(grid, block, stream, d, cnt);
        if (variant == 576) return launch_p<false, false, HDM_ROLE_CONG2, 576>(grid, block, stream, d, cnt);
#endif
    } else if (role == HDM_ROLE_CONG2D && !a_kmajor && !b_kmajor) {
        if (variant == 64) return launch_p<false, false, HDM_ROLE_CONG2D, 64>(grid, block, stream, d, cnt);
#ifdef HDM_DIAGNOSTICS
        if (variant == 96) return launch_p<false, false, HDM_ROLE_CONG2D, 96>(grid, block, stream, d, cnt);
        if (variant == 192) return launch_p<false, false, HDM_ROLE_CONG2D, 192>(grid, block, stream, d, cnt);
#endif
    } else if (role == HDM_ROLE_GRAM && a_kmajor && b_kmajor) {
        if (variant == 64) return launch_p<true, true, HDM_ROLE_GRAM, 64>(grid, block, stream, d, cnt);
#ifdef HDM_DIAGNOSTICS
        if (variant == 96) return launch_p<true, true, HDM_ROLE_GRAM, 96>(grid, block, stream, d, cnt);
        if (variant == 192) return launch_p<true, true, HDM_ROLE_GRAM, 192>(grid, block, stream, d, cnt);
#endif
    }
    return 1;
}

// one kernel of this translation unit (= one code object): what the preload thread asks the runtime about (engine.hip: preload_modules)
const void *hdm_module_handle_gemm_persist() { return (const void *) hdm_gemm_persist_kernel<false, false, HDM_ROLE_CONG2, 64>; }
