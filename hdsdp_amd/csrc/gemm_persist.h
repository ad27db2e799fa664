// gemm_persist.h -- the persistent-workgroup launches of the GEMM family live in their own translation unit (gemm_persist.hip)
#pragma once
#include "gemm_tile.h"

// does (layout, role, variant) have a persistent form?
bool hdm_persist_supported(bool a_kmajor, bool b_kmajor, int role, int variant);
// launches hdm_gemm_persist_kernel<AKmajor, BKmajor, role, variant>; returns 1 when that combination has no persistent
// form (the caller then launches one workgroup per tile)
int hdm_launch_persist(bool a_kmajor, bool b_kmajor, int role, int variant, dim3 grid, dim3 block, hipStream_t stream,
                       const HdmGemmDev &d, int *cnt);
