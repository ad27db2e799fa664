// alloc.cpp -- the engine's one way to device memory (hdm_common.h: hdm_malloc).
#define HDM_MALLOC_IMPL       // this translation unit calls the runtime's own hipMalloc
#include "hdm_common.h"

hipError_t hdm_malloc(void **p, size_t bytes) {
    static const bool poison = [] { const char *e = getenv("HDM_POISON"); return e && atoi(e) != 0; }();
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && poison && bytes) {
        // (hipMemset of device memory may return before the fill has run, and the engine's streams are non-blocking: without
        // the synchronisation the poison could land AFTER the caller's own first writes)
        hipError_t f = hipMemset(*p, 0xFF, bytes);
        if (f == hipSuccess) f = hipDeviceSynchronize();
        if (f != hipSuccess) return f;
    }
    return e;
}
