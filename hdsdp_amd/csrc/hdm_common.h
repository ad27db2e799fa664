// hdm_common.h -- shared declarations for the MI355X (gfx950) HDSDP Schur engine.
// Internal header (C++/HIP). The public C ABI lives in include/hdsdp_mi355x.h.
#pragma once
#include <cstddef>
// Slack (bytes) appended to every device buffer that the role 1-3 GEMM kernels read as an operand: their staging loads
// carry no row mask (gemm_tile.h, SStager::load_nomask), so the last tile of the last matrix in a buffer may read up
// to 127 rows past its end.  `ld` = elements between consecutive rows (K-major) or 1 (M-major).
static inline size_t hdm_operand_pad(long ld) { return (size_t) 128 * (size_t) (ld < 16 ? 16 : ld) * 8 + 4096; }
#define HDM_OPERAND_PAD_DOUBLES 8192   /* the same slack for the [p-block][row][16] congruence output, in doubles */

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

// Every device allocation of the engine goes through hdm_malloc: plain hipMalloc, and -- with HDM_POISON=1 in the environment,
// a diagnostic -- the new memory filled with 0xFF bytes (NaN as a double, -1 as an int), so that a kernel that reads what
// nobody has written shows up as NaNs in the results instead of as whatever the allocator recycled (round 4: one run of the
// ingest test read an earlier cone's data out of a fresh buffer and nothing said so).
hipError_t hdm_malloc(void **p, size_t bytes);   // alloc.cpp
#ifndef HDM_MALLOC_IMPL
#define hipMalloc(p, bytes) hdm_malloc((void **) (p), (bytes))
#endif

#define HDM_TILE 128          // workgroup tile edge of the fp64 MFMA GEMM family
#define HDM_BK 16             // k-depth of one LDS stage
#define HDM_SUB 16            // MFMA sub-tile edge (v_mfma_f64_16x16x4_f64)

typedef double hdm_d4 __attribute__((ext_vector_type(4)));

// "Skyline" storage of a matrix in A_L form (strict lower triangle + half the diagonal; the constraint matrices and the
// objective as the congruence reads them): only the 128-column panels from their diagonal block downwards are stored --
// panel t holds rows 128 t .. n-1 of columns 128 t .. 128 t + 127 as a plain column-major (n - 128 t) x 128 matrix, the
// panels follow each other.  53 % of the square at n = 2000 (34 GB instead of 64 GB for 2000 matrices); inside a panel
// every column is contiguous and 128-byte aligned (n is a multiple of 16), so tile loads stay full lines, and the one
// GEMM operand that reads A_L (congruence step 1, B side: rows = columns of panel tn, k = rows from the panel's top)
// sees panel tn as an ordinary K-major matrix with leading dimension n - 128 tn.  The strict upper triangle of each
// panel's top block is stored and stays zero.
__host__ __device__ inline long hdm_sky_panel(int t, int n) { return 128L * ((long) t * n - 64L * t * (t - 1)); }   // start of panel t
__host__ __device__ inline long hdm_sky_off(int i, int j, int n) {   // element (i, j), i >= 128 * (j / 128)
    const int t = j >> 7;
    return hdm_sky_panel(t, n) + (long) (j & 127) * (n - 128 * t) + (i - 128 * t);
}
__host__ __device__ inline long hdm_sky_size(int n) {                // elements of one matrix
    const int t = (n + 127) / 128 - 1;
    const long w = n - 128L * t;
    return hdm_sky_panel(t, n) + w * w;
}

#define HDM_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            fprintf(stderr, "[hdsdp_mi355x] HIP error %s at %s:%d: %s\n", hipGetErrorName(_e),  \
                    __FILE__, __LINE__, hipGetErrorString(_e));                                 \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

// hipMemset on the legacy stream may still be in flight when work is queued on the engine's non-blocking
// stream: always drain it before anyone else touches the buffer.
static inline hipError_t hdm_memset_sync(void *p, int v, size_t bytes) {
    hipError_t e = hipMemset(p, v, bytes);
    if (e != hipSuccess) return e;
    return hipDeviceSynchronize();
}

static inline hipError_t hdm_memcpy_h2d_sync(void *dst, const void *src, size_t bytes) {
    hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    return hipDeviceSynchronize();
}

static inline long hdm_roundup(long x, long q) { return (x + q - 1) / q * q; }

// ---------------------------------------------------------------------------------------------
// GEMM family (gemm_f64.hip).  Everything is column-major fp64.
//   C[M x N] = alpha * A[M x K] * B[N x K]^T + beta * C
// Operand storage is selected per operand:
//   M-major ("N"): element (i,k) at X[i + k*ld]  (rows contiguous: a column-major M x K matrix)
//   K-major ("T"): element (i,k) at X[i*ld + k]  (k contiguous: the transpose is column-major)
// ---------------------------------------------------------------------------------------------
enum HdmKLimit { HDM_KLIM_NONE = 0, HDM_KLIM_BY_M = 1, HDM_KLIM_BY_N = 2, HDM_KLIM_BAND = 3 };  // BAND: k in [tn*128, (tm+1)*128)
enum HdmEpilogue {
    HDM_EPI_STORE = 0,    // C = alpha*acc + beta*C, column-major
    HDM_EPI_BLOCKED = 1,  // congruence output: 16x16-blocked lower triangle, sqrt(2) off-diagonal blocks
    HDM_EPI_SLAB = 2      // split-K partial sums into slab[blockIdx.z]
};

// kernel roles: a distinct kernel symbol per role so that rocprofv3 --stats separates the hot-path
// launches (congruence step 1/2, Gram) from the small Cholesky/TRTRI helper GEMMs
// HDM_ROLE_CONG2D is internal to the launcher: a role-2 launch is issued as two kernels, the full diagonal tiles (computed
// as P + P^T from one product, gemm_tile.h) and everything else; callers never ask for it
enum HdmRole { HDM_ROLE_GENERIC = 0, HDM_ROLE_CONG1 = 1, HDM_ROLE_CONG2 = 2, HDM_ROLE_GRAM = 3, HDM_ROLE_CONG2D = 4, HDM_NROLES = 5 };

struct HdmGemmArgs {
    const double *A, *B;
    double *C;
    // optional second product accumulated into the same tile (SYR2K form): C = alpha (A B^T + A2 B2^T) + beta C,
    // same shapes, storage classes and K range as the first pair; A2 == nullptr: single product
    const double *A2, *B2;
    long lda2, ldb2, strideA2, strideB2;
    int b_sky;        // congruence step 1: the B operand is a batch of skyline-stored A_L matrices (N = their dimension)
    // roles 1-3 (unmasked tile loads): elements readable from each operand pointer, slack included; checked at launch
    long spanA, spanB, spanA2, spanB2;
    long lda, ldb, ldc;
    long strideA, strideB, strideC;  // batch strides (elements) along blockIdx.z (batch) -- 0 = shared
    int M, N, K;
    int a_kmajor, b_kmajor;
    long a_kblk, b_kblk;  // K-major operands: elements between consecutive 16-deep k blocks (16 for a plain matrix)
    // K-major operands may be cut into row segments (one per source rank after the multi-GPU transpose):
    // element offset += (row / seg_rows) * seg_extra.  seg_rows == 0: one segment.
    long seg_rows, seg_extra;
    int klimit;       // HdmKLimit: triangular operand => shorter K loop for early tiles
    int lower_only;   // only tiles with tile_m >= tile_n are computed (C symmetric / lower)
    unsigned long long tile_col_mask;  // != 0: only tile columns whose bit is set are computed (N <= 64 tiles; multi-GPU
                                       // builds run congruence step 2 by packed-index range, see engine.hip)
    int epilogue;     // HdmEpilogue
    int batch;        // number of batch entries (grid z for STORE/BLOCKED), or #K-splits for SLAB
    int queue_global; // persistent launches: ONE job queue for the whole chip, batch entry (K split) by batch entry in order,
                      // instead of one queue per XCD over the entries x, x + 8, ... (gemm_tile.h: hdm_gemm_persist_kernel)
    double alpha, beta;
    int role;         // HdmRole
    double flops;     // algorithmic flops of this launch (valid data only), for the live roofline
    // BLOCKED epilogue: destination chunk layout  dst[((blk*16 + c_local) * rowStride + row) * 16 + r_local]
    long blk_row_stride;  // = m_pad (number of constraint rows per 16-wide p-block)
    long blk_row0;        // constraint row of batch entry 0
    int nblk;             // n/16: sub-blocks per matrix edge
    // SLAB epilogue / split-K
    long k_chunk;         // K range per split (multiple of HDM_BK)
    long k_base;          // first k of split 0 (a launch may cover a sub-range of the splits; C then points at its first slab)
    long slab_stride;     // elements between slabs
};

int hdm_launch_gemm(const HdmGemmArgs &args, hipStream_t stream);
// while a stream capture is recording the launches (chol.hip), the launcher must not record timing events
void hdm_gemm_capture_mode(int on);
void hdm_gemm_reserve_cus(int cus);   // CUs the persistent GEMM launches leave to concurrent kernels (collectives of a sharded build)
// per-role live timing with HIP events on the launch stream (off by default)
void hdm_timing_enable(int on);
void hdm_set_debug_buffer(unsigned long long *dev, int role);
int hdm_timing_collect(double *ms, double *flops, long *launches, double *issued = nullptr);  // arrays of HDM_NROLES; resets.  issued: flops the launches' MFMA instructions executed (gemm_f64.hip: issued_mfma_flops)

// One kernel handle per translation unit with device code.  The HIP runtime loads a translation unit's code object when its
// first kernel is launched (10 ms for the first HKKTBuildUp of a process, which is a fifth of a whole solve of a 100 x 100
// block); the engine asks for one function attribute per unit on a helper thread when its first context opens instead, beside
// whatever the caller does next (presolve, the other cones' creation).  HDSDP_MI355X_PRELOAD=0: load on first use.
const void *hdm_module_handle_gemm_f64();
const void *hdm_module_handle_gemm_persist();
const void *hdm_module_handle_chol();
const void *hdm_module_handle_schur();
const void *hdm_module_handle_lanczos();
const void *hdm_module_handle_lu();
const void *hdm_module_handle_small();
const void *hdm_module_handle_bsparse();
