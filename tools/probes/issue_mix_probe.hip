// diagnostic (not part of the library): what the instructions BETWEEN the MFMAs of the GEMM K loop cost the matrix pipe.
// A register-only loop of v_mfma_f64_16x16x4_f64 on 16 accumulators (the GEMM kernels' inner shape) sustains 0.98-0.99 of the
// fp64 peak; the real K loops sit at 92-94 % MFMA busy with one OR two workgroups per CU (tools/lone_wg.sh: half the grid loses
// only 6 %), and in the LDS-free body the waves wait for the matrix pipe 92 % of their cycles while the pipe is 91 % busy.  So the
// idle cycles are not waits a partner wave could fill.  This probe puts N instructions of ONE class between the 16 MFMAs of a
// group (everything is asm volatile: the order is the order written) and reports the MFMA rate:
//   V  v_add_u32            A  v_add_co/v_addc pair (64-bit pointer bump)     M  v_mul_lo_u32        F  v_fma_f64
//   S  s_add_u32            L  ds_read_b64        R  ds_read2_b64             W  ds_write_b128
//   G  global_load_dwordx4 (1 KiB per wave, L2-resident)                      B  s_barrier (N per 64 MFMAs)
//   C  v_cndmask_b32        U  v_lshl_add_u64     N  s_nop                    X  buffer_load_dwordx4 (offen, SGPR soffset)
//   K  the mix of a stage body without vector ALU work: 8 ds_read_b64 + 2 ds_write_b128 + 2 buffer loads + scalar adds per 16
//      MFMAs and a barrier per 64
// 1 and 2 workgroups of 256 threads per CU.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/issue_mix_probe.hip -o tools/probes/issue_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define MFMA(ACC, A, B) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

template <char CLS, int N>
__global__ __launch_bounds__(256, 2) void probe(double *out, const double *gbuf, int iters) {
    __shared__ __attribute__((aligned(16))) double lds[4096];
    d4 acc[16];
    const unsigned tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    for (int i = 0; i < 16; ++i) acc[i] = (d4){1e-3 * (gid & 15), 0.5, 0.25, 0.125};
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = 1.0 + ((gid + i) & 63) * 1e-9; b[i] = 1.0 - ((gid + 3 * i) & 31) * 1e-9; }
    for (int i = tid; i < 4096; i += 256) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    unsigned x0 = gid, x1 = gid * 3 + 1, y = 12345u | gid;
    unsigned s0 = blockIdx.x;
    double f0 = 1.0 + 1e-9 * (gid & 255), f1 = 0.999999, f2 = 1e-12;
    const unsigned laddr = (unsigned) (size_t) lds + (tid & 63) * 8 + (tid >> 6) * 2048;   // per-lane byte address, conflict-free
    const unsigned waddr = (unsigned) (size_t) lds + tid * 16;
    double r0 = 0.0; d2 r2 = (d2){0, 0}; u4 wdata = (u4){gid, gid + 1, gid + 2, gid + 3};
    const double *gp = gbuf + (size_t) (gid & 4095) * 2;       // 64 KiB window: L2 (and mostly L1) resident
    u4 g0 = (u4){0, 0, 0, 0};
    unsigned long long p64 = (unsigned long long) gp, inc64 = 16;
    const unsigned boff = (gid & 4095) * 16;
    u4 rsrc;   // raw buffer descriptor over gbuf: base, stride 0, 1 MiB, dword data format
    rsrc[0] = (unsigned) (size_t) gbuf; rsrc[1] = (unsigned) ((size_t) gbuf >> 32) & 0xffff; rsrc[2] = 1u << 20; rsrc[3] = 0x00020000u;
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (CLS == 'L' || CLS == 'R') asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (CLS == 'G' || CLS == 'X' || CLS == 'K') asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                MFMA(acc[m], a[m >> 2], b[m & 3]);
                // N instructions spread over the 16 slots: slot m takes instruction numbers m, m + 16, ...
#pragma unroll
                for (int q = m; q < N; q += 16) {
                    if (CLS == 'V') asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
                    if (CLS == 'A') asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(x0), "+v"(x1) : "v"(y) : "vcc");
                    if (CLS == 'M') asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
                    if (CLS == 'F') asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
                    if (CLS == 'S') asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc");
                    if (CLS == 'C') asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(y) : "vcc");
                    if (CLS == 'U') asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p64) : "s"(inc64));
                    if (CLS == 'N') asm volatile("s_nop 0");
                    if (CLS == 'X') asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g0) : "v"(boff), "s"(rsrc), "s"(s0 & 0xfff0) : "memory");
                    if (CLS == 'K') {   // the mix a VALU-free stage body would have per 16 MFMAs: 8 LDS reads, 2 LDS writes, 2 loads, 8 scalar ops
                        if (q < 8) asm volatile("ds_read_b64 %0, %1" : "=v"(r0) : "v"(laddr) : "memory");
                        else if (q < 10) asm volatile("ds_write_b128 %0, %1" :: "v"(waddr), "v"(wdata) : "memory");
                        else if (q < 12) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g0) : "v"(boff), "s"(rsrc), "s"(s0 & 0xfff0) : "memory");
                        else asm volatile("s_add_u32 %0, %0, 16" : "+s"(s0) :: "scc");
                    }
                    if (CLS == 'L') asm volatile("ds_read_b64 %0, %1" : "=v"(r0) : "v"(laddr) : "memory");
                    if (CLS == 'R') asm volatile("ds_read2_b64 %0, %1 offset1:32" : "=v"(r2) : "v"(laddr) : "memory");
                    if (CLS == 'W') asm volatile("ds_write_b128 %0, %1" :: "v"(waddr), "v"(wdata) : "memory");
                    if (CLS == 'G') asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g0) : "v"(gp) : "memory");
                }
            }
        }
        if (CLS == 'K') asm volatile("s_barrier" ::: "memory");
        if (CLS == 'B') {
#pragma unroll
            for (int q = 0; q < N; ++q) asm volatile("s_barrier" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += (double) p64 + x0 + x1 + s0 + f0 + r0 + r2[0] + r2[1] + g0[0] + g0[3];
    if (s == 12345.678) out[0] = s;
}

static double *g_out, *g_buf;
template <char CLS, int N>
static void run(int wg, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wg;
    hipLaunchKernelGGL((probe<CLS, N>), dim3(blocks), dim3(256), 0, 0, g_out, g_buf, iters / 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<CLS, N>), dim3(blocks), dim3(256), 0, 0, g_out, g_buf, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double) blocks * 4 * (double) iters * 64 * 2.0 * 16 * 16 * 4;
    const double tf = flops / (ms * 1e-3) / 1e12;
    // cycles of matrix pipe lost per inserted instruction (at 2.4 GHz nominal; per SIMD, per group of 16 MFMAs = 1024 cycles)
    printf("  %c x %2d per 16 MFMAs, %d WG/CU: %7.2f ms  %6.2f TFLOP/s  %.4f of 78.6\n", CLS, N, wg, ms, tf, tf / 78.6);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <char CLS>
static void sweep(int iters) {
    for (int wg = 1; wg <= 2; ++wg) {
        run<CLS, 0>(wg, iters);
        run<CLS, 4>(wg, iters);
        run<CLS, 8>(wg, iters);
        run<CLS, 16>(wg, iters);
        run<CLS, 32>(wg, iters);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;
    hipMalloc((void **) &g_out, 8);
    hipMalloc((void **) &g_buf, 1 << 20);
    hipMemset(g_buf, 0, 1 << 20);
    const char *which = argc > 2 ? argv[2] : "VAMFSLRWGBCUNXK";
    for (const char *c = which; *c; ++c) {
        printf("class %c\n", *c);
        switch (*c) {
            case 'V': sweep<'V'>(iters); break;
            case 'A': sweep<'A'>(iters); break;
            case 'M': sweep<'M'>(iters); break;
            case 'F': sweep<'F'>(iters); break;
            case 'S': sweep<'S'>(iters); break;
            case 'L': sweep<'L'>(iters); break;
            case 'R': sweep<'R'>(iters); break;
            case 'W': sweep<'W'>(iters); break;
            case 'G': sweep<'G'>(iters); break;
            case 'C': sweep<'C'>(iters); break;
            case 'U': sweep<'U'>(iters); break;
            case 'N': sweep<'N'>(iters); break;
            case 'X': sweep<'X'>(iters); break;
            case 'K': for (int wg = 1; wg <= 2; ++wg) { run<'K', 0>(wg, iters); run<'K', 16>(wg, iters); run<'K', 20>(wg, iters); } break;
            case 'B': for (int wg = 1; wg <= 2; ++wg) { run<'B', 0>(wg, iters); run<'B', 1>(wg, iters); run<'B', 2>(wg, iters); } break;
            default: break;
        }
    }
    return 0;
}
