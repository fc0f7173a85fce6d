// Dev microbenchmark (gfx950): what does an LDS-DMA piece cost the wave that issues it between its MFMAs?  One workgroup
// of 4 waves per CU (one wave per SIMD, as in gemm_nt_wide); a round = 64 MFMAs 16x16x32 f16 (1024 matrix-pipe cycles)
// with NP pieces of 1 KiB (64 lanes x 16 B) issued one behind every 64/NP-th MFMA:
//   global_load_lds_dwordx4 with an SGPR base + 32-bit lane offset (what the kernels use), the same with a 64-bit
//   per-lane address, and buffer_load_dwordx4 ... lds (raw buffer resource + lane offset).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/dma_issue.hip -o tools/_abl/dma_issue && tools/_abl/dma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int ITER = 1000;

template <int NP, int MODE, bool BIG = false>
__global__ __launch_bounds__(256) void k(const char *src, float *out, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(lane * 0.001f + i);
        b[i] = (_Float16)(lane * 0.002f - i);
    }
    f32x4 acc[16];
    for (int k2 = 0; k2 < 16; ++k2) acc[k2] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 accb[4];
    for (int k2 = 0; k2 < 4; ++k2)
        for (int e = 0; e < 16; ++e) accb[k2][e] = 0.f;
    // per-lane source offset: 16 rows x 64 B of a 2 KiB-pitch matrix (the GEMM pattern)
    uint32_t off = (uint32_t)((lane >> 2) * 2048 + (lane & 3) * 16);
    const char *base = src + (blockIdx.x & 63) * 65536 + wave * 16384;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
    char *dst = lds + wave * 16384;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            if (BIG) {
                if ((m & 1) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(accb[(m >> 1) & 3]) : "v"(a), "v"(b));
            } else {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 15]) : "v"(a), "v"(b));
            }
            const bool here = MODE == 3 ? (m >= 8 && m < 8 + NP) : (MODE == 4 ? (m >= 8 && m < 8 + 2 * NP && (m & 1) == 0)
                                                                                 : (m % (64 / (NP > 0 ? NP : 1))) == 0);
            if (NP > 0 && here) {
                const int pc = MODE == 3 ? m - 8 : (MODE == 4 ? (m - 8) / 2 : m / (64 / (NP > 0 ? NP : 1)));
                uint32_t o = off + (uint32_t)((pc & 3) * 64);
                asm volatile("" : "+v"(o));
                if (MODE == 0 || MODE == 3 || MODE == 4) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (uint64_t)o),
                                                     (__attribute__((address_space(3))) void *)(dst + (pc & 7) * 1024), 16, 0, 0);
                } else if (MODE == 1) {
                    const char *pl = base + (uint64_t)o + (uint64_t)(lane & 1);  // (defeats the SGPR-base form)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pl - (lane & 1)),
                                                     (__attribute__((address_space(3))) void *)(dst + (pc & 7) * 1024), 16, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst + (pc & 7) * 1024),
                                                         16, (int)o, 0, 0, 0);
                }
            }
        }
        if (NP > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    float s = 0.f;
    for (int k2 = 0; k2 < 16; ++k2) s += acc[k2][0] + acc[k2][3];
    for (int k2 = 0; k2 < 4; ++k2) s += accb[k2][0] + accb[k2][15];
    s += (float)lds[threadIdx.x * 16];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NP, int MODE, bool BIG = false>
static void run(const char *what, const char *src, float *out, unsigned long long *cyc) {
    hipLaunchKernelGGL((k<NP, MODE, BIG>), dim3(256), dim3(256), 0, 0, src, out, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NP, MODE, BIG>), dim3(256), dim3(256), 0, 0, src, out, cyc);
    hipDeviceSynchronize();
    unsigned long long h[4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-72s: %7.0f cycles per round of 64 MFMAs (1024 matrix-pipe cycles)\n", what, (double)h[0] / ITER);
}

int main() {
    char *src;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&src, 8 << 20);
    hipMemset(src, 0, 8 << 20);
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 4 * 8);
    run<0, 0>("MFMAs only", src, out, cyc);
    run<8, 0>("+ 8 global_load_lds_dwordx4, SGPR base + lane offset", src, out, cyc);
    run<8, 1>("+ 8 global_load_lds_dwordx4, 64-bit lane addresses", src, out, cyc);
    run<8, 2>("+ 8 buffer_load_dwordx4 ... lds", src, out, cyc);
    run<16, 0>("+ 16 global_load_lds_dwordx4, SGPR base + lane offset", src, out, cyc);
    run<16, 2>("+ 16 buffer_load_dwordx4 ... lds", src, out, cyc);
    run<4, 0>("+ 4 global_load_lds_dwordx4, SGPR base + lane offset", src, out, cyc);
    run<8, 3>("+ 8 global_load_lds_dwordx4 behind 8 consecutive MFMAs", src, out, cyc);
    run<8, 4>("+ 8 global_load_lds_dwordx4 behind every other of 16 consecutive MFMAs", src, out, cyc);
    printf("the same round as 32 MFMAs 32x32x16 (32 cycles each):\n");
    run<0, 0, true>("MFMAs only", src, out, cyc);
    run<4, 0, true>("+ 4 global_load_lds_dwordx4", src, out, cyc);
    run<8, 0, true>("+ 8 global_load_lds_dwordx4", src, out, cyc);
    run<16, 0, true>("+ 16 global_load_lds_dwordx4", src, out, cyc);
    run<8, 3, true>("+ 8 global_load_lds_dwordx4 behind 8 consecutive slots", src, out, cyc);
    return 0;
}
