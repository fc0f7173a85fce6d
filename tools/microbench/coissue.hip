// Dev microbenchmark (gfx950): can the matrix pipe and the VALU work at the same time -- from ONE wave, and from two
// waves of the same SIMD?  One workgroup per CU; each variant runs ITER rounds of (NM MFMAs 32x32x16 f16 on four
// independent accumulators) and / or (NV VALU operations: packed FMAs or v_exp) and reports cycles per round.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/coissue.hip -o tools/_abl/coissue && tools/_abl/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int ITER = 2000;

// KIND: 0 v_pk_fma_f32, 1 v_exp_f32, 2 v_fma_f32, 3 v_max3_f32, 4 v_cvt_pk_f16_f32, 5 v_pk_add_f32, 6 v_pk_mul_f32,
// 7 v_add_f32, 8 v_mov_b32, 9 v_dot2_f32_f16, 10 v_pk_add_f16, 11 v_fma_mix_f32
template <int KIND>
__device__ __forceinline__ void valu(f32x2 &x, const f32x2 &c1, const f32x2 &c2) {
    if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    else if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[0]));
    else if constexpr (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c1[0]), "v"(c2[0]));
    else if constexpr (KIND == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c1[0]), "v"(c2[0]));
    else if constexpr (KIND == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[0]) : "v"(c1[0]));
    else if constexpr (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(c2));
    else if constexpr (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
    else if constexpr (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(c2[0]));
    else if constexpr (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(x[0]) : "v"(c2[0]));
    else if constexpr (KIND == 9) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[0]) : "v"(c1[0]), "v"(c2[0]));
    else if constexpr (KIND == 10) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x[0]) : "v"(c2[0]));
    else if constexpr (KIND == 11) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x[0]) : "v"(c2[0]));
}

// MODE bit 0: this wave issues MFMAs; bit 1: it issues VALU work; KIND 0 = packed FMA, 1 = v_exp_f32
template <int NM, int NV, int KIND>
__device__ __forceinline__ void body(int mode, float *out, unsigned long long *cyc) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(threadIdx.x * 0.001f + i);
        b[i] = (_Float16)(threadIdx.x * 0.002f - i);
    }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 0.01f + i, threadIdx.x * 0.02f - i};
    const f32x2 c1 = {0.999f, 1.001f}, c2 = {1e-3f, -1e-3f};
    const bool do_m = mode & 1, do_v = mode & 2;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (do_m && do_v) {
#pragma unroll 1
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) {
                    const int r = (m * (NV / NM) + j) & 7;
                    valu<KIND>(v[r], c1, c2);
                }
            }
        }
    } else if (do_m) {
#pragma unroll 1
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
        }
    } else if (do_v) {
#pragma unroll 1
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                valu<KIND>(v[j & 7], c1, c2);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][7];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// waves 0..3 (one per SIMD) get mode_a, waves 4..7 (the second wave of each SIMD) mode_b
template <int NM, int NV, int KIND>
__global__ __launch_bounds__(512) void k(int mode_a, int mode_b, float *out, unsigned long long *cyc) {
    const int wave = threadIdx.x >> 6;
    body<NM, NV, KIND>(wave < 4 ? mode_a : mode_b, out, cyc);
}

template <int NM, int NV, int KIND>
static void run(const char *what, int threads, int mode_a, int mode_b, float *out, unsigned long long *cyc) {
    hipLaunchKernelGGL((k<NM, NV, KIND>), dim3(256), dim3(threads), 0, 0, mode_a, mode_b, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, KIND>), dim3(256), dim3(threads), 0, 0, mode_a, mode_b, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-86s: %8.1f us | cycles per round: wave0 %6.0f", what, ms * 1e3, (double)h[0] / ITER);
    if (threads > 256) printf(", wave4 %6.0f", (double)h[4] / ITER);
    printf("\n");
}

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8 * 8);
    printf("one round = 16 MFMA 32x32x16 f16 (16 x 32 = 512 matrix-pipe cycles) and/or NV VALU operations\n");
    run<16, 64, 0>("1 wave/SIMD: 16 MFMA only", 256, 1, 0, out, cyc);
#define TRIO(KIND, NAME)                                                                                          \
    run<16, 64, KIND>("1 wave/SIMD: 64 " NAME " only", 256, 2, 0, out, cyc);                                       \
    run<16, 64, KIND>("1 wave/SIMD: 16 MFMA interleaved with 64 " NAME " (4 behind each MFMA)", 256, 3, 0, out, cyc); \
    run<16, 64, KIND>("2 waves/SIMD: wave A 16 MFMA, wave B 64 " NAME, 512, 1, 2, out, cyc);
    TRIO(0, "v_pk_fma_f32")
    TRIO(1, "v_exp_f32")
    TRIO(2, "v_fma_f32")
    TRIO(3, "v_max3_f32")
    TRIO(4, "v_cvt_pk_f16_f32")
    TRIO(5, "v_pk_add_f32")
    TRIO(6, "v_pk_mul_f32")
    TRIO(7, "v_add_f32")
    TRIO(8, "v_mov_b32")
    TRIO(9, "v_dot2_f32_f16")
    TRIO(10, "v_pk_add_f16")
    TRIO(11, "v_fma_mix_f32")
    run<16, 64, 0>("2 waves/SIMD: both 16 MFMA only", 512, 1, 1, out, cyc);
    run<16, 64, 2>("2 waves/SIMD: both interleave 16 MFMA with 64 v_fma_f32", 512, 3, 3, out, cyc);
    run<16, 64, 2>("2 waves/SIMD: both 64 v_fma_f32 only", 512, 2, 2, out, cyc);
    return 0;
}
