// Dev kernels for tools/r4_ceiling.py (gfx950), built into tools/_abl/libenergy.so:
//   ek_spin   -- every CU holds one wave per SIMD in a scalar loop for a given number of shader cycles: the chip is "active
//                at clock" with no vector, matrix or memory work -- the power floor a busy kernel's joules sit on
//   ek_mlp64  -- the main loop a FUSED fc1 -> GELU -> fc2 kernel would have (64 rows per CU, the 64 x 1024 fp32 result in
//                the accumulator file, hidden activations chunk by chunk through LDS), as a timing / energy SKELETON on the
//                caller's (random) operands: fc1 half = 4 hidden chunks x 32 steps of K = 32, each step A (64 rows x 64 B)
//                + W1 (256 rows x 64 B) through a four-stage LDS-DMA ring and 16 MFMAs 16x16x32 per wave; fc2 half =
//                4 x 4 x 8 steps fetching W2 only (its A operand, the hidden chunk, would come from LDS: the fragment reads
//                are issued all the same).  GELU and the residual epilogue are NOT in it (they add time and energy on top):
//                it is a lower bound of what the fusion costs, for the comparison with the separate fc1 + fc2 launches.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/microbench/energy_kernels.hip -o tools/_abl/libenergy.so
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void spin_kernel(unsigned long long cycles, int busy, unsigned long long *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while (t - t0 < cycles) {
        if (!busy) __builtin_amdgcn_s_sleep(1);   // (busy: the scalar unit polls the clock without a pause)
        t = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t - t0;
}

constexpr int STAGE = 20 * 1024;  // A 4 KiB | B 16 KiB
constexpr int NST = 4;

__device__ __forceinline__ void dma16(const void *src, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// X [16384][1024] fp16 rows (this workgroup: rows 64 * blockIdx.x ..), W1 / W2 [1024][1024] fp16 (K contiguous)
__global__ __launch_bounds__(256) void mlp64_kernel(const char *X, const char *W1, const char *W2, float *out) {
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + 4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int prow = lane >> 2;
    const int qsrc = (lane & 3) ^ ((0x78 >> (2 * ((prow >> 2) & 3))) & 3);
    const uint32_t oa = (uint32_t)((wave * 16 + prow) * 2048 + qsrc * 16);
    uint32_t ob[4];
    for (int i = 0; i < 4; ++i) ob[i] = (uint32_t)(((wave * 4 + i) * 16 + prow) * 2048 + qsrc * 16);
    const char *a0 = X + (size_t)blockIdx.x * 64 * 2048;
    const int fr = lane & 15, fq = lane >> 4;
    auto fsw = [](int g) { return (0x78 >> (2 * (g & 3))) & 3; };
    const int fa = fr * 64 + ((fq ^ fsw(fr >> 2)) << 4);
    f16x8 af[2][4], bf[2][4];
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    char *sink = smem + NST * STAGE + wave * 1024;
    // step s of 256: s < 128 = fc1 half (hidden chunk s / 32, K slice s % 32: A + W1), else fc2 half (W2 only)
    auto issue = [&](int s) {
        char *st = smem + (s & (NST - 1)) * STAGE;
        const bool fc1 = s < 128;
        const int koff = (s & 31) * 64;
        const char *b0 = fc1 ? W1 + (size_t)(s >> 5) * 256 * 2048 : W2 + (size_t)(((s - 128) >> 3) & 3) * 256 * 2048;
        const int kb = fc1 ? koff : ((s - 128) >> 5) * 512 + (s & 7) * 64;
        uint32_t o = oa;
        asm volatile("" : "+v"(o));
        if (s < 256 && fc1) dma16(a0 + koff + (uint64_t)o, st + wave * 1024);
        else dma16(a0 + (uint64_t)o, sink);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t o2 = ob[i];
            asm volatile("" : "+v"(o2));
            if (s < 256) dma16(b0 + kb + (uint64_t)o2, st + 4096 + (wave * 4 + i) * 1024);
            else dma16(a0 + (uint64_t)o, sink);
        }
    };
    for (int s = 0; s < NST - 1; ++s) issue(s);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < 4; ++i) {
        af[0][i] = *(const f16x8 *)(smem + i * 1024 + fa);
        bf[0][i] = *(const f16x8 *)(smem + 4096 + (wave * 4 + i) * 1024 + fa);
    }
    auto step = [&](auto PC, int s) {
        constexpr int P = decltype(PC)::value;
        const char *nx = smem + ((s + 1) & (NST - 1)) * STAGE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[g][t]) : "v"(bf[P][t]), "v"(af[P][g]));
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0) {
                asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue(s + NST - 1);
            }
            if (g == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[P ^ 1][i] = *(const f16x8 *)(nx + i * 1024 + fa);
            }
            if (g == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bf[P ^ 1][i] = *(const f16x8 *)(nx + 4096 + (wave * 4 + i) * 1024 + fa);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll 1
    for (int s = 0; s < 256; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        step(std::integral_constant<int, 1>{}, s + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float sum = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

extern "C" int ek_spin(void *stream, unsigned long long cycles, int busy, unsigned long long *out) {
    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, cycles, busy, out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
extern "C" int ek_mlp64(void *stream, const void *X, const void *W1, const void *W2, float *out) {
    hipLaunchKernelGGL(mlp64_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const char *)X, (const char *)W1,
                       (const char *)W2, out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
