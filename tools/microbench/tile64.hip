// Dev microbenchmark (gfx950): the inner structure a fused fc1 -> GELU -> fc2 kernel would have -- 64 rows per CU, one
// wave per SIMD owning 64 x 64 of a 64 x 256 step tile: per step of K = 32 a wave runs 16 MFMAs 16x16x32 (256
// matrix-pipe cycles), reads 8 fragments and issues its share of the step three ahead through a four-stage LDS-DMA
// ring: A 64 rows x 64 B (4 pieces, one per wave) + B 256 rows x 64 B (16 pieces, four per wave) = 20 KiB per step,
// i.e. 80 B/clk against the CU's 64 B/clk vector-memory path; the fc2 half streams B only (16 KiB = 64 B/clk).
// Timing only (operands are whatever the buffers hold); prints cycles per step against the 256 of the MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/tile64.hip -o tools/_abl/tile64 && tools/_abl/tile64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int STEPS = 2048;       // K = 65536 worth of steps
constexpr int STAGE = 20 * 1024;  // A 4 KiB | B 16 KiB
constexpr int NST = 4;

__device__ __forceinline__ void dma16(const void *src, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// WITH_A: steps fetch A too (fc1 half); otherwise B only (fc2 half: A would come from the hidden block in LDS)
template <bool WITH_A, int NPIECE>
__global__ __launch_bounds__(256) void k(const char *A, const char *B, float *out, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + 4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int prow = lane >> 2;
    const int qsrc = (lane & 3) ^ ((0x78 >> (2 * ((prow >> 2) & 3))) & 3);
    // lane offsets of a piece (16 rows x 64 B of a K-contiguous matrix with 2 KiB rows)
    uint32_t oa = (uint32_t)((wave * 16 + prow) * 2048 + qsrc * 16);
    uint32_t ob[4];
    for (int i = 0; i < 4; ++i) ob[i] = (uint32_t)(((wave * 4 + i) * 16 + prow) * 2048 + qsrc * 16);
    const char *a0 = A + (size_t)(blockIdx.x & 63) * 64 * 2048;
    const char *b0 = B;
    const int fr = lane & 15, fq = lane >> 4;
    auto fsw = [](int g) { return (0x78 >> (2 * (g & 3))) & 3; };
    const int fa = fr * 64 + ((fq ^ fsw(fr >> 2)) << 4);
    f16x8 af[2][4], bf[2][4];
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    char *sink = smem + NST * STAGE + wave * 1024;
    auto issue = [&](int s) {  // this wave's pieces of step s into stage s % NST
        char *st = smem + (s & (NST - 1)) * STAGE;
        const int koff = (s & 31) * 64;  // bytes along K (wraps inside a 2 KiB row: the operands stay in L2)
        uint32_t o = oa;
        asm volatile("" : "+v"(o));
        if (WITH_A) dma16(a0 + koff + (uint64_t)o, st + wave * 1024);
        else if (NPIECE == 5) dma16(a0 + (uint64_t)o, sink);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t o2 = ob[i];
            asm volatile("" : "+v"(o2));
            dma16(b0 + koff + (uint64_t)o2, st + 4096 + (wave * 4 + i) * 1024);
        }
    };
    for (int s = 0; s < NST - 1; ++s) issue(s);
    if (NPIECE == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < 4; ++i) {
        af[0][i] = *(const f16x8 *)(smem + i * 1024 + fa);
        bf[0][i] = *(const f16x8 *)(smem + 4096 + (wave * 4 + i) * 1024 + fa);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
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
                // meeting point: slice s + 1 has landed everywhere, slice s - 1's stage is free
                if (NPIECE == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
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
    for (int s = 0; s < STEPS; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        step(std::integral_constant<int, 1>{}, s + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <bool WITH_A, int NPIECE>
static void run(const char *what, const char *A, const char *B, float *out, unsigned long long *cyc) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<WITH_A, NPIECE>), dim3(256), dim3(256), 0, 0, A, B, out, cyc);
        (void)hipDeviceSynchronize();
    }
    unsigned long long h[4];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-64s: %6.0f cycles per step (16 MFMAs = 256 matrix-pipe cycles): %4.1f %%\n", what, (double)h[0] / STEPS,
           100.0 * 256.0 * STEPS / (double)h[0]);
}

int main() {
    char *A, *B;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&A, 64 * 64 * 2048 + (1 << 20));
    (void)hipMalloc(&B, 256 * 2048 + (1 << 20));
    (void)hipMemset(A, 0, 64 * 64 * 2048);
    (void)hipMemset(B, 0, 256 * 2048);
    (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMalloc(&cyc, 256 * 4 * 8);
    run<true, 5>("fc1 half: A (4 KiB) + B (16 KiB) per step, 5 pieces per wave", A, B, out, cyc);
    run<false, 4>("fc2 half: B (16 KiB) per step, 4 pieces per wave", A, B, out, cyc);
    return 0;
}
