// L2 -> LDS operand-stream microbenchmark (dev tool): how fast can all CUs pull GEMM-shaped operand pieces into
// LDS with global_load_lds_dwordx4, and what does it depend on?  No consumers: the pieces land in a ring and are
// overwritten.  Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/dma_bench.hip -o /tmp/dma_bench &&
// /tmp/dma_bench
//
// Patterns (source addresses of the 1-KiB pieces one wave issues per step):
//   0  GEMM-like, the streaming kernel's tile walk: A = 256 rows x 128 B (row stride ld), B = 128 rows, 8 tile_n
//      neighbours share an A panel and every tile of an XCD run shares B panels      (shared lines, strided rows)
//   1  the same piece shape (8 rows x 128 B, stride ld) but every workgroup has its OWN rows (no sharing)
// Footprints are sized to stay L2 / Infinity-Cache resident across the timed repeats.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int RING = 144 * 1024;
static size_t g_bytes = 512ull << 20;  // bytes of LDS used as the landing ring

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PIECES = pieces per wave per step, DEPTH = steps kept in flight (per wave: PIECES * DEPTH loads outstanding)
// DRY: record the lowest / highest source address instead of loading (host checks them against the buffer first)
template <int PIECES, int DEPTH, bool DRY>
__global__ __launch_bounds__(1024) void dma_stream(const char *src, int pattern, int steps, long ld, int tiles_n,
                                                  long region, unsigned long long *range) {
    __shared__ __attribute__((aligned(16))) char smem[RING];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int bx = blockIdx.x, G = gridDim.x;
    // the streaming GEMM's tile assignment: XCD x owns a contiguous run of tile ids
    const int run = (G + 7) >> 3;
    const int tile = (bx & 7) * run + (bx >> 3);
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const char *base[PIECES];
    unsigned voff[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int piece = wave * PIECES + i;  // piece id inside the stage
        const int row8 = piece * 8 + (lane >> 3);
        if (pattern == 0) {
            // first 2/3 of a stage's pieces are A rows (tile tm), the last third B rows (tile tn)
            const int na = (nw * PIECES * 2) / 3;
            const long row = piece < na ? (long)tm * 256 + row8 : (long)(tiles_n * 0 + 16384) + (long)tn * 128 + (row8 - na * 8);
            base[i] = src;
            voff[i] = (unsigned)(row * ld + (lane & 7) * 16);
        } else {
            const long row = (long)bx * (nw * PIECES * 8) + row8;
            base[i] = src;
            voff[i] = (unsigned)(row * ld + (lane & 7) * 16);
        }
    }
    const int stage_bytes = nw * PIECES * 1024;
    const int nslots = RING / stage_bytes;
    int slot = 0;
    long koff = 0;
    for (int s = 0; s < steps; ++s) {
        char *sb = smem + slot * stage_bytes;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const char *p = base[i] + koff + (unsigned long)voff[i];
            if constexpr (DRY) {
                atomicMin(&range[0], (unsigned long long)p);
                atomicMax(&range[1], (unsigned long long)p + 16);
                continue;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)(sb + (wave * PIECES + i) * 1024),
                                             16, 0, 0);
        }
        if constexpr (!DRY) wait_vm<PIECES *(DEPTH - 1)>();
        slot = slot + 1 == nslots ? 0 : slot + 1;
        // advance along K: 128 B per step inside a row window
        koff = (koff + 128) % (pattern == 1 ? 256 : 2048);  // pattern 1: a 256-B window keeps all rows L2 resident
    }
    wait_vm<0>();
}

template <int PIECES, int DEPTH>
static void run(const char *name, const char *buf, int pattern, int waves, int steps, long ld, int tiles_n, long region) {
    const int G = 256;
    if (region <= 0) region = 1;
    // address check first: a stray DMA would fault the GPU
    static unsigned long long *range = nullptr;
    if (!range) CK(hipMalloc(&range, 16));
    unsigned long long init[2] = {~0ull, 0ull}, got[2];
    CK(hipMemcpy(range, init, 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((dma_stream<PIECES, DEPTH, true>), dim3(G), dim3(waves * 64), 0, 0, buf, pattern, 64, ld, tiles_n, region, range);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got, range, 16, hipMemcpyDeviceToHost));
    if (got[0] < (unsigned long long)buf || got[1] > (unsigned long long)buf + g_bytes) {
        printf("%-34s SKIPPED: addresses [%llx, %llx) outside the buffer [%llx, %llx)\n", name, got[0], got[1],
               (unsigned long long)buf, (unsigned long long)buf + g_bytes);
        fflush(stdout);
        return;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((dma_stream<PIECES, DEPTH, false>), dim3(G), dim3(waves * 64), 0, 0, buf, pattern, steps, ld, tiles_n, region, range);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((dma_stream<PIECES, DEPTH, false>), dim3(G), dim3(waves * 64), 0, 0, buf, pattern, steps, ld, tiles_n, region, range);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)G * waves * PIECES * 1024.0 * steps * reps;
    printf("%-34s waves=%2d pieces/wave/step=%d depth=%d : %7.2f TB/s  (%.1f GB/s per CU, %.1f us per launch)\n", name,
           waves, PIECES, DEPTH, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / G, ms * 1e3 / reps);
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main() {
    const size_t bytes = g_bytes;
    char *buf;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 1, bytes));
    const long ld = 2048;  // K = 1024 fp16
    const int steps = 512;
    // GEMM-like sharing, as the streaming kernel issues it: 8 waves x 6 pieces, 2 steps in flight
    run<6, 2>("0 gemm-like (shared, strided)", buf, 0, 8, steps, ld, 8, 0);
    run<6, 3>("0 gemm-like (shared, strided)", buf, 0, 8, steps, ld, 8, 0);
    run<4, 3>("0 gemm-like (shared, strided)", buf, 0, 8, steps, ld, 8, 0);
    run<4, 4>("0 gemm-like (shared, strided)", buf, 0, 8, steps, ld, 8, 0);
    run<2, 8>("0 gemm-like (shared, strided)", buf, 0, 8, steps, ld, 8, 0);
    run<3, 4>("0 gemm-like (shared, strided)", buf, 0, 16, steps, ld, 8, 0);
    // private rows, same piece shape
    run<6, 2>("1 private rows (strided)", buf, 1, 8, steps, ld, 8, 0);
    run<4, 4>("1 private rows (strided)", buf, 1, 8, steps, ld, 8, 0);
    run<3, 4>("1 private rows (strided)", buf, 1, 16, steps, ld, 8, 0);
    CK(hipFree(buf));
    return 0;
}
