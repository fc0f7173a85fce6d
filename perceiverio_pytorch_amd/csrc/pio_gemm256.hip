// NT GEMM, large-tile variant for the weight GEMMs of the latent stack (M = B*N_latents in the thousands):
//   256x256 output tile, 8 waves as 2(M) x 4(N), 128x64 per wave (8x4 MFMA 16x16x32 accumulators = 128 VGPRs),
//   K advanced in 32-deep tiles through a 4-slot LDS ring (4 x 32 KiB = 128 KiB, one workgroup per CU):
//     * both operands arrive by LDS-DMA (global_load_lds_dwordx4, 4 per wave per K tile); up to THREE K tiles are
//       in flight while one is multiplied: the wait is a COUNTED s_waitcnt vmcnt(8) and the barrier a raw
//       s_barrier, so the DMA of later tiles stays in flight across it (one barrier per K tile);
//     * 64-byte LDS rows, 16-byte chunk XOR swizzle f(row) = (-(row>>2)) & 3 applied on the DMA source address
//       and on the ds_read_b128 side: conflict-free fragment reads;
//     * per 32 MFMAs a wave issues 12 ds_read_b128 (0.375 per MFMA) and 4 DMA pieces;
//     * blockIdx is remapped so that the 8 XCDs each own a contiguous run of tiles (neighbouring tiles share
//       their A panel / the weight in that XCD's L2);
//     * epilogue in two 128-row halves through the (now idle) ring: whole 1-KiB rows per wave instruction.
//   Schedule variants (template VAR, env PIO_GEMM256_VAR for A/B runs in one process):
//     bit 0: fragments of tile kt+1 are read from LDS while the MFMAs of tile kt run (register double buffer)
//     bit 1: the 4 DMA pieces of the refill are issued between the four 8-MFMA groups instead of up front
// Same GemmParams / epilogue semantics as the 128x128 kernel (pio_gemm.hip).
#include "pio_gemm_common.h"

namespace pio {

static __device__ __attribute__((aligned(16))) uint32_t g_zero256[4] = {0, 0, 0, 0};

#ifdef PIO_GEMM_STAMPS
// Dev-only phase stamps (tools/gemm_stamps.py --build makes a private copy of the library with this flag; the shipped
// library never has it): wave 0 of tile 0 records s_memtime at the phase edges, g_epi_mode selects an epilogue
// ablation (1: no global stores, 2: registers -> global without the LDS pass, 3: no epilogue).
__device__ unsigned long long g_stamps[8];
__device__ int g_epi_mode;
#define PIO_STAMP(i)                                                                                  \
    do {                                                                                              \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define PIO_STAMP(i)
#endif

constexpr int L_BM = 256, L_BN = 256, L_BK = 32, L_STAGES = 4;
constexpr int L_OP = L_BM * L_BK * 2;  // 16 KiB per operand per stage
constexpr int L_STAGE = 2 * L_OP;      // 32 KiB

template <int N>
__device__ __forceinline__ void wait_vm_tiles(int tiles_in_flight) {  // 4 DMA instructions per tile per wave
    if (tiles_in_flight >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (tiles_in_flight == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (tiles_in_flight == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DT, int KIND, int VAR>
__global__ __launch_bounds__(512) void gemm_nt_256(const GemmParams p) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    constexpr bool PREFETCH = (VAR & 1) != 0;
    constexpr bool INTERLEAVE = (VAR & 2) != 0;
    constexpr int AHEAD = PREFETCH ? L_STAGES : L_STAGES - 1;  // refill distance (tiles)
    __shared__ __attribute__((aligned(16))) char smem[L_STAGES * L_STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    int bid = blockIdx.x;
    const int nt = gridDim.x;
    if ((nt & 7) == 0) bid = (bid & 7) * (nt >> 3) + (bid >> 3);  // XCD-contiguous tile runs (bijective)
    const int tile_n = bid % p.tiles_n;
    const int tile_m = bid / p.tiles_n;
    const int z = blockIdx.y;
    const int zb = z / p.nh, zh = z % p.nh;

    const T *A = (const T *)p.A + zb * p.sAb + zh * p.sAh;
    const T *B = (const T *)p.B + zb * p.sBb + zh * p.sBh;
    const T *zsrc = (const T *)g_zero256;

    // ---- staging: stage = A[256][32] | B[256][32]; a 1-KiB piece = 16 rows; wave w owns pieces 2w, 2w+1
    const T *a_src[2];
    const T *b_src[2];
    int s_koff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((-(r >> 2)) & 3);
        s_koff[i] = c * 8;
        int gm = tile_m * L_BM + r;
        gm = gm < p.M ? gm : p.M - 1;
        int gn = tile_n * L_BN + r;
        gn = gn < p.N ? gn : p.N - 1;
        a_src[i] = A + (int64_t)gm * p.lda + c * 8;
        b_src[i] = B + (int64_t)gn * p.ldb + c * 8;
    }
    const int nk1 = (p.K + L_BK - 1) / L_BK;
    const int nk = p.npass * nk1;

    auto stage_piece = [&](int kt, int i, bool isB) {
        const int pass = (kt >= nk1) + (kt >= 2 * nk1);
        const int k0 = (kt - pass * nk1) * L_BK;
        const int64_t d = pass == 0 ? 0 : (isB ? (pass == 1 ? p.dB1 : p.dB2) : (pass == 1 ? p.dA1 : p.dA2));
        char *base = smem + (kt & (L_STAGES - 1)) * L_STAGE + (isB ? L_OP : 0);
        const bool kin = (k0 + s_koff[i]) < p.K;
        const T *src = kin ? ((isB ? b_src[i] : a_src[i]) + k0 + d) : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(base + (wave * 2 + i) * 1024), 16,
                                         0, 0);
    };
    auto stage = [&](int kt) {
        stage_piece(kt, 0, false);
        stage_piece(kt, 0, true);
        stage_piece(kt, 1, false);
        stage_piece(kt, 1, true);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses: row = base16 + (lane&15), chunk = lane>>4, swizzle f = (-(row>>2)) & 3
    const int frow = lane & 15;
    const int fcoff = (((lane >> 4) ^ ((-(frow >> 2)) & 3)) << 4);
    const int a_off0 = (wr * 128 + frow) * 64 + fcoff;
    const int b_off0 = (wc * 64 + frow) * 64 + fcoff;

    auto load_frags = [&](int kt, V8 (&af)[8], V8 (&bf)[4]) {
        const char *abase = smem + (kt & (L_STAGES - 1)) * L_STAGE;
        const char *bbase = abase + L_OP;
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = *(const V8 *)(bbase + b_off0 + i * 16 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *(const V8 *)(abase + a_off0 + i * 16 * 64);
    };
    auto mma = [&](int kt, V8 (&afc)[8], V8 (&bfc)[4]) {
        const bool more = kt + AHEAD < nk;
        if (!INTERLEAVE && more) stage(kt + AHEAD);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int mi = 2 * g; mi < 2 * g + 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Op<DT>::mfma16(bfc[ni], afc[mi], acc[mi][ni]);
            if (INTERLEAVE && more) stage_piece(kt + AHEAD, g >> 1, (g & 1) != 0);
        }
    };

    PIO_STAMP(0);
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
        if (t < nk) stage(t);

    if constexpr ((VAR & 4) != 0) {
        // PING-PONG: the two waves that share a SIMD (wave w and w+4 = the two wave rows wr 0/1) run half a K tile
        // out of phase: while one row issues its LDS reads + DMA refill (LOAD) the other owns the matrix pipe (MFMA).
        // Both rows run the same instruction stream LOAD(0) MFMA(0) LOAD(1) MFMA(1) ... with ONE barrier per K tile
        // and wave, at different points: wr1 meets wr0 after its LOAD, wr0 after its MFMA phase -- so between two
        // barriers wr0 runs LOAD(t) MFMA(t) while wr1 runs MFMA(t-1) LOAD(t).  Tile t+1 is complete before barrier t
        // (wr0 waits for its pieces at the end of MFMA(t), wr1 at the end of LOAD(t)); the slot refilled in LOAD(t)
        // (tile t+3) held tile t-1, whose last reads (wr1's LOAD(t-1)) drained before barrier t-1.
        V8 af[8], bf[4];
        wait_vm_tiles<0>(nk - 1 < AHEAD - 1 ? nk - 1 : AHEAD - 1);
        __builtin_amdgcn_s_barrier();  // tile 0 visible to everyone
        PIO_STAMP(1);
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = nk - 2 - kt;
            const int infl = rem < 0 ? 0 : (rem > 2 ? 2 : rem);
            load_frags(kt, af, bf);
            if (kt + AHEAD < nk) stage(kt + AHEAD);
            if (wr == 1) wait_vm_tiles<0>(infl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wr == 1) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Op<DT>::mfma16(bf[ni], af[mi], acc[mi][ni]);
            __builtin_amdgcn_s_setprio(0);
            if (wr == 0) {
                wait_vm_tiles<0>(infl);
                __builtin_amdgcn_s_barrier();
            }
        }
    } else if constexpr (PREFETCH) {
        // tile kt+1's fragments are read while tile kt is multiplied; the refill goes into tile kt's own slot
        V8 af0[8], bf0[4], af1[8], bf1[4];
        wait_vm_tiles<0>(nk - 1 < AHEAD - 1 ? nk - 1 : AHEAD - 1);
        __builtin_amdgcn_s_barrier();
        load_frags(0, af0, bf0);
        auto iter = [&](int kt, V8 (&afc)[8], V8 (&bfc)[4], V8 (&afn)[8], V8 (&bfn)[4]) {
            const int rem = nk - 2 - kt;  // issued tiles after kt+1 (<= 2) may stay in flight
            wait_vm_tiles<0>(rem < 0 ? 0 : (rem > 2 ? 2 : rem));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile kt have RETURNED
            __builtin_amdgcn_s_barrier();  // tile kt+1 visible to all; tile kt's slot is free
            if (kt + 1 < nk) load_frags(kt + 1, afn, bfn);
            mma(kt, afc, bfc);
        };
        for (int kt = 0; kt < nk; kt += 2) {
            iter(kt, af0, bf0, af1, bf1);
            if (kt + 1 < nk) iter(kt + 1, af1, bf1, af0, bf0);
        }
    } else {
        V8 af[8], bf[4];
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = nk - 1 - kt;  // issued tiles after kt (<= 2) may stay in flight
            wait_vm_tiles<0>(rem > 2 ? 2 : rem);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // tile kt visible to all; tile kt-1's slot is free
            load_frags(kt, af, bf);
            mma(kt, af, bf);
        }
    }

    // ---- epilogue, two halves of 128 rows through LDS ([128][256] fp32 = 128 KiB, chunk ^= row & 15)
    const int64_t coffz = zb * p.sCb + zh * p.sCh;
    float *cs = (float *)smem;
    const int n0 = tile_n * L_BN + lane * 4;
    const bool ncol = n0 < p.n_store;
    const f32x4 bias_n = ncol ? load_bias4<DT>(p, n0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    // Barriers here order LDS accesses only (raw s_barrier behind lgkmcnt(0)): __syncthreads() would also drain
    // vmcnt, i.e. make every wave wait for the HBM stores (and residual loads) of the previous half to COMPLETE
    // before the next half may start -- measured with in-kernel stamps: 26k of a tile's 79k cycles.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (all operand DMA has long landed; keeps the ring quiescent)
    PIO_STAMP(2);
#ifdef PIO_GEMM_STAMPS
    const int epi_mode = g_epi_mode;
    if (epi_mode == 3) return;
    if (epi_mode == 2) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int nn = tile_n * L_BN + wc * 64 + ni * 16 + (lane >> 4) * 4;
            const f32x4 bb = nn < p.n_store ? load_bias4<DT>(p, nn) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = tile_m * L_BM + wr * 128 + mi * 16 + (lane & 15);
                if (m < p.M && nn < p.n_store) epilogue_store4<DT>(p, coffz, m, nn, acc[mi][ni], bb);
            }
        }
        PIO_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PIO_STAMP(4);
        return;
    }
#endif
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == h) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int ml = mi * 16 + (lane & 15);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int c = wc * 16 + ni * 4 + (lane >> 4);
                    *(f32x4 *)(cs + ml * L_BN + ((c ^ (ml & 15)) << 2)) = acc[mi][ni];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ncol) {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int ml = it * 8 + wave;
                const int m = tile_m * L_BM + h * 128 + ml;
                if (m < p.M) {
                    const f32x4 v = *(const f32x4 *)(cs + ml * L_BN + ((lane ^ (ml & 15)) << 2));
#ifdef PIO_GEMM_STAMPS
                    if (epi_mode == 1) {
                        if (v[0] == 1.2345e33f) epilogue_store4<DT>(p, coffz, m, n0, v, bias_n);
                        continue;
                    }
#endif
                    epilogue_store4<DT>(p, coffz, m, n0, v, bias_n);
                }
            }
        }
    }
    PIO_STAMP(3);
#ifdef PIO_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PIO_STAMP(4);
#endif
}

#ifdef PIO_GEMM_STAMPS
extern "C" int pio_debug_gemm_stamps(unsigned long long *out8, int epi_mode) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return 1;
    if (epi_mode >= 0 && hipMemcpyToSymbol(HIP_SYMBOL(g_epi_mode), &epi_mode, sizeof(int)) != hipSuccess) return 2;
    return 0;
}
#endif

void gemm256_launch(const GemmParams &p, int dtype, bool attn, int tiles_m, int tiles_n, int batch, hipStream_t s) {
    static const int var = [] {
        const char *e = getenv("PIO_GEMM256_VAR");
        return e ? atoi(e) : 4;  // default: ping-pong schedule (fastest in A/B runs)
    }();
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)batch, 1), block(512, 1, 1);
#define PIO_G256(DTV, KINDV)                                                                          \
    switch (var & 7) {                                                                                \
        case 4: hipLaunchKernelGGL((gemm_nt_256<DTV, KINDV, 4>), grid, block, 0, s, p); break;       \
        case 1: hipLaunchKernelGGL((gemm_nt_256<DTV, KINDV, 1>), grid, block, 0, s, p); break;       \
        case 2: hipLaunchKernelGGL((gemm_nt_256<DTV, KINDV, 2>), grid, block, 0, s, p); break;       \
        case 3: hipLaunchKernelGGL((gemm_nt_256<DTV, KINDV, 3>), grid, block, 0, s, p); break;       \
        default: hipLaunchKernelGGL((gemm_nt_256<DTV, KINDV, 0>), grid, block, 0, s, p); break;      \
    }
    if (dtype == PIO_DT_F16) {
        if (attn) { PIO_G256(PIO_DT_F16, 1) } else { PIO_G256(PIO_DT_F16, 0) }
    } else {
        if (attn) { PIO_G256(PIO_DT_BF16, 1) } else { PIO_G256(PIO_DT_BF16, 0) }
    }
#undef PIO_G256
}

}  // namespace pio
