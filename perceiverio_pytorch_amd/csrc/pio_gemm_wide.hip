// gemm_nt_wide: C[M,N] = epi(alpha * A[M,K] B[N,K]^T + bias) -- persistent 256x256 tiles, FOUR waves (one per SIMD),
// each wave a 128x128 quarter with its 256 accumulator registers in AGPRs.  Epilogues: 16-bit out (optionally
// erf-GELU), fp32 out (optionally + fp32 residual), and the two halves of the LayerNorm fold (template LNF): the
// PRODUCER adds the residual given as a 16-bit pair, leaves its result as a 16-bit pair (+ fp32 if asked) together with
// per-row partial sums; the CONSUMER turns those sums into mean / rstd and finishes LayerNorm(x) W^T + b from the
// un-normalised 16-bit x (pio_gemm_t, pio_ln_fold_t).  It carries every weight GEMM of the latent self-attend stack.
//
// Why a third GEMM: the streaming kernel (pio_gemm_stream.hip, 256x128 tiles, eight waves in two teams) spends a
// step of 1024 MFMA-cycles in ~2000 cycles: each wave alternates a load phase (16 LDS reads + 6 DMA pieces, bound
// by the CU's 64 B/clk vector-memory path: 48 KiB per step = 768 cycles at full rate) and an MFMA phase, and the two
// phases of the two teams only overlap pairwise.  Here a wave never leaves its MFMA stream: per phase (one K slice
// of 32: 64 MFMAs = 1024 cycles) it issues, BETWEEN its MFMAs, the 16 LDS reads of the next slice's fragments into
// a second fragment register set and 8 DMA pieces of the slice four ahead -- 64 KiB of operands per 2048
// MFMA-cycles, i.e. half the vector-memory rate and two thirds of the LDS-read bytes of the streaming kernel per
// FLOP.  One barrier per phase, behind the phase's first eight MFMAs.  The price: with one accumulator set the tile's
// result leaves in an exposed epilogue (no second set to drip from).  That epilogue is bound by its memory traffic,
// not by its arithmetic (GELU or the fold's extra FMAs hide behind the stores); for the 16-bit outputs the caches
// absorb it at ~7 TB/s, for the residual GEMMs (64 MB in, 64 MB out per launch) it is the larger half of the kernel.
//
// LDS: ring of 4 slices x (A 256 rows x 64 B | B 256 rows x 64 B) = 128 KiB; a row's four 16-B chunks are stored at
// position chunk ^ f((row >> 2) & 3), f = {0, 2, 3, 1}: conflict-free for ds_read_b128's lane groups
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md, LDS table) both for fragment rows lane & 15 (A)
// and for the permuted B rows below.  Filled by global_load_lds_dwordx4 pieces of 16 rows x 64 B whose per-lane
// SOURCE address carries the swizzle (the DMA writes LDS linearly).
//
// B rows are permuted inside each 32-column group so that a lane's two units (ni even / odd) hold 8 CONSECUTIVE
// output columns: MFMA column index c (0..15) of unit ni is column 32 * (ni / 2) + 8 * (c / 4) + 4 * (ni & 1) + c % 4,
// so a lane (row lane & 15, index group lane >> 4) stores 16 bytes per unit pair instead of two 8-byte pieces.
//
// Pipeline, per wave, phase g (K slice g of the flattened slice sequence over this workgroup's tiles):
//   MFMAs on fragment set g & 1 | LDS reads of slice g+1 into set (g+1) & 1 | DMA of slice g+4 into ring slot g & 3
//   (slice g left that slot for the registers during phase g-1) | s_waitcnt vmcnt(16): all but this and the
//   previous phase's pieces have landed, i.e. slice g+2 is complete | lgkmcnt(0) | barrier.
// Every phase issues exactly 8 pieces (past the end of the work they go to a sink region), and the epilogue always
// issues its 32 stores (out-of-range ones to a global sink), so the counted waits are exact.
//
// Reference semantics: y = x W^T + b of nn.Linear (perceiver_io/transformer_primitives.py:93-95, 110).
#include <type_traits>

#include "pio_gemm_common.h"

namespace pio {

constexpr int W_BM = 256, W_BN = 256, W_BK = 32, W_NST = 4;
constexpr int W_AB = W_BM * W_BK * 2;    // 16 KiB of A per slice
constexpr int W_STAGE = 2 * W_AB;        // 32 KiB
constexpr int W_RING = W_NST * W_STAGE;  // 128 KiB
constexpr int W_BIAS = W_RING + 4 * 1024;  // after the per-wave sinks for the pieces past the end: per-wave bias rows
constexpr int W_SMEM = W_BIAS + 4 * 1024;  // per wave: bias row (512 B) + the LayerNorm fold's c row (512 B)

static __device__ __attribute__((aligned(16))) uint32_t g_sink_w[64 * 4];
static __device__ __attribute__((aligned(16))) uint32_t g_zero_w[4] = {0, 0, 0, 0};

#ifdef PIO_GEMM_STAMPS
// Dev-only (tools/gemm_stamps.py --wide): wave 0 of workgroup 0 records s_memtime inside its ordinary phases (the
// last one executed survives) and, around the kernel, s_memtime / s_memrealtime (the clock held under load).
__device__ unsigned long long g_wstamps[8];
__device__ unsigned long long g_wclk[4];
__device__ unsigned long long g_westamps[16];  // staged producer epilogue: start, after the barrier, after each row block
#define PIO_WESTAMP(i)                                                                          \
    do {                                                                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_westamps[i] = __builtin_readcyclecounter();  \
    } while (0)
#ifndef PIO_WIDE_ABL
#define PIO_WIDE_ABL 0  // timing-only ablations, compile time: bit 0 = no LDS fragment reads, 1 = no DMA pieces, 2 = no barrier
#endif
#ifdef PIO_WIDE_PHASE_STAMPS  // (they cost ~76 cycles each and pull the stamped wave out of step with the others)
#define PIO_WSTAMP(i)                                                            \
    do {                                                                         \
        if (!FIRST && P == 0 && blockIdx.x == 0 && threadIdx.x == 0)             \
            g_wstamps[i] = __builtin_readcyclecounter();                         \
    } while (0)
#else
#define PIO_WSTAMP(i)
#endif
#else
#define PIO_WSTAMP(i)
#define PIO_WESTAMP(i)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void wide_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wide_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void wide_dma16(const void *src, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// Reads one accumulator element where it lives (an AGPR): without this the register allocator moves all 256 to VGPRs
// at the end of the K loop and a handful of address registers spill.
__device__ __forceinline__ float acc_read(float a) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

// The MFMA as inline assembly with its accumulator pinned to AGPRs ("+a"), accumulating in place: left to the
// register allocator, the 256 accumulator registers wander between the two files and a few of them spill.  (The
// compiler's hazard recognizer does not look inside: the only dependent reader of an accumulator within fewer than
// 64 MFMAs is the epilogue, which starts behind the last phase's waits and barrier.)
template <int DT>
__device__ __forceinline__ void mfma_acc(f32x4 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {
    if constexpr (DT == PIO_DT_F16) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int DT>
__device__ __forceinline__ void mfma_first(f32x4 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {
    if constexpr (DT == PIO_DT_F16) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

// The same with the 32x32x16 MFMA (MF == 1 below): an MFMA twice as long hides the issue stall of an LDS-DMA piece twice
// as well -- tools/microbench/dma_issue.hip: 8 pieces beside 64 MFMAs 16x16x32 cost 312 cycles per 1024, beside 32
// MFMAs 32x32x16 133.
template <int DT>
__device__ __forceinline__ void mfma_acc32(f32x16 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {
    if constexpr (DT == PIO_DT_F16) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int DT>
__device__ __forceinline__ void mfma_first32(f32x16 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {
    if constexpr (DT == PIO_DT_F16) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

// OUT: 0 = 16-bit C, 2 = fp32 C; RES: 0 = no residual, 1 = fp32 residual R, 2 = residual as a 16-bit pair
// R16_hi + R16_lo (LNF == 1 only; the result then leaves as the pair X16 / X16_lo, and as fp32 only if C is given).
// LNF, the LayerNorm fold (pio_gemm_t): 1 = producer (OUT == 2): also stores a 16-bit copy of the result and per-row
// partial (sum, sum of squares) per 128-column block; 2 = consumer (OUT == 0, K == 1024): the result is
// rstd_m * acc - rstd_m * mean_m * c[n] + bias[n], with mean / rstd of row m of A from the producer's partial sums.
// MF: 0 = MFMA 16x16x32 (every variant, any shape), 1 = MFMA 32x32x16 (the LayerNorm fold's consumer and its staged
// producer on problems of whole 256 x 256 tiles: the launcher decides).  Same LDS layout, DMA and phase structure; a
// wave's 128 x 128 quarter is 4 x 4 blocks of 32 x 32, a fragment = 32 rows x 16 k (row lane & 31, 16-byte chunk
// 2 s + (lane >> 5) of the 64-byte row: conflict-free with the same swizzle -- ds_read_b128's lane groups hold rows that
// are distinct mod 16), the B rows of a 32-column group permuted so that a lane (row lane & 31, half lane >> 5) holds
// 16 CONSECUTIVE output columns 16 (lane >> 5) + e of the block in its 16 accumulator registers.
template <int DT, int ACT, int OUT, int RES, int LNF, int MF = 0>
__global__ __launch_bounds__(256) void gemm_nt_wide(const GemmParams p, int tiles_m, int tiles_n) {
    static_assert(MF == 0 || (OUT == 0 && LNF == 2) || (OUT == 2 && RES == 2 && LNF == 1), "MF == 1: fold variants only");
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    __shared__ __attribute__((aligned(16))) char smem[W_SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
#ifdef PIO_GEMM_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_wclk[0] = __builtin_amdgcn_s_memtime();
        g_wclk[1] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    // ---- this workgroup's tiles.  Blocked walk: XCD x (= blockIdx & 7, its own L2) owns tile rows
    // [x * RM, (x + 1) * RM); its C workgroups cover RM rows x CW columns per round, so the tiles in flight in one L2
    // share RM A panels and CW B panels.  Otherwise: a contiguous run of row-major tile ids per XCD, round-robin.
    const int total = tiles_m * tiles_n, G = gridDim.x, bx = blockIdx.x;
    int ntl, first = 0, stride = 1, b_tm = 0, b_tn = 0, b_cw = 0;
    {
        const int x = bx & 7, c = bx >> 3, C = G >> 3;
        const int RM = tiles_m >> 3;
        if ((G & 7) == 0 && (tiles_m & 7) == 0 && RM <= C && C % RM == 0 && tiles_n % (C / RM) == 0) {
            b_cw = C / RM;
            b_tm = x * RM + c % RM;
            b_tn = c / RM;
            ntl = tiles_n / b_cw;
        } else if ((G & 7) == 0) {
            const int run = (total + 7) >> 3;
            first = x * run + c;
            stride = C;
            int end = x * run + run;
            end = end < total ? end : total;
            ntl = first < end ? (end - first + stride - 1) / stride : 0;
        } else {
            first = bx;
            stride = G;
            ntl = first < total ? (total - first + stride - 1) / stride : 0;
        }
    }
    if (ntl <= 0) return;
    auto tile_mn = [&](int j, int &tm, int &tn) {
        if (b_cw > 0) {
            tm = b_tm;
            tn = b_tn + b_cw * j;
        } else {
            const int id = first + j * stride;
            tm = id / tiles_n;
            tn = id - tm * tiles_n;
        }
    };
    // K slices per tile: K / 32 (>= 4 and even: launcher); with a second weight image (p.npass == 2: C += A B_lo^T, the
    // rounding residual of the weights) the tiles whose columns have one run the K range twice, the second time
    // against B_lo -- same A slices, same accumulators.  Columns below p.lo_n0 (a multiple of 256) have no B_lo.
    const int nphK = p.K / W_BK;
    // (p.npass sweeps in general -- (A, B) [, (A, B_lo)] [, (A_lo, B)]: split activations and / or split weights -- each at
    //  its operand offsets dA / dB; only the B_lo-only form knows columns without a second image)
    // GEN: the instantiations the general form is for (plain epilogues; the fp32-residual producer form of the decoders'
    // fc2).  The latent stack's fold kernels keep the two-way form they were tuned with: the general bookkeeping cost them
    // 0.4-1.6 % (same box, rocprofv3).
    constexpr bool GEN = LNF == 0 || RES == 1;
    auto nph_of = [&](int tn) {
        if (!GEN || (p.npass == 2 && p.dA1 == 0)) return (p.npass == 2 && tn * W_BN >= p.lo_n0) ? 2 * nphK : nphK;
        return p.npass * nphK;
    };

    // ---- DMA side: a 1-KiB piece = 16 rows x 64 B; wave w owns A pieces 4w..4w+3 and B pieces 4w..4w+3 of a slice.
    const int prow = lane >> 2;
    const int qsrc = (lane & 3) ^ ((0x78 >> (2 * ((prow >> 2) & 3))) & 3);  // source chunk of LDS position lane & 3
    int dj = 0, dph = 0, dslot = 0, dnph = nphK;
    uint32_t voa[4], vob[4];  // per-lane byte offsets (row, chunk) inside A / B (< 2^32: launcher)
    auto dma_tile = [&](int j) {
        int tm, tn;
        tile_mn(j, tm, tn);
        dnph = nph_of(tn);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int gm = tm * W_BM + wave * 64 + i * 16 + prow;
            gm = gm < p.M ? gm : p.M - 1;
            voa[i] = ((uint32_t)gm * (uint32_t)p.lda + (uint32_t)(qsrc * 8)) * 2u;
            int gn = tn * W_BN + wave * 64 + i * 16 + prow;
            gn = gn < p.N ? gn : p.N - 1;
            vob[i] = ((uint32_t)gn * (uint32_t)p.ldb + (uint32_t)(qsrc * 8)) * 2u;
#ifdef PIO_GEMM_STAMPS
            if constexpr (PIO_WIDE_ABL & 8) {  // timing only: every piece reads 1 KiB of CONTIGUOUS memory (8 full lines)
                voa[i] = (uint32_t)((tm * 64 + wave * 4 + i) * 1024 + lane * 16);
                vob[i] = (uint32_t)((tn * 64 + wave * 4 + i) * 1024 + lane * 16);
            }
#endif
        }
    };
    const char *ia = nullptr, *ib = nullptr;  // wave-uniform bases: the DMA is "sgpr base + vgpr offset"
    char *isb = nullptr;
    int istep = 0, ibo = 0;
    // the DMA's position inside its tile as RUNNING state (a phase is 64 MFMAs: a dozen more scalar instructions per phase
    // to re-derive sweep, slice and operand offsets from dph cost the layer's GEMMs 1.5 %): K slice of the current sweep,
    // the sweep's operand bases
    int dk0 = 0;
    const T *swA = (const T *)p.A, *swB = (const T *)p.B;
    auto issue_begin = [&]() {
        if (dj < ntl) {
            // second sweep: the same A slices against B_lo.  k_rev: the workgroup's odd tiles run the K range backwards --
            // in the blocked walk consecutive tiles of a workgroup share their A panel, and the slices a tile used LAST
            // are the ones still in the XCD's L2 when the next tile starts (the panel as a whole is not: a round of
            // 8 x 4 tiles streams 6 MB through a 4 MB L2)
            if constexpr (GEN) {
                const int kph = (p.k_rev && (dj & 1)) ? nphK - 1 - dk0 : dk0;
                ia = (const char *)(swA + kph * W_BK);
                ib = (const char *)(swB + kph * W_BK);
            } else {
                const int kph0 = dph < nphK ? dph : dph - nphK;
                const int kph = (p.k_rev && (dj & 1)) ? nphK - 1 - kph0 : kph0;
                ia = (const char *)((const T *)p.A + kph * W_BK);
                ib = (const char *)((const T *)p.B + (dph < nphK ? 0 : p.dB1) + kph * W_BK);
            }
            isb = smem + dslot * W_STAGE + wave * 4096;
            istep = 1024;
            ibo = W_AB;
        } else {  // past the end of the work: same number of pieces, into the sink
            ia = (const char *)p.A;
            ib = (const char *)p.B;
            isb = smem + W_RING + wave * 1024;
            istep = 0;
            ibo = 0;
        }
    };
    auto piece = [&](int i) {  // 0..3: A, 4..7: B
        // (the offset is made opaque here so that its zero-extension is not hoisted out of the loop as a 64-bit
        //  register pair: "uniform base + zext(32-bit VGPR)" at the use is what selects the SGPR-base form of the DMA)
        uint32_t &o = i < 4 ? voa[i] : vob[i - 4];
        asm volatile("" : "+v"(o));  // (in place: no copy)
        if (i < 4) wide_dma16(ia + (uint64_t)o, isb + i * istep);
        else wide_dma16(ib + (uint64_t)o, isb + ibo + (i - 4) * istep);
    };
    auto issue_end = [&]() {
        dslot = (dslot + 1) & 3;
        if constexpr (!GEN) {
            if (dj < ntl && ++dph == dnph) {
                dph = 0;
                if (++dj < ntl) dma_tile(dj);
            }
        } else if (dj < ntl) {
            ++dph;
            if (++dk0 == nphK) {   // next sweep of this tile (or the next tile: reset below)
                dk0 = 0;
                const bool second = dph == nphK;
                swA = (const T *)p.A + (second ? p.dA1 : p.dA2);
                swB = (const T *)p.B + (second ? p.dB1 : p.dB2);
            }
            if (dph == dnph) {
                dph = 0;
                dk0 = 0;
                swA = (const T *)p.A;
                swB = (const T *)p.B;
                if (++dj < ntl) dma_tile(dj);
            }
        }
    };

    // ---- MFMA side
    const int fr = lane & 15, fq = lane >> 4;
    auto fsw = [](int g) { return (0x78 >> (2 * (g & 3))) & 3; };
    const int fa = fr * 64 + ((fq ^ fsw(fr >> 2)) << 4);
    const int fb0 = (8 * (fr >> 2) + (fr & 3)) * 64 + ((fq ^ fsw(2 * (fr >> 2))) << 4);
    const int fb1 = (8 * (fr >> 2) + 4 + (fr & 3)) * 64 + ((fq ^ fsw(2 * (fr >> 2) + 1)) << 4);
    const int a_base = wm * 128 * 64, b_base = W_AB + wn * 128 * 64;
    V8 af[2][8], bf[2][8];
    using AccT = std::conditional_t<MF == 1, f32x16, f32x4>;
    AccT acc[MF ? 4 : 8][MF ? 4 : 8];
    // MF == 1: fragment j = 2 * block + s (k-step s of the 32-deep slice)
    const int r31 = lane & 31, hq = lane >> 5;
    const int pi31 = (r31 & 3) + 4 * (r31 >> 3) + 16 * ((r31 >> 2) & 1);  // B row of MFMA row index r31
    const int fa32_0 = r31 * 64 + (((0 + hq) ^ fsw(r31 >> 2)) << 4), fa32_1 = r31 * 64 + (((2 + hq) ^ fsw(r31 >> 2)) << 4);
    const int fb32_0 = pi31 * 64 + (((0 + hq) ^ fsw(pi31 >> 2)) << 4), fb32_1 = pi31 * 64 + (((2 + hq) ^ fsw(pi31 >> 2)) << 4);
    auto a_frag = [&](const char *st, int j) __attribute__((always_inline)) {
        if constexpr (MF == 1) return *(const V8 *)(st + a_base + (j >> 1) * 2048 + ((j & 1) ? fa32_1 : fa32_0));
        else return *(const V8 *)(st + a_base + j * 1024 + fa);
    };
    auto b_frag = [&](const char *st, int j) __attribute__((always_inline)) {
        if constexpr (MF == 1) return *(const V8 *)(st + b_base + (j >> 1) * 2048 + ((j & 1) ? fb32_1 : fb32_0));
        else return *(const V8 *)(st + b_base + (j >> 1) * 2048 + ((j & 1) ? fb1 : fb0));
    };

    // ---- prologue: slices 0..3 in flight, slice 0 landed and in fragment set 0
    dma_tile(0);
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        issue_begin();
#pragma unroll
        for (int i = 0; i < 8; ++i) piece(i);
        issue_end();
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        af[0][i] = a_frag(smem, i);
        bf[0][i] = b_frag(smem, i);
    }

    int rslot = 1;      // ring slot of the slice whose fragments the next phase reads
    // The bias row of a tile (this wave's 128 columns) is parked in LDS by ONE DMA at the start of the tile's first
    // phase (no register held across the tile) and read back in the epilogue.
    float *const bstash = (float *)(smem + W_BIAS + wave * 1024);
    float *const cstash = bstash + 128;
    int o_m = 0, o_n = 0, o_n0 = 0;  // this lane's first row / column of the current tile; the wave's first column
    char *const sink = (char *)g_sink_w + lane * 16;

    // One phase.  The wave is alone on its SIMD, so whatever it issues outside an MFMA's shadow idles the matrix pipe:
    // the meeting point of the phase (counted wait for the DMA, wait for the LDS reads, barrier) sits BEHIND the
    // phase's first eight MFMAs, which keep the pipe busy while the wave waits, and the scalar bookkeeping rides
    // between MFMA groups.  `extra`: vector-memory operations issued between the last two phases' pieces and this
    // phase's meeting point (the previous tile's 32 stores and this tile's bias DMA: first two phases of a tile).
    auto phase = [&](auto PC, auto FC, int extra) {
        constexpr int P = decltype(PC)::value;
        constexpr bool FIRST = decltype(FC)::value;
        const char *rb = smem + rslot * W_STAGE;
        PIO_WSTAMP(0);
        if constexpr (FIRST) {
            if (lane < 32) {
                const int n = o_n0 + lane * 4;
                const float *src = (p.bias_mode == 1 && n < p.N) ? p.bias + n : (const float *)g_zero_w;
                wide_dma16(src, bstash);
                if constexpr (LNF == 2) {
                    const float *srcc = n < p.N ? p.ln_c + n : (const float *)g_zero_w;
                    wide_dma16(srcc, cstash);
                }
            }
        }
        wide_for<0, 16>([&](auto GI) {
            constexpr int g = decltype(GI)::value;
#ifdef PIO_GEMM_STAMPS
            constexpr bool skip_rd = PIO_WIDE_ABL & 1, skip_dma = PIO_WIDE_ABL & 2;
#else
            constexpr bool skip_rd = false, skip_dma = false;
#endif
            // DMA pieces ride behind groups 2, 4, ..., 14 and 15.  A piece costs the wave ~40 cycles of MFMA issue
            // (8 pieces: 325 of a phase's ~1450 cycles; the 16 LDS reads cost 50) wherever it is put: issuing the four
            // waves' pieces 16 cycles apart (tile loop compiled once per wave index) changed nothing, and a branch on
            // the wave index between MFMAs costs far more than it saves.
            constexpr int PIECE = (g >= 2 && !skip_dma) ? (g == 15 ? 7 : ((g & 1) == 0 ? (g - 2) >> 1 : -1)) : -1;
            if constexpr (MF == 1) {
                // 32 MFMAs 32x32x16: groups 0..7 run k-step 0 of the 16 blocks, groups 8..15 k-step 1
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    constexpr int ks = g >> 3;
                    const int blk = (2 * g + t) & 15, mi = blk >> 2, ni = blk & 3;
                    if constexpr (FIRST && ks == 0) mfma_first32<DT>(acc[mi][ni], bf[P][2 * ni + ks], af[P][2 * mi + ks]);
                    else mfma_acc32<DT>(acc[mi][ni], bf[P][2 * ni + ks], af[P][2 * mi + ks]);
                    if (PIECE >= 0 && t == 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(PIECE >= 0 ? PIECE : 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int mi = (4 * g + t) >> 3, ni = (4 * g + t) & 7;
                if constexpr (FIRST) mfma_first<DT>(acc[mi][ni], bf[P][ni], af[P][mi]);
                else mfma_acc<DT>(acc[mi][ni], bf[P][ni], af[P][mi]);
                if (PIECE >= 0 && t == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(PIECE >= 0 ? PIECE : 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
            __builtin_amdgcn_sched_barrier(0);  // (reads placed BEFORE the group's MFMAs would be waited for by them)
            if constexpr (g == 0) issue_begin();
            if constexpr (g == 1) {
                // ---- meeting point.  All but the last two phases' pieces (+ extra) have landed: slice g+1 is complete
                // in every wave's share; this wave's reads of slice g (issued in the previous phase) are done, so
                // after the barrier slot g & 3 may be refilled (slice g+4) and slice g+1 may be read.
                PIO_WSTAMP(1);
                if (extra == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (extra == 1) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
                else if (extra == 10) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
                else if (extra == 18) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
                else if (extra == 33) asm volatile("s_waitcnt vmcnt(49)" ::: "memory");
                else if (extra == 42) asm volatile("s_waitcnt vmcnt(58)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");  // (the counter's ceiling: stricter than needed)
                PIO_WSTAMP(2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PIO_WSTAMP(3);
#ifdef PIO_GEMM_STAMPS
                if constexpr (!(PIO_WIDE_ABL & 4))
#endif
                __builtin_amdgcn_s_barrier();
                PIO_WSTAMP(4);
            }
            if constexpr (g >= 2 && g < 6 && !skip_rd) {
                af[P ^ 1][2 * (g - 2)] = a_frag(rb, 2 * (g - 2));
                af[P ^ 1][2 * (g - 2) + 1] = a_frag(rb, 2 * (g - 2) + 1);
            } else if constexpr (g >= 6 && g < 10 && !skip_rd) {
                bf[P ^ 1][2 * (g - 6)] = b_frag(rb, 2 * (g - 6));
                bf[P ^ 1][2 * (g - 6) + 1] = b_frag(rb, 2 * (g - 6) + 1);
            }
            if constexpr (g == 15) {
                issue_end();
                rslot = (rslot + 1) & 3;
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        PIO_WSTAMP(5);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using Tt = std::true_type;
    using Ff = std::false_type;

    {
#pragma unroll 1
    for (int j = 0; j < ntl; ++j) {
        int nph;
        {
            int tm, tn;
            tile_mn(j, tm, tn);
            o_m = tm * W_BM + wm * 128 + fr;
            o_n0 = tn * W_BN + wn * 128;
            o_n = o_n0 + fq * 8;
            nph = nph_of(tn);
        }
        // stores of one epilogue per wave (a lower bound is enough once it exceeds the counter's ceiling) + the DMAs
        // of the bias / c rows at the head of the tile's first phase
        // (LNF == 2: + the 8 loads of the row statistics, issued here at the head of the tile and consumed in its
        //  epilogue -- they ride through the main loop in 32 of the ~100 free VGPRs instead of exposing their latency)
        // (K > 1024: slots 8 .. 11 ride in part2 -- one more 8-byte load per row block)
        constexpr int NDMA0 = LNF == 2 ? (MF == 1 ? 10 : 18) : 1;
        f32x4 part[8];
        pio_f32x2 part2[8];
        if constexpr (LNF == 2 && MF == 1) {
            // (row 32 mi + (lane & 31): the (sum, sum of squares) of column blocks 4 (lane >> 5) .. + 3 -- two loads)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = o_m - fr + 32 * mi + r31;
                const float *src = p.ln_part + ((int64_t)m * 8 + 4 * hq) * 2;
                part[2 * mi] = *(const f32x4 *)src;
                part[2 * mi + 1] = *(const f32x4 *)(src + 4);
            }
        } else if constexpr (LNF == 2) {
            // lane group fq takes slots 2 fq, 2 fq + 1 and (K > 1024) 8 + fq of its rows; slots the row does not have read
            // a zero block (ln_slots is even: the wide producer writes K / 128 of them)
            const int ns = p.ln_slots;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                int m = o_m + mi * 16;
                m = m < p.M ? m : p.M - 1;
                const float *pr = p.ln_part + (int64_t)m * ns * 2;
                part[mi] = *(const f32x4 *)(2 * fq < ns ? pr + 4 * fq : (const float *)g_zero_w);
                part2[mi] = *(const pio_f32x2 *)(8 + fq < ns ? pr + 16 + 2 * fq : (const float *)g_zero_w);
            }
        }
        constexpr int NSTORE = OUT == 2 ? 64 : 32;
        // (phase g waits for the pieces of phase g-3: in phases 0..2 of a tile those are older than the previous
        //  tile's stores, which may therefore stay outstanding; from phase 3 on the in-order counter makes them finish)
        phase(I0{}, Tt{}, j > 0 ? NSTORE + NDMA0 : NDMA0);
        phase(I1{}, Ff{}, j > 0 ? NSTORE + NDMA0 : NDMA0);
        phase(I0{}, Ff{}, j > 0 ? NSTORE + NDMA0 : NDMA0);
        phase(I1{}, Ff{}, 0);
#pragma unroll 1
        for (int ph = 4; ph < nph; ph += 2) {
            phase(I0{}, Ff{}, 0);
            phase(I1{}, Ff{}, 0);
        }
        // ---- exposed epilogue: 32 stores of 8 columns each
#ifdef PIO_GEMM_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) g_wstamps[6] = __builtin_readcyclecounter();
#endif
        const int t_m0 = o_m - (wm * 128 + fr), t_n0 = o_n0 - wn * 128;  // the tile's origin (wave-uniform)
        const bool interior = t_m0 + W_BM <= p.M && t_n0 + W_BN <= p.N;
        if constexpr (LNF == 1 && RES == 2) {
            // ---- LDS-STAGED epilogue of the LayerNorm fold's producer (the last tile of this workgroup, i.e. every
            // tile of the one-tile-per-CU out / fc2 projections): the ring is free now, so the accumulators of a row
            // block go through this wave's 32 KiB of it and come back in a ROW-COALESCED arrangement -- lane -> (row
            // lane >> 4, eight consecutive columns (lane & 15) * 8).  Every residual load and every store then covers
            // 4 rows x 256 contiguous bytes (whole 128-byte lines) instead of 16 rows x 64 bytes (half lines, each line
            // written by two instructions): the direct epilogue below moves its 128 MB per launch at 4.4 TB/s.
            if (MF == 1 || (!p.C && p.X16_lo && p.staged_epi && interior && j == ntl - 1)) {  // (MF == 1: launcher)
                PIO_WESTAMP(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // every wave has read its last fragments out of the ring
                PIO_WESTAMP(1);
                constexpr int ROWP = 132;      // floats per staged row: 128 + 4 (conflict-free b128 writes and reads)
                float *const stg = (float *)(smem + wave * 32768);
                const int cr = lane >> 4, cc = (lane & 15) * 8;
                const f32x4 bc0 = *(const f32x4 *)(bstash + cc), bc1 = *(const f32x4 *)(bstash + cc + 4);
                const int m_w = o_m - fr;      // first row of this wave's quarter
                const int n_c = o_n0 + cc;     // this lane's first column in the coalesced arrangement
                const T *const r_hi = (const T *)p.R16_hi + (int64_t)(m_w + cr) * p.ld16 + n_c;
                const T *const r_lo = (const T *)p.R16_lo + (int64_t)(m_w + cr) * p.ld16 + n_c;
                T *const x_hi = (T *)p.X16 + (int64_t)(m_w + cr) * p.ld16 + n_c;
                T *const x_lo = (T *)p.X16_lo + (int64_t)(m_w + cr) * p.ld16 + n_c;
                float *const part = p.row_part + ((int64_t)(m_w + cr) * (p.N >> 7) + (o_n0 >> 7)) * 2;
                const int64_t rstep4 = 4 * p.ld16;          // elements per 4 rows
                // The residual of row block mi is requested RD blocks ahead: with one block in flight (8 KiB per wave,
                // 32 KiB per CU) against ~2 us of loaded-memory latency a CU drew 16-19 GB/s -- the whole chip 4.7 TB/s
                // by lack of requests in flight, not by HBM.
                constexpr int RD = 2;
                V8 rh[RD + 1][4], rl[RD + 1][4];
                auto r_load = [&](int set, int mi) {
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        rh[set][st] = *(const V8 *)(r_hi + (int64_t)(mi * 4 + st) * rstep4);
                        rl[set][st] = *(const V8 *)(r_lo + (int64_t)(mi * 4 + st) * rstep4);
                    }
                };
#pragma unroll
                for (int mi = 0; mi < RD; ++mi) r_load(mi, mi);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    if (mi + RD < 8) r_load((mi + RD) % (RD + 1), mi + RD);
                    // accumulators of row block mi -> LDS, accumulator arrangement (row fr, 8 columns at 32 pp + 8 fq)
                    if constexpr (MF == 1) {
                        // (32-row blocks: every other 16-row step writes rows 32 (mi / 2) + (lane & 31), 16 columns at
                        //  32 ni + 16 (lane >> 5); the staging rows hold 32 rows then)
                        if ((mi & 1) == 0) {
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                                for (int e4 = 0; e4 < 4; ++e4) {
                                    f32x4 x;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) x[r] = acc_read(acc[mi >> 1][ni][4 * e4 + r]);
                                    *(f32x4 *)(stg + r31 * ROWP + 32 * ni + 16 * hq + 4 * e4) = x;
                                }
                        }
                    } else {
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        f32x4 x0, x1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            x0[r] = acc_read(acc[mi][2 * pp][r]);
                            x1[r] = acc_read(acc[mi][2 * pp + 1][r]);
                        }
                        *(f32x4 *)(stg + fr * ROWP + 32 * pp + 8 * fq) = x0;
                        *(f32x4 *)(stg + fr * ROWP + 32 * pp + 8 * fq + 4) = x1;
                    }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ... and back, row-coalesced: 4 steps of 4 rows (LDS operations of one wave execute in order)
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const int srow = (MF == 1 ? (mi & 1) * 16 : 0) + st * 4 + cr;
                        const f32x4 a0 = *(const f32x4 *)(stg + srow * ROWP + cc);
                        const f32x4 a1 = *(const f32x4 *)(stg + srow * ROWP + cc + 4);
                        const V8 hh = rh[mi % (RD + 1)][st], ll = rl[mi % (RD + 1)][st];
                        f32x4 x0, x1;
                        float rsum, rsq;
                        V8 h, l;
                        if constexpr (DT == PIO_DT_F16) {
                            // This epilogue is bound by the number of instructions one wave can issue (~5 cycles each),
                            // not by its bytes: the residual halves enter through v_fma_mix_f32 (16-bit operand read in
                            // place: no unpack + convert), the lo half of the result leaves the same way, the row
                            // statistics run two columns per instruction, and the 16-lane reduction below uses DPP row
                            // rotations instead of ds_bpermute.
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            const u32x4 hw = __builtin_bit_cast(u32x4, hh), lw = __builtin_bit_cast(u32x4, ll);
                            auto mix_add = [](uint32_t pair, int hi, float t, float sgn) {  // t + sgn * half(pair, hi)
                                float d;
                                if (hi) {
                                    if (sgn > 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(t));
                                    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(t));
                                } else {
                                    if (sgn > 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(t));
                                    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(t));
                                }
                                return d;
                            };
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float t0 = fmaf(a0[r], p.alpha, bc0[r]), t1 = fmaf(a1[r], p.alpha, bc1[r]);
                                t0 = mix_add(hw[r >> 1], r & 1, t0, 1.f);
                                t1 = mix_add(hw[2 + (r >> 1)], r & 1, t1, 1.f);
                                x0[r] = mix_add(lw[r >> 1], r & 1, t0, 1.f);
                                x1[r] = mix_add(lw[2 + (r >> 1)], r & 1, t1, 1.f);
                                h[r] = Op<DT>::from_f32(x0[r]);
                                h[4 + r] = Op<DT>::from_f32(x1[r]);
                            }
                            const u32x4 hp = __builtin_bit_cast(u32x4, h);
                            pio_f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                l[r] = Op<DT>::from_f32(mix_add(hp[r >> 1], r & 1, x0[r], -1.f));
                                l[4 + r] = Op<DT>::from_f32(mix_add(hp[2 + (r >> 1)], r & 1, x1[r], -1.f));
                                const pio_f32x2 xx = {x0[r], x1[r]};
                                s2 += xx;
                                q2 = __builtin_elementwise_fma(xx, xx, q2);
                            }
                            rsum = s2[0] + s2[1];
                            rsq = q2[0] + q2[1];
                        } else {
                            rsum = 0.f;
                            rsq = 0.f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                x0[r] = a0[r] * p.alpha + bc0[r] + (Op<DT>::to_f32(hh[r]) + Op<DT>::to_f32(ll[r]));
                                x1[r] = a1[r] * p.alpha + bc1[r] + (Op<DT>::to_f32(hh[4 + r]) + Op<DT>::to_f32(ll[4 + r]));
                                rsum += x0[r] + x1[r];
                                rsq += x0[r] * x0[r] + x1[r] * x1[r];
                                h[r] = Op<DT>::from_f32(x0[r]);
                                h[4 + r] = Op<DT>::from_f32(x1[r]);
                                l[r] = Op<DT>::from_f32(x0[r] - Op<DT>::to_f32(h[r]));
                                l[4 + r] = Op<DT>::from_f32(x1[r] - Op<DT>::to_f32(h[4 + r]));
                            }
                        }
                        const int64_t ro = (int64_t)(mi * 4 + st) * rstep4;
                        *(V8 *)(x_hi + ro) = h;
                        *(V8 *)(x_lo + ro) = l;
                        // the 16 lanes of a row hold 8 columns each: four rotations inside the 16-lane DPP row leave the
                        // row's (sum, sum of squares) over this wave's 128-column block in every lane; lane 0 of the
                        // row writes it
                        {
                            auto ror_add = [](float v, auto NC) {
                                constexpr int n = decltype(NC)::value;
                                const int moved = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x120 + n, 0xf, 0xf, true);
                                return v + __builtin_bit_cast(float, moved);
                            };
                            rsum = ror_add(rsum, std::integral_constant<int, 8>{});
                            rsq = ror_add(rsq, std::integral_constant<int, 8>{});
                            rsum = ror_add(rsum, std::integral_constant<int, 4>{});
                            rsq = ror_add(rsq, std::integral_constant<int, 4>{});
                            rsum = ror_add(rsum, std::integral_constant<int, 2>{});
                            rsq = ror_add(rsq, std::integral_constant<int, 2>{});
                            rsum = ror_add(rsum, std::integral_constant<int, 1>{});
                            rsq = ror_add(rsq, std::integral_constant<int, 1>{});
                        }
                        if ((lane & 15) == 0) {
                            float *dstp = part + (int64_t)(mi * 16 + st * 4) * (p.N >> 7) * 2;
                            dstp[0] = rsum;
                            dstp[1] = rsq;
                            // range guard of the folded stack: an overflowed 16-bit half shows as a non-finite sum
                            // (rare store; an extra vector-memory operation only makes the counted waits stricter)
                            if (p.range_flag && !(rsq <= 3.0e38f)) *p.range_flag = 1;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    PIO_WESTAMP(2 + mi);
                }
#ifdef PIO_GEMM_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (dev build: the last stamp includes the stores' completion)
#endif
                PIO_WESTAMP(10);
                continue;
            }
        }
        if constexpr (OUT == 2 && MF == 1) {
            // (every tile of an MF == 1 producer leaves through the staged epilogue above)
        } else if constexpr (OUT == 2) {
            // fp32 out [+ residual]: 64 stores of 4 columns; the residual of row block mi+1 is loaded while block mi
            // is converted and stored
            float *const cf = (float *)p.C;
            f32x4 b0[4], b1[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                b0[pp] = *(const f32x4 *)(bstash + pp * 32 + fq * 8);
                b1[pp] = *(const f32x4 *)(bstash + pp * 32 + fq * 8 + 4);
            }
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            f32x4 rb[2][8];   // RES == 1: residual of a row block as fp32; RES == 2: rp = (hi, lo) 16-bit pairs
            V8 rp[2][8];
            const V8 zero8 = {};
            auto r_load = [&](int set, int mi) {
                const int m = o_m + mi * 16;
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int n = o_n + pp * 32;
                    const bool in = interior || (m < p.M && n < p.N);
                    if constexpr (RES == 1) {
                        if (in) {
                            const float *src = p.R + (int64_t)m * p.ldr + n;
                            rb[set][2 * pp] = *(const f32x4 *)src;
                            rb[set][2 * pp + 1] = *(const f32x4 *)(src + 4);
                        } else {
                            rb[set][2 * pp] = zero4;
                            rb[set][2 * pp + 1] = zero4;
                        }
                    } else {
                        if (in) {
                            rp[set][2 * pp] = *(const V8 *)((const T *)p.R16_hi + (int64_t)m * p.ld16 + n);
                            rp[set][2 * pp + 1] = *(const V8 *)((const T *)p.R16_lo + (int64_t)m * p.ld16 + n);
                        } else {
                            rp[set][2 * pp] = zero8;
                            rp[set][2 * pp + 1] = zero8;
                        }
                    }
                }
            };
            if constexpr (RES != 0) r_load(0, 0);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = o_m + mi * 16;
                if constexpr (RES != 0) {
                    if (mi < 7) r_load((mi + 1) & 1, mi + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float rsum = 0.f, rsq = 0.f;  // (LNF == 1) this lane's share of row m: 32 columns
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int n = o_n + pp * 32;
                    f32x4 x0, x1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        x0[r] = acc_read(acc[mi][2 * pp][r]) * p.alpha + b0[pp][r];
                        x1[r] = acc_read(acc[mi][2 * pp + 1][r]) * p.alpha + b1[pp][r];
                    }
                    if constexpr (RES == 1) {
                        x0 += rb[mi & 1][2 * pp];
                        x1 += rb[mi & 1][2 * pp + 1];
                    } else if constexpr (RES == 2) {
                        const V8 hh = rp[mi & 1][2 * pp], ll = rp[mi & 1][2 * pp + 1];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            x0[r] += Op<DT>::to_f32(hh[r]) + Op<DT>::to_f32(ll[r]);
                            x1[r] += Op<DT>::to_f32(hh[4 + r]) + Op<DT>::to_f32(ll[4 + r]);
                        }
                    }
                    if (!interior && n >= p.N) {
                        x0 = zero4;  // columns [N, n_store): zeros
                        x1 = zero4;
                    }
                    const bool ok = interior || (m < p.M && n < p.n_store);
                    if (LNF != 1 || cf) {  // (beside a 16-bit pair the fp32 result is optional)
                        char *dst = ok ? (char *)(cf + (int64_t)m * p.ldc + n) : sink;
                        *(f32x4 *)dst = x0;
                        *(f32x4 *)(ok ? dst + 16 : dst) = x1;
                    }
                    if constexpr (LNF == 1) {
                        V8 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            rsum += x0[r] + x1[r];
                            rsq += x0[r] * x0[r] + x1[r] * x1[r];
                            h[r] = Op<DT>::from_f32(x0[r]);
                            h[4 + r] = Op<DT>::from_f32(x1[r]);
                        }
                        *(V8 *)(ok ? (char *)((T *)p.X16 + (int64_t)m * p.ld16 + n) : sink) = h;
                        if (p.X16_lo) {
                            V8 l;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                l[r] = Op<DT>::from_f32(x0[r] - Op<DT>::to_f32(h[r]));
                                l[4 + r] = Op<DT>::from_f32(x1[r] - Op<DT>::to_f32(h[4 + r]));
                            }
                            *(V8 *)(ok ? (char *)((T *)p.X16_lo + (int64_t)m * p.ld16 + n) : sink) = l;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (LNF == 1) {
                    // the four lanes (fq = 0..3) of a row hold 32 columns each: two butterfly steps, then lane fq == 0
                    // writes the row's (sum, sum of squares) over this wave's 128-column block
                    rsum += __shfl_xor(rsum, 16);
                    rsq += __shfl_xor(rsq, 16);
                    rsum += __shfl_xor(rsum, 32);
                    rsq += __shfl_xor(rsq, 32);
                    if (p.row_part && fq == 0 && m < p.M && o_n0 < p.N) {   // (no statistics asked: a plain pair result)
                        float *dstp = p.row_part + ((int64_t)m * (p.N >> 7) + (o_n0 >> 7)) * 2;
                        dstp[0] = rsum;
                        dstp[1] = rsq;
                        if (p.range_flag && !(rsq <= 3.0e38f)) *p.range_flag = 1;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (MF == 1) {
            // ---- LayerNorm fold's consumer, 32 x 32 blocks, whole tiles only (launcher): lane = row 32 mi + (lane & 31),
            // 16 consecutive columns 32 ni + 16 (lane >> 5) + e per block: two 16-byte stores per block
            T *const cbase = (T *)p.C;
            float rs[4], nmr[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                float sm = part[2 * mi][0] + part[2 * mi][2] + part[2 * mi + 1][0] + part[2 * mi + 1][2];
                float sq = part[2 * mi][1] + part[2 * mi][3] + part[2 * mi + 1][1] + part[2 * mi + 1][3];
                sm += __shfl_xor(sm, 32);
                sq += __shfl_xor(sq, 32);
                const float mean = sm * (1.0f / 1024.0f);
                float var = sq * (1.0f / 1024.0f) - mean * mean;
                var = var > 0.f ? var : 0.f;
                rs[mi] = 1.0f / sqrtf(var + p.ln_eps);
                nmr[mi] = -mean * rs[mi];
            }
            auto val4m = [&](f32x4 a, f32x4 b, f32x4 c, int mi) {
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = a[r] * rs[mi] + (nmr[mi] * c[r] + b[r]);
                if constexpr (ACT == 1) {
                    const pio_f32x2 lo = gelu_erf2(pio_f32x2{x[0], x[1]}), hi = gelu_erf2(pio_f32x2{x[2], x[3]});
                    x = f32x4{lo[0], lo[1], hi[0], hi[1]};
                }
                return x;
            };
            const int m0 = o_m - fr + r31;
            const int n0 = o_n0 + 16 * hq;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 b4[4], c4[4];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    b4[e4] = *(const f32x4 *)(bstash + 32 * ni + 16 * hq + 4 * e4);
                    c4[e4] = *(const f32x4 *)(cstash + 32 * ni + 16 * hq + 4 * e4);
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    V8 h0, h1;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        f32x4 a;
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[r] = acc_read(acc[mi][ni][4 * e4 + r]);
                        const f32x4 y = val4m(a, b4[e4], c4[e4], mi);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (e4 < 2) h0[4 * e4 + r] = Op<DT>::from_f32(y[r]);
                            else h1[4 * (e4 - 2) + r] = Op<DT>::from_f32(y[r]);
                        }
                    }
                    T *dst = cbase + (int64_t)(m0 + 32 * mi) * p.ldc + n0 + 32 * ni;
                    *(V8 *)dst = h0;
                    *(V8 *)(dst + 8) = h1;
                    __builtin_amdgcn_sched_barrier(0);  // (else all 256 accumulators are read out before the first store)
                }
            }
        } else {
        T *const cbase = (T *)p.C;
        // (LNF == 2) mean / rstd of this lane's eight rows from the producer's partial sums: lane group fq adds slots
        // 2 fq, 2 fq + 1 and 8 + fq (K / 128 slots of 128 columns, K <= 1536), two butterfly steps add the four groups
        float rs[8], nmr[8];
        if constexpr (LNF == 2) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                float sm = part[mi][0] + part[mi][2] + part2[mi][0], sq = part[mi][1] + part[mi][3] + part2[mi][1];
                sm += __shfl_xor(sm, 16);
                sq += __shfl_xor(sq, 16);
                sm += __shfl_xor(sm, 32);
                sq += __shfl_xor(sq, 32);
                const float mean = sm * p.ln_inv_k;
                float var = sq * p.ln_inv_k - mean * mean;
                var = var > 0.f ? var : 0.f;
                rs[mi] = 1.0f / sqrtf(var + p.ln_eps);
                nmr[mi] = -mean * rs[mi];
            }
        }
        // one output element: alpha * acc + bias, or the folded LayerNorm form
        auto val = [&](float a, float b, float c, int mi) {
            float x;
            if constexpr (LNF == 2) x = a * rs[mi] + (nmr[mi] * c + b);
            else x = a * p.alpha + b;
            if constexpr (ACT == 1) x = gelu_erf(x);
            return x;
        };
        // four output elements (the epilogue is bound by its instruction count: GELU goes two elements at a time)
        auto val4 = [&](f32x4 a, f32x4 b, f32x4 c, int mi) {
            f32x4 x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (LNF == 2) x[r] = a[r] * rs[mi] + (nmr[mi] * c[r] + b[r]);
                else x[r] = a[r] * p.alpha + b[r];
            }
            if constexpr (ACT == 1) {
                const pio_f32x2 lo = gelu_erf2(pio_f32x2{x[0], x[1]}), hi = gelu_erf2(pio_f32x2{x[2], x[3]});
                x = f32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            return x;
        };
        // (the lo image of a hi + lo result lives at a wave-uniform byte distance from the hi image: no second address)
        const int64_t lo_delta = (LNF == 0 && p.C_lo) ? (int64_t)((const char *)p.C_lo - (const char *)p.C) : 0;
        if (interior) {
            // interior tile (every column < N <= n_store): one row pointer per mi, the four column groups are
            // immediate offsets of the store
            f32x4 b0[4], b1[4], c0[4], c1[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                b0[pp] = *(const f32x4 *)(bstash + pp * 32 + fq * 8);
                b1[pp] = *(const f32x4 *)(bstash + pp * 32 + fq * 8 + 4);
                if constexpr (LNF == 2) {
                    c0[pp] = *(const f32x4 *)(cstash + pp * 32 + fq * 8);
                    c1[pp] = *(const f32x4 *)(cstash + pp * 32 + fq * 8 + 4);
                } else {
                    c0[pp] = b0[pp];
                    c1[pp] = b1[pp];
                }
            }
            char *crow = (char *)(cbase + (int64_t)o_m * p.ldc + o_n);
            const int64_t rstep = (int64_t)p.ldc * 32;  // 16 rows, bytes
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    V8 h;
                    f32x4 a0, a1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a0[r] = acc_read(acc[mi][2 * pp][r]);
                        a1[r] = acc_read(acc[mi][2 * pp + 1][r]);
                    }
                    const f32x4 y0 = val4(a0, b0[pp], c0[pp], mi), y1 = val4(a1, b1[pp], c1[pp], mi);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        h[r] = Op<DT>::from_f32(y0[r]);
                        h[4 + r] = Op<DT>::from_f32(y1[r]);
                    }
                    *(V8 *)(crow + pp * 64) = h;
                    if constexpr (LNF == 0) {
                        if (lo_delta) {   // (uniform) the result's lo image, at a constant distance from the hi one
                            V8 l;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                l[r] = Op<DT>::from_f32(y0[r] - Op<DT>::to_f32(h[r]));
                                l[4 + r] = Op<DT>::from_f32(y1[r] - Op<DT>::to_f32(h[4 + r]));
                            }
                            *(V8 *)(crow + lo_delta + pp * 64) = l;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // (else all 256 accumulators are read out before the first store)
                }
                crow += rstep;
            }
        } else {
        T *const cbase = (T *)p.C;
    #pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int n = o_n + pp * 32;
                const f32x4 b0 = *(const f32x4 *)(bstash + pp * 32 + fq * 8);
                const f32x4 b1 = *(const f32x4 *)(bstash + pp * 32 + fq * 8 + 4);
                const f32x4 c0 = LNF == 2 ? *(const f32x4 *)(cstash + pp * 32 + fq * 8) : b0;
                const f32x4 c1 = LNF == 2 ? *(const f32x4 *)(cstash + pp * 32 + fq * 8 + 4) : b1;
    #pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    const int m = o_m + mi * 16;
                    V8 h, l;
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float y0 = val(acc_read(acc[mi][2 * pp][r]), b0[r], c0[r], mi);
                        const float y1 = val(acc_read(acc[mi][2 * pp + 1][r]), b1[r], c1[r], mi);
                        h[r] = Op<DT>::from_f32(y0);
                        h[4 + r] = Op<DT>::from_f32(y1);
                        l[r] = Op<DT>::from_f32(y0 - Op<DT>::to_f32(h[r]));
                        l[4 + r] = Op<DT>::from_f32(y1 - Op<DT>::to_f32(h[4 + r]));
                    }
                    if (n >= p.N) {
    #pragma unroll
                        for (int r = 0; r < 8; ++r) h[r] = l[r] = Op<DT>::from_f32(0.f);  // columns [N, n_store): zeros
                    }
                    const bool ok = m < p.M && n < p.n_store;
                    *(V8 *)(ok ? (char *)(cbase + (int64_t)m * p.ldc + n) : sink) = h;
                    if constexpr (LNF == 0) {
                        if (lo_delta) *(V8 *)(ok ? (char *)(cbase + (int64_t)m * p.ldc + n) + lo_delta : sink) = l;
                    }
                    __builtin_amdgcn_sched_barrier(0);  // (else all 256 accumulators are read out before the first store)
                }
            }
        }
        }
#ifdef PIO_GEMM_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) g_wstamps[7] = __builtin_readcyclecounter();
#endif
    }
    }
#ifdef PIO_GEMM_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_wclk[2] = __builtin_amdgcn_s_memtime();
        g_wclk[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef PIO_GEMM_STAMPS
extern "C" int pio_debug_wide_mode(void) { return PIO_WIDE_ABL; }
extern "C" int pio_debug_wide_epi_stamps(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_westamps), sizeof(g_westamps)) == hipSuccess ? 0 : 1;
}
extern "C" int pio_debug_wide_stamps(unsigned long long *out12) {
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_wstamps), sizeof(g_wstamps)) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out12 + 8, HIP_SYMBOL(g_wclk), sizeof(g_wclk)) == hipSuccess ? 0 : 1;
}
#endif

static int wide_grid(int64_t total) {
    const int n_cu = cu_budget();
    int G = (int)(total < n_cu ? total : n_cu);
    if (G >= 8) G &= ~7;
    return G;
}

bool gemm_wide_ok(const GemmParams &p, int batch) {
    if (batch != 1 || p.npass > 3) return false;
    // a second sweep against B_lo only (weights as hi + lo, single activations) may start at a 256-aligned column; the
    // sweeps with an A_lo image (split activations: dA1 or dA2) cover every column
    const bool b_lo_only = p.npass == 2 && p.dA1 == 0;
    // (the fold forms run the two-way sweep code: B_lo only)
    if (!b_lo_only && p.npass > 1 && (p.ln_part || p.ln_c || p.R16_hi || p.R16_lo || (p.row_part && !p.R))) return false;
    if (b_lo_only && (p.dB1 == 0 || p.lo_n0 < 0 || (p.lo_n0 % W_BN))) return false;
    if (!b_lo_only && p.lo_n0 != 0) return false;
    if (p.K < 4 * W_BK || (p.K % (2 * W_BK))) return false;
    if (p.act != 0 && p.act != 1) return false;
    // the hi + lo pair of the result: the plain 16-bit epilogue only (not the fold forms, not fp32 out)
    if (p.C_lo && (p.out_f32 || p.ln_part || p.row_part || p.X16 || ((uintptr_t)p.C_lo & 15))) return false;
    if (p.out_f32 && (p.act != 0 || (p.C && (p.ldc & 3)))) return false;
    // an fp32 residual: with an fp32 result, or -- the dense decoders' fc2 -- with the hi + lo pair of the result (the
    // fold producer's epilogue without its statistics: gemm_wide_launch maps C / C_lo onto X16 / X16_lo)
    const bool pair_res = p.R && !p.out_f32 && p.C_lo && p.C && p.act == 0 && !p.X16 && !p.row_part && !p.ln_part;
    if (p.R && ((!p.out_f32 && !pair_res) || !p.r_vec || p.r_rows != 0 || (p.ldr & 3))) return false;
    if (pair_res && (((uintptr_t)p.C & 15) || ((uintptr_t)p.C_lo & 15) || (p.ldc & 7))) return false;
    if (p.X16 || p.row_part || p.X16_lo || p.R16_hi || p.R16_lo) {  // LayerNorm-fold producer
        const bool pair_r = p.R16_hi || p.R16_lo;
        if (!p.X16 || !p.row_part || !p.out_f32 || (p.N & 127) || p.slot_w != 128 || (p.ld16 & 7) || ((uintptr_t)p.X16 & 15) ||
            ((uintptr_t)p.row_part & 7) || p.ln_part || p.ln_c)
            return false;
        if (pair_r ? (!p.R16_hi || !p.R16_lo || p.R || ((uintptr_t)p.R16_hi & 15) || ((uintptr_t)p.R16_lo & 15)) : !p.R)
            return false;
        if (((uintptr_t)p.X16_lo & 15) || (!p.C && !p.X16_lo)) return false;
    }
    if (p.ln_part || p.ln_c) {  // LayerNorm-fold consumer
        if (!p.ln_part || !p.ln_c || p.out_f32 || (p.K & 127) || p.K > 1536 || p.ln_slots != p.K / 128 ||
            (p.ln_slots & 1) || p.alpha != 1.0f || ((uintptr_t)p.ln_c & 15) || ((uintptr_t)p.ln_part & 15))
            return false;
    }
    if (p.bias_mode > 1 || (p.bias_mode == 1 && !p.bias_vec)) return false;
    if ((p.N & 7) || (p.n_store & 7) || (p.C && ((p.ldc & 7) || ((uintptr_t)p.C & 15)))) return false;
    if ((p.lda & 7) || (p.ldb & 7) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return false;
    if (((int64_t)p.M * p.lda + p.K) * 2 >= (1ll << 32) || ((int64_t)p.N * p.ldb + p.K) * 2 >= (1ll << 32)) return false;
    return true;
}

void gemm_wide_launch(const GemmParams &p, int dtype, hipStream_t s) {
    const int tiles_m = (p.M + W_BM - 1) / W_BM, tiles_n = (p.n_store + W_BN - 1) / W_BN;
    const int G = wide_grid((int64_t)tiles_m * tiles_n);
    dim3 grid((unsigned)G, 1, 1), block(256, 1, 1);
    GemmParams pk = p;
    {
        const char *e = getenv("PIO_WIDE_KREV");  // (read per launch: A/B switch for tools/ab_env.py)
        pk.k_rev = e ? atoi(e) : 0;
    }
    const bool pair_res = p.R && !p.out_f32 && p.C_lo;
    if (pair_res) {   // fp32 residual in, hi + lo pair out: the fold producer's epilogue (<0, 2, 1, 1>) without statistics
        pk.X16 = p.C;
        pk.X16_lo = p.C_lo;
        pk.ld16 = p.ldc;
        pk.C = nullptr;
        pk.C_lo = nullptr;
        pk.out_f32 = 1;
        pk.row_part = nullptr;
        pk.range_flag = nullptr;
    }
#define PIO_WK(DTV, ACT, OUT, R, LNF) \
    hipLaunchKernelGGL((gemm_nt_wide<DTV, ACT, OUT, R, LNF>), grid, block, 0, s, pk, tiles_m, tiles_n)
    // MFMA 32x32x16 variants of the fold GEMMs (template parameter MF): an experiment that measured level -- 7 % fewer
    // cycles, 5 % less clock (DESIGN_LOG.md) -- instantiated in the experiments build only (-DPIO_EXPERIMENTS).
#ifdef PIO_EXPERIMENTS
    static const bool mf32_on = [] {
        const char *e = getenv("PIO_WIDE_MF32");
        return e && atoi(e) != 0;
    }();
    const bool whole = (p.M % W_BM) == 0 && (p.N % W_BN) == 0 && p.n_store == p.N;
    const bool mf_any = mf32_on || p.mf32;
    const bool mf_cons = mf_any && whole && p.ln_part && !p.out_f32 && p.K == 1024;
    const bool mf_prod = mf_any && whole && p.row_part && p.R16_hi && !p.C && p.X16_lo && p.staged_epi &&
                         (int64_t)tiles_m * tiles_n <= G;
#define PIO_WKM(DTV, ACT, OUT, R, LNF) \
    hipLaunchKernelGGL((gemm_nt_wide<DTV, ACT, OUT, R, LNF, 1>), grid, block, 0, s, pk, tiles_m, tiles_n)
#else
    const bool mf_cons = false, mf_prod = false;
#define PIO_WKM(DTV, ACT, OUT, R, LNF) ((void)0)
#endif
#define PIO_WS(DTV)                                                  \
    do {                                                             \
        if (mf_prod) PIO_WKM(DTV, 0, 2, 2, 1);                       \
        else if (mf_cons && p.act == 1) PIO_WKM(DTV, 1, 0, 0, 2);    \
        else if (mf_cons) PIO_WKM(DTV, 0, 0, 0, 2);                  \
        else if (p.row_part && p.R16_hi) PIO_WK(DTV, 0, 2, 2, 1);    \
        else if (p.row_part || pair_res) PIO_WK(DTV, 0, 2, 1, 1);    \
        else if (p.out_f32 && p.R) PIO_WK(DTV, 0, 2, 1, 0);          \
        else if (p.out_f32) PIO_WK(DTV, 0, 2, 0, 0);                 \
        else if (p.ln_part && p.act == 1) PIO_WK(DTV, 1, 0, 0, 2);   \
        else if (p.ln_part) PIO_WK(DTV, 0, 0, 0, 2);                 \
        else if (p.act == 1) PIO_WK(DTV, 1, 0, 0, 0);                \
        else PIO_WK(DTV, 0, 0, 0, 0);                                \
    } while (0)
    if (dtype == PIO_DT_F16) PIO_WS(PIO_DT_F16);
    else PIO_WS(PIO_DT_BF16);
#undef PIO_WS
#undef PIO_WK
#undef PIO_WKM
}

}  // namespace pio
