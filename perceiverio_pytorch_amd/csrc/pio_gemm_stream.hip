// NT GEMM, persistent "streaming" variant for the weight GEMMs of the latent stack (K*passes >= 1024).
//
// Why: with one 256x256 tile per workgroup the matrix pipe idles while the tile's result goes out (bias / GELU /
// convert / residual / 128-256 KiB of stores): in-kernel stamps put 27k-48k of a K=1024 tile's 80k-100k cycles in
// that epilogue (tools/gemm_stamps.py), and every CU is in it at the same time, so HBM idles during the main loops
// and the MFMAs idle during the epilogues.  Here one workgroup per CU walks a LIST of 256x128 tiles and the result
// of tile j leaves the registers WHILE tile j+1 is being multiplied:
//   * 8 waves = 4 (M) x 2 (N), 64x64 per wave; TWO accumulator sets (2 x 64 VGPRs): MFMAs of tile j+1 go to one
//     set while the other set (tile j) is written out, one 16x16 unit per K step;
//   * a wave alternates between a LOAD phase (LDS fragment reads + DMA refill, nothing else: the partner wave's
//     MFMAs hold the vector issue, a VALU instruction costs ~20 cycles there) and an MFMA phase whose 32 MFMAs carry
//     the epilogue arithmetic of the previous tile among them (an MFMA holds the issue 8 of its 16 cycles);
//   * the two waves of a SIMD (wave w and w+4, "teams" 0/1) run the SAME instruction stream LOAD(0) MFMA(0) LOAD(1)
//     MFMA(1) ... with ONE barrier per step and wave, at different points: team 1 meets team 0 after its LOAD
//     phase, team 0 after its MFMA phase -- between two barriers team 0 runs LOAD(g) MFMA(g), team 1 MFMA(g-1)
//     LOAD(g): one team's MFMAs beside the other's loads (the GELU variant keeps two meeting points per step);
//   * K advances in 64-deep steps through a 3-slot LDS ring (3 x 48 KiB) filled by LDS-DMA two steps ahead; the
//     step sequence is flattened over the tile list, so the ring never drains between tiles (no per-tile prologue);
//     the DMA address is a wave-uniform base (tile, K step, pass) plus a per-lane 32-bit offset fixed per tile;
//   * 128-byte LDS rows, 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7): conflict-free ds_read_b128;
//   * the first 16 steps of a tile are unrolled, so which accumulator registers leave in which step is static;
//   * residual GEMMs (out = A*B^T + bias + R): R is not an epilogue read -- right after a unit of tile j has been
//     stored, the residual of tile j+1 is LOADED into the registers it vacated (global_load straight into the MFMA C
//     registers, at the end of that MFMA phase), so the R traffic overlaps the MFMAs as well;
//   * the bias is not an epilogue operand either: the bias row of tile j+2 is parked in a per-wave LDS stash
//     (three slots) by one more DMA at step 13 of tile j, and bias/alpha is added to the accumulators of a tile
//     during its own first steps;
//   * the tile list of a workgroup is decoded once into an LDS table (tile ids walk an XCD's run of the tile grid in
//     rounds of (C/8) x 8 tiles, C = workgroups per XCD);
//   * every wait on memory is a COUNTED s_waitcnt vmcnt(n): n = the vector-memory operations this wave issued after
//     the stage it needs.  Loads return in order; stores are counted too -- lanes outside the matrix store to a sink
//     so the number of stores per step is fixed -- and the build fails if a variant spills (scratch traffic would
//     join the counted stream unseen).
// Results are identical to gemm_nt_256 / gemm_nt_128 (same MFMA, same K order, fp32 epilogue), except that bias and
// residual enter the fp32 accumulation early instead of last.
// Shapes: K % 64 == 0, K*passes >= 1024, N % 4 == 0 (gemm_stream_ok); everything else stays on the tile kernels.
#include <type_traits>

#include "pio_gemm_common.h"

namespace pio {

static __device__ __attribute__((aligned(16))) uint32_t g_zero_s[4] = {0, 0, 0, 0};
// Stores of lanes outside the matrix (and of the first tile's empty predecessor) go here instead of being
// predicated off: the store instruction is then ALWAYS issued, so the counted waits can count it.
static __device__ __attribute__((aligned(16))) uint32_t g_sink_s[64 * 4];

#ifdef PIO_GEMM_STAMPS
// Dev-only (tools/gemm_stamps.py): wave 0 of workgroup 0 records s_memtime inside one unrolled step with epilogue
// work (row 0: KT == 9 of a tile that has a predecessor) and inside the late steps of a long K (row 1).
__device__ unsigned long long g_sstamps[2][8];
// ... and, around the whole kernel, s_memtime (shader cycles) and s_memrealtime (100 MHz): their ratio is the clock the
// chip holds under this kernel's load (MI355X_MICROARCH.md, "DVFS give-back" item 6).
__device__ unsigned long long g_sclk[4];
__device__ int g_smode;  // ablations: bit 0 = no fragment reads / MFMAs, bit 1 = no DMA after the prologue,
                         // bit 2 = residual loads from one small (cache-resident) region, bit 3 = no residual loads,
                         // bit 4 = no fragment reads but the MFMAs still run
#define PIO_SSTAMP(i)                                                                                         \
    do {                                                                                                      \
        if ((KT == 9 ? has_prev : KT == 16) && blockIdx.x == 0 && threadIdx.x == 0)                           \
            g_sstamps[KT == 16][i] = __builtin_readcyclecounter();                                            \
    } while (0)
#else
#define PIO_SSTAMP(i)
#endif

constexpr int S_BM = 256, S_BN = 128, S_BK = 64, S_NST = 3;
constexpr int S_AB = S_BM * S_BK * 2;     // 32 KiB of A per stage
constexpr int S_BB = S_BN * S_BK * 2;     // 16 KiB of B per stage
constexpr int S_STAGE = S_AB + S_BB;      // 48 KiB
constexpr int S_RING = S_NST * S_STAGE;   // 144 KiB
constexpr int S_BIAS_W = 768;             // per wave: 3 slots x 64 floats (bias rows of tiles j, j+1, j+2)
constexpr int S_TAB_N = 256;            // tile-table entries (32 B each): this workgroup's tile list, decoded once
constexpr int S_TAB = S_RING + 8 * S_BIAS_W;
constexpr int S_SMEM = S_TAB + S_TAB_N * 32;

__device__ __forceinline__ void wait_vm_n(int n) {  // n: wave-uniform
#define PIO_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        PIO_W(0) PIO_W(1) PIO_W(2) PIO_W(3) PIO_W(4) PIO_W(5) PIO_W(6) PIO_W(7) PIO_W(8) PIO_W(9) PIO_W(10)
        PIO_W(11) PIO_W(12) PIO_W(13) PIO_W(14) PIO_W(15) PIO_W(16) PIO_W(17) PIO_W(18) PIO_W(19)
        default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    }
#undef PIO_W
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int64_t uniform64(int64_t v) {  // a value every lane holds -> scalar registers
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void lds_dma16(const void *src, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

struct STile {  // wave-uniform description of one output tile
    int64_t a_off, b_off, c_off;  // element offsets of the (batch, head) slice in A / B / C
    int m0, n0;
};

__device__ __forceinline__ STile stile(const GemmParams &p, int t, int tiles_mn, int tiles_n) {
    const int z = t / tiles_mn;
    const int r = t - z * tiles_mn;
    const int tm = r / tiles_n;
    const int tn = r - tm * tiles_n;
    const int zb = z / p.nh, zh = z - zb * p.nh;
    STile s;
    s.a_off = zb * p.sAb + zh * p.sAh;
    s.b_off = zb * p.sBb + zh * p.sBh;
    s.c_off = zb * p.sCb + zh * p.sCh;
    s.m0 = tm * S_BM;
    s.n0 = tn * S_BN;
    return s;
}

// OUT: 0 = 16-bit C, 1 = 16-bit C + C_lo (hi/lo split of the fp32 result), 2 = fp32 C
template <int DT, bool HAS_R, int ACT, int OUT>
__global__ __launch_bounds__(512) void gemm_nt_stream(const GemmParams p, int tiles_m, int tiles_n, int nz) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    __shared__ __attribute__((aligned(16))) char smem[S_SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, wm = wave & 3;  // team doubles as the wave's N half

    // ---- this workgroup's tile list: the 8 XCDs each own a contiguous run of tile ids (tile_n fastest), and the
    // workgroups of an XCD take ids of that run round-robin, so the tiles in flight in one L2 share A / B panels.
    // (Walking a run in rounds of 4 x 8 tiles instead -- fewer distinct B panels per round, A panels resident across
    // rounds -- cut the L2 fetch traffic of the fused q|k|v projection but ran 4 % slower: the round-robin order has
    // more tiles reading the same lines at the same time.)
#ifdef PIO_GEMM_STAMPS
    const int smode = __builtin_amdgcn_readfirstlane(g_smode);  // (read once: a load per step would distort the phases)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_sclk[0] = __builtin_amdgcn_s_memtime();
        g_sclk[1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const int tiles_mn = tiles_m * tiles_n, total = tiles_mn * nz;
    const int G = gridDim.x, bx = blockIdx.x;
    int first, stride, end;
    int rn = 0, ngrp = 1, id0 = 0, idm = 0;  // blocked walk (rn > 0): id(j) = id0 + (j / ngrp) * idm + (j % ngrp) * rn
    if ((G & 7) == 0) {
        const int run = (total + 7) >> 3;
        first = (bx & 7) * run + (bx >> 3);
        stride = G >> 3;
        end = (bx & 7) * run + run;
        end = end < total ? end : total;
        // rounds of RM x RN tiles (RN = up to 8 tile columns, RM = C / RN tile rows), all column groups of a row
        // group before the next row group
        const int C = G >> 3, RN = tiles_n < 8 ? tiles_n : 8;
        if (nz == 1 && (total & 7) == 0 && run % tiles_n == 0 && C % RN == 0 && tiles_n % RN == 0 &&
            (run / tiles_n) % (C / RN) == 0) {
            rn = RN;
            ngrp = tiles_n / RN;
            id0 = ((bx & 7) * (run / tiles_n) + (bx >> 3) / RN) * tiles_n + (bx >> 3) % RN;
            idm = (C / RN) * tiles_n;
        }
    } else {
        first = bx;
        stride = G;
        end = total;
    }
    if (first >= end) return;
    const int ntl = (end - first + stride - 1) / stride;
    // The list is decoded ONCE into an LDS table (every division below costs a wave ~150 cycles of scalar latency,
    // and the steps that look a tile up sit on the critical path of all eight waves).
    auto tile_id = [&](int j) {  // id of this workgroup's j-th tile
        if (rn > 0) {
            const int mg = j / ngrp;
            return id0 + mg * idm + (j - mg * ngrp) * rn;
        }
        return first + j * stride;
    };

    STile *table = (STile *)(smem + S_TAB);
    for (int j = tid; j < ntl; j += 512) table[j] = stile(p, tile_id(j), tiles_mn, tiles_n);
    __syncthreads();
    // field readers (j < ntl <= S_TAB_N: launcher); wave-uniform addresses, i.e. broadcast reads, moved to SGPRs
    const int *tab32 = (const int *)table;
    const int64_t *tab64 = (const int64_t *)table;
    auto tile_m0 = [&](int j) { return __builtin_amdgcn_readfirstlane(tab32[j * 8 + 6]); };
    auto tile_n0 = [&](int j) { return __builtin_amdgcn_readfirstlane(tab32[j * 8 + 7]); };
    auto tile_off = [&](int j, int which) { return uniform64(tab64[j * 4 + which]); };  // 0: A, 1: B, 2: C

    const int nk1 = (p.K + S_BK - 1) / S_BK;
    const int nk = p.npass * nk1;  // >= 16 (launcher)

    // ---- DMA side: a 1-KiB piece = 8 rows x 128 B; wave w owns A pieces 4w..4w+3 and B pieces 2w, 2w+1.
    // LDS chunk position (lane & 7) of row r holds source chunk (lane & 7) ^ ((r >> 1) & 7).
    const int lrow = lane >> 3;
    const int ce = ((lane & 7) ^ (lane >> 4)) * 8;  // source element inside the 64-wide step, even pieces
    const int co = ce ^ 32;                           // odd pieces ((r >> 1) & 4 set)
    int dj = 0, dkt = 0, dslot = 0;                   // next stage to issue: tile index, K step, ring slot
    const T *dA = nullptr, *dB = nullptr;             // (batch, head) slice of the DMA tile (uniform)
    uint32_t voa[4], vob[2];                          // per-lane byte offsets of this lane's rows + chunk (< 2^32)
    auto dma_tile = [&](int j) {
        dA = (const T *)p.A + tile_off(j, 0);
        dB = (const T *)p.B + tile_off(j, 1);
        const int tm0 = tile_m0(j), tn0 = tile_n0(j);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int gm = tm0 + (wave * 4 + i) * 8 + lrow;
            gm = gm < p.M ? gm : p.M - 1;
            voa[i] = ((uint32_t)gm * (uint32_t)p.lda + (uint32_t)((i & 1) ? co : ce)) * 2u;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int gn = tn0 + (wave * 2 + i) * 8 + lrow;
            gn = gn < p.N ? gn : p.N - 1;
            vob[i] = ((uint32_t)gn * (uint32_t)p.ldb + (uint32_t)((i & 1) ? co : ce)) * 2u;
        }
    };
    auto issue = [&](auto WC) {  // WC: may this stage be the last one of a tile (nk >= 16: never before step 13)
        const int pass = (dkt >= nk1) + (dkt >= 2 * nk1);
        const int k0 = (dkt - pass * nk1) * S_BK;
        const int64_t da = pass == 0 ? 0 : (pass == 1 ? p.dA1 : p.dA2);
        const int64_t db = pass == 0 ? 0 : (pass == 1 ? p.dB1 : p.dB2);
        const char *ua = (const char *)(dA + k0 + da);  // wave-uniform bases: the DMA is "sgpr base + vgpr offset"
        const char *ub = (const char *)(dB + k0 + db);
        char *sb = smem + dslot * S_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(ua + (uint64_t)voa[i], sb + (wave * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(ub + (uint64_t)vob[i], sb + S_AB + (wave * 2 + i) * 1024);
        dslot = dslot == S_NST - 1 ? 0 : dslot + 1;
        ++dkt;
        if constexpr (decltype(WC)::value) {
            if (dkt == nk) {
                dkt = 0;
                if (++dj < ntl) dma_tile(dj);
            }
        }
    };

    // ---- MFMA side: fragment row = base16 + (lane & 15), K chunk = ks*4 + (lane >> 4), swizzled by (row >> 1) & 7
    const int frow = lane & 15, fq = lane >> 4;
    const int fo0 = frow * 128 + ((fq ^ (frow >> 1)) << 4);
    const int fo1 = frow * 128 + (((4 + fq) ^ (frow >> 1)) << 4);
    const int a_base = wm * 64 * 128;
    const int b_base = S_AB + team * 64 * 128;
    V8 af[4][2], bf[4][2];
    auto load_frags = [&](int slot) {
        const char *sb = smem + slot * S_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf[i][0] = *(const V8 *)(sb + b_base + i * 2048 + fo0);
            bf[i][1] = *(const V8 *)(sb + b_base + i * 2048 + fo1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const V8 *)(sb + a_base + i * 2048 + fo0);
            af[i][1] = *(const V8 *)(sb + a_base + i * 2048 + fo1);
        }
    };

    // ---- epilogue side.  Unit (mi, ni) of a wave = rows m0 + wm*64 + mi*16 + (lane & 15),
    // 4 columns from n0 + team*64 + ni*16 + (lane >> 4)*4.  Per tile and lane: row / column of unit (0,0) and the
    // element offset of that position in C (o_*: the tile whose result is leaving).
    float *bstash = (float *)(smem + S_RING + wave * S_BIAS_W);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int lm = wm * 64 + frow, ln = team * 64 + fq * 4;
    int o_m = 0, o_n = 0;
    int64_t o_off = 0;
    auto out_tile = [&](int j) {
        o_m = tile_m0(j) + lm;
        o_n = tile_n0(j) + ln;
        o_off = tile_off(j, 2) + (int64_t)o_m * p.ldc + o_n;
    };
    // The bias is not an epilogue operand either: bias/alpha is added to the accumulators of a tile in the load
    // phases of its steps 1..8 (two units per step), so a leaving unit needs alpha, the activation and the
    // conversion only.  The bias row of tile j+2 is parked in the stash (slot (j+2) % 3) by one DMA at step 13 of
    // tile j.
    const float inv_alpha = 1.0f / p.alpha;  // (alpha != 0: launcher)
    auto clamp_j = [&](int j) { return j < ntl ? j : ntl - 1; };  // (a tile past the list: the last one again)
    auto bias_dma = [&](int j, int slot) {  // one DMA: this wave's 64 bias values of tile j (zeros if no bias)
        const int n0 = tile_n0(clamp_j(j));
        // the lane id is recomputed here from an opaque zero, so that nothing lane-dependent has to be kept in a
        // register between two tiles for this once-per-tile DMA (the residual variant has no register to spare)
        int zero = 0;
        asm volatile("" : "+v"(zero));
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        if (lane < 16) {
            const int n = n0 + team * 64 + lane * 4;
            const float *src = (p.bias_mode == 1 && n < p.N) ? p.bias + n : (const float *)g_zero_s;
            lds_dma16(src, bstash + slot * 64);
        }
    };
    auto bias_of = [&](int slot, int ni) { return *(const f32x4 *)(bstash + slot * 64 + ni * 16 + fq * 4); };
    int bslot = 0;  // stash slot of the CURRENT tile's bias row (j % 3); tile j+2's row goes to (bslot + 2) % 3
    char *const sink = (char *)g_sink_s + lane * 16;
    auto store_unit = [&](f32x4 v, int mi, int ni, bool live) {
        const int m = o_m + mi * 16, n = o_n + ni * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = HAS_R ? v[r] : v[r] * p.alpha;  // (alpha == 1 with a residual: launcher)
            if constexpr (ACT == 1) x = gelu_erf(x);
            v[r] = x;
        }
        if (n >= p.N) v = zero4;  // columns [N, n_store) are written as zeros
        const bool ok = live && m < p.M && n < p.n_store;
        const int64_t o = o_off + (int64_t)(mi * 16) * p.ldc + ni * 16;
        if constexpr (OUT == 2) {
            *(f32x4 *)(ok ? (char *)((float *)p.C + o) : sink) = v;
        } else {
            V4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = Op<DT>::from_f32(v[r]);
            *(V4 *)(ok ? (char *)((T *)p.C + o) : sink) = h;
            if constexpr (OUT == 1) {
                V4 l;
#pragma unroll
                for (int r = 0; r < 4; ++r) l[r] = Op<DT>::from_f32(v[r] - Op<DT>::to_f32(h[r]));
                *(V4 *)(ok ? (char *)((T *)p.C_lo + o) : sink) = l;
            }
        }
    };
    int nx_m0 = tile_m0(0), nx_n0 = tile_n0(0);  // origin of the tile whose residual is being loaded (the next one)
    auto r_load = [&](f32x4 &dst, int mi, int ni) {  // accumulator <- residual of the next tile (clamped: always valid)
        int m = nx_m0 + lm + mi * 16;
        m = m < p.M ? m : p.M - 1;
        int n = nx_n0 + ln + ni * 16;
        n = n < p.N ? n : 0;
        const float *rrow = p.R + (int64_t)m * p.ldr;  // (r_rows == 0: launcher)
#ifdef PIO_GEMM_STAMPS
        const float *src = (smode & 4) ? p.R + (lm & 63) * p.ldr + (n & 127) : rrow + n;  // ablation: cache-resident residual
#else
        const float *src = rrow + n;
#endif
#ifdef PIO_GEMM_STAMPS
        if (smode & 8) return;  // ablation: no residual loads at all (wrong results, the loads are not counted either)
#endif
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
    };

    f32x4 acc[2][4][4];
    int cslot = 0;
    int s_prev = 0;  // stores this wave issued in its previous MFMA phase (they sit between two stages' DMA)

    // ---- prologue: bias rows of tiles 0 and 1, accumulators of tile 0 (and of tile 1 without a residual: with one
    // they are loaded during tile 0), stages 0 and 1
    bias_dma(0, 0);
    bias_dma(1, 1);
    if constexpr (HAS_R) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) r_load(acc[0][mi][ni], mi, ni);
    }
    dma_tile(0);
    issue(std::false_type{});
    issue(std::false_type{});  // nk >= 16
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // stage 0 and everything issued before it has landed
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if constexpr (!HAS_R) acc[0][mi][ni] = zero4;
            acc[1][mi][ni] = zero4;
        }
    __builtin_amdgcn_s_barrier();
    // (GELU variant: its MFMA phase, ~1000 cycles with the activation among the MFMAs, is the longer one; there the
    //  classic two meeting points per step -- team 1 half a step behind -- measured 5 % faster than one)
    constexpr bool ONE_BARRIER = ACT == 0;
    if (!ONE_BARRIER && team == 1) __builtin_amdgcn_s_barrier();

    // One K step.  KT (compile time) = position inside the tile for the first 16 steps -- it fixes which
    // accumulator registers leave / are loaded in this step, so every register index is static -- and 16 for the
    // later steps of a long K, which carry no epilogue work.
    auto step = [&](auto PC, auto KC, bool has_prev, int j) {
        constexpr int P = decltype(PC)::value, Q = P ^ 1, KT = decltype(KC)::value;
        // Units of the previous tile leaving in this step: one per step (0..15); with a residual one in steps 0..13
        // and two in step 14 (where most of that register set is already free), each followed by the load of the
        // NEXT tile's residual into the registers it vacated -- so every residual load is at least one full step
        // old at the end of step 15.
        constexpr int ND = HAS_R ? (KT <= 13 ? 1 : (KT == 14 ? 2 : 0)) : (KT < 16 ? 1 : 0);
        constexpr int D0 = KT;
        constexpr int NR = HAS_R ? ND : 0;
        constexpr int NR_PREV = (HAS_R && KT >= 1 && KT <= 14) ? 1 : 0;
        // bias joins THIS tile: two units per step in steps 1..8; with a residual (tightest on registers) one
        // unit per step in steps 0..15
        constexpr bool BA = HAS_R ? KT < 16 : (KT >= 1 && KT <= 8);
        constexpr int B0 = HAS_R ? KT : 2 * (KT - 1);
        constexpr int E_NOW = (KT == 13 ? 1 : 0) + NR;                    // loads issued after this step's stage
        constexpr int E_PREV = (KT == 14 ? 1 : 0) + NR_PREV;
        constexpr int E_BIAS = (KT == 13 ? 1 : 0);                        // ... of them before the load-phase wait
        // ================= LOAD phase (the other team of this SIMD owns the matrix pipe and, with its raised
        // priority, nearly all vector issue: VALU work here costs ~20 cycles an instruction, so this phase is LDS
        // reads and DMA only -- everything else rides in this wave's own MFMA phase)
        PIO_SSTAMP(0);
#ifdef PIO_GEMM_STAMPS
        const bool iss = dj < ntl && !(smode & 2);
        const bool rd = !(smode & 17);  // bit 4: MFMAs run on whatever the fragment registers hold
#else
        const bool iss = dj < ntl;
        const bool rd = true;
#endif
        if (rd) load_frags(cslot);
        if (iss) issue(std::integral_constant<bool, (KT >= 13)>{});
        if constexpr (KT == 13) bias_dma(j + 2, bslot == 0 ? 2 : bslot - 1);  // (j + 2) % 3
        PIO_SSTAMP(1);
        PIO_SSTAMP(2);
        // loads AND stores issued after the stage needed next (stage g+1, issued one load phase ago): the previous
        // MFMA phase's stores and residual loads, this phase's stage and bias row
        // (step 15 with a residual is STRICT: the residual loads of step 14 -- and with them everything older --
        //  must have landed before the next tile multiplies into those registers)
        constexpr bool STRICT = HAS_R && KT == 15;
        const int allow1 = (STRICT ? 0 : E_PREV + s_prev) + (iss ? 6 : 0) + E_BIAS;
        if (team == 1) wait_vm_n(allow1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIO_SSTAMP(3);
        // ONE barrier per step and wave: team 1 meets team 0 here, after its load phase, team 0 after its MFMA phase
        // (below).  Both teams run the same instruction stream LOAD(0) MFMA(0) LOAD(1) MFMA(1) ...; cutting it at
        // different points puts them half a step apart: between two barriers team 0 runs LOAD(g) MFMA(g) while team
        // 1 runs MFMA(g-1) LOAD(g) -- one team's MFMAs beside the other's loads without a second meeting point.
        // Stage g+1 is complete before the barrier (both teams waited for their pieces), and the slot refilled in
        // LOAD(g) held stage g-1, whose last reads (team 1's LOAD(g-1)) finished before the previous barrier.
        if (!ONE_BARRIER || team == 1) __builtin_amdgcn_s_barrier();
        // ================= MFMA phase: 32 MFMAs; beside them (an MFMA holds the vector issue for 8 of its 16 cycles)
        // the bias of two units of this tile, the unit(s) of the previous tile that leave and, into their registers,
        // the residual of the next tile
        PIO_SSTAMP(4);
        __builtin_amdgcn_s_setprio(1);
#ifdef PIO_GEMM_STAMPS
        if (!(smode & 1)) {
#endif
        if constexpr (BA) {
            acc[P][B0 >> 2][B0 & 3] += bias_of(bslot, B0 & 3) * inv_alpha;
            if constexpr (!HAS_R) acc[P][(B0 + 1) >> 2][(B0 + 1) & 3] += bias_of(bslot, (B0 + 1) & 3) * inv_alpha;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[P][mi][ni] = Op<DT>::mfma16(bf[ni][0], af[mi][0], acc[P][mi][ni]);
#pragma unroll
        for (int d = 0; d < ND; ++d) {  // (no branch on has_prev: the first tile's "predecessor" goes to the sink)
            store_unit(acc[Q][(D0 + d) >> 2][(D0 + d) & 3], (D0 + d) >> 2, (D0 + d) & 3, has_prev);
            if constexpr (!HAS_R) acc[Q][(D0 + d) >> 2][(D0 + d) & 3] = zero4;
        }
        // With a residual the epilogue arithmetic stays among the FIRST sixteen MFMAs (letting it spread over all 32
        // costs the registers this variant does not have), and the residual loads come LAST: inline asm is a
        // scheduling boundary, in the middle of the phase it kept the arithmetic out from among the MFMAs altogether.
        if constexpr (HAS_R) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[P][mi][ni] = Op<DT>::mfma16(bf[ni][1], af[mi][1], acc[P][mi][ni]);
        if constexpr (HAS_R) {
#pragma unroll
            for (int d = 0; d < ND; ++d) r_load(acc[Q][(D0 + d) >> 2][(D0 + d) & 3], (D0 + d) >> 2, (D0 + d) & 3);
        }
#ifdef PIO_GEMM_STAMPS
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        PIO_SSTAMP(5);
        constexpr int S_NOW = ND * (OUT == 1 ? 2 : 1);
        if (team == 0) wait_vm_n(allow1 + S_NOW + (E_NOW - E_BIAS));
        s_prev = S_NOW;
        PIO_SSTAMP(6);
        if (!ONE_BARRIER || team == 0) __builtin_amdgcn_s_barrier();
        PIO_SSTAMP(7);
        cslot = cslot == S_NST - 1 ? 0 : cslot + 1;
    };
    auto run_tile = [&](auto PC, int j) {
        const bool has_prev = j > 0;
        // (without a next tile the residual loads still run, from the last tile's addresses into the idle
        //  registers: the number of loads per step stays fixed, which is what the counted waits assume)
        if constexpr (HAS_R) {
            nx_m0 = tile_m0(clamp_j(j + 1));
            nx_n0 = tile_n0(clamp_j(j + 1));
        }
        static_for<0, 16>([&](auto kc) { step(PC, kc, has_prev, j); });
#pragma unroll 1
        for (int kt = 16; kt < nk; ++kt) step(PC, std::integral_constant<int, 16>{}, false, j);
        out_tile(j);  // this tile's result leaves during the next tile (or in the tail)
        bslot = bslot == 2 ? 0 : bslot + 1;
    };

    for (int j = 0; j < ntl; j += 2) {
        run_tile(std::integral_constant<int, 0>{}, j);
        if (j + 1 < ntl) run_tile(std::integral_constant<int, 1>{}, j + 1);
    }
    if (!ONE_BARRIER && team == 0) __builtin_amdgcn_s_barrier();  // pairs with team 1's last barrier

    // ---- the last tile's result leaves without cover
    const int lp = (ntl - 1) & 1;
#define PIO_TAIL(PAR)                                                                                     \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)     \
        store_unit(acc[PAR][mi][ni], mi, ni, true);
    if (lp == 0) { PIO_TAIL(0) } else { PIO_TAIL(1) }
#undef PIO_TAIL
#ifdef PIO_GEMM_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_sclk[2] = __builtin_amdgcn_s_memtime();
        g_sclk[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef PIO_GEMM_STAMPS
extern "C" int pio_debug_stream_mode(int mode) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_smode), &mode, sizeof(int)) == hipSuccess ? 0 : 1;
}
extern "C" int pio_debug_stream_clock(unsigned long long *out4) {
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_sclk), sizeof(g_sclk)) == hipSuccess ? 0 : 1;
}
extern "C" int pio_debug_stream_stamps(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sstamps), sizeof(g_sstamps)) == hipSuccess ? 0 : 1;
}
#endif

static int stream_grid(int64_t total) {  // one workgroup per CU, a multiple of 8 (the XCD count) when possible
    const int n_cu = cu_budget();
    int G = (int)(total < n_cu ? total : n_cu);
    if (G >= 8) G &= ~7;
    return G;
}

bool gemm_stream_ok(const GemmParams &p, int batch) {
    const int nk = p.npass * ((p.K + S_BK - 1) / S_BK);
    if (nk < 16 || (p.K % S_BK)) return false;
    if (p.bias_mode > 1 || (p.bias_mode == 1 && !p.bias_vec)) return false;
    if ((p.N & 3) || (p.n_store & 3) || !p.vec_ok) return false;
    if (p.act != 0 && p.act != 1) return false;
    if (p.alpha == 0.0f) return false;
    // (a residual with a 16-bit result -- the decoder's fc2 leaving y as the operand of the final Linear -- rides the same
    //  way: the residual is loaded into the vacated accumulators, the output form is the epilogue's business)
    if (p.R && (!p.r_vec || p.alpha != 1.0f || p.act != 0 || p.r_rows != 0)) return false;
    // (... as ONE 16-bit array: the hi + lo pair form of that variant spills 11 registers, which the counted waits of this
    //  kernel cannot carry -- those launches go to gemm_nt_256)
    if (p.R && !p.out_f32 && p.C_lo) return false;
    if (p.out_f32 && p.act != 0) return false;
    if (p.out_f32 && p.C_lo) return false;
    // per-lane DMA offsets are 32-bit byte offsets inside one (batch, head) slice
    if (((int64_t)p.M * p.lda + p.K) * 2 >= (1ll << 32) || ((int64_t)p.N * p.ldb + p.K) * 2 >= (1ll << 32)) return false;
    // the per-workgroup tile list is decoded into an LDS table of S_TAB_N entries
    const int64_t tiles = (int64_t)((p.M + S_BM - 1) / S_BM) * ((p.n_store + S_BN - 1) / S_BN) * batch;
    const int G = stream_grid(tiles);
    if ((tiles + G - 1) / G + 8 > S_TAB_N) return false;
    return true;
}

void gemm_stream_launch(const GemmParams &p, int dtype, int batch, hipStream_t s) {
    const int tiles_m = (p.M + S_BM - 1) / S_BM, tiles_n = (p.n_store + S_BN - 1) / S_BN;
    const int G = stream_grid((int64_t)tiles_m * tiles_n * batch);
    dim3 grid((unsigned)G, 1, 1), block(512, 1, 1);
#define PIO_GK(DTV, R, ACT, OUT) hipLaunchKernelGGL((gemm_nt_stream<DTV, R, ACT, OUT>), grid, block, 0, s, p, tiles_m, tiles_n, batch)
#define PIO_GS(DTV)                                                      \
    if (p.R && p.out_f32) PIO_GK(DTV, true, 0, 2);                       \
    else if (p.R) PIO_GK(DTV, true, 0, 0);                               \
    else if (p.out_f32) PIO_GK(DTV, false, 0, 2);                        \
    else if (p.act == 1 && p.C_lo) PIO_GK(DTV, false, 1, 1);             \
    else if (p.act == 1) PIO_GK(DTV, false, 1, 0);                       \
    else if (p.C_lo) PIO_GK(DTV, false, 0, 1);                           \
    else PIO_GK(DTV, false, 0, 0);
    if (dtype == PIO_DT_F16) { PIO_GS(PIO_DT_F16) } else { PIO_GS(PIO_DT_BF16) }
#undef PIO_GK
#undef PIO_GS
}

}  // namespace pio
