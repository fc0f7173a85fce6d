// NT GEMM on gfx950 matrix cores:  C = epilogue(alpha * A B^T)
//   A [M,K], B [N,K]: 16-bit operands (fp16 / bf16), K contiguous -- exactly the layout of an
//   nn.Linear weight ([out,in], transformer_primitives.py:73-75) and of a row-major activation, so
//   neither operand is ever transposed in memory.
//   Replaces F.linear / matmul at transformer_primitives.py:93-95, 110, 138, 163, 213-215.
//
// v1 structure (128x128x64 tile, 4 waves as 2x2, one 64x64 sub-tile per wave, MFMA 16x16x32):
//   * both operand tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4): the LDS image is lane-linear,
//     the XOR swizzle is applied to the per-lane SOURCE address and again on the ds_read_b128 side;
//   * rows past M / N are clamped (their results are never stored), 8-element K chunks past K read a
//     16-byte zero block, so no operand needs padding beyond a multiple of 8;
//   * double-buffered, ONE barrier per K step: the DMA of tile t+1 is in flight while tile t is multiplied;
//   * the MFMA is issued as (B-fragment, A-fragment) so that each lane ends up with 4 CONSECUTIVE output
//     columns of one row: bias / residual / stores are 8- or 16-byte vectors;
//   * optional second K sweep with one operand replaced by its *_lo image (W = W_hi + W_lo, the "x2w"
//     precision policy) accumulating into the same registers.
#include "pio_gemm_common.h"

namespace pio {

__device__ __attribute__((aligned(16))) uint32_t g_zero_chunk[4] = {0, 0, 0, 0};

#ifdef PIO_G128_STAMPS
// Dev-only (tools/g128_stamps.py builds a private copy of the library with this flag): thread 0 of one workgroup records
// s_memtime at four points of each of the first 30 K steps (in LDS: stores would count in vmcnt), and at the kernel's edges.
__device__ unsigned long long g128_stamps[128];
__device__ int g128_block;
#define G128_STAMP(i)                                                                  \
    do {                                                                               \
        if (stamp_on && (i) < 128) stamp_lds[i] = __builtin_readcyclecounter();        \
    } while (0)
#else
#define G128_STAMP(i)
#endif

constexpr int BM = 128, BN = 128, BK = 64;   // the default tile (launcher arithmetic); the kernel is templated on TM x TN

// KIND only tags the instantiation (0: one flat [rows,K]x[N,K] linear, 1: batched attention product) so that
// profilers report the two uses under different kernel names.
// TM x TN: 128 x 128 (four waves of 64 x 64) or 64 x 64 (four waves of 32 x 32: problems whose 128 x 128 tiling would
// leave most of the 256 CUs idle -- a 2048-row latent stack at batch 1 has 64 such tiles per GEMM).
// NS: operand stages in LDS.  2 = double buffer, the DMA of step t+1 against the MFMAs of step t (two workgroups per CU
// hide each other's waits: the 128 x 128 tile on problems with many tiles).  4 (the 64 x 64 tile: problems of at most a
// tile per CU, where nobody else covers a wait) = a RING: the pieces of step t+3 are issued at the top of step t, a
// counted s_waitcnt vmcnt leaves two steps in flight across a raw s_barrier, every wave issues the same number of pieces
// per step (steps past the end re-read valid memory into the slot the ring would use next: never read) -- with the
// double buffer a 64 x 64 x 1024 tile spent ~1.9 k cycles per 64-deep step (the latency of one LDS-DMA round trip under
// load) on 128 cycles of MFMAs.  (A six-stage ring -- four steps in flight, 96 KiB, one workgroup per CU -- measured no
// better than four stages with two workgroups per CU: DESIGN_LOG R3.3.)
// KG: K groups.  2 = EIGHT waves, two teams of four that each run the whole loop below on every other 64-deep K step
// (own ring, own accumulators; the row pass sums the two epilogue images).  An LDS-DMA piece of 16 bytes a lane holds its
// wave for ~150 cycles (tools/g128_stamps.py: the 3-4 pieces of a step are ~440 of its ~700 cycles, four waves together
// reach 27 B/clk where tools/dma_bench.hip needs eight for the path's 54): a tile that owns its CU alone gets the second
// team instead of a second workgroup.
template <int DT, int KIND, int TM, int TN, int NS, int KG = 1>
__global__ __launch_bounds__(256 * KG) void gemm_nt_128(const GemmParams p) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    constexpr int A_BYTES = TM * BK * 2, B_BYTES = TN * BK * 2;     // operand tiles: rows x 128 bytes
    constexpr int A_PPW = TM / 32, B_PPW = TN / 32;                  // 1-KiB pieces (8 rows) per wave
    constexpr int MI = TM / 32, NI = TN / 32;                        // 16 x 16 units per wave (waves as 2 x 2)
    constexpr int EPI_BYTES = TM * TN * 4;
    // (+ TM x (rstd, -mean rstd) behind the epilogue image: the LayerNorm fold's consumer)
    constexpr int TEAM_BYTES = NS * (A_BYTES + B_BYTES);           // one team's ring
    constexpr int SMEM = KG * (TEAM_BYTES > EPI_BYTES ? TEAM_BYTES : EPI_BYTES) + TM * 8;
    static_assert(NS == 2 || (NS - 2) * (A_PPW + B_PPW) <= 63, "the counted wait must fit vmcnt");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];        // A stages, B stages (then the epilogue image)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);   // inside the team
    const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 8);
    char *const ring = smem + kg * TEAM_BYTES;
    const int wm = wave >> 1, wn = wave & 1;
#ifdef PIO_G128_STAMPS
    __shared__ unsigned long long stamp_lds[128];
    const bool stamp_on = (int)blockIdx.x == g128_block && blockIdx.y == 0 && tid == 0;
    G128_STAMP(120);
#endif

    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int z = blockIdx.y;
    const int zb = z / p.nh, zh = z % p.nh;

    const T *A = (const T *)p.A + zb * p.sAb + zh * p.sAh;
    const T *B = (const T *)p.B + zb * p.sBb + zh * p.sBh;

    // ---- staging addresses: wave w, piece i covers tile rows (w*4+i)*8 .. +8, lane -> (row, 16-B slot)
    const int srow = lane >> 3;
    const int sslot = lane & 7;
    const T *a_src[A_PPW];
    const T *b_src[B_PPW];
    int a_koff[A_PPW], b_koff[B_PPW];
#pragma unroll
    for (int i = 0; i < A_PPW; ++i) {
        const int r = (wave * A_PPW + i) * 8 + srow;       // row inside the tile
        const int c = sslot ^ ((r >> 1) & 7);              // source chunk that lands in this slot
        a_koff[i] = c * 8;
        int gm = tile_m * TM + r;
        gm = gm < p.M ? gm : p.M - 1;
        a_src[i] = A + (int64_t)gm * p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < B_PPW; ++i) {
        const int r = (wave * B_PPW + i) * 8 + srow;
        const int c = sslot ^ ((r >> 1) & 7);
        b_koff[i] = c * 8;
        int gn = tile_n * TN + r;
        gn = gn < p.N ? gn : p.N - 1;
        b_src[i] = B + (int64_t)gn * p.ldb + c * 8;
    }
    const T *zsrc = (const T *)g_zero_chunk;

    const int nk1 = (p.K + BK * KG - 1) / (BK * KG);   // steps per pass (of one team)
    const int nk = p.npass * nk1;

    auto stage = [&](int kt, int buf) {
        const int pass = (kt >= nk1) + (kt >= 2 * nk1);   // wave-uniform
        const int k0 = ((kt - pass * nk1) * KG + kg) * BK;   // (a team's step past K reads the zero chunk)
        const int64_t dA = pass == 0 ? 0 : (pass == 1 ? p.dA1 : p.dA2);
        const int64_t dB = pass == 0 ? 0 : (pass == 1 ? p.dB1 : p.dB2);
        char *abase = ring + buf * A_BYTES;
        char *bbase = ring + NS * A_BYTES + buf * B_BYTES;
#pragma unroll
        for (int i = 0; i < A_PPW; ++i) {
            const T *sa = (k0 + a_koff[i]) < p.K ? (a_src[i] + k0 + dA) : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sa,
                                             (__attribute__((address_space(3))) void *)(abase + (wave * A_PPW + i) * 1024),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_PPW; ++i) {
            const T *sb = (k0 + b_koff[i]) < p.K ? (b_src[i] + k0 + dB) : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)sb,
                                             (__attribute__((address_space(3))) void *)(bbase + (wave * B_PPW + i) * 1024),
                                             16, 0, 0);
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- fragment read addresses (bytes inside a tile): row = base16 + (lane&15), chunk = ks*4 + (lane>>4)
    const int frow = lane & 15;
    const int fswz = (frow >> 1) & 7;
    const int fchunk = lane >> 4;
    int a_off[MI], b_off[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) a_off[i] = (wm * (TM / 2) + i * 16 + frow) * 128;
#pragma unroll
    for (int i = 0; i < NI; ++i) b_off[i] = (wn * (TN / 2) + i * 16 + frow) * 128;

    // Epilogue operands fetched HERE, ahead of the ring (older than every DMA piece: the counted waits still hold): the
    // bias / LayerNorm-fold column constants of this thread's four columns and, for the fold's consumer, the row's partial
    // sums -- on the small tiles their round trips (two dependent ones: statistics -> barrier -> constants) were ~1.5 us of
    // a ~9 us launch, exposed behind the last K step.
    constexpr int CPR_ = TN / 4;
    const int pre_n0 = tile_n * TN + (tid & (CPR_ - 1)) * 4;
    f32x4 bias_pre = {0.f, 0.f, 0.f, 0.f}, lnc_pre = {0.f, 0.f, 0.f, 0.f};
    constexpr int MAXS = 24;                     // statistics slots held in registers (K <= 1536 in 64-column slots)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 part_pre[MAXS];
    // (branch-free: whole 16-byte loads of interior columns, slot indices clamped -- a conditional load costs a branch and
    //  a drained vmcnt each; edge tiles and unaligned operands fetch in the epilogue as before)
    // (only the 32 x 64 tile: same-box A/B of the ImageNet forward, graph replay, B = 1 2.940 -> 2.899 ms; with it on the
    //  64 x 64 tile B = 4 4.66 -> 4.73, B = 8 7.20 -> 7.23 -- two workgroups per CU cover each other's epilogue, the 48
    //  registers only cost)
    constexpr bool PRE = TM == 32;
    const bool pre_cols = PRE && tid < 256 && pre_n0 + 3 < p.N && (p.bias_mode != 1 || p.bias_vec) &&
                          (!p.ln_part || ((uintptr_t)p.ln_c & 15) == 0);
    const bool pre_stat = PRE && p.ln_part && p.ln_slots <= MAXS && ((uintptr_t)p.ln_part & 7) == 0;
    if (pre_cols) {
        if (p.bias_mode == 1) bias_pre = *(const f32x4 *)(p.bias + pre_n0);
        if (p.ln_part) lnc_pre = *(const f32x4 *)(p.ln_c + pre_n0);
    }
    if (pre_stat && tid < TM) {
        int m = tile_m * TM + tid;
        m = m < p.M ? m : p.M - 1;
        const f32x2 *pr = (const f32x2 *)(p.ln_part + (int64_t)m * p.ln_slots * 2);
        const int last = p.ln_slots - 1;
#pragma unroll
        for (int i = 0; i < MAXS; ++i) part_pre[i] = pr[i < last ? i : last];
    }
    // (nothing of the above may sink below a DMA piece: a younger plain load would shift the window of the counted waits)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NS == 2) stage(0, 0);
    else {
        // ring prologue: steps 0 .. NS-2 in flight (a step past the end repeats step 0's valid sources: the counts stay
        // exact, the slot is never read)
#pragma unroll
        for (int j = 0; j < NS - 1; ++j) stage(j < nk ? j : 0, j);
    }
    int slot = 0;  // (NS > 2) ring slot of step kt
    G128_STAMP(121);
    for (int kt = 0; kt < nk; ++kt) {
        G128_STAMP(kt < 30 ? kt * 4 : 127);
        const int cur = NS == 2 ? (kt & 1) : slot;
        const char *abase = ring + cur * A_BYTES;
        const char *bbase = ring + NS * A_BYTES + cur * B_BYTES;
        V8 af[2][MI], bf[2][NI];
        // the step's fragment reads go out BEFORE the next stage's DMA pieces (whose issue holds the wave for ~130 cycles a
        // piece with all four waves at the TA together: tools/g128_stamps.py), so the LDS latency passes under the issue
        auto frags = [&]() {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int coff = ((ks * 4 + fchunk) ^ fswz) << 4;
#pragma unroll
                for (int i = 0; i < MI; ++i) af[ks][i] = *(const V8 *)(abase + a_off[i] + coff);
#pragma unroll
                for (int i = 0; i < NI; ++i) bf[ks][i] = *(const V8 *)(bbase + b_off[i] + coff);
            }
        };
        if constexpr (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            G128_STAMP(kt < 30 ? kt * 4 + 1 : 127);
            frags();
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        } else {
            // all but the youngest NS-2 steps' pieces have landed: step kt is complete in this wave's share, after the
            // barrier in every wave's; and every wave has left step kt-1 (its fragment reads were consumed by its
            // MFMAs), whose slot takes step kt+NS-1.  A raw s_barrier: __syncthreads() would drain vmcnt to zero.
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (A_PPW + B_PPW)) : "memory");
            __builtin_amdgcn_s_barrier();
            G128_STAMP(kt < 30 ? kt * 4 + 1 : 127);
            const int nslot = slot == 0 ? NS - 1 : slot - 1;
            frags();
            G128_STAMP(kt < 30 ? kt * 4 + 3 : 127);   // (the stamp's own lgkmcnt(0) waits for the fragment reads)
            stage(kt + NS - 1 < nk ? kt + NS - 1 : 0, nslot);
        }
        G128_STAMP(kt < 30 ? kt * 4 + 2 : 127);
        if constexpr (NS != 2) slot = slot + 1 == NS ? 0 : slot + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Op<DT>::mfma16(bf[ks][ni], af[ks][mi], acc[mi][ni]);
    }

    // ---- epilogue through LDS: the accumulators (lane = one row, 4 consecutive columns) are parked in a
    // [TM][TN] fp32 image (16-byte chunks XOR-swizzled by row, conflict-free both ways) so that the
    // bias / GELU / residual / store pass walks whole rows: a wave touches 2 rows x 512 contiguous bytes
    // (fp32) or 2 rows x 256 bytes (16-bit) per instruction instead of 16 rows x 64 bytes.
    G128_STAMP(122);
    if constexpr (NS != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the ring's surplus pieces: land before the image)
    __syncthreads();  // every wave is done reading the last operand tile
    G128_STAMP(123);
    float *cs = (float *)smem + kg * (TM * TN);   // (team kg's image; both are summed in the row pass)
    constexpr int CPR = TN / 4;  // 16-byte chunks per image row (32 or 16)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int ml = wm * (TM / 2) + mi * 16 + (lane & 15);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int c = wn * (CPR / 2) + ni * 4 + (lane >> 4);
            *(f32x4 *)(cs + ml * TN + ((c ^ (ml & (CPR - 1))) << 2)) = acc[mi][ni];
        }
    }
    // LayerNorm fold on the small tiles (stacks too short for the 256 x 256 kernel: the flow / multimodal latents, small
    // batches).  Consumer: row m's (mean, rstd) from the ln_slots partial (sum, sum of squares) pairs its producer left,
    // once per tile row, parked behind the image.  Producer (below): residual pair in, result pair out, 64-column sums.
    float *const stat = (float *)(smem + SMEM - TM * 8);
    if (p.ln_part && tid < TM) {
        int m = tile_m * TM + tid;
        m = m < p.M ? m : p.M - 1;
        const float *pr = p.ln_part + (int64_t)m * p.ln_slots * 2;
        float sm = 0.f, sq = 0.f;
        if (pre_stat) {
#pragma unroll
            for (int i = 0; i < MAXS; ++i) {     // (same order of additions as the loop below)
                sm += i < p.ln_slots ? part_pre[i][0] : 0.f;
                sq += i < p.ln_slots ? part_pre[i][1] : 0.f;
            }
        } else {
            for (int i = 0; i < p.ln_slots; ++i) {
                sm += pr[2 * i];
                sq += pr[2 * i + 1];
            }
        }
        const float mean = sm * p.ln_inv_k;
        float var = sq * p.ln_inv_k - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float rstd = 1.0f / sqrtf(var + p.ln_eps);
        stat[2 * tid] = rstd;
        stat[2 * tid + 1] = -mean * rstd;
    }
    __syncthreads();
    if (KG > 1 && tid >= 256) return;   // the row pass belongs to the first team (no barrier below)
    cs = (float *)smem;
    const int64_t coffz = zb * p.sCb + zh * p.sCh;
    const int c = tid & (CPR - 1);
    constexpr int RPI = 256 / CPR;  // rows per pass of the 256 threads (8 or 16)
    const int n0 = tile_n * TN + c * 4;
    if (n0 < p.n_store) {
        const bool nfull = n0 + 3 < p.N;
        f32x4 bias_n = bias_pre, lnc = lnc_pre;   // (fetched ahead of the K loop ...
        if (!pre_cols) {                          //  ... except on edge tiles / unaligned operands)
            if (p.bias_mode == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < p.N) bias_n[r] = p.bias[n0 + r];
            }
            if (p.ln_part) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < p.N) lnc[r] = p.ln_c[n0 + r];
            }
        }
#pragma unroll 4
        for (int it = 0; it < TM / RPI; ++it) {
            const int ml = it * RPI + tid / CPR;
            const int m = tile_m * TM + ml;
            if (m >= p.M) break;  // rows are visited in increasing order
            f32x4 v = *(const f32x4 *)(cs + ml * TN + ((c ^ (ml & (CPR - 1))) << 2));
            if constexpr (KG > 1) {
                const f32x4 v2 = *(const f32x4 *)(cs + TM * TN + ml * TN + ((c ^ (ml & (CPR - 1))) << 2));
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += v2[r];
            }
            if (p.row_part) {
                // ---- fold producer (launcher: N % 64 == 0, n_store == N, 16-bit rows 8-byte aligned): x = acc + bias +
                // residual pair, leaves as the pair (X16, X16_lo) [+ fp32 C] with the row's (sum, sum of squares) over
                // this thread group's 64 columns; the 16 lanes of a group sit in one wave and share the row
                typedef typename Op<DT>::V4 V4;
                const V4 hh = *(const V4 *)((const T *)p.R16_hi + (int64_t)m * p.ld16 + n0);
                const V4 ll = *(const V4 *)((const T *)p.R16_lo + (int64_t)m * p.ld16 + n0);
                V4 h, l;
                float rsum = 0.f, rsq = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = v[r] * p.alpha + bias_n[r] + (Op<DT>::to_f32(hh[r]) + Op<DT>::to_f32(ll[r]));
                    v[r] = x;
                    h[r] = Op<DT>::from_f32(x);
                    l[r] = Op<DT>::from_f32(x - Op<DT>::to_f32(h[r]));
                    rsum += x;
                    rsq += x * x;
                }
                *(V4 *)((T *)p.X16 + (int64_t)m * p.ld16 + n0) = h;
                *(V4 *)((T *)p.X16_lo + (int64_t)m * p.ld16 + n0) = l;
                if (p.C) *(f32x4 *)((float *)p.C + (int64_t)m * p.ldc + n0) = v;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    rsum += __shfl_xor(rsum, o, 64);
                    rsq += __shfl_xor(rsq, o, 64);
                }
                if ((tid & 15) == 0) {
                    float *dst = p.row_part + ((int64_t)m * (p.N >> 6) + (n0 >> 6)) * 2;
                    dst[0] = rsum;
                    dst[1] = rsq;
                    if (p.range_flag && !(rsq <= 3.0e38f)) *p.range_flag = 1;
                }
                continue;
            }
            if (p.ln_part) {
                // ---- fold consumer: LayerNorm(x) W^T + b = rstd (x W'^T) - rstd mean c + b'   (alpha == 1: launcher)
                const float rs = stat[2 * ml], nm = stat[2 * ml + 1];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * rs + nm * lnc[r];
            }
            const float bias_m = (p.bias_mode == 2) ? p.bias[m] : 0.f;
            const float *rrow = nullptr;
            if (p.R) {
                rrow = (p.r_rows > 0)
                           ? p.R + (int64_t)(m / p.r_rows) * p.r_stride_b + (int64_t)(m % p.r_rows) * p.ldr
                           : p.R + (int64_t)m * p.ldr;
            }
            f32x4 rv = {0.f, 0.f, 0.f, 0.f};
            if (rrow) {
                if (nfull && p.r_vec) {
                    rv = *(const f32x4 *)(rrow + n0);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + r < p.N) rv[r] = rrow[n0 + r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = v[r] * p.alpha + bias_n[r] + bias_m;
                if (p.act == 1) x = gelu_erf(x);
                x += rv[r];
                v[r] = (n0 + r < p.N) ? x : 0.f;  // columns [N, n_store) are written as zeros
            }
            const bool sfull = n0 + 3 < p.n_store;
            if (p.out_f32) {
                float *crow = (float *)p.C + coffz + (int64_t)m * p.ldc;
                if (p.vec_ok && sfull) {
                    *(f32x4 *)(crow + n0) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + r < p.n_store) crow[n0 + r] = v[r];
                }
            } else {
                T *crow = (T *)p.C + coffz + (int64_t)m * p.ldc;
                T *lrow = p.C_lo ? (T *)p.C_lo + coffz + (int64_t)m * p.ldc : nullptr;
                typename Op<DT>::V4 h, l;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h[r] = Op<DT>::from_f32(v[r]);
                    l[r] = Op<DT>::from_f32(v[r] - Op<DT>::to_f32(h[r]));
                }
                if (p.vec_ok && sfull) {
                    *(typename Op<DT>::V4 *)(crow + n0) = h;
                    if (lrow) *(typename Op<DT>::V4 *)(lrow + n0) = l;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + r < p.n_store) {
                            crow[n0 + r] = h[r];
                            if (lrow) lrow[n0 + r] = l[r];
                        }
                }
            }
        }
    }
#ifdef PIO_G128_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G128_STAMP(124);
    if (stamp_on)
        for (int i = 0; i < 128; ++i) g128_stamps[i] = stamp_lds[i];
#endif
}

#ifdef PIO_G128_STAMPS
extern "C" int pio_debug_g128_stamps(unsigned long long *out128, int block) {
    if (out128 && hipMemcpyFromSymbol(out128, HIP_SYMBOL(g128_stamps), sizeof(g128_stamps)) != hipSuccess) return 1;
    if (block >= 0 && hipMemcpyToSymbol(HIP_SYMBOL(g128_block), &block, sizeof(int)) != hipSuccess) return 2;
    return 0;
}
#endif

// C[M, n_store <= NMAX] = A B^T for a handful of output columns (the flow decoder's final Linear: 182 528 x 328 -> 2; on
// a 128 x 128 tile 98 % of the MFMAs multiply padding: 234 us for a read of 240 MB).  No matrix pipe: a wave takes one row
// at a time, lane j the 16-byte chunk j of that row (of every K sweep's A image), multiplies it with the same chunk of the
// NMAX weight rows (LDS) in fp32, and the 64 lanes are summed by xor-shuffles.  Bound by the read of A.
template <int DT, int NMAX>
__global__ __launch_bounds__(256) void gemm_nt_skinny(const GemmParams p) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];   // [npass][NMAX][K] weight rows (zeros past N)
    T *const bs = (T *)sk_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = p.K, cpr = K >> 3;   // 16-byte chunks per row (K % 8 == 0: launcher)
    for (int i = tid; i < p.npass * NMAX * cpr; i += 256) {
        const int pass = i / (NMAX * cpr), r = i - pass * NMAX * cpr;
        const int n = r / cpr, c = r - n * cpr;
        const int64_t dB = pass == 0 ? 0 : (pass == 1 ? p.dB1 : p.dB2);
        V8 w;
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = Op<DT>::from_f32(0.f);
        if (n < p.N) w = *(const V8 *)((const T *)p.B + dB + (int64_t)n * p.ldb + c * 8);
        *(V8 *)(bs + ((int64_t)pass * NMAX + n) * K + c * 8) = w;
    }
    __syncthreads();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (tid >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t m = wave0; m < p.M; m += nwaves) {
        float acc[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
        for (int c = lane; c < cpr; c += 64) {
            for (int pass = 0; pass < p.npass; ++pass) {
                const int64_t dA = pass == 0 ? 0 : (pass == 1 ? p.dA1 : p.dA2);
                const V8 a = *(const V8 *)((const T *)p.A + dA + m * p.lda + c * 8);
                if constexpr (DT == PIO_DT_F16) {
                    // v_dot2_f32_f16: two products and the sum in fp32 per instruction
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int n = 0; n < NMAX; ++n) {
                        const V8 w = *(const V8 *)(bs + ((int64_t)pass * NMAX + n) * K + c * 8);
#pragma unroll
                        for (int e = 0; e < 8; e += 2)
                            acc[n] = __builtin_amdgcn_fdot2(h2{a[e], a[e + 1]}, h2{w[e], w[e + 1]}, acc[n], false);
                    }
                } else {
                    float af[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) af[e] = Op<DT>::to_f32(a[e]);
#pragma unroll
                    for (int n = 0; n < NMAX; ++n) {
                        const V8 w = *(const V8 *)(bs + ((int64_t)pass * NMAX + n) * K + c * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[n] = fmaf(af[e], Op<DT>::to_f32(w[e]), acc[n]);
                    }
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[n] += __shfl_xor(acc[n], o, 64);
        if (lane < p.n_store) {
            float x = 0.f;
#pragma unroll
            for (int n = 0; n < NMAX; ++n)
                if (n == lane) x = acc[n];
            x = x * p.alpha + ((p.bias_mode == 1 && lane < p.N) ? p.bias[lane] : 0.f);
            if (lane >= p.N) x = 0.f;   // columns [N, n_store) are written as zeros
            if (p.out_f32) ((float *)p.C)[m * p.ldc + lane] = x;
            else ((T *)p.C)[m * p.ldc + lane] = Op<DT>::from_f32(x);
        }
    }
}

// kernel selection override (pio_gemm_kernel_override; initial value from env PIO_GEMM_TILE)
static int &gemm_kernel_choice() {
    static int choice = [] {
        const char *e = getenv("PIO_GEMM_TILE");
        return e ? atoi(e) : 0;
    }();
    return choice;
}
// A/B switch for benchmarks: env PIO_GEMM_WIDE_R=1 sends residual GEMMs to the wide kernel as well
static bool wide_residual() {
    static const bool on = [] {
        const char *e = getenv("PIO_GEMM_WIDE_R");
        return e && atoi(e) != 0;
    }();
    return on;
}
int gemm_kernel_override(int which) {
    int &c = gemm_kernel_choice();
    const int prev = c;
    c = which;
    return prev;
}

// Dev aid: PIO_GEMM_LOG=1 prints one line per launch (shape, sweeps, epilogue form, kernel chosen) to stderr.
static bool gemm_log_on() {
    static const bool on = [] {
        const char *e = getenv("PIO_GEMM_LOG");
        return e && atoi(e) != 0;
    }();
    return on;
}
static void gemm_log(const char *kernel, const pio_gemm_t &g, const GemmParams &p) {
    if (!gemm_log_on()) return;
    fprintf(stderr, "pio_gemm %-10s M=%d N=%d K=%d batch=%d npass=%d act=%d out_f32=%d res=%d fold=%s n_store=%d\n", kernel, g.M,
            g.N, g.K, g.batch, p.npass, g.act, g.out_f32, (g.R || g.R16_hi) ? 1 : 0,
            g.ln_part ? "consumer" : (g.row_part ? "producer" : "-"), p.n_store);
}

int gemm_nt_launch(const pio_gemm_t &g, hipStream_t s) {
    if (!g.A || !g.B) return PIO_E_ARG;
    if (!g.C && !(g.X16 && g.X16_lo && g.out_f32)) return PIO_E_ARG;  // (fp32 C is optional beside a 16-bit pair)
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.batch <= 0 || g.nh <= 0) return PIO_E_SHAPE;
    if (g.K % 8) return PIO_E_SHAPE;
    if (g.batch % g.nh) return PIO_E_SHAPE;
    if ((g.lda % 8) || (g.ldb % 8) || (g.sAb % 8) || (g.sAh % 8) || (g.sBb % 8) || (g.sBh % 8)) return PIO_E_ALIGN;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.A_lo & 15) || ((uintptr_t)g.B_lo & 15) ||
        ((uintptr_t)g.C_lo & 7))
        return PIO_E_ALIGN;
    if (g.batch > 65535) return PIO_E_SHAPE;
    if (g.bias_mode && !g.bias) return PIO_E_ARG;
    if (g.dtype != PIO_DT_F16 && g.dtype != PIO_DT_BF16) return PIO_E_ARG;

    GemmParams p;
    p.A = g.A; p.B = g.B; p.C = g.C;
    p.C_lo = g.out_f32 ? nullptr : g.C_lo;
    // K sweeps: (A,B) [+ (A,B_lo)] [+ (A_lo,B)]  -- the dropped A_lo*B_lo term is ~2^-22 relative
    p.npass = 1;
    p.dA1 = p.dB1 = p.dA2 = p.dB2 = 0;
    auto delta = [](const void *lo, const void *hi) { return (int64_t)(((intptr_t)lo - (intptr_t)hi) / 2); };
    if (g.B_lo) { p.dB1 = delta(g.B_lo, g.B); p.npass = 2; }
    if (g.A_lo) {
        if (p.npass == 1) { p.dA1 = delta(g.A_lo, g.A); p.npass = 2; }
        else              { p.dA2 = delta(g.A_lo, g.A); p.npass = 3; }
    }
    p.M = g.M; p.N = g.N; p.K = g.K;
    p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc;
    p.nh = g.nh;
    p.sAb = g.sAb; p.sAh = g.sAh; p.sBb = g.sBb; p.sBh = g.sBh; p.sCb = g.sCb; p.sCh = g.sCh;
    p.bias = g.bias; p.bias_mode = g.bias_mode; p.act = g.act; p.alpha = g.alpha;
    p.R = g.R; p.ldr = g.ldr; p.r_stride_b = g.r_stride_b; p.r_rows = g.r_rows_per_batch;
    // a residual whose batches are contiguous ([B, T, C] with stride_b == T * ld) is one flat [B*T, C] matrix
    if (p.R && p.r_rows > 0 && p.r_stride_b == (int64_t)p.r_rows * p.ldr) p.r_rows = 0;
    p.out_f32 = g.out_f32;
    p.X16 = g.X16; p.ld16 = g.ld16; p.row_part = g.row_part;
    p.ln_part = g.ln_part; p.ln_c = g.ln_c; p.ln_eps = g.ln_eps;
    p.X16_lo = g.X16_lo; p.R16_hi = g.R16_hi; p.R16_lo = g.R16_lo;
    p.range_flag = g.row_part ? g.range_flag : nullptr;
    p.lo_n0 = g.B_lo ? g.b_lo_n0 : 0;
    p.k_rev = 0;
    p.ln_slots = g.ln_part ? (g.ln_slots > 0 ? g.ln_slots : g.K / 128) : 0;
    p.ln_inv_k = 1.0f / (float)g.K;
    p.slot_w = g.row_part ? (g.row_slot_w > 0 ? g.row_slot_w : 128) : 0;
    {
        // A/B switch: env PIO_WIDE_STAGED_EPI=0 keeps the direct (16 rows x 64 bytes per instruction) epilogue
        static const int staged = [] {
            const char *e = getenv("PIO_WIDE_STAGED_EPI");
            return (e && atoi(e) == 0) ? 0 : 1;
        }();
        p.staged_epi = staged;
#ifdef PIO_EXPERIMENTS
        p.mf32 = gemm_kernel_choice() == 4 ? 1 : 0;   // (experiments build: the MFMA 32x32x16 variants of the fold GEMMs)
#else
        p.mf32 = 0;
#endif
    }
    if (g.b_lo_n0 && (!g.B_lo || g.A_lo)) return PIO_E_ARG;
    p.n_store = g.n_store > g.N ? g.n_store : g.N;
    if (g.C && p.n_store > g.ldc) return PIO_E_SHAPE;
    p.tiles_n = (p.n_store + BN - 1) / BN;
    const int tiles_m = (g.M + BM - 1) / BM;
    const size_t esz = g.out_f32 ? 4 : 2;
    const size_t valign = g.out_f32 ? 16 : 8;
    bool vec = (((uintptr_t)g.C) % valign == 0) && ((g.ldc * esz) % valign == 0) && ((g.sCb * esz) % valign == 0) &&
               ((g.sCh * esz) % valign == 0);
    p.vec_ok = vec ? 1 : 0;
    p.r_vec = (g.R && ((uintptr_t)g.R % 16 == 0) && (g.ldr % 4 == 0) && (g.r_stride_b % 4 == 0)) ? 1 : 0;
    p.bias_vec = (g.bias && ((uintptr_t)g.bias % 16 == 0)) ? 1 : 0;

    dim3 grid((unsigned)(tiles_m * p.tiles_n), (unsigned)g.batch, 1);
    dim3 block(256, 1, 1);
    const bool attn = g.batch > 1;
    const double elems = (double)g.batch * ((double)g.M * g.K + (double)g.N * g.K);
    const double algo_flops = 2.0 * g.M * g.N * (double)g.K * g.batch;
    // algorithmic bytes: operands once (every precision sweep reads its own image), the result in every form it
    // leaves in (fp32 C if written, 16-bit C [+ C_lo], the LayerNorm fold's 16-bit pair X16 [+ X16_lo] and row sums),
    // the residual in the form it arrives in (fp32 R or the 16-bit pair R16_hi + R16_lo), the consumer's row sums
    const double mn = (double)g.batch * g.M * g.N;
    double algo_bytes = 2.0 * elems + (g.B_lo ? 2.0 * g.batch * (double)g.N * g.K : 0.0) +
                        (g.A_lo ? 2.0 * g.batch * (double)g.M * g.K : 0.0);
    if (g.out_f32) algo_bytes += g.C ? 4.0 * mn : 0.0;
    else algo_bytes += (g.C_lo ? 4.0 : 2.0) * mn;
    if (g.X16) algo_bytes += (g.X16_lo ? 4.0 : 2.0) * mn;
    if (g.row_part) algo_bytes += 8.0 * g.M * (double)(g.N / 128);
    if (g.R16_hi) algo_bytes += 4.0 * mn;
    else if (g.R) algo_bytes += 4.0 * mn;
    if (g.ln_part) algo_bytes += 64.0 * g.M;
    // A handful of output columns over many rows: no tile kernel (see gemm_nt_skinny); env PIO_GEMM_SKINNY=0: A/B switch
    {
        static const bool skinny_on = [] {
            const char *e = getenv("PIO_GEMM_SKINNY");
            return !e || atoi(e) != 0;
        }();
        const bool plain = !p.X16 && !p.row_part && !p.ln_part && !p.ln_c && !p.X16_lo && !p.R16_hi && !p.R16_lo && !p.R &&
                           !p.C_lo && g.act == 0 && g.bias_mode <= 1 && !g.b_lo_n0 && g.C;
        // (up to four columns: 182 528 x 328 -> 2 with split activations 42 us against 70 on the 128 x 128 tile; at eight
        //  columns x K = 1032 the fp32 FMAs of this kernel cost more than the tile's padding: 190 against 125 us)
        if (skinny_on && gemm_kernel_choice() == 0 && plain && g.batch == 1 && p.n_store <= 4 && g.M >= 2048 &&
            g.K <= 2048 && p.n_store <= g.ldc) {
            ProfScope prof(PROF_GEMM_SMALL, algo_flops, algo_bytes, s);
            gemm_log("skinny", g, p);
            const int nmax = p.n_store <= 2 ? 2 : 4;
            const size_t lds = (size_t)p.npass * nmax * g.K * 2;   // <= 48 KiB
            const unsigned blocks = (unsigned)(g.M / 4 < 256 * 8 ? (g.M + 3) / 4 : 256 * 8);
#define PIO_SK(DTV)                                                                                                  \
    do {                                                                                                             \
        if (nmax == 2) hipLaunchKernelGGL((gemm_nt_skinny<DTV, 2>), dim3(blocks), dim3(256), lds, s, p);             \
        else hipLaunchKernelGGL((gemm_nt_skinny<DTV, 4>), dim3(blocks), dim3(256), lds, s, p);                       \
    } while (0)
            if (g.dtype == PIO_DT_F16) PIO_SK(PIO_DT_F16);
            else PIO_SK(PIO_DT_BF16);
#undef PIO_SK
            return launch_status();
        }
    }
    // Large problems go to the 256x256-tile / 4-slot-ring kernel: enough rows, and an N that fills whole
    // 256-column tiles reasonably (<= 25 % padding).  PIO_GEMM_TILE=128|256 forces one (benchmarks).
    {
        const int forced = gemm_kernel_choice();
        const int tn256 = (p.n_store + 255) / 256, tm256 = (g.M + 255) / 256;
        bool fold_small = false;
        bool big = g.batch == 1 && g.M >= 1024 && p.n_store >= 256 && (double)tn256 * 256.0 <= 1.25 * p.n_store &&
                   (int64_t)tm256 * tn256 >= 128;
        if (forced == 128) big = false;
        if (forced == 256) big = true;
        // Persistent 256x256 four-wave kernel (pio_gemm_wide.hip): 16-bit-out projections (bias, optional GELU) with
        // at least one tile per CU.  Its epilogue is exposed, but it is bound by the stores, which the caches absorb
        // at ~7 TB/s, and the GELU arithmetic hides behind them: 16384x1024x1024 takes 36 us (GELU: 42) against 42
        // (48) on the streaming kernel.  Override 2 forces it wherever it is legal.
        {
            // (With a residual the kernel is legal but not chosen: the 64 MB residual read of a 16384x1024 launch is
            //  exposed in its epilogue -- 49 us against 39 without -- where the streaming kernel hides most of it.)
            // (from 128 tiles on: below one tile per CU the four-wave kernel still beats gemm_nt_256 tile for tile -- the
            //  language model's 8192 x 1280 projections have 160 -- and with fewer than two 256x128 tiles per CU the
            //  streaming kernel has nothing to hide a residual epilogue behind, so those come here too)
            const int64_t t256 = (int64_t)tm256 * tn256;
            static const int wide_min = [] {
                const char *e = getenv("PIO_WIDE_MIN_TILES");
                return e ? atoi(e) : 128;
            }();
            // (a residual with the hi + lo pair of the result -- the dense decoders' fc2 -- has no streaming variant: here
            //  rather than on gemm_nt_256)
            // (multimodal forward 23.23 -> 22.00 ms, in-process A/B)
            const bool pair_res = p.R && !g.out_f32 && p.C_lo;
            // (N = 384 = 1.5 tile columns: a third of the MFMAs multiply padding and the kernel still beats the 128 x 128
            //  tile's exact three columns -- 182 528 x 384 x 384, split activations, GELU, pair out: 204 against 291 us)
            const bool fill_ok = (double)tn256 * 256.0 <= 1.25 * p.n_store ||
                                 (p.n_store >= 384 && (double)tn256 * 256.0 <= 1.34 * p.n_store);
            bool wide = g.batch == 1 && g.M >= 2048 && fill_ok && t256 >= wide_min &&
                        (!p.R || wide_residual() || 2 * t256 < 448 || pair_res);
            if (forced == 2) wide = true;
            if (forced == 1 || forced == 128 || forced == 256) wide = false;
            const bool fold = p.X16 || p.row_part || p.ln_part || p.ln_c || p.X16_lo || p.R16_hi || p.R16_lo;
            // The fold on the small tiles (a stack too short for 256 x 256 tiles): chosen by the caller through the slot
            // form -- a producer asked for 64-column slots, a consumer given anything but K / 128 slots (or a shape the
            // wide kernel does not take).
            fold_small = fold && (p.row_part ? p.slot_w == 64
                                             : (p.ln_slots != g.K / 128 || g.M < 2048 || !gemm_wide_ok(p, g.batch)));
            if (fold_small) {
                if (g.batch != 1 || p.npass != 1 || p.lo_n0 || g.bias_mode > 1) return PIO_E_SHAPE;
                if (p.row_part) {  // producer
                    if (!p.X16 || !p.X16_lo || !p.R16_hi || !p.R16_lo || !g.out_f32 || p.R || (g.N & 63) || p.n_store != g.N ||
                        (p.ld16 & 3) || ((uintptr_t)p.X16 & 7) || ((uintptr_t)p.X16_lo & 7) || ((uintptr_t)p.R16_hi & 7) ||
                        ((uintptr_t)p.R16_lo & 7) || p.ln_part || g.act != 0 || (g.C && ((g.ldc & 3) || ((uintptr_t)g.C & 15))))
                        return PIO_E_SHAPE;
                } else {           // consumer
                    if (!p.ln_part || !p.ln_c || g.out_f32 || g.alpha != 1.0f || p.ln_slots <= 0) return PIO_E_SHAPE;
                }
                wide = false;
            } else if (fold) {
#ifdef PIO_EXPERIMENTS
                // (experiments build) the producer on two half-height workgroups per CU: override 3 / env PIO_GEMM_DUO=1
                static const bool duo_env = [] {
                    const char *e = getenv("PIO_GEMM_DUO");
                    return e && atoi(e) != 0;
                }();
                if ((forced == 3 || (forced == 0 && duo_env)) && gemm_duo_ok(p, g.batch)) {
                    ProfScope prof(PROF_GEMM_WIDE, algo_flops, algo_bytes, s);
                    gemm_log("duo", g, p);
                    gemm_duo_launch(p, g.dtype, s);
                    return launch_status();
                }
#endif
                if (!gemm_wide_ok(p, g.batch)) return PIO_E_SHAPE;
                wide = true;
            }
            if (wide && gemm_wide_ok(p, g.batch)) {
                ProfScope prof(PROF_GEMM_WIDE, algo_flops, algo_bytes, s);
                gemm_log("wide", g, p);
                gemm_wide_launch(p, g.dtype, s);
                return launch_status();
            }
        }
        if (p.lo_n0) return PIO_E_SHAPE;  // (a partial B_lo is a gemm_nt_wide feature)
        if (fold_small) big = false;
        // Persistent 256x128 streaming kernel (epilogue of tile j hidden behind the MFMAs of tile j+1): deep-K flat
        // problems with about two or more tiles per CU (with fewer there is nothing to hide an epilogue behind and
        // the 256x256 tile's lower operand traffic wins).  Override 1 forces it wherever it is legal.
        {
            const int tn128 = (p.n_store + 127) / 128;
            bool stream = !fold_small && g.batch == 1 && g.M >= 1024 && p.n_store >= 128 &&
                          (double)tn128 * 128.0 <= 1.25 * p.n_store && (int64_t)tm256 * tn128 >= 448;
            if (forced == 1) stream = true;
            if (forced == 128 || forced == 256) stream = false;
            if (stream && gemm_stream_ok(p, g.batch)) {
                ProfScope prof(PROF_GEMM_STREAM, algo_flops, algo_bytes, s);
                gemm_log("stream", g, p);
                gemm_stream_launch(p, g.dtype, g.batch, s);
                return launch_status();
            }
        }
        if (big) {
            ProfScope prof(PROF_GEMM_LINEAR, algo_flops, algo_bytes, s);  // class 0 == kernel gemm_nt_256
            p.tiles_n = tn256;
            gemm_log("t256", g, p);
            gemm256_launch(p, g.dtype, attn, tm256, tn256, g.batch, s);
            return launch_status();
        }
    }
    ProfScope prof(attn ? PROF_GEMM_ATTN : PROF_GEMM_SMALL, algo_flops, algo_bytes, s);  // kernel gemm_nt_128
    // 64 x 64 tiles when the 128 x 128 tiling would leave most CUs without a tile (small batches: the 2048-row latent
    // stack of the flow model at B = 1 has 64 tiles of 128 x 128 per GEMM, 256 of 64 x 64); override 64 forces them
    const int64_t tiles128 = (int64_t)tiles_m * p.tiles_n * g.batch;
    // ... and up to ONE 128 x 128 tile per CU (the double buffer then has no second workgroup to hide its waits behind) when
    // the 64 x 64 tiling fills whole rounds of the chip's 512 resident workgroups or K is short: same box, ms per forward
    // (tools/latency_probe.py / ab_env.py): ImageNet B = 8 (4096 x 1024 x 1024 projections: 256 -> 1024 tiles) 7.63 -> 7.27,
    // flow (q|k|v 2048 x 1536 x 512: 192 -> 768 tiles) 5.31 -> 5.24; NOT the B = 2 q|k|v (1024 x 3072 x 1024: 192 -> 768
    // tiles = 1.5 rounds of 16 K steps): 4.11 -> 4.18.
    const int64_t tiles64_all = (int64_t)((g.M + 63) / 64) * ((p.n_store + 63) / 64) * g.batch;
    const bool one_round = tiles128 <= cu_budget() && (g.K <= 512 || tiles64_all % (2 * (int64_t)cu_budget()) == 0);
    const bool small = gemm_kernel_choice() == 64 || (gemm_kernel_choice() == 0 && (tiles128 < 192 || one_round));
    // ... and 32 x 64 tiles when even the 64 x 64 tiling leaves a third of the CUs without one (ImageNet B = 1: the
    // 512 x 1024 projections have 128 tiles of 64 x 64, 256 of 32 x 64); env PIO_GEMM_T32=0: A/B switch
    static const bool t32_on = [] {
        const char *e = getenv("PIO_GEMM_T32");
        return !e || atoi(e) != 0;
    }();
    const int64_t tiles64 = tiles64_all;
    const bool tiny = small && t32_on && gemm_kernel_choice() == 0 && 3 * tiles64 < 2 * (int64_t)cu_budget();
    gemm_log(tiny ? "t32" : small ? "t64" : "t128", g, p);
    if (small) {
        p.tiles_n = (p.n_store + 63) / 64;
        grid = dim3((unsigned)(((g.M + (tiny ? 31 : 63)) / (tiny ? 32 : 64)) * p.tiles_n), (unsigned)g.batch, 1);
    }
    // 128 x 128 tiles on problems of at most ONE workgroup per CU (nobody covers a wave's wait for its next operand
    // stage): the four-stage LDS-DMA ring instead of the double buffer -- measured level on the flow stack's q|k|v GEMM
    // (2048 x 1536 x 512: 5.69 -> 5.68 ms per forward) and not better anywhere, so opt-in (env PIO_GEMM_RING128=1)
    static const bool ring128_on = [] {
        const char *e = getenv("PIO_GEMM_RING128");
        return e && atoi(e) != 0;
    }();
    const bool ring128 = !small && ring128_on && tiles128 <= cu_budget();
    // two K teams (eight waves) for small tiles that own their CU alone (see the kernel); env PIO_GEMM_KG2=0: A/B switch
    static const bool kg2_on = [] {
        const char *e = getenv("PIO_GEMM_KG2");
        return !e || atoi(e) != 0;
    }();
    // (from K = 1024: the flow stack's 2048 x 512 x 512 projections -- four steps a team -- lose 2 % of a forward to it,
    //  ImageNet B = 1 gains 5.6 %: tools/ab_env.py)
    const bool kg2 = small && kg2_on && gemm_kernel_choice() == 0 && p.npass * g.K >= 1024 &&
                     (int64_t)grid.x * grid.y <= (int64_t)cu_budget();
    // ... and the 128 x 128 tile likewise when it owns its CU alone (ImageNet B = 2: the 1024 x 3072 x 1024 q|k|v GEMM has
    // 192 tiles; forward 4.62 -> 4.46 ms, tools/ab_env.py)
    const bool kg2_128 = !small && !ring128 && kg2_on && gemm_kernel_choice() == 0 && p.npass * g.K >= 1024 &&
                         (int64_t)grid.x * grid.y <= (int64_t)cu_budget();
#define PIO_G128(DTV, KINDV)                                                                            \
    do {                                                                                                \
        if (kg2_128) hipLaunchKernelGGL((gemm_nt_128<DTV, KINDV, 128, 128, 2, 2>), grid, dim3(512, 1, 1), 0, s, p); \
        else if (tiny && kg2) hipLaunchKernelGGL((gemm_nt_128<DTV, KINDV, 32, 64, 4, 2>), grid, dim3(512, 1, 1), 0, s, p); \
        else if (small && kg2) hipLaunchKernelGGL((gemm_nt_128<DTV, KINDV, 64, 64, 4, 2>), grid, dim3(512, 1, 1), 0, s, p); \
        else if (tiny) hipLaunchKernelGGL((gemm_nt_128<DTV, KINDV, 32, 64, 4>), grid, block, 0, s, p);  \
        else if (small) hipLaunchKernelGGL((gemm_nt_128<DTV, KINDV, 64, 64, 4>), grid, block, 0, s, p); \
        else if (ring128) hipLaunchKernelGGL((gemm_nt_128<DTV, KINDV, 128, 128, 4>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((gemm_nt_128<DTV, KINDV, 128, 128, 2>), grid, block, 0, s, p);          \
    } while (0)
    if (g.dtype == PIO_DT_F16) {
        if (attn) PIO_G128(PIO_DT_F16, 1);
        else PIO_G128(PIO_DT_F16, 0);
    } else {
        if (attn) PIO_G128(PIO_DT_BF16, 1);
        else PIO_G128(PIO_DT_BF16, 0);
    }
#undef PIO_G128
    return launch_status();
}

}  // namespace pio
