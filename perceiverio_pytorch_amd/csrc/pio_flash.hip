// Fused multi-head attention core for gfx950: O = softmax(Q K^T / sqrt(dk)) V without materialising the
// score matrix.  Replaces transformer_primitives.py:138-166 (scores, scale, softmax, P.v, head merge) for the
// un-masked case with per-head widths 32/64/128 (qk) and 32..160 (v) -- the latent self-attention stack.
//
// Structure (one workgroup = 4 waves = 128 query rows of one (batch, head); one wave = 32 query rows):
//   * "swapped" products so that the softmax axis never crosses lanes:
//       S^T[k][q] = K[k][:] . Q[q][:]      mfma_32x32x16(A = K fragment, B = Q fragment)  -> column q on the lane
//       O^T[d][q] = V^T[d][:] . P^T[:][q]  mfma_32x32x16(A = V^T fragment, B = P^T)       -> P^T is the S^T
//     accumulator itself, converted to 16 bit in place (registers 8s..8s+7 are the B fragment of k-step s):
//     no LDS round trip, no shuffles for P.  Lanes l and l+32 hold the two halves of a column; only the
//     running max crosses them (one ds_bpermute per 64-key tile), the row sums are combined once at the end.
//   * Q fragments stay in registers for the whole kernel; K [64 keys][dk] and V^T [dv][64 keys] tiles are
//     staged HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), double buffered, one barrier per tile, with the
//     XOR swizzle applied on the source address and on the ds_read side (conflict-free K reads, 2-way V^T reads).
//   * V is consumed K-contiguous (V^T [dv][keys]), which is exactly what the V projection GEMM writes.
//   * online softmax in base 2 (scale folded into the exponent constant), fp32 statistics and accumulators.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "pio_internal.h"

namespace pio {

#ifndef PIO_FLASH_DEFAULT_VARIANT
#define PIO_FLASH_DEFAULT_VARIANT 0
#endif

struct FlashParams {
    const void *Q, *K, *VT;
    void *O;
    int Tq, Tk, H, nqt;
    int64_t ldq, ldk, ldvt, ldo;
    int64_t sQb, sKb, sVb, sOb;  // batch strides (elements); 0 for a batch-invariant Q
    float scale_log2;            // log2(e) / sqrt(dk)
    int o_rows16;                // O rows are 16-byte aligned: the epilogue may store whole 16-byte chunks
};

static __device__ __attribute__((aligned(16))) uint32_t g_zero_chunk_f[4] = {0, 0, 0, 0};

// VROW = true (DV = 128 only): V is consumed ROW-major ([keys][H*dv], the layout a plain projection GEMM writes, so
// Q | K | V come out of ONE fused GEMM) through transposed LDS reads; VROW = false: V^T [dv][keys] as before.
// NW = waves per workgroup: 4 (128 query rows, two workgroups per CU) or 8 (256 query rows, one workgroup per CU:
// every staged K / V tile then serves twice the queries, i.e. half the L2 -> LDS traffic per flop).
template <int DT, int DK, int DV, bool VROW, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void flash_attn_kernel(const FlashParams p) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    typedef short tr4 __attribute__((__vector_size__(4 * sizeof(short))));
    static_assert(!VROW || DV == 128, "row-major V path is written for 256-byte V rows");
    constexpr int KT = 64;                       // keys per tile
    constexpr int K_TILE = KT * DK * 2;          // bytes
    constexpr int V_TILE = DV * KT * 2;          // bytes
    constexpr int KCPR = DK / 8;                 // 16-byte chunks per K row (4, 8, 16)
    constexpr int KRPB = 16 / KCPR;              // K rows per 256-byte bank row
    constexpr int K_PIECES = K_TILE / 1024;      // 1-KiB LDS-DMA pieces per tile
    constexpr int V_PIECES = V_TILE / 1024;
    constexpr int NDT = DV / 32;                 // 32-row O^T tiles
    constexpr int NQS = DK / 16;                 // k-steps of the Q.K product
    constexpr int OROW = DV * 2 + 16;            // epilogue staging: bytes per output row (+16: rows spread over banks)
    constexpr int RING = 2 * (K_TILE + V_TILE), OSTG = NW * 32 * OROW;
    __shared__ __attribute__((aligned(16))) char smem[RING > OSTG ? RING : OSTG];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    // 1-D grid, remapped so that the q-tiles of one (batch, head) -- which stream the same K / V^T -- are
    // consecutive on ONE XCD (ids are dealt round-robin over the 8 XCDs): their K/V re-reads hit that L2.
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int qt = bid % p.nqt, bh = bid / p.nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qt * (NW * 32) + wave * 32;

    const T *Qg = (const T *)p.Q + b * p.sQb + (int64_t)h * DK;
    const T *Kg = (const T *)p.K + b * p.sKb + (int64_t)h * DK;
    const T *Vg = (const T *)p.VT + b * p.sVb + (VROW ? (int64_t)h * DV : (int64_t)h * DV * p.ldvt);
    const T *zsrc = (const T *)g_zero_chunk_f;

    V8 qf[NQS];  // Q fragments (B operand), loaded behind the first tile's request: the two latencies overlap

    const int ntiles = (p.Tk + KT - 1) / KT;

    auto stage = [&](int kt, int buf) {
        char *kb = smem + buf * (K_TILE + V_TILE);
        char *vb = kb + K_TILE;
        const int k0 = kt * KT;
        // K tile: piece = 64 chunks = 64/KCPR rows
        for (int pc = wave; pc < K_PIECES; pc += NW) {
            const int row = pc * (64 / KCPR) + lane / KCPR;
            const int slot = lane % KCPR;
            const int c = slot ^ ((row / KRPB) & (KCPR - 1));
            int key = k0 + row;
            key = key < p.Tk ? key : p.Tk - 1;
            const T *src = Kg + (int64_t)key * p.ldk + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(kb + pc * 1024), 16, 0, 0);
        }
        if constexpr (VROW) {
            // row-major V tile [64 keys][DV = 128]: 256-byte rows (16 chunks), a 1-KiB piece = 4 key rows
            for (int pc = wave; pc < V_PIECES; pc += NW) {
                const int row = pc * 4 + (lane >> 4);
                const int slot = lane & 15;
                const int c = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
                int key = k0 + row;
                key = key < p.Tk ? key : p.Tk - 1;
                const T *src = Vg + (int64_t)key * p.ldvt + c * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(vb + pc * 1024), 16, 0,
                                                 0);
            }
        } else
        // V^T tile: rows = dv (128-byte rows, 8 chunks), piece = 8 rows
        for (int pc = wave; pc < V_PIECES; pc += NW) {
            const int row = pc * 8 + (lane >> 3);
            const int slot = lane & 7;
            const int c = slot ^ ((row >> 1) & 7);
            const int kcol = k0 + c * 8;
            const T *src = (kcol < p.ldvt) ? Vg + (int64_t)row * p.ldvt + kcol : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(vb + pc * 1024), 16, 0, 0);
        }
    };

    f32x16 oacc[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[i][j] = 0.f;
    float m_run = -INFINITY;  // running max (already in base-2 scaled units), identical in lanes l and l+32
    float l_run = 0.f;        // this lane's partial row sum

    // per-lane read offsets
    int k_off[2];  // K fragment row byte offsets for the two 32-key halves (chunk part added per step)
#pragma unroll
    for (int t = 0; t < 2; ++t) k_off[t] = (32 * t + r32) * (DK * 2);
    const int k_swz = (r32 / KRPB) & (KCPR - 1);  // 32t is a multiple of 16*KRPB?  (32t + r32)/KRPB & mask:
    // (32/KRPB) is a multiple of KCPR for every supported DK (KCPR*KRPB = 16 divides 32), so t drops out.
    const int v_swz = (r32 >> 1) & 7;             // 32*dvt is even and a multiple of 16 rows -> drops out as well

    stage(0, 0);
    {  // lane holds Q[q0 + r32][16*s + 8*hh + 0..7]
        int q = q0 + r32;
        q = q < p.Tq ? q : p.Tq - 1;
        const T *qrow = Qg + (int64_t)q * p.ldq + 8 * hh;
#pragma unroll
        for (int s = 0; s < NQS; ++s) qf[s] = *(const V8 *)(qrow + 16 * s);
    }
    for (int kt = 0; kt < ntiles; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < ntiles) stage(kt + 1, (kt + 1) & 1);
        const char *kb = smem + (kt & 1) * (K_TILE + V_TILE);
        const char *vb = kb + K_TILE;

        // ---- S^T = K Q^T for the two 32-key halves
        f32x16 sacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int j = 0; j < 16; ++j) sacc[t][j] = 0.f;
#pragma unroll
            for (int s = 0; s < NQS; ++s) {
                const int chunk = (2 * s + hh) ^ k_swz;
                const V8 kf = *(const V8 *)(kb + k_off[t] + chunk * 16);
                sacc[t] = Op<DT>::mfma32(kf, qf[s], sacc[t]);
            }
        }
        // ---- scale to base 2, mask keys past Tk (last tile only), tile max
        const bool tail = (kt == ntiles - 1) && (p.Tk % KT != 0);
        // The max is taken on the RAW scores (scale > 0 commutes with max); the scale and the max subtraction are
        // then one FMA per element:  p = exp2(s * c - m * c).
        float mx = -INFINITY;
        if (tail) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * KT + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key >= p.Tk) sacc[t][i] = -INFINITY;
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[t][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * p.scale_log2;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first tile: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sacc[t][i], p.scale_log2, -m_new));
                sacc[t][i] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
        // rescale the output accumulators only when some row's running max actually moved (alpha == 1
        // exactly otherwise): after the first few tiles this is rare and saves NDT*16 multiplies per tile
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int j = 0; j < 16; ++j) oacc[d][j] *= alpha;
        }

        // ---- P^T fragments: registers 8s..8s+7 of the S^T accumulator are the B operand of k-step s
        V8 pf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[t][s][j] = Op<DT>::from_f32(sacc[t][8 * s + j]);

        // ---- O^T += V^T P^T.  A fragment element j <-> key 32t + 16s + 8(j>>2) + 4hh + (j&3)
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
            const char *vrow = vb + (32 * d + r32) * 128;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    V4 lo, hi;
                    if constexpr (VROW) {
                        // V tile is ROW-major [64 keys][DV] (256-byte rows); the fragment is read TRANSPOSED by
                        // ds_read_b64_tr_b16: per 16-lane group a 4-key x 16-column block, lane 4q+p supplies the
                        // address of key q / columns 4p..4p+3 and receives its own column of the 4 keys.
                        // chunk swizzle f(row) = ((row&3)<<2) | ((row>>2)&3) (conflict-free for row and tr reads)
                        const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
                        const int chunk = d * 4 + 2 * g1 + (p4 >> 1);
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int key = 32 * t + 16 * s + 8 * jj + 4 * hh + q4;
                            const int f = (q4 << 2) | ((hh + 2 * jj) & 3);
                            const char *a = vb + key * 256 + ((chunk ^ f) << 4) + 8 * (p4 & 1);
                            const tr4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) tr4 *)a);
                            if (jj == 0) lo = __builtin_bit_cast(V4, r);
                            else hi = __builtin_bit_cast(V4, r);
                        }
                    } else {
                        const int g = 8 * t + 4 * s + hh;  // 8-byte granule of keys [4g, 4g+4); second one is g+2
                        lo = *(const V4 *)(vrow + ((((g >> 1)) ^ v_swz) << 4) + (g & 1) * 8);
                        hi = *(const V4 *)(vrow + ((((g + 2) >> 1) ^ v_swz) << 4) + (g & 1) * 8);
                    }
                    V8 vf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vf[j] = lo[j];
                        vf[4 + j] = hi[j];
                    }
                    oacc[d] = Op<DT>::mfma32(vf, pf[t][s], oacc[d]);
                }
        }
    }

    // ---- epilogue: combine the two half-column sums, normalise, store O[q][h*DV + d]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (p.o_rows16) {
        // Through LDS (the ring is free after the barrier), so that this wave's 32 rows leave as whole lines -- a lane
        // stores 16 contiguous bytes, DV / 8 lanes one row -- instead of 8-byte pieces of 32 different rows per
        // instruction (stamps of the pipelined kernel below: that form spent 8.7 k cycles per workgroup issuing stores).
        __syncthreads();
        char *const ost = smem + wave * (32 * OROW);
        char *const orow = ost + r32 * OROW;
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                V4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = Op<DT>::from_f32(oacc[d][4 * g4 + j] * inv);
                *(V4 *)(orow + (32 * d + 8 * g4 + 4 * hh) * 2) = o;
            }
        constexpr int CPR = DV / 8;  // 16-byte chunks per row
        T *const og = (T *)p.O + b * p.sOb + (int64_t)h * DV;
#pragma unroll
        for (int idx0 = 0; idx0 < 32 * CPR; idx0 += 64) {
            const int idx = idx0 + lane;
            const int row = idx / CPR, ch = idx % CPR;
            if (idx < 32 * CPR && q0 + row < p.Tq)
                *(V8 *)(og + (int64_t)(q0 + row) * p.ldo + ch * 8) = *(const V8 *)(ost + row * OROW + ch * 16);
        }
        return;
    }
    const int q = q0 + r32;
    if (q < p.Tq) {
        T *orow = (T *)p.O + b * p.sOb + (int64_t)q * p.ldo + (int64_t)h * DV;
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                V4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = Op<DT>::from_f32(oacc[d][4 * g4 + j] * inv);
                *(V4 *)(orow + 32 * d + 8 * g4 + 4 * hh) = o;
            }
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void fl_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        fl_for<I + 1, N>(f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same attention for the hot shape of the latent self-attend stack (128-wide heads, V row-major out of the fused
// q|k|v GEMM, 256-query workgroups of 8 waves = 2 per SIMD) with the two waves of a SIMD STAGGERED by half a key tile.
// In flash_attn_kernel all eight waves move through S = K Q^T, the softmax and O += P V together: matrix pipe, VALU
// and LDS take turns (stamps: ~5.1 k cycles per 64-key tile against 2 k of MFMA time).  Here waves 0-3 ("A") run
// [S(k), softmax(k), PV(k)] between two barriers while waves 4-7 ("B") run [PV(k-1), S(k), softmax(k)]: A's softmax
// (VALU) overlaps B's S (MFMA), A's PV overlaps B's softmax; only A's S and B's PV meet on the matrix pipe.  ONE
// barrier per tile, passed by A at the start of tile k and by B between softmax(k-1) and PV(k-1): the V tile of k-1 is
// still being read by B when tile k+1 is staged, hence THREE K/V stages.  Both groups issue their share of tile
// k + 1's LDS-DMA pieces right behind barrier k and wait for them before barrier k + 1.
// ---------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(512, 1) void flash_attn_stag_kernel(const FlashParams p) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    typedef short tr4 __attribute__((__vector_size__(4 * sizeof(short))));
    constexpr int DK = 128, DV = 128, NW = 8;
    constexpr int KT = 64;
    constexpr int K_TILE = KT * DK * 2, V_TILE = DV * KT * 2;
    constexpr int STG = K_TILE + V_TILE;
    constexpr int KCPR = DK / 8, KRPB = 16 / KCPR;
    constexpr int K_PIECES = K_TILE / 1024, V_PIECES = V_TILE / 1024;
    constexpr int NDT = DV / 32, NQS = DK / 16;
    __shared__ __attribute__((aligned(16))) char smem[3 * STG];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int qt = bid % p.nqt, bh = bid / p.nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qt * (NW * 32) + wave * 32;

    const T *Qg = (const T *)p.Q + b * p.sQb + (int64_t)h * DK;
    const T *Kg = (const T *)p.K + b * p.sKb + (int64_t)h * DK;
    const T *Vg = (const T *)p.VT + b * p.sVb + (int64_t)h * DV;

    V8 qf[NQS];
    {
        int q = q0 + r32;
        q = q < p.Tq ? q : p.Tq - 1;
        const T *qrow = Qg + (int64_t)q * p.ldq + 8 * hh;
#pragma unroll
        for (int s = 0; s < NQS; ++s) qf[s] = *(const V8 *)(qrow + 16 * s);
    }
    const int ntiles = (p.Tk + KT - 1) / KT;

    auto stage = [&](int kt) {  // this wave's share (1 / 8) of tile kt's pieces into stage kt % 3
        char *kb = smem + (kt % 3) * STG;
        char *vb = kb + K_TILE;
        const int k0 = kt * KT;
        for (int pc = wave; pc < K_PIECES; pc += NW) {
            const int row = pc * (64 / KCPR) + lane / KCPR;
            const int slot = lane % KCPR;
            const int c = slot ^ ((row / KRPB) & (KCPR - 1));
            int key = k0 + row;
            key = key < p.Tk ? key : p.Tk - 1;
            const T *src = Kg + (int64_t)key * p.ldk + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(kb + pc * 1024), 16, 0, 0);
        }
        for (int pc = wave; pc < V_PIECES; pc += NW) {  // row-major V tile [64 keys][128]: a piece = 4 key rows
            const int row = pc * 4 + (lane >> 4);
            const int slot = lane & 15;
            const int c = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
            int key = k0 + row;
            key = key < p.Tk ? key : p.Tk - 1;
            const T *src = Vg + (int64_t)key * p.ldvt + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(vb + pc * 1024), 16, 0, 0);
        }
    };

    f32x16 oacc[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[i][j] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 sacc[2];
    V8 pf[2][2];

    int k_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) k_off[t] = (32 * t + r32) * (DK * 2);
    const int k_swz = (r32 / KRPB) & (KCPR - 1);

    // LDS fragment reads as inline assembly with hand-placed waits (see fp_lds_read128 below: the compiler puts
    // s_waitcnt vmcnt(0) in front of the transposed reads that follow a stage(), i.e. group B would wait for the tile it
    // has just requested at every barrier).  A wait is tied to the registers it covers ("+v"), so nothing that uses
    // them can be scheduled in front of it.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    uint32_t ka[NQS];  // K fragment of k-step s: key row r32 (+ 32 t through the offset), chunk (2 s + hh) ^ swizzle
#pragma unroll
    for (int s = 0; s < NQS; ++s) ka[s] = lds0 + r32 * (DK * 2) + ((((2 * s + hh) ^ k_swz)) << 4);
    uint32_t va[NDT][2];  // transposed V reads: d tile, key group jj (keys 8 jj + 4 hh + q4 of a 16-key step)
    {
        const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
        const int c0 = 2 * g1 + (p4 >> 1);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int key0 = 8 * jj + 4 * hh + q4;
            const int lo = key0 * 256 + ((c0 ^ ((hh + 2 * jj) & 3)) << 4) + 8 * (p4 & 1);
#pragma unroll
            for (int d = 0; d < NDT; ++d) va[d][jj] = lds0 + K_TILE + lo + ((d ^ q4) << 6);
        }
    }
#define PIO_STAG_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define PIO_STAG_RDTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto phase_s = [&](int kt) __attribute__((always_inline)) {  // S^T = K Q^T for the two 32-key halves of tile kt
        const uint32_t so = (uint32_t)((kt % 3) * STG);
        V8 k0f[NQS], k1f[NQS];
#pragma unroll
        for (int s = 0; s < NQS; ++s) PIO_STAG_RD128(k0f[s], ka[s] + so, 0);
#pragma unroll
        for (int s = 0; s < NQS; ++s) PIO_STAG_RD128(k1f[s], ka[s] + so, 32 * DK * 2);
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(k0f[0]), "+v"(k0f[1]), "+v"(k0f[2]), "+v"(k0f[3]), "+v"(k0f[4]), "+v"(k0f[5]), "+v"(k0f[6]),
                       "+v"(k0f[7]));
#pragma unroll
        for (int j = 0; j < 16; ++j) sacc[0][j] = 0.f;
#pragma unroll
        for (int s = 0; s < NQS; ++s) sacc[0] = Op<DT>::mfma32(k0f[s], qf[s], sacc[0]);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(k1f[0]), "+v"(k1f[1]), "+v"(k1f[2]), "+v"(k1f[3]), "+v"(k1f[4]), "+v"(k1f[5]), "+v"(k1f[6]),
                       "+v"(k1f[7]));
#pragma unroll
        for (int j = 0; j < 16; ++j) sacc[1][j] = 0.f;
#pragma unroll
        for (int s = 0; s < NQS; ++s) sacc[1] = Op<DT>::mfma32(k1f[s], qf[s], sacc[1]);
    };
    auto phase_softmax = [&](int kt) {  // online softmax of tile kt: sacc -> pf, rescales oacc when a maximum moved
        const bool tail = (kt == ntiles - 1) && (p.Tk % KT != 0);
        float mx = -INFINITY;
        if (tail) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * KT + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key >= p.Tk) sacc[t][i] = -INFINITY;
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[t][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * p.scale_log2;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sacc[t][i], p.scale_log2, -m_new));
                sacc[t][i] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int j = 0; j < 16; ++j) oacc[d][j] *= alpha;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[t][s][j] = Op<DT>::from_f32(sacc[t][8 * s + j]);
    };
    auto phase_pv = [&](int kt) __attribute__((always_inline)) {  // O^T += V^T P^T, the row-major V tile read transposed
        const uint32_t so = (uint32_t)((kt % 3) * STG);
        V4 vl[2][4], vh[2][4];  // [buffer][2 t + s]: the two halves of a fragment
        auto rd = [&](auto DC, int bufi) __attribute__((always_inline)) {
            constexpr int d = decltype(DC)::value;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c == 0) { PIO_STAG_RDTR(vl[bufi][0], va[d][0] + so, 0); PIO_STAG_RDTR(vh[bufi][0], va[d][1] + so, 0); }
                if (c == 1) { PIO_STAG_RDTR(vl[bufi][1], va[d][0] + so, 4096); PIO_STAG_RDTR(vh[bufi][1], va[d][1] + so, 4096); }
                if (c == 2) { PIO_STAG_RDTR(vl[bufi][2], va[d][0] + so, 8192); PIO_STAG_RDTR(vh[bufi][2], va[d][1] + so, 8192); }
                if (c == 3) { PIO_STAG_RDTR(vl[bufi][3], va[d][0] + so, 12288); PIO_STAG_RDTR(vh[bufi][3], va[d][1] + so, 12288); }
            }
        };
        rd(std::integral_constant<int, 0>{}, 0);
        fl_for<0, NDT>([&](auto DC) __attribute__((always_inline)) {
            constexpr int d = decltype(DC)::value;
            constexpr int bi = d & 1;
            if constexpr (d + 1 < NDT) {
                rd(std::integral_constant<int, (d + 1) % NDT>{}, bi ^ 1);
                asm volatile("s_waitcnt lgkmcnt(8)"
                             : "+v"(vl[bi][0]), "+v"(vl[bi][1]), "+v"(vl[bi][2]), "+v"(vl[bi][3]), "+v"(vh[bi][0]),
                               "+v"(vh[bi][1]), "+v"(vh[bi][2]), "+v"(vh[bi][3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(vl[bi][0]), "+v"(vl[bi][1]), "+v"(vl[bi][2]), "+v"(vl[bi][3]), "+v"(vh[bi][0]),
                               "+v"(vh[bi][1]), "+v"(vh[bi][2]), "+v"(vh[bi][3]));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                V8 vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vf[j] = vl[bi][c][j];
                    vf[4 + j] = vh[bi][c][j];
                }
                oacc[d] = Op<DT>::mfma32(vf, pf[c >> 1][c & 1], oacc[d]);
            }
        });
    };
    auto meet = [&]() {  // this wave's pieces have landed; after the barrier everybody's have
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    stage(0);
    if (wave < 4) {
        // ---- group A: [barrier k] stage(k+1) S(k) softmax(k) PV(k)
        for (int kt = 0; kt < ntiles; ++kt) {
            meet();
            if (kt + 1 < ntiles) stage(kt + 1);
            phase_s(kt);
            phase_softmax(kt);
            phase_pv(kt);
        }
        meet();  // barrier ntiles: group B passes it before its last PV
    } else {
        // ---- group B: [barrier 0] stage(1) S(0) softmax(0); then [barrier k] stage(k+1) PV(k-1) S(k) softmax(k)
        meet();
        if (1 < ntiles) stage(1);
        phase_s(0);
        phase_softmax(0);
        for (int kt = 1; kt <= ntiles; ++kt) {
            meet();
            if (kt + 1 < ntiles) stage(kt + 1);
            phase_pv(kt - 1);
            if (kt < ntiles) {
                phase_s(kt);
                phase_softmax(kt);
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + r32;
    if (q < p.Tq) {
        T *orow = (T *)p.O + b * p.sOb + (int64_t)q * p.ldo + (int64_t)h * DV;
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                V4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = Op<DT>::from_f32(oacc[d][4 * g4 + j] * inv);
                *(V4 *)(orow + 32 * d + 8 * g4 + 4 * hh) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The hot shape once more (128-wide heads, V row-major, Tq >= 256), built around what the two kernels above showed:
// with 32 query rows per wave every wave reads the whole K and V tile out of LDS (256 KiB per 64-key tile and CU =
// 2048 cycles of the 128 B/clk LDS port -- exactly the tile's MFMA time), and the per-tile barrier keeps the two waves
// of a SIMD in the same stage, so matrix pipe, VALU and LDS take turns (5.1 k cycles per tile).  Here ONE wave per
// SIMD owns 64 query rows as two 32-row blocks A and B:
//   * a K fragment read from LDS feeds two MFMAs (A and B): half the LDS bytes per flop;
//   * the wave's own instruction stream is software-pipelined so that the softmax (VALU, ~800 cycles per block and
//     tile: 32 v_exp at quarter rate + packed FMAs / adds / converts) runs in the shadow of MFMAs that do not depend on
//     it:    W1: S_A(k+1), S_B(k+1) (32 MFMAs)  ||  softmax_A(k)
//            W2: PV_A(k)            (16 MFMAs)  ||  softmax_B(k)
//            W3: PV_B(k)            (16 MFMAs)
//     (S^T accumulators double buffered; P^T converted in place, as above);
//   * lazy softmax reference point (as in pio_xattn.hip): the running reference only moves when a tile's maximum
//     exceeds it by more than 2^10, so the O^T rescale (64 multiplies per block) is a rare branch;
//   * K / V tiles in a ring of three stages, ONE barrier per tile, tile k+2 requested right behind barrier k.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fl_max_halves(float x) {  // max of lanes l and l ^ 32, in both
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float fl_sum_halves(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

#ifdef PIO_FLASH_STAMPS
// Dev-only (tools/flash_stamps.py, private build): wave 0 of workgroup 0 records s_memtime at the top of its third key
// tile and behind W1 / W2 / W3 of it, around the kernel s_memtime / s_memrealtime (held clock).  PIO_FLASH_ABL, timing
// only: bit 0 = no exponentials, 1 = no LDS fragment reads, 2 = no MFMAs, 3 = no LDS-DMA.
__device__ unsigned long long g_fstamps[24];
#ifndef PIO_FLASH_ABL
#define PIO_FLASH_ABL 0
#endif
#define PIO_FSTAMP(i)                                                                           \
    do {                                                                                        \
        if (kt == 2 && blockIdx.x == 0 && threadIdx.x == 0) g_fstamps[i] = __builtin_readcyclecounter(); \
    } while (0)
#define PIO_FSTAMP_K(i)                                                                         \
    do {                                                                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_fstamps[i] = __builtin_readcyclecounter();   \
    } while (0)
extern "C" int pio_debug_flash_stamps(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fstamps), sizeof(g_fstamps)) == hipSuccess ? 0 : 1;
}
extern "C" int pio_debug_flash_mode(void) { return PIO_FLASH_ABL; }
#else
#define PIO_FLASH_ABL 0
#define PIO_FSTAMP(i)
#define PIO_FSTAMP_K(i)
#endif

// MFMAs as inline assembly with their register files pinned (the pattern of pio_xattn.hip / pio_gemm_wide.hip): the
// O^T accumulators (128 registers) and the Q fragments (64) live in AGPRs, S^T / P^T and the streamed K / V fragments in
// the architectural VGPRs the VALU can reach.  Left to the register allocator this kernel spilled 269 registers.
// The hazard recogniser does not look inside: the few places where the VALU reads what an MFMA has just written carry
// explicit s_nop padding below.
template <int DT>
__device__ __forceinline__ void fp_mfma_s0(f32x16 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {  // c = a b, b in AGPRs
    if constexpr (PIO_FLASH_ABL & 4) asm volatile("; no mfma %0 %1 %2" : "=&v"(c) : "v"(a), "a"(b));
    else if constexpr (DT == PIO_DT_F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
}
template <int DT>
__device__ __forceinline__ void fp_mfma_s(f32x16 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {  // c += a b
    if constexpr (PIO_FLASH_ABL & 4) asm volatile("; no mfma %0 %1 %2" : "+v"(c) : "v"(a), "a"(b));
    else if constexpr (DT == PIO_DT_F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
template <int DT>
__device__ __forceinline__ void fp_mfma_o(f32x16 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {  // acc in AGPRs
    if constexpr (PIO_FLASH_ABL & 4) asm volatile("; no mfma %0 %1 %2" : "+a"(c) : "v"(a), "v"(b));
    else if constexpr (DT == PIO_DT_F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// LDS fragment reads as inline assembly, waited for by hand (s_waitcnt lgkmcnt(N) below; LDS returns in order).  Two
// reasons: (1) the compiler cannot tell an LDS-DMA destination from the address of the transposed-read intrinsic and
// puts s_waitcnt vmcnt(0) in front of the first ds_read_b64_tr_b16 behind a stage() -- i.e. it waits for the tile that
// was just requested, every tile (flash_attn_kernel above pays exactly that); (2) its lgkmcnt waits for a ring of
// fragments come out as lgkmcnt(0) every third fragment, exposing the LDS latency each time.
template <int OFF, class V8>
__device__ __forceinline__ V8 fp_lds_read128(uint32_t addr) {
    V8 r;
    if constexpr (PIO_FLASH_ABL & 2) asm volatile("; no read %0 %1" : "=v"(r) : "v"(addr));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF, class V4>
__device__ __forceinline__ V4 fp_lds_read_tr64(uint32_t addr) {
    V4 r;
    if constexpr (PIO_FLASH_ABL & 2) asm volatile("; no read %0 %1" : "=v"(r) : "v"(addr));
    else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int N>
__device__ __forceinline__ void fp_wait_lds() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

template <int DT>
__global__ __launch_bounds__(256, 1) void flash_attn_pipe_kernel(const FlashParams p) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    typedef short tr4 __attribute__((__vector_size__(4 * sizeof(short))));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int DK = 128, DV = 128, KT = 64;
    constexpr int K_TILE = KT * DK * 2, STG = 2 * K_TILE;  // 16 KiB K + 16 KiB V per stage
#ifdef PIO_FLASH_THR0
    constexpr float THR = 0.0f;
#else
    constexpr float THR = 10.0f;                           // lazy reference: p <= 2^10
#endif
    __shared__ __attribute__((aligned(16))) char smem[3 * STG];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int qt = bid % p.nqt, bh = bid / p.nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qt * 256 + wave * 64;
#ifdef PIO_FLASH_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_fstamps[4] = __builtin_amdgcn_s_memtime();
        g_fstamps[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    const T *Qg = (const T *)p.Q + b * p.sQb + (int64_t)h * DK;
    const T *Kg = (const T *)p.K + b * p.sKb + (int64_t)h * DK;
    const T *Vg = (const T *)p.VT + b * p.sVb + (int64_t)h * DV;

    V8 qf[2][8];  // Q fragments (B operand) of the two blocks, loaded behind the first tiles' requests
    const int ntiles = (p.Tk + KT - 1) / KT;
    PIO_FSTAMP_K(8);

    // ---- LDS-DMA: a 1-KiB piece = 4 rows of 256 B; wave w owns pieces w, w + 4, w + 8, w + 12 of the K tile and of the
    // V tile: rows 4 w + 16 i + (lane >> 4), whose swizzle does not depend on i (same layouts as flash_attn_kernel).
    const int drow = 4 * wave + (lane >> 4);
    const int kc = (lane & 15) ^ (drow & 15);
    const int vc = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | wave);
    const uint32_t ko = ((uint32_t)drow * (uint32_t)p.ldk + (uint32_t)kc * 8u) * 2u;
    const uint32_t vo = ((uint32_t)drow * (uint32_t)p.ldvt + (uint32_t)vc * 8u) * 2u;
    auto dma16 = [](const char *src, char *dst) {
        if constexpr (PIO_FLASH_ABL & 8) return;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    auto stage = [&](int kt, int slot) __attribute__((always_inline)) {
        char *kd = smem + slot * STG + wave * 1024;
        char *vd = kd + K_TILE;
        const int k0 = kt * KT;
        const char *kbase = (const char *)(Kg + (int64_t)k0 * p.ldk);
        const char *vbase = (const char *)(Vg + (int64_t)k0 * p.ldvt);
        {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t o = ko;
                asm volatile("" : "+v"(o));
                dma16(kbase + (int64_t)(32 * i) * p.ldk + (uint64_t)o, kd + i * 4096);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t o = vo;
                asm volatile("" : "+v"(o));
                dma16(vbase + (int64_t)(32 * i) * p.ldvt + (uint64_t)o, vd + i * 4096);
            }
        }
    };

    // ---- fragment read addresses (LDS byte addresses inside stage 0; + slot * STG per tile)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    uint32_t ka[8];  // K fragment (A operand of S^T = K Q^T): key row r32 (+ 32 t), chunk (2 s + hh) ^ (row & 15)
#pragma unroll
    for (int s = 0; s < 8; ++s) ka[s] = lds0 + r32 * 256 + (((2 * s + hh) ^ (r32 & 15)) << 4);
    uint32_t va[4][2];  // V fragment halves (transposed reads, see flash_attn_kernel): d tile, key group jj
    {
        const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
        const int c0 = 2 * g1 + (p4 >> 1);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int key0 = 8 * jj + 4 * hh + q4;
            const int lo = key0 * 256 + ((c0 ^ ((hh + 2 * jj) & 3)) << 4) + 8 * (p4 & 1);
#pragma unroll
            for (int d = 0; d < 4; ++d) va[d][jj] = lds0 + K_TILE + lo + ((d ^ q4) << 6);
        }
    }
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    f32x16 oacc[2][4];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int j = 0; j < 16; ++j) oacc[X][d][j] = 0.f;
    f32x16 sc[2][2][2];   // [buffer][block][32-key half]: S^T, then (in place) the exponentials
    uint32_t pk[2][2][8]; // [block][half][pair]: P^T as 16-bit pairs = the B operands of PV
    float mref[2] = {-INFINITY, -INFINITY};  // reference point of the exponentials (base-2 units), per query column
    float lsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    const float cs = p.scale_log2;

    // softmax of block X on buffer P, part 1: tile maximum; moves the reference point (and rescales O^T, l) when the
    // tile exceeds it by more than THR.  Returns -reference.
    auto sm_head = [&](auto PC, auto XC) __attribute__((always_inline)) -> float {
        constexpr int P = decltype(PC)::value, X = decltype(XC)::value;
        float mx = sc[P][X][0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[P][X][t][i]);
        mx = fl_max_halves(mx) * cs;
        const bool need = mx > mref[X] + THR;
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
            const float m_new = need ? mx : mref[X];
            const float alpha = __builtin_amdgcn_exp2f(mref[X] - m_new);  // first tile: exp2(-inf) = 0
            mref[X] = m_new;
            lsum[X][0] *= alpha;
            lsum[X][1] *= alpha;
            // (explicit AGPR -> VGPR -> AGPR round trip INSIDE the branch, one d tile at a time: see pio_xattn.hip)
            // The PV MFMAs of the previous tile may still be in the matrix pipe's queue (two tiles back to back at the
            // end of the key loop have nothing but ~200 cycles of VALU work between them): MFMA write of O^T -> the
            // VALU reads it.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float v;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(oacc[X][d][j]));
                    v *= alpha;
                    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(oacc[X][d][j]) : "v"(v));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return -mref[X];
    };
    // part 2, pair m (0..15): two exponentials, their sum, their 16-bit pair.  Plain (un-packed) fp32 instructions on
    // purpose: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 execute on the matrix pipe's datapath -- behind the MFMAs in
    // flight, from this wave or the other wave of the SIMD -- while v_fma_f32, v_add_f32, v_max3_f32, v_cvt_pk_f16_f32
    // and v_exp_f32 issue in an MFMA's shadow at no cost (tools/microbench/coissue.hip: 16 MFMAs + 64 v_pk_fma_f32 from
    // one wave take 996 cycles, 16 MFMAs + 64 v_fma_f32 take 524).  Written as inline assembly so that the compiler's
    // vectoriser cannot pair them up again.
    // A pair is issued in two halves one MFMA apart: front = the two FMAs and exponentials, back = the two adds into
    // the row sum and the 16-bit conversion -- v_exp_f32 has a long latency and a dependent instruction right behind
    // it stalls the wave (measured: 65 cycles per pair back to back against ~48 issued apart).
    float ex[2][2];  // exponentials in flight: [step parity][2]
    auto sm_front = [&](auto PC, auto XC, auto MC, auto EC, float nm) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value, X = decltype(XC)::value, m = decltype(MC)::value, E = decltype(EC)::value;
        constexpr int t = m >> 3, i = 2 * (m & 7);
        float x0, x1;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(sc[P][X][t][i]), "s"(cs), "v"(nm));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(sc[P][X][t][i + 1]), "s"(cs), "v"(nm));
        if constexpr (PIO_FLASH_ABL & 1) {
            ex[E][0] = x0;
            ex[E][1] = x1;
        } else {
            ex[E][0] = __builtin_amdgcn_exp2f(x0);
            ex[E][1] = __builtin_amdgcn_exp2f(x1);
        }
        // (pins the arithmetic HERE, behind the MFMA it was written behind: without a use in place the compiler sinks
        //  the whole softmax down to the first MFMA that consumes P^T)
        asm volatile("" : "+v"(ex[E][0]), "+v"(ex[E][1]));
    };
    auto sm_back = [&](auto XC, auto MC, auto EC) __attribute__((always_inline)) {
        constexpr int X = decltype(XC)::value, m = decltype(MC)::value, E = decltype(EC)::value;
        constexpr int t = m >> 3;
        asm("v_add_f32 %0, %0, %1" : "+v"(lsum[X][0]) : "v"(ex[E][0]));
        asm("v_add_f32 %0, %0, %1" : "+v"(lsum[X][1]) : "v"(ex[E][1]));
        typedef T T2 __attribute__((ext_vector_type(2)));
        T2 h2;
        h2[0] = Op<DT>::from_f32(ex[E][0]);
        h2[1] = Op<DT>::from_f32(ex[E][1]);
        pk[X][t][m & 7] = __builtin_bit_cast(uint32_t, h2);
        asm volatile("" : "+v"(pk[X][t][m & 7]), "+v"(lsum[X][0]), "+v"(lsum[X][1]));
    };
    // step n of a tile's softmax: 0..15 = the pairs of block A, 16..31 = the pairs of block B
    auto sm_step_front = [&](auto PC, auto NC, float nmA, float nmB) __attribute__((always_inline)) {
        constexpr int n = decltype(NC)::value;
        if constexpr (n < 16) sm_front(PC, I0{}, std::integral_constant<int, n>{}, std::integral_constant<int, n & 1>{}, nmA);
        else sm_front(PC, I1{}, std::integral_constant<int, n - 16>{}, std::integral_constant<int, n & 1>{}, nmB);
    };
    auto sm_step_back = [&](auto NC) __attribute__((always_inline)) {
        constexpr int n = decltype(NC)::value;
        if constexpr (n < 16) sm_back(I0{}, std::integral_constant<int, n>{}, std::integral_constant<int, n & 1>{});
        else sm_back(I1{}, std::integral_constant<int, n - 16>{}, std::integral_constant<int, n & 1>{});
    };
    auto pfrag = [&](int X, int c) __attribute__((always_inline)) -> V8 {  // c = 2 t + s2
        const u32x4 u = {pk[X][c >> 1][4 * (c & 1)], pk[X][c >> 1][4 * (c & 1) + 1], pk[X][c >> 1][4 * (c & 1) + 2],
                         pk[X][c >> 1][4 * (c & 1) + 3]};
        return __builtin_bit_cast(V8, u);
    };
    // one key tile: buffer P holds S(kt) of both blocks; DO_S: S(kt + 1) goes to buffer P ^ 1 meanwhile.
    //   W1: 32 MFMAs of S(kt + 1) (each K fragment feeds blocks A and B) || both softmax heads, the 16 exponential pairs
    //       of block A and the first 8 of block B (one pair behind each of the first 24 MFMAs);
    //   W2: 32 MFMAs of PV (each V fragment feeds A and B; key chunk c = 16 keys needs pairs <= 4 c + 3) || the last 8
    //       pairs of block B behind the MFMAs of chunks 0 and 1.
    auto tile = [&](auto PC, auto DS, int kt, uint32_t so_cur, uint32_t so_next) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
        constexpr bool DO_S = decltype(DS)::value;
        uint32_t kad[8], vad[4][2];  // this tile's fragment addresses (one add each instead of one per read)
        if constexpr (DO_S) {
#pragma unroll
            for (int s = 0; s < 8; ++s) kad[s] = ka[s] + so_next;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            vad[d][0] = va[d][0] + so_cur;
            vad[d][1] = va[d][1] + so_cur;
        }
        auto kread = [&](auto FC) __attribute__((always_inline)) -> V8 {  // fragment f = 8 t + s of the K tile
            constexpr int f = decltype(FC)::value;
            return fp_lds_read128<(f >> 3) * 8192, V8>(kad[f & 7]);
        };
        auto vread = [&](auto JC) __attribute__((always_inline)) -> V8 {  // j = 4 c + d: keys 16 c .., rows 32 d ..
            constexpr int j = decltype(JC)::value;
            constexpr int c = j >> 2, d = j & 3;
            const V4 l4 = fp_lds_read_tr64<c * 4096, V4>(vad[d][0]);
            const V4 h4 = fp_lds_read_tr64<c * 4096, V4>(vad[d][1]);
            V8 vf;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                vf[j2] = l4[j2];
                vf[4 + j2] = h4[j2];
            }
            return vf;
        };
        PIO_FSTAMP(0);
#ifdef PIO_FLASH_DRAIN
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
        V8 kfr[4];
        if constexpr (DO_S) {
            kfr[0] = kread(I0{});
            kfr[1] = kread(I1{});
        }
        const float nmA = sm_head(PC, I0{});
        const float nmB = sm_head(PC, I1{});
        __builtin_amdgcn_sched_barrier(0);
        fl_for<0, 32>([&](auto IC) __attribute__((always_inline)) {
            constexpr int i = decltype(IC)::value;
            constexpr int f = i >> 1, X = i & 1, t = f >> 3, s = f & 7;
            if constexpr (DO_S) {
                if constexpr (X == 0) {
                    if constexpr (f + 2 < 16) kfr[(f + 2) & 3] = kread(std::integral_constant<int, (f + 2) & 15>{});
                    fp_wait_lds<(f + 2 < 16) ? 2 : 15 - f>();  // fragment f has landed (f + 1, f + 2 may be in flight)
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (s == 0) fp_mfma_s0<DT>(sc[P ^ 1][X][t], kfr[f & 3], qf[X][s]);
                else fp_mfma_s<DT>(sc[P ^ 1][X][t], kfr[f & 3], qf[X][s]);
            }
            if constexpr (i < 24) sm_step_front(PC, IC, nmA, nmB);
            if constexpr (i >= 1 && i < 25) sm_step_back(std::integral_constant<int, (i + 31) & 31>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        PIO_FSTAMP(1);
        V8 vfr[4];
        vfr[0] = vread(I0{});
        vfr[1] = vread(I1{});
        __builtin_amdgcn_sched_barrier(0);
        fl_for<0, 16>([&](auto JC) __attribute__((always_inline)) {
            constexpr int j = decltype(JC)::value;
            if constexpr (j + 2 < 16) vfr[(j + 2) & 3] = vread(std::integral_constant<int, (j + 2) & 15>{});
            fp_wait_lds<(j + 2 < 16) ? 4 : 2 * (15 - j)>();
            __builtin_amdgcn_sched_barrier(0);
            fp_mfma_o<DT>(oacc[0][j & 3], vfr[j & 3], pfrag(0, j >> 2));
            if constexpr (j < 9) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (j < 8) sm_step_front(PC, std::integral_constant<int, 24 + (j & 7)>{}, nmA, nmB);
                if constexpr (j >= 1) sm_step_back(std::integral_constant<int, 23 + (j % 9)>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            fp_mfma_o<DT>(oacc[1][j & 3], vfr[j & 3], pfrag(1, j >> 2));
            __builtin_amdgcn_sched_barrier(0);
        });
        PIO_FSTAMP(2);
        // The MFMAs are inline assembly: the compiler takes their results for available at once.  Where it moves an
        // accumulator tuple between tiles (it does, on the edges out of the key loop: v_accvgpr_mov of whole O^T tiles)
        // a copy placed right behind the last MFMA would read the accumulator before that MFMA has written it.  Let the
        // matrix pipe run dry first (the last two MFMAs: 2 x 8 passes).
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        PIO_FSTAMP(3);
    };

    // ---- prologue: tiles 0 and 1 requested, S(0) computed without anything to overlap with
    PIO_FSTAMP_K(9);
    stage(0, 0);
    stage(1, 1);
    // Q fragments: lane holds Q[q0 + 32 X + r32][16 s + 8 hh + 0..7] (their latency overlaps the tiles')
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        int q = q0 + 32 * X + r32;
        q = q < p.Tq ? q : p.Tq - 1;
        const T *qrow = Qg + (int64_t)q * p.ldq + 8 * hh;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[X][s] = *(const V8 *)(qrow + 16 * s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PIO_FSTAMP_K(10);
    {
        auto kread = [&](auto FC) __attribute__((always_inline)) -> V8 {
            constexpr int f = decltype(FC)::value;
            return fp_lds_read128<(f >> 3) * 8192, V8>(ka[f & 7]);
        };
        V8 kfr[4];
        kfr[0] = kread(I0{});
        kfr[1] = kread(I1{});
        fl_for<0, 32>([&](auto IC) __attribute__((always_inline)) {
            constexpr int i = decltype(IC)::value;
            constexpr int f = i >> 1, X = i & 1, t = f >> 3, s = f & 7;
            if constexpr (X == 0) {
                if constexpr (f + 2 < 16) kfr[(f + 2) & 3] = kread(std::integral_constant<int, (f + 2) & 15>{});
                fp_wait_lds<(f + 2 < 16) ? 2 : 15 - f>();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (s == 0) fp_mfma_s0<DT>(sc[0][X][t], kfr[f & 3], qf[X][s]);
            else fp_mfma_s<DT>(sc[0][X][t], kfr[f & 3], qf[X][s]);
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA write of S^T(0) -> the VALU reads it
        __builtin_amdgcn_sched_barrier(0);
    }
    PIO_FSTAMP_K(11);
    int s_cur = 0, s_next = 1, s_free = 2;
    auto rotate = [&]() __attribute__((always_inline)) {
        const int t3 = s_cur;
        s_cur = s_next;
        s_next = s_free;
        s_free = t3;
    };
    // The key loop has ONE body (two tiles: S^T buffers 0 / 1) and every tile goes through it: the launcher takes this
    // kernel for an even number of whole key tiles only (Tk % 128 == 0).  The last tile computes "S of the tile after"
    // from whatever its ring slot holds (landed, finite data; the result is never read) -- in the shadow of that tile's
    // softmax arithmetic it costs next to nothing, and separate no-successor bodies would double the code and, worse,
    // make the register allocator move accumulator tuples between differently allocated bodies right behind the
    // MFMAs that produce them (it does not know that an inline-assembly MFMA's result arrives later).
#pragma unroll 1
    for (int kt = 0; kt < ntiles; kt += 2) {
        // (top of tile kt: tile kt + 1 must have landed everywhere; the stage of tile kt - 1 is free for tile kt + 2)
        PIO_FSTAMP(16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PIO_FSTAMP(17);
        __builtin_amdgcn_s_barrier();
        PIO_FSTAMP(18);
        if (kt + 2 < ntiles) stage(kt + 2, s_free);
        PIO_FSTAMP(19);
        tile(I0{}, std::true_type{}, kt, (uint32_t)(s_cur * STG), (uint32_t)(s_next * STG));
        rotate();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < ntiles) stage(kt + 3, s_free);
        tile(I1{}, std::true_type{}, kt + 1, (uint32_t)(s_cur * STG), (uint32_t)(s_next * STG));
        rotate();
    }
    PIO_FSTAMP_K(12);

    // ---- epilogue: combine the half-column sums, normalise, store O[q][h * DV + d]
    PIO_FSTAMP_K(13);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA write of O^T -> the VALU reads it
    __builtin_amdgcn_sched_barrier(0);
    // Through LDS (the ring is free now: each wave takes 17 KiB of it), so that the 64 rows of this wave leave as whole
    // 256-byte lines -- a lane stores 16 contiguous bytes, 16 lanes one row -- instead of 8-byte pieces of 32 different
    // rows per instruction (measured: 8.7 k cycles of a 63 k-cycle workgroup spent issuing those).
    __builtin_amdgcn_s_barrier();  // every wave has read its last V fragments
    constexpr int OROW = 272;      // bytes per staged row: 256 + 16 (rows r and r + 16 share a bank: 2-way at worst)
    char *const ost = smem + wave * (64 * OROW);
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const float inv = 1.0f / fl_sum_halves(lsum[X][0] + lsum[X][1]);
        char *const orow = ost + (32 * X + r32) * OROW;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                V4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = Op<DT>::from_f32(oacc[X][d][4 * g4 + j] * inv);
                *(V4 *)(orow + (32 * d + 8 * g4 + 4 * hh) * 2) = o;
            }
    }
    {
        const int orow4 = lane >> 4, och = lane & 15;
        T *const og = (T *)p.O + b * p.sOb + (int64_t)h * DV + och * 8;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = 4 * it + orow4;
            const V8 v = *(const V8 *)(ost + row * OROW + och * 16);
            const int q = q0 + row;
            if (q < p.Tq) *(V8 *)(og + (int64_t)q * p.ldo) = v;
        }
    }
#ifdef PIO_FLASH_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_fstamps[6] = __builtin_amdgcn_s_memtime();
        g_fstamps[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

bool flash_supported(int dkp, int dvp) {
    return (dkp == 128 && dvp == 128) || (dkp == 64 && dvp == 64) || (dkp == 32 && dvp == 32) ||
           (dkp == 32 && dvp == 160);
}

// Variant of the row-major-V kernel: 0 = lock-step waves (flash_attn_kernel, default), 1 = one wave per SIMD with the
// softmax in the MFMAs' shadow (flash_attn_pipe_kernel), 2 = two staggered wave groups (flash_attn_stag_kernel).
// which == -2 only reads; returns the previous setting.  Initial value: env PIO_FLASH_VARIANT.
int flash_variant_override(int which) {
    static int cur = [] {
        const char *e = getenv("PIO_FLASH_VARIANT");
        const int v = e ? atoi(e) : PIO_FLASH_DEFAULT_VARIANT;
        return (v >= 0 && v <= 2) ? v : PIO_FLASH_DEFAULT_VARIANT;
    }();
    const int prev = cur;
    if (which >= 0 && which <= 2) cur = which;
    return prev;
}

// v_rowmajor: VT points at V [B][Tk][.. h*dvp ..] (row stride ldvt) instead of V^T [B][H*dvp][Tk]; dvp == 128 only.
int flash_attention_launch(int dtype, int dkp, int dvp, int dk_logical, const void *Q, const void *K, const void *VT,
                           void *O, int B, int H, int Tq, int Tk, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo,
                           int64_t sQb, int64_t sKb, int64_t sVb, int64_t sOb, bool v_rowmajor, hipStream_t s) {
    if (!flash_supported(dkp, dvp)) return PIO_E_SHAPE;
    if (v_rowmajor && !(dkp == 128 && dvp == 128)) return PIO_E_SHAPE;
    if (!Q || !K || !VT || !O) return PIO_E_ARG;
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || (int64_t)B * H * ((Tq + 127) / 128) > 0x7fffffffLL) return PIO_E_SHAPE;
    if ((ldq % 8) || (ldk % 8) || (ldvt % 8) || (ldo % 4) || (sQb % 8) || (sKb % 8) || (sVb % 8) || (sOb % 4))
        return PIO_E_ALIGN;
    if (((uintptr_t)Q & 15) || ((uintptr_t)K & 15) || ((uintptr_t)VT & 15) || ((uintptr_t)O & 7)) return PIO_E_ALIGN;
    // 256-row workgroups when that still gives every CU a workgroup (row-major-V flagship path only)
    const bool wide = v_rowmajor && Tq >= 256 && (int64_t)B * H * ((Tq + 255) / 256) >= 256;
    const int nqt = wide ? (Tq + 255) / 256 : (Tq + 127) / 128;
    const int o_rows16 = (ldo % 8 == 0 && sOb % 8 == 0 && ((uintptr_t)O & 15) == 0) ? 1 : 0;
    FlashParams p{Q, K, VT, O, Tq, Tk, H, nqt, ldq, ldk, ldvt, ldo, sQb, sKb, sVb, sOb,
                  1.4426950408889634f / sqrtf((float)dk_logical), o_rows16};
    dim3 grid((unsigned)(nqt * B * H), 1, 1), block(wide ? 512 : 256, 1, 1);
    ProfScope prof(PROF_FLASH, 2.0 * B * H * (double)Tq * Tk * (dkp + dvp),
                   2.0 * B * H * ((double)Tq * (dkp + dvp) + (double)Tk * (dkp + dvp)), s);
#define PIO_FLASH(DTV, DKV, DVV) hipLaunchKernelGGL((flash_attn_kernel<DTV, DKV, DVV, false, 4>), grid, block, 0, s, p)
#define PIO_FLASH_DT(DKV, DVV)                                         \
    do {                                                               \
        if (dtype == PIO_DT_F16) PIO_FLASH(PIO_DT_F16, DKV, DVV);      \
        else PIO_FLASH(PIO_DT_BF16, DKV, DVV);                         \
    } while (0)
    // Kernel variant of the hot shape (128-wide heads, V row-major): pio_flash_variant_override / env PIO_FLASH_VARIANT.
    const int variant = flash_variant_override(-2);
    const bool pipe = variant == 1, stagger = variant == 2;
    static const bool debug = [] {
        const char *e = getenv("PIO_FLASH_DEBUG");
        return e && atoi(e) != 0;
    }();
    if (debug)
        fprintf(stderr, "[pio] fused attention B=%d H=%d Tq=%d Tk=%d dkp=%d dvp=%d v_rowmajor=%d wide=%d variant=%d\n", B, H,
                Tq, Tk, dkp, dvp, (int)v_rowmajor, (int)wide, variant);
    if (v_rowmajor && Tq >= 256 && Tk % 128 == 0 && pipe && o_rows16) {
        FlashParams pp = p;
        pp.nqt = (Tq + 255) / 256;
        dim3 pgrid((unsigned)(pp.nqt * B * H), 1, 1);
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((flash_attn_pipe_kernel<PIO_DT_F16>), pgrid, dim3(256), 0, s, pp);
        else hipLaunchKernelGGL((flash_attn_pipe_kernel<PIO_DT_BF16>), pgrid, dim3(256), 0, s, pp);
    } else if (v_rowmajor && wide && stagger) {
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((flash_attn_stag_kernel<PIO_DT_F16>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((flash_attn_stag_kernel<PIO_DT_BF16>), grid, block, 0, s, p);
    } else if (v_rowmajor && wide) {
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_F16, 128, 128, true, 8>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_BF16, 128, 128, true, 8>), grid, block, 0, s, p);
    } else if (v_rowmajor) {
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_F16, 128, 128, true, 4>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_BF16, 128, 128, true, 4>), grid, block, 0, s, p);
    } else if (dkp == 128 && dvp == 128) PIO_FLASH_DT(128, 128);
    else if (dkp == 64 && dvp == 64) PIO_FLASH_DT(64, 64);
    else if (dkp == 32 && dvp == 32) PIO_FLASH_DT(32, 32);
    else PIO_FLASH_DT(32, 160);
#undef PIO_FLASH_DT
#undef PIO_FLASH
    return launch_status();
}

}  // namespace pio
