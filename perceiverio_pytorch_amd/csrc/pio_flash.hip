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

// VROW = true: V is consumed ROW-major ([keys][H*dv], the layout a plain projection GEMM writes, so Q | K | V come out
// of ONE fused GEMM) through transposed LDS reads; VROW = false: V^T [dv][keys] as before.
// Row-major tile in LDS: key row r at byte r * DV * 2, its 16-byte chunk c at position c ^ vrow_swz<DV>(r).  One
// ds_read_b64_tr_b16 touches, per half wave, 4 consecutive keys x 4 consecutive chunks (4d .. 4d+3): conflict-free when
// those 16 units fall into 16 different 16-byte bank slots, slot = (r * DV / 8 + position) mod 16 --
//   DV = 128 (16 chunks per row): slot = position: f(r) = ((r & 3) << 2) | ((r >> 2) & 3) (also conflict-free row reads)
//   DV = 64  ( 8 chunks per row): slot = 8 (r & 1) + position: f(r) = ((r >> 1) & 1) << 2
//   DV = 32, 160 (4, 20 chunks):  slot = 4 (r & 3) + position mod 16: no swizzle needed
// NW = waves per workgroup: 4 (128 query rows, two workgroups per CU) or 8 (256 query rows, one workgroup per CU:
// every staged K / V tile then serves twice the queries, i.e. half the L2 -> LDS traffic per flop).
template <int DV>
__device__ __forceinline__ int vrow_swz(int r) {
    if constexpr (DV == 128) return ((r & 3) << 2) | ((r >> 2) & 3);
    else if constexpr (DV == 64) return ((r >> 1) & 1) << 2;
    else return 0;
}

// KS = 2 (NW = 8; launcher: Tk % 128 == 0 and at most one 128-row workgroup per CU): the KEY axis is split over two wave
// groups of the workgroup -- wave w owns query block w % 4 and key half w / 4 -- which run their halves side by side
// (two waves per SIMD: one wave's dependent MFMA -> exp2 -> MFMA chain covers the other's) and merge (m, l, O) through
// LDS at the end.  For the shapes that offer only one 4-wave workgroup per CU: the flow stack (16 heads x 2048 latents:
// 256 workgroups of ~1 900 cycles per 64-key tile against 256 cycles of MFMAs), small ImageNet batches.
template <int DT, int DK, int DV, bool VROW, int NW, int KS = 1>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 1 : 2) void flash_attn_kernel(const FlashParams p) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    typedef short tr4 __attribute__((__vector_size__(4 * sizeof(short))));
    static_assert(!VROW || DV == 128 || DV == 64 || DV == 32 || DV == 160, "row-major V: swizzles exist for these widths");
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
    static_assert(KS == 1 || ((KS == 2 || KS == 4) && NW == 4 * KS), "key split: KS groups of four waves");
    constexpr int NWQ = NW / KS;                 // waves along the query axis
    constexpr int MRG = KS > 1 ? (KS - 1) * NWQ * 64 * (NDT * 16 + 2) * 4 : 0;   // merge area: (O^T, m, l) of the other key parts
    constexpr int RING = 2 * KS * (K_TILE + V_TILE), OSTG = NW * 32 * OROW;
    constexpr int SM0 = RING > OSTG ? RING : OSTG;
    __shared__ __attribute__((aligned(16))) char smem[SM0 > MRG ? SM0 : MRG];

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
    const int qb = KS == 1 ? wave : wave % NWQ, kh = KS == 1 ? 0 : wave / NWQ;   // query block, key half of this wave
    const int q0 = qt * (NWQ * 32) + qb * 32;

    const T *Qg = (const T *)p.Q + b * p.sQb + (int64_t)h * DK;
    const T *Kg = (const T *)p.K + b * p.sKb + (int64_t)h * DK;
    const T *Vg = (const T *)p.VT + b * p.sVb + (VROW ? (int64_t)h * DV : (int64_t)h * DV * p.ldvt);
    const T *zsrc = (const T *)g_zero_chunk_f;

    V8 qf[NQS];  // Q fragments (B operand), loaded behind the first tile's request: the two latencies overlap

    const int ntiles_all = (p.Tk + KT - 1) / KT;
    const int ntiles = ntiles_all / KS;          // (KS == 2: an even tile count -- launcher) tiles of this key half
    const int tile0 = kh * ntiles;               // its first tile

    auto stage = [&](int kt, int buf) {          // tile kt of THIS key half, staged by the half's NWQ waves
        char *kb = smem + (buf * KS + kh) * (K_TILE + V_TILE);
        char *vb = kb + K_TILE;
        const int k0 = (tile0 + kt) * KT;
        // K tile: piece = 64 chunks = 64/KCPR rows
        for (int pc = qb; pc < K_PIECES; pc += NWQ) {
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
            // row-major V tile [64 keys][DV]: DV / 8 chunks per row, a 1-KiB piece = 64 consecutive chunks
            constexpr int VCPR = DV / 8;
            for (int pc = qb; pc < V_PIECES; pc += NWQ) {
                const int idx = pc * 64 + lane;
                const int row = idx / VCPR;
                const int slot = idx - row * VCPR;
                const int c = slot ^ vrow_swz<DV>(row);
                int key = k0 + row;
                key = key < p.Tk ? key : p.Tk - 1;
                const T *src = Vg + (int64_t)key * p.ldvt + c * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(vb + pc * 1024), 16, 0,
                                                 0);
            }
        } else
        // V^T tile: rows = dv (128-byte rows, 8 chunks), piece = 8 rows
        for (int pc = qb; pc < V_PIECES; pc += NWQ) {
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
        const char *kb = smem + ((kt & 1) * KS + kh) * (K_TILE + V_TILE);
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
        const bool tail = KS == 1 && (kt == ntiles - 1) && (p.Tk % KT != 0);
        // The max is taken on the RAW scores (scale > 0 commutes with max); the scale and the max subtraction are
        // then one FMA per element:  p = exp2(s * c - m * c).
        float mx = -INFINITY;
        if (tail) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = (tile0 + kt) * KT + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
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
                        // V tile is ROW-major [64 keys][DV]; the fragment is read TRANSPOSED by
                        // ds_read_b64_tr_b16: per 16-lane group a 4-key x 16-column block, lane 4q+p supplies the
                        // address of key q / columns 4p..4p+3 and receives its own column of the 4 keys.
                        // chunk swizzle vrow_swz<DV>(row): conflict-free transposed reads (see the top of the kernel)
                        const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
                        const int chunk = d * 4 + 2 * g1 + (p4 >> 1);
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int key = 32 * t + 16 * s + 8 * jj + 4 * hh + q4;
                            const int f = vrow_swz<DV>(key);
                            const char *a = vb + key * (DV * 2) + ((chunk ^ f) << 4) + 8 * (p4 & 1);
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

    if constexpr (KS > 1) {
        // ---- merge the key parts: parts 1 .. KS-1 park (O^T, m, l) in LDS (the ring is free behind the barrier), part 0
        // folds them in one after the other: m = max(m0, m1), O = O0 2^(m0 - m) + O1 2^(m1 - m), l likewise (per-lane
        // partials)
        __syncthreads();
        constexpr int MW = NDT * 16 + 2;
        if (kh > 0) {
            float *mg = (float *)smem + (size_t)(((kh - 1) * NWQ + qb) * 64 + lane) * MW;
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int j = 0; j < 16; ++j) mg[d * 16 + j] = oacc[d][j];
            mg[NDT * 16] = m_run;
            mg[NDT * 16 + 1] = l_run;
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int part = 1; part < KS; ++part) {
                const float *mg = (const float *)smem + (size_t)(((part - 1) * NWQ + qb) * 64 + lane) * MW;
                const float m1 = mg[NDT * 16], l1 = mg[NDT * 16 + 1];
                const float m = fmaxf(m_run, m1);
                const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
#pragma unroll
                for (int d = 0; d < NDT; ++d)
#pragma unroll
                    for (int j = 0; j < 16; ++j) oacc[d][j] = oacc[d][j] * a0 + mg[d * 16 + j] * a1;
                l_run = l_run * a0 + l1 * a1;
                m_run = m;
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
        if (kh != 0) return;                     // (KS > 1: the first key part's waves hold the merged result)
        char *const ost = smem + qb * (32 * OROW);
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
    if (q < p.Tq && kh == 0) {
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

#ifdef PIO_EXPERIMENTS
#include "../../tools/experiments/pio_flash_variants.inc"
#endif

bool flash_supported(int dkp, int dvp) {
    return (dkp == 128 && dvp == 128) || (dkp == 64 && dvp == 64) || (dkp == 32 && dvp == 32) ||
           (dkp == 32 && dvp == 160);
}

#ifdef PIO_EXPERIMENTS
// (experiments build only) variant of the row-major-V kernel: 0 = lock-step waves (flash_attn_kernel, the shipped one),
// 1 = one wave per SIMD with the softmax in the MFMAs' shadow (flash_attn_pipe_kernel), 2 = two staggered wave groups
// (flash_attn_stag_kernel).  which == -2 only reads; returns the previous setting.  Initial value: env PIO_FLASH_VARIANT.
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
extern "C" int pio_debug_flash_variant(int which) { return flash_variant_override(which); }
#endif

// v_rowmajor: VT points at V [B][Tk][.. h*dvp ..] (row stride ldvt) instead of V^T [B][H*dvp][Tk].
int flash_attention_launch(int dtype, int dkp, int dvp, int dk_logical, const void *Q, const void *K, const void *VT,
                           void *O, int B, int H, int Tq, int Tk, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo,
                           int64_t sQb, int64_t sKb, int64_t sVb, int64_t sOb, bool v_rowmajor, hipStream_t s) {
    if (!flash_supported(dkp, dvp)) return PIO_E_SHAPE;
    if (!Q || !K || !VT || !O) return PIO_E_ARG;
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || (int64_t)B * H * ((Tq + 127) / 128) > 0x7fffffffLL) return PIO_E_SHAPE;
    if ((ldq % 8) || (ldk % 8) || (ldvt % 8) || (ldo % 4) || (sQb % 8) || (sKb % 8) || (sVb % 8) || (sOb % 4))
        return PIO_E_ALIGN;
    if (((uintptr_t)Q & 15) || ((uintptr_t)K & 15) || ((uintptr_t)VT & 15) || ((uintptr_t)O & 7)) return PIO_E_ALIGN;
    // 256-row workgroups when that still gives every CU a workgroup (row-major-V flagship path only)
    const bool wide = v_rowmajor && Tq >= 256 && (int64_t)B * H * ((Tq + 255) / 256) >= 256;
    // (64-row workgroups for small batches -- B = 1: 64 workgroups instead of 32 -- were measured in round 3: 4.00 ms
    //  against 3.82 ms per B = 1 forward; two waves issuing a whole tile's DMA cost more than the idle CUs: not kept)
    const int nqt = wide ? (Tq + 255) / 256 : (Tq + 127) / 128;
    const int o_rows16 = (ldo % 8 == 0 && sOb % 8 == 0 && ((uintptr_t)O & 15) == 0) ? 1 : 0;
    FlashParams p{Q, K, VT, O, Tq, Tk, H, nqt, ldq, ldk, ldvt, ldo, sQb, sKb, sVb, sOb,
                  1.4426950408889634f / sqrtf((float)dk_logical), o_rows16};
    dim3 grid((unsigned)(nqt * B * H), 1, 1);
    dim3 block(wide ? 512 : 256, 1, 1);
    ProfScope prof(PROF_FLASH, 2.0 * B * H * (double)Tq * Tk * (dkp + dvp),
                   2.0 * B * H * ((double)Tq * (dkp + dvp) + (double)Tk * (dkp + dvp)), s);
#define PIO_FLASH(DTV, DKV, DVV) hipLaunchKernelGGL((flash_attn_kernel<DTV, DKV, DVV, false, 4>), grid, block, 0, s, p)
#define PIO_FLASH_DT(DKV, DVV)                                         \
    do {                                                               \
        if (dtype == PIO_DT_F16) PIO_FLASH(PIO_DT_F16, DKV, DVV);      \
        else PIO_FLASH(PIO_DT_BF16, DKV, DVV);                         \
    } while (0)
#ifdef PIO_EXPERIMENTS
    const int variant = flash_variant_override(-2);   // (experiments build: pio_debug_flash_variant / env PIO_FLASH_VARIANT)
#else
    const int variant = 0;
#endif
    static const bool debug = [] {
        const char *e = getenv("PIO_FLASH_DEBUG");
        return e && atoi(e) != 0;
    }();
    if (debug)
        fprintf(stderr, "[pio] fused attention B=%d H=%d Tq=%d Tk=%d dkp=%d dvp=%d v_rowmajor=%d wide=%d variant=%d\n", B, H,
                Tq, Tk, dkp, dvp, (int)v_rowmajor, (int)wide, variant);
#ifdef PIO_EXPERIMENTS
    const bool pipe = variant == 1, stagger = variant == 2;
    if (v_rowmajor && Tq >= 256 && Tk % 128 == 0 && pipe && o_rows16) {
        FlashParams pp = p;
        pp.nqt = (Tq + 255) / 256;
        dim3 pgrid((unsigned)(pp.nqt * B * H), 1, 1);
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((flash_attn_pipe_kernel<PIO_DT_F16>), pgrid, dim3(256), 0, s, pp);
        else hipLaunchKernelGGL((flash_attn_pipe_kernel<PIO_DT_BF16>), pgrid, dim3(256), 0, s, pp);
    } else if (v_rowmajor && wide && stagger) {
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((flash_attn_stag_kernel<PIO_DT_F16>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((flash_attn_stag_kernel<PIO_DT_BF16>), grid, block, 0, s, p);
    } else
#endif
    // key split over two wave groups (KS = 2): when the launch offers at most one 128-row workgroup per CU
    static const bool ksplit_on = [] {
        const char *e = getenv("PIO_FLASH_KSPLIT");
        return !e || atoi(e) != 0;
    }();
    const bool ksplit = ksplit_on && v_rowmajor && !wide && Tk >= 256 && (Tk % 128) == 0 &&
                        (int64_t)B * H * nqt <= cu_budget();
    // (four key parts = 16 waves per workgroup, four per SIMD: the narrow heads, whose waves need < 128 registers)
    const bool ksplit4 = ksplit && dkp <= 64 && dvp <= 64 && Tk >= 1024 && (Tk % 256) == 0;
    if (ksplit4) {
        block = dim3(1024, 1, 1);
        if (dkp == 64) {
            if (dtype == PIO_DT_F16) hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_F16, 64, 64, true, 16, 4>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_BF16, 64, 64, true, 16, 4>), grid, block, 0, s, p);
        } else {
            if (dtype == PIO_DT_F16) hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_F16, 32, 32, true, 16, 4>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_BF16, 32, 32, true, 16, 4>), grid, block, 0, s, p);
        }
    } else if (ksplit) {
        block = dim3(512, 1, 1);
#define PIO_FLASH_KS(DKV, DVV)                                                                                        \
    do {                                                                                                              \
        if (dtype == PIO_DT_F16)                                                                                      \
            hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_F16, DKV, DVV, true, 8, 2>), grid, block, 0, s, p);          \
        else hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_BF16, DKV, DVV, true, 8, 2>), grid, block, 0, s, p);        \
    } while (0)
        if (dkp == 128 && dvp == 128) PIO_FLASH_KS(128, 128);
        else if (dkp == 64 && dvp == 64) PIO_FLASH_KS(64, 64);
        else if (dkp == 32 && dvp == 32) PIO_FLASH_KS(32, 32);
        else PIO_FLASH_KS(32, 160);
#undef PIO_FLASH_KS
    } else if (v_rowmajor) {
#define PIO_FLASH_ROW(DKV, DVV)                                                                                        \
    do {                                                                                                               \
        if (wide) {                                                                                                    \
            if (dtype == PIO_DT_F16)                                                                                   \
                hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_F16, DKV, DVV, true, 8>), grid, block, 0, s, p);          \
            else hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_BF16, DKV, DVV, true, 8>), grid, block, 0, s, p);        \
        } else {                                                                                                       \
            if (dtype == PIO_DT_F16)                                                                                   \
                hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_F16, DKV, DVV, true, 4>), grid, block, 0, s, p);          \
            else hipLaunchKernelGGL((flash_attn_kernel<PIO_DT_BF16, DKV, DVV, true, 4>), grid, block, 0, s, p);        \
        }                                                                                                              \
    } while (0)
        if (dkp == 128 && dvp == 128) PIO_FLASH_ROW(128, 128);
        else if (dkp == 64 && dvp == 64) PIO_FLASH_ROW(64, 64);
        else if (dkp == 32 && dvp == 32) PIO_FLASH_ROW(32, 32);
        else PIO_FLASH_ROW(32, 160);
#undef PIO_FLASH_ROW
    } else if (dkp == 128 && dvp == 128) PIO_FLASH_DT(128, 128);
    else if (dkp == 64 && dvp == 64) PIO_FLASH_DT(64, 64);
    else if (dkp == 32 && dvp == 32) PIO_FLASH_DT(32, 32);
    else PIO_FLASH_DT(32, 160);
#undef PIO_FLASH_DT
#undef PIO_FLASH
    return launch_status();
}

}  // namespace pio
