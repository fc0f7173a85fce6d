// Fused cross-attention core for a head that is WIDER than the key axis is long: the ImageNet decoder's single
// 1024-channel head over 512 latents (perceiver.py:145-154, 177 -> transformer_primitives.py:138-175).  The tiled
// kernel of pio_xattn.hip keeps Q fragments + an O^T accumulator in registers and would have to cut a 1024-wide head
// into four dv slices that each recompute S (2.5x the flops); here the roles are turned around -- the whole score row
// of a query fits the register file (Tk <= 512: 32 queries x 512 keys per wave = 256 accumulator registers), so
//
//   phase A   S^T[512 keys x 32 queries] = K Q^T, accumulated over dk in 32-deep chunks: K chunk [512][32] and Q chunk
//             [128 queries][32] stream through LDS (a GEMM main loop; S^T lives in the 256 AGPRs, Q is NOT register
//             resident);
//   softmax   once per query row, EXACT maximum (no online rescale): 2 passes over the AGPRs; the probabilities stay in
//             128 VGPRs as the P^T operand fragments of phase B;
//   phase B   O^T[dv x 32 queries] = V^T P^T in passes of 256 dv rows (128 AGPRs), V^T [256][64 keys] chunks through the
//             same LDS ring; each pass ends with its share of the output row (normalised, 16-bit [+ lo]).
//
// S is computed ONCE, nothing is materialised in HBM, executed flops = algorithmic flops.  One workgroup = 4 waves x 32
// query rows of one (sample, head); 1000 queries x 32 samples = 256 workgroups = one per CU.
//
// LDS: ring of 3 stages x 40 KiB (K [512 rows x 64 B] | Q [128 x 64 B]; phase B: V^T [256 rows x 128 B]) filled by
// LDS-DMA pieces (global_load_lds_dwordx4, swizzle on the SOURCE address) with COUNTED waits: every wave issues exactly
// ten pieces per chunk (phase B: eight + two into a sink), the meeting point of chunk c (s_waitcnt vmcnt(10), lgkmcnt(0),
// one raw s_barrier) sits four MFMAs before the chunk's end, and the first fragments of chunk c+1 are read right behind
// it -- the fragment stream (a ring of eight register sets, read eight MFMAs ahead) never drains at a chunk boundary.
// K rows are permuted inside each 32-key block (bits 2 and 3 of the row swapped) so that accumulator registers
// 8 s .. 8 s + 7 of an S^T block are eight CONSECUTIVE keys: the block's two k-steps of phase B take them as their B
// operand with no data movement, and the matching V^T fragment is one 16-byte LDS read.
//
// Keys beyond Tk (Tk < 512) read clamped (finite) K rows / V^T columns and are masked to -inf / p = 0; a key mask
// arrives as one 32-bit word per 32 keys (xattn_keybits_kernel), a query mask wipes the row -- the same semantics as
// pio_xattn.hip.
#include <stdlib.h>

#include <type_traits>

#include "pio_internal.h"

namespace pio {

template <int I, int N, class F>
__device__ __forceinline__ void xt_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        xt_for<I + 1, N>(f);
    }
}

struct XtallParams {
    const void *Q, *K, *VT;
    void *O, *O_lo;
    const uint8_t *q_mask;
    const uint32_t *key_bits;  // [B][16]: bit j of word t = key 32 t + j is attendable (NULL: every key < Tk is)
    int Tq, Tk, H, nqt, dkp, dvp;
    int nka;     // dk chunks of 32
    int npass;   // dv passes of 256
    int64_t ldq, ldk, ldvt, ldo, sQb, sKb, sVb, sOb;
    float scale_log2;
};

constexpr int T_NKB = 16;                 // key blocks of 32: the kernel always covers 512 keys
constexpr int T_STAGE = 40 * 1024;        // K [512][64 B] + Q [128][64 B]; phase B: V^T [256][128 B]
constexpr int T_QOFF = 32 * 1024;
constexpr int T_NST = 3;
constexpr int T_NP = 10;                  // LDS-DMA pieces per wave and chunk -- ALWAYS exactly this many
constexpr int T_SINK = T_NST * T_STAGE;   // 1 KiB per wave
constexpr int T_SMEM = T_SINK + 4096;

__device__ __attribute__((aligned(16))) uint32_t g_sink_t[64 * 4];   // stores of rows past Tq (so that they can be counted)

__device__ __forceinline__ void xt_dma16(const void *src, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

template <int DT>
__device__ __forceinline__ void xt_mfma(f32x16 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {  // acc in AGPRs
    if constexpr (DT == PIO_DT_F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// (macros: an asm operand may name a vector element, a reference may not)
__device__ __forceinline__ float xt_acc_read(float a) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
#define XT_ACC_WRITE(dst, val) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(val))

template <int DT>
__global__ __launch_bounds__(256, 1) void xattn_tall_kernel(const XtallParams p) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    __shared__ __attribute__((aligned(16))) char smem[T_SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    // consecutive ids of ONE XCD walk the query tiles of one (sample, head): they stream the same K / V^T through that L2
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int qt = bid % p.nqt;
    const int bh = bid / p.nqt;
    const int b = bh / p.H, h = bh % p.H;
    const int q0w = qt * 128;            // the workgroup's first query row
    const int q0 = q0w + wave * 32;      // this wave's

    const T *Qg = (const T *)p.Q + b * p.sQb + (int64_t)h * p.dkp;
    const T *Kg = (const T *)p.K + b * p.sKb + (int64_t)h * p.dkp;
    const T *Vg = (const T *)p.VT + b * p.sVb + (int64_t)h * p.dvp * p.ldvt;

    // ---- per-lane source offsets (elements) of this wave's pieces.  64-byte LDS rows (K, Q): 16-byte chunk c of row r
    // is stored at position c ^ f((r >> 2) & 3), f = {0, 2, 3, 1}; 128-byte rows (V^T): chunk c at c ^ ((r >> 1) & 7) --
    // both conflict-free for ds_read_b128's lane groups with the 32-row fragments below.
    auto fsw = [](int g) { return (0x78 >> (2 * (g & 3))) & 3; };
    uint32_t koff[8], qoff[2], voff[8];
    {
        const int prow = lane >> 2, pos = lane & 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 16 * (wave + 4 * i) + prow;                                     // LDS row of the K chunk
            const int rr = r & 31;
            int key = (r & ~31) | (rr & ~12) | ((rr & 4) << 1) | ((rr & 8) >> 1);         // bits 2, 3 swapped
            key = key < p.Tk ? key : p.Tk - 1;                                            // finite data; masked below
            koff[i] = (uint32_t)(key * (int)p.ldk + 8 * (pos ^ fsw(r >> 2)));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 16 * (wave + 4 * i) + prow;                                     // row of the Q chunk
            int q = q0w + r;
            q = q < p.Tq ? q : p.Tq - 1;
            qoff[i] = (uint32_t)(q * (int)p.ldq + 8 * (pos ^ fsw(r >> 2)));
        }
        const int vrow = lane >> 3, vpos = lane & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * (wave + 4 * i) + vrow;                                      // row of the V^T chunk
            voff[i] = (uint32_t)(r * (int)p.ldvt + 8 * (vpos ^ ((r >> 1) & 7)));
        }
    }
    const int n_a = p.nka, n_all = p.nka + 8 * p.npass;
    // the row's query mask byte, fetched (and waited for) before anything enters the counted vector-memory queue
    const int q = q0 + r32;
    bool q_live = q < p.Tq;
    if (p.q_mask && q_live) q_live = p.q_mask[(int64_t)b * p.Tq + q] != 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // Chunk g of the unified sequence: g < nka: dk chunk g of phase A (pieces 0..7: K rows, 8..9: Q rows); then the
    // (pass, 64-key chunk) pairs of phase B (pieces 0..7: V^T rows, 8..9: into the sink); past the end: a phase-B chunk
    // whose pieces all go to the sink (same count, valid sources).  Everything that differs between the kinds is a
    // scalar (bases, steps) or a per-lane SELECT -- no branch: the chunk loops have ONE body each (differently allocated
    // bodies make the register allocator move accumulator tuples between them right behind the inline-assembly MFMAs,
    // whose results it takes for available at once).
    const char *i_src0 = nullptr, *i_src1 = nullptr;
    char *i_dst = nullptr, *i_dstq = nullptr;
    int i_step = 0, i_stepq = 0;
    bool i_b = false;
    int i_vadj = 0;               // phase B, Tk < 512: per-lane column correction that keeps the read inside the row pitch
    const int c8 = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);   // logical 16-byte chunk of this lane's V^T pieces
    char *const sinkw = smem + T_SINK + wave * 1024;
    auto issue_begin = [&](int g, int slot) {
        char *stage = smem + slot * T_STAGE + wave * 1024;
        if (g < n_a) {
            i_b = false;
            i_src0 = (const char *)(Kg + g * 32);
            i_src1 = (const char *)(Qg + g * 32);
            i_dst = stage;
            i_dstq = stage + T_QOFF;
            i_step = i_stepq = 4096;
            i_vadj = 0;
        } else {
            const int j = g - n_a, pass = j >> 3, cb = j & 7;
            const bool real = g < n_all;
            i_b = true;
            i_src0 = real ? (const char *)(Vg + (int64_t)pass * 256 * p.ldvt + cb * 64) : (const char *)Vg;
            i_src1 = (const char *)Qg;
            i_dst = real ? stage : sinkw;
            i_step = real ? 4096 : 0;
            i_dstq = sinkw;
            i_stepq = 0;
            i_vadj = 0;
            if (real && cb * 64 + 64 > (int)p.ldvt) {   // (rare: Tk < 512) keys past the pitch have p = 0: any finite data
                const int col = cb * 64 + 8 * c8, lim = (int)p.ldvt - 8;
                i_vadj = col > lim ? lim - col : 0;
            }
        }
    };
    auto issue_piece = [&](auto PI) {
        constexpr int i = decltype(PI)::value;
        if constexpr (i < 8) {
            // (signed: the column correction of a chunk past the row pitch is negative and may exceed a small row offset --
            //  the chunk's base pointer, which already carries cb * 64, makes up for it)
            const int32_t o = i_b ? (int32_t)voff[i] + i_vadj : (int32_t)koff[i];
            xt_dma16(i_src0 + 2 * (int64_t)o, i_dst + i * i_step);
        } else {
            uint32_t &o = qoff[i & 1];
            asm volatile("" : "+v"(o));
            xt_dma16(i_src1 + 2 * (uint64_t)o, i_dstq + (i - 8) * i_stepq);
        }
    };

    // ---- LDS read addresses (bytes inside a stage)
    int ka[2], va[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        ka[s] = r32 * 64 + ((((2 * s) | hh) ^ fsw(r32 >> 2)) << 4);
        asm volatile("" : "+v"(ka[s]));
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        va[s4] = r32 * 128 + ((((2 * s4) | hh) ^ ((r32 >> 1) & 7)) << 4);
        asm volatile("" : "+v"(va[s4]));
    }
    const int qrow_off = T_QOFF + wave * 32 * 64;

    // ---- prologue: chunks 0, 1 in flight (both phase A: nka >= 2); chunk 0 landed
    issue_begin(0, 0);
    xt_for<0, T_NP>([&](auto PI) { issue_piece(PI); });
    issue_begin(1, 1);
    xt_for<0, T_NP>([&](auto PI) { issue_piece(PI); });

    f32x16 sacc[T_NKB];
    {
        const float zero = 0.f;
#pragma unroll
        for (int kb = 0; kb < T_NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) XT_ACC_WRITE(sacc[kb][i], zero);
    }

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T_NP) : "memory");
    __builtin_amdgcn_s_barrier();

    // =================================================== phase A ===================================================
    // fragment f of a chunk = 16 s + kb: K rows of key block kb, k-step s (16 of the chunk's 32 dk), against Q fragment s
    V8 kf[8], qf[2], qn[2] = {};
    auto k_read = [&](const char *st, int f) { return *(const V8 *)(st + ka[f >> 4] + 2048 * (f & 15)); };
    auto q_read = [&](const char *st, int s) { return *(const V8 *)(st + qrow_off + ka[s]); };
    {
        const char *st0 = smem;
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[i] = k_read(st0, i);
        qf[0] = q_read(st0, 0);
        qf[1] = q_read(st0, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7" ::: "memory");  // VALU write of the zeroed S^T -> MFMA reads it as SrcC
    __builtin_amdgcn_sched_barrier(0);

    int slot = 0;
    // one dk chunk: 32 MFMAs, the ten pieces of chunk c + 2 between them, the meeting point for chunk c + 1 behind MFMA 27,
    // the first fragments of chunk c + 1 read right behind it.  ONE body for every chunk: the last chunk's prefetch reads
    // the (landed) first V^T chunk as if it were K rows -- harmless, never used.
#pragma unroll 1
    for (int c = 0; c < n_a; ++c) {
        const char *st = smem + slot * T_STAGE;
        const int nslot1 = slot == 2 ? 0 : slot + 1;          // stage of chunk c + 1
        const int nslot2 = slot == 0 ? 2 : slot - 1;          // stage of chunk c + 2 (= of chunk c - 1: free)
        const char *stn = smem + nslot1 * T_STAGE;
        issue_begin(c + 2, nslot2);
        xt_for<0, 32>([&](auto FI) {
            constexpr int f = decltype(FI)::value;
            xt_mfma<DT>(sacc[f & 15], kf[f & 7], qf[f >> 4]);
            if constexpr ((f & 1) == 0 && f < 2 * T_NP) issue_piece(std::integral_constant<int, f / 2>{});
            if constexpr (f + 8 < 32) kf[f & 7] = k_read(st, f + 8);
            if constexpr (f == 27) {
                // ---- meeting point: chunk c + 1 has landed (all but the ten pieces just issued), this wave's reads of
                // chunk c are done (the last one was issued four MFMAs ago); behind the barrier the stage of chunk c is
                // free for chunk c + 3 and chunk c + 1 may be read
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T_NP) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                kf[24 & 7] = k_read(stn, 0);
                kf[25 & 7] = k_read(stn, 1);
                kf[26 & 7] = k_read(stn, 2);
                kf[27 & 7] = k_read(stn, 3);
                qn[0] = q_read(stn, 0);
                qn[1] = q_read(stn, 1);
            }
            if constexpr (f >= 28) kf[f & 7] = k_read(stn, f + 8 - 32);
            __builtin_amdgcn_sched_barrier(0);
        });
        qf[0] = qn[0];
        qf[1] = qn[1];
        slot = nslot1;
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA write of S^T -> v_accvgpr_read
    __builtin_amdgcn_sched_barrier(0);

    // =================================================== softmax ===================================================
    // accumulator register i of block kb, lane half hh = key 32 kb + 16 (i >> 3) + 8 hh + (i & 7).  The accumulators are
    // only ever READ here (asm operands): a conditional write into an AGPR-resident vector makes the compiler shuttle
    // whole accumulator tuples through VGPRs right behind the MFMAs, whose results it takes for available at once.
    // Two straight-line variants under one uniform branch: with a key mask / a tail past Tk every score passes through a
    // select against the block's 32-bit word, without (the decoder) it does not.
    float mx = -INFINITY, psum = 0.f, m_use;
    V8 pf[2 * T_NKB];  // P^T operand fragments: pf[2 kb + s] = keys 32 kb + 16 s + 8 hh + 0..7
    const bool masked = p.key_bits || p.Tk < 32 * T_NKB;
    auto softmax = [&](auto MC) {
        constexpr bool MASKED = decltype(MC)::value;
        typedef const __attribute__((address_space(4))) uint32_t *cbits_t;
        auto word_of = [&](int kb) -> uint32_t {
            const int k0 = 32 * kb;
            uint32_t word = k0 + 32 <= p.Tk ? 0xffffffffu : (k0 >= p.Tk ? 0u : ((1u << (p.Tk - k0)) - 1u));
            if (p.key_bits) word &= ((cbits_t)(uintptr_t)p.key_bits)[(int64_t)b * T_NKB + kb];
            return word >> (8 * hh);   // bit 16 (i >> 3) + (i & 7) = accumulator register i of this lane half
        };
#pragma unroll
        for (int kb = 0; kb < T_NKB; ++kb) {
            uint32_t bits = 0;
            if constexpr (MASKED) bits = word_of(kb);
            float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float v0 = xt_acc_read(sacc[kb][i]), v1 = xt_acc_read(sacc[kb][i + 1]);
                if constexpr (MASKED) {
                    v0 = ((bits >> (16 * (i >> 3) + (i & 7))) & 1u) ? v0 : -INFINITY;
                    v1 = ((bits >> (16 * ((i + 1) >> 3) + ((i + 1) & 7))) & 1u) ? v1 : -INFINITY;
                }
                m0 = fmaxf(m0, v0);
                m1 = fmaxf(m1, v1);
            }
            mx = fmaxf(mx, fmaxf(m0, m1));
            __builtin_amdgcn_sched_barrier(0);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * p.scale_log2;
        m_use = mx == -INFINITY ? 0.f : mx;   // no attendable key at all: every p = exp2(-inf) = 0, l = 0
#pragma unroll
        for (int kb = 0; kb < T_NKB; ++kb) {
            uint32_t bits = 0;
            if constexpr (MASKED) bits = word_of(kb);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = xt_acc_read(sacc[kb][8 * s2 + j]);
                    if constexpr (MASKED) v = ((bits >> (16 * s2 + j)) & 1u) ? v : -INFINITY;
                    const float e = __builtin_amdgcn_exp2f(fmaf(v, p.scale_log2, -m_use));
                    psum += e;
                    pf[2 * kb + s2][j] = Op<DT>::from_f32(e);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (masked) softmax(std::true_type{});
    else softmax(std::false_type{});
    const float l_tot = psum + __shfl_xor(psum, 32, 64);
    const bool live = l_tot > 0.f && q_live;  // no attendable key / masked query: the reference wipes the row to zeros
    const float inv = live ? 1.0f / l_tot : 0.f;

    // =================================================== phase B ===================================================
    // fragment f of a chunk = 8 s4 + d: V^T rows of dv block d, k-step s4 (16 of the chunk's 64 keys), against P^T
    // fragment 4 cb + s4
    f32x16 oacc[8];
    V8 vf[8];
    auto v_read = [&](const char *st, int f) { return *(const V8 *)(st + va[f >> 3] + 4096 * (f & 7)); };
    const int64_t obase = b * p.sOb + (int64_t)q * p.ldo + (int64_t)h * p.dvp;
    bool after_epilogue = false;
    char *const gsink = (char *)g_sink_t + lane * 16;
#pragma unroll 1
    for (int pass = 0; pass < p.npass; ++pass) {
        {
            const float zero = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) XT_ACC_WRITE(oacc[d][i], zero);
        }
        {
            const char *st0 = smem + slot * T_STAGE;
#pragma unroll
            for (int i = 0; i < 8; ++i) vf[i] = v_read(st0, i);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7" ::: "memory");  // VALU write of the zeroed O^T (and of P^T) -> MFMA operands
        __builtin_amdgcn_sched_barrier(0);
        xt_for<0, 8>([&](auto CBI) {
            constexpr int cb = decltype(CBI)::value;
            const char *st = smem + slot * T_STAGE;
            const int nslot1 = slot == 2 ? 0 : slot + 1;
            const int nslot2 = slot == 0 ? 2 : slot - 1;
            const char *stn = smem + nslot1 * T_STAGE;
            issue_begin(n_a + pass * 8 + cb + 2, nslot2);
            xt_for<0, 32>([&](auto FI) {
                constexpr int f = decltype(FI)::value;
                xt_mfma<DT>(oacc[f & 7], vf[f & 7], pf[4 * cb + (f >> 3)]);
                if constexpr ((f & 1) == 0 && f < 2 * T_NP) issue_piece(std::integral_constant<int, f / 2>{});
                if constexpr (f + 8 < 32) vf[f & 7] = v_read(st, f + 8);
                if constexpr (f == 27) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (cb == 0) {
                        // (first meeting point of a pass: the previous pass's stores are younger than the awaited pieces)
                        if (!after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T_NP) : "memory");
                        else if (p.O_lo) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(32 + T_NP) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T_NP) : "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (cb < 7) {
                        vf[24 & 7] = v_read(stn, 0);
                        vf[25 & 7] = v_read(stn, 1);
                        vf[26 & 7] = v_read(stn, 2);
                        vf[27 & 7] = v_read(stn, 3);
                    }
                }
                if constexpr (f >= 28 && cb < 7) vf[f & 7] = v_read(stn, f + 8 - 32);
                __builtin_amdgcn_sched_barrier(0);
            });
            slot = nslot1;
        });
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA write of O^T -> v_accvgpr_read
        __builtin_amdgcn_sched_barrier(0);
        // ---- this pass's 256 columns of the output row: register 4 g4 + j of block d = column 32 d + 8 g4 + 4 hh + j.
        // ALWAYS 32 stores per wave (64 with a lo half) -- rows past Tq store into a sink -- so the next wait can count them
        // (with a lo half the count passes the counter's ceiling: vmcnt(63) there is stricter than needed, still exact).
        {
            T *orow = (T *)p.O + obase + pass * 256;
            T *lrow = p.O_lo ? (T *)p.O_lo + obase + pass * 256 : nullptr;
            const bool in = q < p.Tq;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int col = 32 * d + 8 * g4 + 4 * hh;
                    V4 o, l;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = xt_acc_read(oacc[d][4 * g4 + j]) * inv;
                        o[j] = Op<DT>::from_f32(v);
                        l[j] = Op<DT>::from_f32(v - Op<DT>::to_f32(o[j]));
                    }
                    *(V4 *)(in ? (char *)(orow + col) : gsink) = o;
                    if (lrow) *(V4 *)(in ? (char *)(lrow + col) : gsink + 8) = l;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        after_epilogue = true;
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------
bool xtall_supported(int dkp, int dvp, int Tk) {
    return Tk >= 1 && Tk <= 32 * T_NKB && dkp >= 64 && (dkp % 32) == 0 && dvp >= 256 && (dvp % 256) == 0;
}

size_t xtall_scratch_bytes(int B) { return ((size_t)B * T_NKB * 4 + 255) & ~(size_t)255; }

void xattn_keybits_launch(const uint8_t *kv_mask, uint32_t *bits, int B, int Tk, int ntiles, hipStream_t s);

int xtall_launch(int dtype, int dkp, int dvp, int dk_logical, const void *Q, const void *K, const void *VT, void *O,
                 void *O_lo, int B, int H, int Tq, int Tk, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo, int64_t sQb,
                 int64_t sKb, int64_t sVb, int64_t sOb, const uint8_t *kv_mask, const uint8_t *q_mask, void *scratch,
                 hipStream_t s) {
    if (!xtall_supported(dkp, dvp, Tk)) return PIO_E_SHAPE;
    if (!Q || !K || !VT || !O) return PIO_E_ARG;
    if (B <= 0 || H <= 0 || Tq <= 0) return PIO_E_SHAPE;
    if ((ldq % 8) || (ldk % 8) || (ldvt % 8) || (ldo % 4) || (sQb % 8) || (sKb % 8) || (sVb % 8) || (sOb % 4))
        return PIO_E_ALIGN;
    if (((uintptr_t)Q & 15) || ((uintptr_t)K & 15) || ((uintptr_t)VT & 15) || ((uintptr_t)O & 7) || ((uintptr_t)O_lo & 7))
        return PIO_E_ALIGN;
    if (ldvt < 64 || (int64_t)Tk * ldk >= (1ll << 31) || (int64_t)Tq * ldq >= (1ll << 31) || 256 * ldvt >= (1ll << 31))
        return PIO_E_SHAPE;
    if (kv_mask && !scratch) return PIO_E_WORKSPACE;
    XtallParams p{};
    p.Q = Q; p.K = K; p.VT = VT; p.O = O; p.O_lo = O_lo;
    p.q_mask = q_mask;
    p.key_bits = kv_mask ? (const uint32_t *)scratch : nullptr;
    p.Tq = Tq; p.Tk = Tk; p.H = H; p.nqt = (Tq + 127) / 128; p.dkp = dkp; p.dvp = dvp;
    p.nka = dkp / 32;
    p.npass = dvp / 256;
    p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo; p.sQb = sQb; p.sKb = sKb; p.sVb = sVb; p.sOb = sOb;
    p.scale_log2 = 1.4426950408889634f / sqrtf((float)dk_logical);
    const int64_t nwg = (int64_t)B * H * p.nqt;
    if (nwg > 0x7fffffffLL) return PIO_E_SHAPE;
    if (kv_mask) xattn_keybits_launch(kv_mask, (uint32_t *)scratch, B, Tk, T_NKB, s);
    dim3 grid((unsigned)nwg, 1, 1), block(256, 1, 1);
    {
        ProfScope prof(PROF_FLASH, 2.0 * B * H * (double)Tq * Tk * (dkp + dvp),
                       2.0 * B * H * ((double)Tq * (dkp + dvp) + (double)Tk * (dkp + dvp)), s);
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((xattn_tall_kernel<PIO_DT_F16>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((xattn_tall_kernel<PIO_DT_BF16>), grid, block, 0, s, p);
    }
    return launch_status();
}

}  // namespace pio
