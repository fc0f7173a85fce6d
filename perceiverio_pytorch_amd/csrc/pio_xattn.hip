// Fused CROSS-attention core for gfx950: O = softmax(mask(Q K^T / sqrt(dk))) V without the score matrix in HBM, for the
// shapes the encoder / decoder cross-attends of the shipped models have and the fused self-attention kernel
// (pio_flash.hip) does not cover: ONE wide head (322/328, 512, 704 channels), narrow heads with dv != dk (8 x (32, 160),
// 8 x (32, 96)), a key mask vector (encoder, perceiver.py:99-103) or a query mask vector (decoder, perceiver.py:172-175),
// a batch-invariant Q (the latent / query table projected once), and -- when batch x heads x query tiles do not fill
// 256 CUs (optical flow: one sample, 2048 latents against 182 528 keys) -- a split over the KEYS with a partial-softmax
// reduction (xattn_reduce_kernel).  Replaces transformer_primitives.py:138-175 (scores, scale, mask, softmax, P.v, head
// merge, wipe of rows without an attendable key).
//
// Structure (one workgroup = 4 waves = 128 query rows of one (batch, head, dv slice, key split); one wave = 32 rows):
//   * swapped products as in pio_flash.hip: S^T = K Q^T (the softmax axis is lane-local, lanes l / l+32 hold the two
//     halves of a query column) and O^T = V^T P^T with the S^T accumulator converted in place into the P^T operand.
//   * Q fragments stay in registers (DKL / 4 VGPRs per lane: up to 176 for the 704-wide head), the O^T accumulator of
//     the workgroup's dv slice too (DVS / 2 registers): one wave per SIMD for the wide heads, the whole 512-register
//     file.  A head wider than 352 is cut into dv SLICES over workgroups (each recomputes S: 1.5x the flops at 2
//     slices) because Q fragments + a full-width accumulator do not fit.
//   * keys in tiles of 32: K tile [32][KP] (KP = DKL rounded up to 128 elements = whole 256-byte bank rows, 16-byte
//     chunk c of row r stored at position (c & ~15) | ((c ^ r) & 15): conflict-free ds_read_b128) and V^T tile
//     [DVS][32 keys] (64-byte rows, chunk c of row r at c ^ ((r >> 2) & 3): conflict-free ds_read_b128), both by LDS-DMA,
//     double buffered, ONE barrier per tile, the next tile's pieces issued between this tile's MFMAs from per-lane
//     source offsets computed once.
//   * the K tile's rows are PERMUTED (bits 2 and 3 of the row index swapped) so that the eight S^T accumulator
//     registers of a k-step are eight CONSECUTIVE keys: the matching V^T fragment is one 16-byte read, not two.
//   * masks: the tile's key mask bytes (and the "past Tk" tail) become one ballot; a masked score is -inf, i.e. p = 0
//     exactly as exp(-1e30 - max) is in the reference; a row that never saw an attendable key has l = 0 and is written
//     as zeros -- the reference's "wipe" (transformer_primitives.py:168-175) -- and so is a row whose query mask is 0.
//   * nothing is zero-filled in LDS: pad chunks of K beyond dkp meet zero Q fragments, V^T rows beyond dvp feed output
//     rows that are never stored, keys beyond Tk have p = 0; their sources are clamped to valid (finite) data.
#include <stdlib.h>

#include <type_traits>

#include "pio_internal.h"

namespace pio {

template <int I, int N, class F>
__device__ __forceinline__ void xa_for(F &&f) {  // compile-time loop: the index arrives as an integral_constant
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        xa_for<I + 1, N>(f);
    }
}

struct XattnParams {
    const void *Q, *K, *VT;
    void *O;
    void *O_lo;      // optional: the rounding residual o - float(O) (the output as a 16-bit pair)
    float *part_o;   // key split: [B*H][nsplit][Tq][dvp] un-normalised O
    float *part_ml;  // key split: [B*H][nsplit][Tq][2]   (running max in exp2 units, row sum)
    const uint8_t *q_mask;
    const uint32_t *key_bits;  // [B][ntiles]: bit j of word t = key 32 t + j is attendable (NULL: every key < Tk is)
    int Tq, Tk, H, nqt, nslice, nsplit, tiles_per_split, dkp, dvp;
    int64_t ldq, ldk, ldvt, ldo, sQb, sKb, sVb, sOb;
    float scale_log2;  // log2(e) / sqrt(dk)
};

__device__ __forceinline__ void xattn_dma16(const void *src, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// MFMAs as inline assembly with their register files pinned (the pattern of pio_gemm_wide.hip).  The O^T accumulator
// (up to 176 registers) lives in AGPRs and accumulates in place; the first (256 - |O^T|) / 4 Q fragments live in AGPRs
// too (they are only ever MFMA operands), the rest and everything the VALU touches in the 256 architectural VGPRs.
// Left to the register allocator the wide-head instantiations spilled Q fragments to scratch, and every reload sat in
// the vector-memory queue behind the LDS-DMA pieces.  The compiler's hazard recogniser does not look inside an asm
// statement: the places where a VALU instruction reads what an MFMA wrote (softmax after S^T, the epilogue) or an MFMA
// reads an accumulator the VALU wrote (after a rescale) carry explicit s_nop padding below.
template <int DT, bool B_IN_AGPR>
__device__ __forceinline__ void xa_mfma_v(f32x16 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {  // acc in VGPRs
    if constexpr (DT == PIO_DT_F16) {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    } else {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
}
template <int DT, bool B_IN_AGPR>
__device__ __forceinline__ void xa_mfma_v0(f32x16 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {  // c = a b
    if constexpr (DT == PIO_DT_F16) {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
    } else {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
    }
}
template <int DT>
__device__ __forceinline__ void xa_mfma_a(f32x16 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {  // acc in AGPRs
    if constexpr (DT == PIO_DT_F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int DT, int DKL, int DVS>
__global__ __launch_bounds__(256, (DKL <= 128 && DVS <= 160) ? 2 : 1) void xattn_kernel(const XattnParams p) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    constexpr int KT = 32;                                               // keys per tile
    constexpr bool PIN = DKL > 128;                // wide heads: one wave per SIMD, pinned register files
    constexpr int KP = DKL <= 128 ? 128 : ((DKL + 127) / 128) * 128;     // LDS pitch of a K row, elements
    constexpr int KCH = KP / 8;                                          // 16-byte chunks per K row
    constexpr int K_TILE = KT * KP * 2, V_TILE = DVS * KT * 2;           // bytes
    constexpr int K_PIECES = K_TILE / 1024, V_PIECES = V_TILE / 1024;    // 1-KiB LDS-DMA pieces
    constexpr int KPW = (K_PIECES + 3) / 4, VPW = (V_PIECES + 3) / 4;    // pieces per wave
    constexpr int NQS = DKL / 16, NDT = DVS / 32;
    constexpr int RING = PIN ? (DVS >= 512 ? 3 : 6) : 4;                  // fragment sets in flight from LDS (the
                                                                          // 512-row accumulator leaves room for three)
    constexpr int NQ_A = PIN ? ((256 - NDT * 16) / 4 < NQS ? (256 - NDT * 16) / 4 : NQS) : 0;  // Q fragments in AGPRs
    static_assert(DKL % 16 == 0 && DVS % 32 == 0 && K_TILE % 1024 == 0 && V_TILE % 1024 == 0, "tile shapes");
    constexpr int STAGE = K_TILE + V_TILE;
    constexpr int LDS_BUDGET = 152 * 1024 / ((DKL <= 128 && DVS <= 160) ? 2 : 1);  // two workgroups per CU when narrow
    constexpr int NST = LDS_BUDGET / STAGE >= 4 ? 4 : (LDS_BUDGET / STAGE >= 3 ? 3 : 2);
    constexpr int NP = KPW + VPW;                // DMA pieces per wave and tile -- ALWAYS exactly this many
    static_assert((NST - 2) * NP <= 63, "the counted wait must fit vmcnt");
    constexpr int SINK = NST * STAGE;            // 1 KiB per wave: where pieces without a destination tile go
    __shared__ __attribute__((aligned(16))) char smem[SINK + 4096];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    // 1-D grid; consecutive ids of ONE XCD (ids are dealt round-robin over the 8 XCDs) walk the query tiles and dv
    // slices of one (batch, head, key split), which stream the same K / V^T: their re-reads hit that L2.
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int qt = bid % p.nqt;
    int rest = bid / p.nqt;
    const int ds = rest % p.nslice;
    rest /= p.nslice;
    const int sp = rest % p.nsplit;
    const int bh = rest / p.nsplit;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qt * 128 + wave * 32;
    const int d0 = ds * DVS;

    const T *Qg = (const T *)p.Q + b * p.sQb + (int64_t)h * p.dkp;
    const T *Kg = (const T *)p.K + b * p.sKb + (int64_t)h * p.dkp;
    const T *Vg = (const T *)p.VT + b * p.sVb + ((int64_t)h * p.dvp + d0) * p.ldvt;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + r32][16 s + 8 hh + 0..7]; zero beyond dkp
    V8 qf[NQS];
    {
        int q = q0 + r32;
        q = q < p.Tq ? q : p.Tq - 1;
        const T *qrow = Qg + (int64_t)q * p.ldq + 8 * hh;
        const V8 zero8 = {};
#pragma unroll
        for (int s = 0; s < NQS; ++s) qf[s] = (16 * s + 8 * hh < p.dkp) ? *(const V8 *)(qrow + 16 * s) : zero8;
    }

    const int ntiles = (p.Tk + KT - 1) / KT;
    const int t_begin = sp * p.tiles_per_split;
    int t_end = t_begin + p.tiles_per_split;
    t_end = t_end < ntiles ? t_end : ntiles;

    // ---- per-lane source offsets of this wave's DMA pieces (elements, relative to the tile's first key).  Nothing is
    // clamped per tile: the caller pads V^T rows with zeros to a multiple of 32 keys and guarantees that the 31 K rows
    // behind key Tk - 1 are readable memory (they are masked by assignment, whatever they hold).  A piece index beyond
    // this wave's share (ragged piece counts) re-reads piece 0's source into the per-wave sink: EVERY wave issues
    // exactly NP pieces per tile, so the waits below can be counted.
    auto k_source = [&](int pc) -> uint32_t {
        const int ci = pc * 64 + lane;               // LDS chunk index inside the K tile
        const int row = ci / KCH, pos = ci % KCH;
        int c = (pos & ~15) | ((pos ^ row) & 15);    // the logical chunk stored at this position
        c = c * 8 < p.dkp ? c : (p.dkp >> 3) - 1;    // pad chunks: any finite data (their Q fragment is zero)
        const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);  // tile row -> key: bits 2, 3 swapped
        return (uint32_t)(key * (int)p.ldk + c * 8);
    };
    auto v_source = [&](int pc) -> uint32_t {
        const int ci = pc * 64 + lane;               // LDS chunk index inside the V^T tile: 4 chunks per row
        int row = ci >> 2;
        const int kcol = ((ci & 3) ^ ((row >> 2) & 3)) << 3;
        row = d0 + row < p.dvp ? row : p.dvp - 1 - d0;   // rows beyond dvp: finite data, never stored
        return (uint32_t)(row * (int)p.ldvt + kcol);
    };
    uint32_t koff[KPW], voff[VPW];   // (element offsets < 2^31: launcher)
#pragma unroll
    for (int i = 0; i < KPW; ++i) koff[i] = k_source(wave + 4 * i < K_PIECES ? wave + 4 * i : wave);
#pragma unroll
    for (int i = 0; i < VPW; ++i) voff[i] = v_source(wave + 4 * i < V_PIECES ? wave + 4 * i : wave);

    // Piece i of tile kt = uniform tile base + the lane's precomputed 32-bit offset.  The pieces of tile kt + NST - 1
    // are issued between the MFMAs of tile kt -- always, without a branch: when that tile does not exist they re-read
    // valid memory into the sink.  Ring of NST stages; tile kt + NST - 1 overwrites the stage of tile kt - 1, which
    // every wave left before the barrier at the top of tile kt.
    char *stage_kb = nullptr, *stage_vb = nullptr;
    int stage_step = 0;
    const T *stage_k = Kg, *stage_v = Vg;
    auto stage_begin = [&](int kt, int slot, bool real) {
        if (real) {
            stage_kb = smem + slot * STAGE + wave * 1024;
            stage_vb = stage_kb + K_TILE;
            stage_step = 4096;
            stage_k = Kg + (int64_t)kt * KT * p.ldk;
            stage_v = Vg + kt * KT;
        } else {
            stage_kb = stage_vb = smem + SINK + wave * 1024;
            stage_step = 0;
            stage_k = Kg;
            stage_v = Vg;
        }
    };
    auto stage_piece = [&](auto PI) {  // 0..KPW-1: K pieces, KPW..NP-1: V^T pieces (compile-time index)
        constexpr int i = decltype(PI)::value;
        // (the offset is made opaque so that its zero-extension is not hoisted out of the tile loop as a 64-bit
        //  register pair per piece: "uniform base + zext(32-bit VGPR)" at the use selects the SGPR-base form of the DMA)
        if constexpr (i < KPW) {
            uint32_t &o = koff[i];
            asm volatile("" : "+v"(o));
            char *dst = (4 * i + 4 <= K_PIECES || wave + 4 * i < K_PIECES) ? stage_kb + i * stage_step
                                                                          : smem + SINK + wave * 1024;
            xattn_dma16((const char *)stage_k + 2 * (uint64_t)o, dst);
        } else {
            constexpr int j = i - KPW;
            uint32_t &o = voff[j];
            asm volatile("" : "+v"(o));
            char *dst = (4 * j + 4 <= V_PIECES || wave + 4 * j < V_PIECES) ? stage_vb + j * stage_step
                                                                          : smem + SINK + wave * 1024;
            xattn_dma16((const char *)stage_v + 2 * (uint64_t)o, dst);
        }
    };

    f32x16 oacc[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[i][j] = 0.f;
    float m_run = -INFINITY;  // running max in exp2 units, identical in lanes l and l + 32
    float l_run = 0.f;        // this lane's partial row sum

    // LDS read addresses.  The swizzles are XORs with a lane constant, so "base + immediate" does not apply directly;
    // but chunk (2 s + hh) ^ r32 only depends on s through (2 s) & 15: EIGHT lane registers cover every k-step of K
    // (the 256-byte block index s >> 3 is an immediate), two cover V^T (the d tile is an immediate).  Kept opaque so
    // that the compiler neither re-derives one address per unrolled read (dozens of registers living across the tile
    // loop: that spilled) nor folds them back into the loop.
    int kaddr[8], vaddr[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        kaddr[e] = r32 * (KP * 2) + ((((2 * e) ^ hh ^ r32) & 15) << 4);
        asm volatile("" : "+v"(kaddr[e]));
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        vaddr[s2] = r32 * 64 + ((((2 * s2) | hh) ^ ((r32 >> 2) & 3)) << 4);
        asm volatile("" : "+v"(vaddr[s2]));
    }

    // ---- prologue: tiles t_begin .. t_begin + NST - 2 in flight
#pragma unroll
    for (int j = 0; j < NST - 1; ++j) {
        stage_begin(t_begin + j, j, t_begin + j < t_end);
        xa_for<0, NP>([&](auto PI) { stage_piece(PI); });
    }
    int slot = 0;  // ring stage of tile kt
    for (int kt = t_begin; kt < t_end; ++kt) {
        // all but the youngest (NST - 2) tiles' pieces have landed: tile kt is complete in this wave's share; after
        // the barrier in every wave's, and the stage of tile kt - 1 is free
        // (a raw s_barrier: __syncthreads() would drain vmcnt to zero and with it the tiles in flight.  No LDS read
        //  of this wave is outstanding here -- the MFMAs of the previous tile consumed them all.)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NP) : "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int nslot = slot == 0 ? NST - 1 : slot - 1;  // = (slot + NST - 1) % NST: the stage tile kt - 1 had
            stage_begin(kt + NST - 1, nslot, kt + NST - 1 < t_end);
        }
        const char *kb = smem + slot * STAGE;
        const char *vb = kb + K_TILE;
        slot = slot + 1 == NST ? 0 : slot + 1;

        // ---- S^T = K Q^T.  The wave is alone on its SIMD (wide heads), so nobody else hides its LDS latency: the K
        // fragments travel through a RING of RK registers sets, the read of k-step s + RK issued right behind the
        // MFMA of k-step s (RK - 1 MFMAs = 160..220 cycles of cover); a scheduling barrier per step keeps the
        // compiler from hoisting every read of the tile to the top (which spills).  The next tile's DMA pieces ride
        // between the MFMAs.
        f32x16 sacc;
        if constexpr (!PIN) {
#pragma unroll
            for (int j = 0; j < 16; ++j) sacc[j] = 0.f;
        }
        constexpr int RK = NQS < RING ? NQS : RING;
        V8 kf[RK];
        auto k_read = [&](int sx) { return *(const V8 *)(kb + kaddr[sx & 7] + 256 * (sx >> 3)); };
#pragma unroll
        for (int i = 0; i < RK; ++i) kf[i] = k_read(i);
        __builtin_amdgcn_sched_barrier(0);
        xa_for<0, NQS>([&](auto SI) {
            constexpr int sx = decltype(SI)::value;
            if constexpr (!PIN) sacc = Op<DT>::mfma32(kf[sx % RK], qf[sx], sacc);
            else if constexpr (sx == 0) xa_mfma_v0<DT, (0 < NQ_A)>(sacc, kf[0], qf[0]);
            else xa_mfma_v<DT, (sx < NQ_A)>(sacc, kf[sx % RK], qf[sx]);
            if constexpr (sx + RK < NQS) kf[sx % RK] = k_read(sx + RK);
            if constexpr (sx < KPW + VPW) stage_piece(std::integral_constant<int, sx>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        xa_for<NQS, KPW + VPW>([&](auto PI) { stage_piece(PI); });
        // first V^T fragments: in flight during the softmax arithmetic.  Fragment f = 2 d + s2 (d tile, k-step).
        constexpr int NVF = 2 * NDT, RV = NVF < RING ? NVF : RING;
        V8 vf[RV];
        auto v_read = [&](int f) { return *(const V8 *)(vb + vaddr[f & 1] + 2048 * (f >> 1)); };
#pragma unroll
        for (int i = 0; i < RV; ++i) vf[i] = v_read(i);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PIN) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA write of S^T -> VALU reads
        __builtin_amdgcn_sched_barrier(0);  // (nothing that reads S^T may be scheduled in front of the padding)

        // ---- key mask / tail: accumulator register i is key k0 + 16 (i >> 3) + 8 hh + (i & 7); one 32-bit word per
        // tile (a SCALAR load: uniform address), bit j = key k0 + j is attendable
        const int k0 = kt * KT;
        if (p.key_bits || k0 + KT > p.Tk) {
            // (constant address space: a SCALAR load -- a vector load would be waited for with vmcnt(0), i.e. with
            //  every tile in flight; the words were written by xattn_keybits_kernel before this launch)
            typedef const __attribute__((address_space(4))) uint32_t *cbits_t;
            uint32_t word = p.key_bits ? ((cbits_t)(uintptr_t)p.key_bits)[(int64_t)b * ntiles + kt]
                                       : (p.Tk - k0 >= 32 ? 0xffffffffu : ((1u << (p.Tk - k0)) - 1u));
            const uint32_t bits = word >> (8 * hh);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (!((bits >> (16 * (i >> 3) + (i & 7))) & 1u)) sacc[i] = -INFINITY;
        }
        // ---- online softmax in base 2: p = exp2(s * c - m_ref * c).  The reference point m_ref follows the running
        // maximum LAZILY: it moves (and the accumulators are rescaled -- ~400 instructions with the accumulator in
        // AGPRs) only when the maximum has grown by more than 2^10 since it was set, so p <= 1024 (fine in 16 bits,
        // same relative precision) and after the first tiles a rescale is rare.  A tile without an attendable key
        // leaves m_ref at -inf; while it is -inf the accumulators are still zero and nothing needs rescaling.
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * p.scale_log2;
        const bool unset = m_run == -INFINITY;
        const float m_new = (unset || mx > m_run + 10.0f) ? fmaxf(m_run, mx) : m_run;
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = unset ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_use);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f(fmaf(sacc[i], p.scale_log2, -m_use));
            sacc[i] = e;
            psum += e;
        }
        l_run = l_run * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
            // (one d tile at a time: with the accumulator in AGPRs every element travels AGPR -> VGPR -> AGPR, and
            //  an unfenced fully unrolled loop would want 176 temporaries at once)
#pragma unroll
            for (int d = 0; d < NDT; ++d) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if constexpr (PIN) {
                        // explicit AGPR -> VGPR -> AGPR round trip INSIDE the branch: written as plain arithmetic
                        // the compiler hoists the copies of the whole accumulator out of it, into every tile
                        float v;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(oacc[d][j]));
                        v *= alpha;
                        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(oacc[d][j]) : "v"(v));
                    } else {
                        oacc[d][j] *= alpha;
                    }
                }
                if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (PIN) asm volatile("s_nop 7" ::: "memory");  // VALU write of O^T -> MFMA reads it as SrcC
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- P^T fragments: registers 8 s .. 8 s + 7 are keys 16 s + 8 hh + 0..7 = the B operand of k-step s
        V8 pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s][j] = Op<DT>::from_f32(sacc[8 * s + j]);
        // ---- O^T += V^T P^T: the A fragment of (d tile, k-step s2) is ONE 16-byte chunk of V^T row 32 d + r32
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PIN) asm volatile("s_nop 3" ::: "memory");  // VALU write of P^T -> MFMA reads it as SrcB
        __builtin_amdgcn_sched_barrier(0);
        xa_for<0, NVF>([&](auto FI) {
            constexpr int f = decltype(FI)::value;
            constexpr int d = f >> 1, s2 = f & 1;
            if constexpr (PIN) xa_mfma_a<DT>(oacc[d], vf[f % RV], pf[s2]);
            else oacc[d] = Op<DT>::mfma32(vf[f % RV], pf[s2], oacc[d]);
            if constexpr (f + RV < NVF) vf[f % RV] = v_read(f + RV);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PIN) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA write of O^T -> VALU reads
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int q = q0 + r32;
    if (q >= p.Tq) return;
    if (p.nsplit > 1) {
        // partial result of this key split: un-normalised O and (m, l); combined by xattn_reduce_kernel
        const int64_t prow = ((int64_t)bh * p.nsplit + sp) * p.Tq + q;
        float *po = p.part_o + prow * p.dvp + d0;
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = 32 * d + 8 * g4 + 4 * hh;
                if (d0 + col < p.dvp) {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = oacc[d][4 * g4 + j];
                    *(f32x4 *)(po + col) = o;
                }
            }
        if (ds == 0 && hh == 0) {
            p.part_ml[prow * 2] = m_run;
            p.part_ml[prow * 2 + 1] = l_tot;
        }
        return;
    }
    bool live = l_tot > 0.f;  // no attendable key at all: the reference wipes the row to zeros
    if (p.q_mask) live = live && p.q_mask[(int64_t)b * p.Tq + q] != 0;
    const float inv = live ? 1.0f / l_tot : 0.f;
    const int64_t ooff = b * p.sOb + (int64_t)q * p.ldo + (int64_t)h * p.dvp + d0;
    T *orow = (T *)p.O + ooff;
    T *lrow = p.O_lo ? (T *)p.O_lo + ooff : nullptr;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int col = 32 * d + 8 * g4 + 4 * hh;
            if (d0 + col < p.dvp) {
                V4 o, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = oacc[d][4 * g4 + j] * inv;
                    o[j] = Op<DT>::from_f32(v);
                    l[j] = Op<DT>::from_f32(v - Op<DT>::to_f32(o[j]));
                }
                *(V4 *)(orow + col) = o;
                if (lrow) *(V4 *)(lrow + col) = l;
            }
        }
}

// One 32-bit word per (sample, key tile): bit j = key 32 t + j exists (< Tk) and its mask byte is non-zero.
__global__ __launch_bounds__(256) void xattn_keybits_kernel(const uint8_t *kv_mask, uint32_t *bits, int Tk, int ntiles,
                                                            int64_t total_words) {
    const int64_t w = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);  // one 32-lane half-wave per word
    const int j = threadIdx.x & 31;
    bool valid = false;
    if (w < total_words) {
        const int64_t b = w / ntiles;
        const int key = (int)(w % ntiles) * 32 + j;
        valid = key < Tk && kv_mask[b * Tk + key] != 0;
    }
    const uint64_t bal = __builtin_amdgcn_ballot_w64(valid);
    if (j == 0 && w < total_words) bits[w] = (threadIdx.x & 32) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
}

// (shared with pio_xtall.hip)
void xattn_keybits_launch(const uint8_t *kv_mask, uint32_t *bits, int B, int Tk, int ntiles, hipStream_t s) {
    const int64_t words = (int64_t)B * ntiles;
    hipLaunchKernelGGL(xattn_keybits_kernel, dim3((unsigned)((words + 7) / 8)), dim3(256), 0, s, kv_mask, bits, Tk, ntiles,
                       words);
}

// Combines the key splits of one (batch, head, query row): O = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M).
template <int DT>
__global__ __launch_bounds__(256) void xattn_reduce_kernel(const float *part_o, const float *part_ml, void *O, void *O_lo,
                                                           const uint8_t *q_mask, int H, int nsplit, int Tq, int dvp,
                                                           int64_t ldo, int64_t sOb, int64_t total) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V4 V4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int d4 = dvp >> 2;
    const int c = (int)(idx % d4) * 4;
    const int64_t row = idx / d4;  // (bh, q)
    const int q = (int)(row % Tq);
    const int bh = (int)(row / Tq);
    const int b = bh / H, h = bh % H;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[(((int64_t)bh * nsplit + s) * Tq + q) * 2]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
    if (M > -INFINITY) {
        for (int s = 0; s < nsplit; ++s) {
            const int64_t pr = ((int64_t)bh * nsplit + s) * Tq + q;
            const float m = part_ml[pr * 2];
            if (m == -INFINITY) continue;  // (this split saw no attendable key: its O is zero, its l is zero)
            const float w = __builtin_amdgcn_exp2f(m - M);
            L += part_ml[pr * 2 + 1] * w;
            const f32x4 o = *(const f32x4 *)(part_o + pr * dvp + c);
            acc += o * w;
        }
    }
    bool live = L > 0.f;
    if (q_mask) live = live && q_mask[(int64_t)b * Tq + q] != 0;
    const float inv = live ? 1.0f / L : 0.f;
    V4 o, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = acc[j] * inv;
        o[j] = Op<DT>::from_f32(v);
        l[j] = Op<DT>::from_f32(v - Op<DT>::to_f32(o[j]));
    }
    const int64_t off = b * sOb + (int64_t)q * ldo + (int64_t)h * dvp + c;
    *(V4 *)((T *)O + off) = o;
    if (O_lo) *(V4 *)((T *)O_lo + off) = l;
}

// ---- host side -------------------------------------------------------------------------------------------------
namespace {
struct XCfg {
    int dkl, dvs;
};
// the kernel instantiations, narrowest first
// (the sliced <352,192> / <512,256> instantiations of round 2 became unreachable when the single-pass <352,352> /
// <512,512> ones were put in front of them in round 3 -- first match wins -- and are gone)
constexpr XCfg kCfgs[] = {{32, 96}, {32, 160}, {128, 128}, {352, 352}, {512, 512}, {704, 256}};

const XCfg *xattn_cfg(int dkp, int dvp) {
    for (const XCfg &c : kCfgs) {
        if (dkp > c.dkl) continue;
        // a head wider than one slice is cut into dv slices (each recomputes S): only where Q + O do not fit otherwise
        const bool sliced = c.dkl >= 352;
        if (dvp <= c.dvs || (sliced && dvp <= c.dkl)) return &c;
    }
    return nullptr;
}
}  // namespace

bool xattn_supported(int dkp, int dvp) { return xattn_cfg(dkp, dvp) != nullptr; }

// key splits for a launch: about one workgroup per CU and resident slot (wide heads hold one workgroup per CU, narrow
// ones two) when batch x heads x query tiles x slices alone give clearly fewer, each split keeping >= 8 key tiles.
// The partials cost HBM traffic (4 dv bytes per query row and split, written and read back): no more splits than that.
int xattn_splits(int dkp, int dvp, int B, int H, int Tq, int Tk) {
    const XCfg *c = xattn_cfg(dkp, dvp);
    if (!c) return 1;
    const int nslice = (dvp + c->dvs - 1) / c->dvs;
    const int64_t base = (int64_t)B * H * ((Tq + 127) / 128) * nslice;
    const int ntiles = (Tk + 31) / 32;
    const int64_t target = (c->dkl <= 128 && c->dvs <= 160) ? 512 : 256;
    if (base * 5 >= target * 4 || ntiles < 16) return 1;
    int64_t s = (target + base - 1) / base;
    if (s > ntiles / 8) s = ntiles / 8;
    return s < 1 ? 1 : (int)s;
}

// scratch of one launch: the key-bit words of a masked launch [B][ntiles], then the fp32 partials of the key splits
static size_t keybits_bytes(int B, int Tk) { return ((size_t)B * ((Tk + 31) / 32) * 4 + 255) & ~(size_t)255; }
size_t xattn_partial_bytes(int dkp, int dvp, int B, int H, int Tq, int Tk) {
    const int s = xattn_splits(dkp, dvp, B, H, Tq, Tk);
    return keybits_bytes(B, Tk) + (s <= 1 ? 0 : (size_t)B * H * s * Tq * ((size_t)dvp * 4 + 8) + 512);
}

int xattn_launch(int dtype, int dkp, int dvp, int dk_logical, const void *Q, const void *K, const void *VT, void *O,
                 void *O_lo, int B, int H, int Tq, int Tk, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo, int64_t sQb,
                 int64_t sKb, int64_t sVb, int64_t sOb, const uint8_t *kv_mask, const uint8_t *q_mask, void *partials,
                 hipStream_t s) {
    const XCfg *c = xattn_cfg(dkp, dvp);
    if (!c) return PIO_E_SHAPE;
    if (!Q || !K || !VT || !O) return PIO_E_ARG;
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || (dkp & 7) || (dvp & 7)) return PIO_E_SHAPE;
    if ((ldq % 8) || (ldk % 8) || (ldvt % 8) || (ldo % 4) || (sQb % 8) || (sKb % 8) || (sVb % 8) || (sOb % 4))
        return PIO_E_ALIGN;
    if (((uintptr_t)Q & 15) || ((uintptr_t)K & 15) || ((uintptr_t)VT & 15) || ((uintptr_t)O & 7) || ((uintptr_t)O_lo & 7))
        return PIO_E_ALIGN;
    if (ldvt < 8 || 32 * ldk >= (1ll << 31) || (int64_t)dvp * ldvt >= (1ll << 31)) return PIO_E_SHAPE;
    const int nslice = (dvp + c->dvs - 1) / c->dvs;
    const int nqt = (Tq + 127) / 128;
    const int nsplit = xattn_splits(dkp, dvp, B, H, Tq, Tk);
    const int ntiles = (Tk + 31) / 32;
    const int tps = (ntiles + nsplit - 1) / nsplit;
    if ((nsplit > 1 || kv_mask) && !partials) return PIO_E_WORKSPACE;
    if (ldvt % 32) return PIO_E_ALIGN;  // V^T rows are read in whole 32-key tiles (zero padded by the caller)
    const int64_t nwg = (int64_t)B * H * nqt * nslice * nsplit;
    if (nwg > 0x7fffffffLL) return PIO_E_SHAPE;
    XattnParams p{};
    p.Q = Q; p.K = K; p.VT = VT; p.O = O; p.O_lo = O_lo;
    p.key_bits = kv_mask ? (const uint32_t *)partials : nullptr;
    p.part_o = (float *)((char *)partials + keybits_bytes(B, Tk));
    p.part_ml = nsplit > 1 ? p.part_o + (size_t)B * H * nsplit * Tq * dvp : nullptr;
    p.q_mask = q_mask;
    p.Tq = Tq; p.Tk = Tk; p.H = H; p.nqt = nqt; p.nslice = nslice; p.nsplit = nsplit; p.tiles_per_split = tps;
    p.dkp = dkp; p.dvp = dvp;
    p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo; p.sQb = sQb; p.sKb = sKb; p.sVb = sVb; p.sOb = sOb;
    p.scale_log2 = 1.4426950408889634f / sqrtf((float)dk_logical);
    dim3 grid((unsigned)nwg, 1, 1), block(256, 1, 1);
    if (kv_mask) {
        const int64_t words = (int64_t)B * ntiles;
        hipLaunchKernelGGL(xattn_keybits_kernel, dim3((unsigned)((words + 7) / 8)), block, 0, s, kv_mask,
                           (uint32_t *)partials, Tk, ntiles, words);
    }
    {
        // S recomputed per dv slice counts once: algorithmic flops of the reference formulation
        ProfScope prof(PROF_FLASH, 2.0 * B * H * (double)Tq * Tk * (dkp + dvp),
                       2.0 * B * H * ((double)Tq * (dkp + dvp) + (double)Tk * (dkp + dvp)), s);
#define PIO_XA(DKLV, DVSV)                                                                                   \
    do {                                                                                                     \
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((xattn_kernel<PIO_DT_F16, DKLV, DVSV>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((xattn_kernel<PIO_DT_BF16, DKLV, DVSV>), grid, block, 0, s, p);               \
    } while (0)
        if (c->dkl == 32 && c->dvs == 96) PIO_XA(32, 96);
        else if (c->dkl == 32) PIO_XA(32, 160);
        else if (c->dkl == 128) PIO_XA(128, 128);
        else if (c->dkl == 352) PIO_XA(352, 352);
        else if (c->dkl == 512) PIO_XA(512, 512);
        else PIO_XA(704, 256);
#undef PIO_XA
        if (nsplit > 1) {
            const int64_t total = (int64_t)B * H * Tq * (dvp / 4);
            dim3 rgrid((unsigned)((total + 255) / 256), 1, 1);
            if (dtype == PIO_DT_F16)
                hipLaunchKernelGGL((xattn_reduce_kernel<PIO_DT_F16>), rgrid, block, 0, s, p.part_o, p.part_ml, O, O_lo,
                                   q_mask, H, nsplit, Tq, dvp, ldo, sOb, total);
            else
                hipLaunchKernelGGL((xattn_reduce_kernel<PIO_DT_BF16>), rgrid, block, 0, s, p.part_o, p.part_ml, O, O_lo,
                                   q_mask, H, nsplit, Tq, dvp, ldo, sOb, total);
        }
    }
    return launch_status();
}

}  // namespace pio
