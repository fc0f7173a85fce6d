// gemm_nt_duo: the LayerNorm fold's PRODUCER GEMM (x = R + A W^T + b with the residual given and the result left as
// 16-bit pairs + per-row partial sums, pio_gemm_t.{R16_hi,R16_lo,X16,X16_lo,row_part}) with TWO workgroups per CU.
//
// Why: on gemm_nt_wide (one 256x256 tile per CU) this GEMM is 24 us of main loop that moves 34 MB followed by a 28 us
// epilogue that moves 128 MB -- every CU reaches its epilogue at the same moment, HBM idles during the main loops and
// the matrix pipes idle during the epilogues.  Here a tile is 128x256 (four waves of 64x128, 128 accumulator
// registers in AGPRs, <= 128 VGPRs), so two workgroups share a CU, and the second one on each CU (odd wave slot)
// starts half a main loop late: while one workgroup streams its residual in and its result out, the other one
// multiplies.  One tile per workgroup, no persistence.
//
// Main loop (per wave and phase = one K slice of 32: 32 MFMAs): A fragments double-buffered (2 x 4), B fragments
// single-buffered and reloaded for the next slice right behind their last MFMA of this one (column-major MFMA order);
// ring of three slices (A 8 KiB | B 16 KiB) filled by global_load_lds_dwordx4 pieces (2 of A + 4 of B per wave and
// phase), the meeting point (vmcnt(6), lgkmcnt(0), barrier) behind the phase's first four MFMAs.  LDS layout, B-row
// permutation and epilogue column mapping are those of pio_gemm_wide.hip.
//
// Reference semantics: x + Linear(...) of transformer_primitives.py:290-292.
#include <type_traits>

#include "../../perceiverio_pytorch_amd/csrc/pio_gemm_common.h"

namespace pio {

constexpr int D_BM = 128, D_BN = 256, D_BK = 32, D_NST = 3;
constexpr int D_AB = D_BM * D_BK * 2;          // 8 KiB of A per slice
constexpr int D_BB = D_BN * D_BK * 2;          // 16 KiB of B
constexpr int D_STAGE = D_AB + D_BB;           // 24 KiB
constexpr int D_RING = D_NST * D_STAGE;        // 72 KiB
constexpr int D_SINK = D_RING;                 // 4 x 1 KiB: pieces past the end of K
constexpr int D_BIAS = D_SINK + 4 * 1024;      // 4 x 512 B: per-wave bias rows
constexpr int D_SMEM = D_BIAS + 4 * 512;       // 78 KiB: two workgroups per CU

static __device__ __attribute__((aligned(16))) uint32_t g_sink_d[64 * 4];
static __device__ __attribute__((aligned(16))) uint32_t g_zero_d[4] = {0, 0, 0, 0};

template <int I, int N, class F>
__device__ __forceinline__ void duo_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        duo_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void duo_dma16(const void *src, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

template <int DT>
__device__ __forceinline__ void duo_mfma(f32x4 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {
    if constexpr (DT == PIO_DT_F16) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int DT>
__device__ __forceinline__ void duo_mfma_first(f32x4 &c, typename Op<DT>::V8 a, typename Op<DT>::V8 b) {
    if constexpr (DT == PIO_DT_F16) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ float duo_acc_read(float a) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

template <int DT>
__global__ __launch_bounds__(256, 2) void gemm_nt_duo(const GemmParams p, int tiles_m, int tiles_n, int stagger) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    __shared__ __attribute__((aligned(16))) char smem[D_SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- the tile: XCD x (= blockIdx & 7) owns tile rows [x * RM, (x + 1) * RM)
    int tm, tn;
    {
        const int b = blockIdx.x, x = b & 7, i = b >> 3;
        if ((tiles_m & 7) == 0 && (gridDim.x & 7) == 0) {
            const int RM = tiles_m >> 3;
            tm = x * RM + i / tiles_n;
            tn = i % tiles_n;
        } else {
            tm = b / tiles_n;
            tn = b - tm * tiles_n;
        }
    }
    // ---- the second workgroup of a CU (its waves sit in odd wave slots: HW_ID[3:0]) starts `stagger` cycles late
    if (stagger > 0) {
        const unsigned slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 1u;  // HW_ID, bits [3:0]
        if (slot) {
            const long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < stagger) __builtin_amdgcn_s_sleep(32);
        }
    }
    const int nph = p.K / D_BK;  // >= 4 (launcher)

    // ---- DMA side: pieces of 16 rows x 64 B; wave w owns A pieces 2w, 2w+1 and B pieces 4w..4w+3 of a slice
    const int prow = lane >> 2;
    const int qsrc = (lane & 3) ^ ((0x78 >> (2 * ((prow >> 2) & 3))) & 3);
    uint32_t voa[2], vob[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int gm = tm * D_BM + (wave * 2 + i) * 16 + prow;
        gm = gm < p.M ? gm : p.M - 1;
        voa[i] = ((uint32_t)gm * (uint32_t)p.lda + (uint32_t)(qsrc * 8)) * 2u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int gn = tn * D_BN + (wave * 4 + i) * 16 + prow;
        gn = gn < p.N ? gn : p.N - 1;
        vob[i] = ((uint32_t)gn * (uint32_t)p.ldb + (uint32_t)(qsrc * 8)) * 2u;
    }
    const char *ia = nullptr, *ib = nullptr;
    char *isb = nullptr;  // LDS base of the slot (or of the sink) + integer offsets: one LDS pointer only
    int istep = 0, ioa = 0, iob = 0;
    auto issue_begin = [&](int dph, int dslot) {  // slice index, ring slot
        if (dph < nph) {
            ia = (const char *)((const T *)p.A + dph * D_BK);
            ib = (const char *)((const T *)p.B + dph * D_BK);
            isb = smem + dslot * D_STAGE;
            ioa = wave * 2048;
            iob = D_AB + wave * 4096;
            istep = 1024;
        } else {  // past the end of K: same number of pieces, into the sink
            ia = (const char *)p.A;
            ib = (const char *)p.B;
            isb = smem + D_SINK + wave * 1024;
            ioa = 0;
            iob = 0;
            istep = 0;
        }
    };
    auto piece = [&](int i) {  // 0..1: A, 2..5: B
        uint32_t &o = i < 2 ? voa[i] : vob[i - 2];
        asm volatile("" : "+v"(o));
        if (i < 2) duo_dma16(ia + (uint64_t)o, isb + ioa + i * istep);
        else duo_dma16(ib + (uint64_t)o, isb + iob + (i - 2) * istep);
    };

    // ---- MFMA side (layout as pio_gemm_wide.hip)
    const int fr = lane & 15, fq = lane >> 4;
    auto fsw = [](int g) { return (0x78 >> (2 * (g & 3))) & 3; };
    const int fa = fr * 64 + ((fq ^ fsw(fr >> 2)) << 4);
    const int fb0 = (8 * (fr >> 2) + (fr & 3)) * 64 + ((fq ^ fsw(2 * (fr >> 2))) << 4);
    const int fb1 = (8 * (fr >> 2) + 4 + (fr & 3)) * 64 + ((fq ^ fsw(2 * (fr >> 2) + 1)) << 4);
    const int a_base = wm * 64 * 64, b_base = D_AB + wn * 128 * 64;
    V8 af[2][4], bf[8];
    f32x4 acc[4][8];

    const int o_m = tm * D_BM + wm * 64 + fr;
    const int o_n0 = tn * D_BN + wn * 128;
    const int o_n = o_n0 + fq * 8;
    float *const bstash = (float *)(smem + D_BIAS + wave * 512);
    char *const sink = (char *)g_sink_d + lane * 16;

    // ---- prologue: bias row, slices 0..2 in flight, slice 0 landed and in the fragment registers
    if (lane < 32) {
        const int n = o_n0 + lane * 4;
        const float *src = (p.bias_mode == 1 && n < p.N) ? p.bias + n : (const float *)g_zero_d;
        duo_dma16(src, bstash);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {  // (unrolled: the offset arrays must stay in registers)
        issue_begin(s, s);
#pragma unroll
        for (int i = 0; i < 6; ++i) piece(i);
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = *(const V8 *)(smem + a_base + i * 1024 + fa);
#pragma unroll
    for (int i = 0; i < 8; ++i) bf[i] = *(const V8 *)(smem + b_base + (i >> 1) * 2048 + ((i & 1) ? fb1 : fb0));

    int rslot = 1;  // ring slot of the slice the next phase reads its fragments from
    auto phase = [&](auto PC, auto FC, int g) {  // g: this phase's K slice
        constexpr int P = decltype(PC)::value;
        constexpr bool FIRST = decltype(FC)::value;
        const char *rb = smem + rslot * D_STAGE;
        duo_for<0, 8>([&](auto NI) {
            constexpr int ni = decltype(NI)::value;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                if constexpr (FIRST) duo_mfma_first<DT>(acc[mi][ni], bf[ni], af[P][mi]);
                else duo_mfma<DT>(acc[mi][ni], bf[ni], af[P][mi]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ni == 0) {
                // ---- meeting point: all but the previous phase's pieces have landed (slice g+1 complete), this
                // wave's reads of slice g are done: after the barrier slot g % 3 may be refilled and slice g+1 read
                issue_begin(g + 3, rslot == 0 ? D_NST - 1 : rslot - 1);  // slice g+3 into the slot of slice g
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // next slice: B fragment ni right behind its last MFMA of this slice, the four A fragments (other set)
            // behind the first four column groups, the six DMA pieces behind column groups 1..6
            bf[ni] = *(const V8 *)(rb + b_base + (ni >> 1) * 2048 + ((ni & 1) ? fb1 : fb0));
            if constexpr (ni < 4) af[P ^ 1][ni] = *(const V8 *)(rb + a_base + ni * 1024 + fa);
            if constexpr (ni >= 1 && ni <= 6) piece(ni - 1);
            if constexpr (ni == 7) rslot = rslot == D_NST - 1 ? 0 : rslot + 1;
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    phase(I0{}, std::true_type{}, 0);
    phase(I1{}, std::false_type{}, 1);
#pragma unroll 1
    for (int ph = 2; ph < nph; ph += 2) {
        phase(I0{}, std::false_type{}, ph);
        phase(I1{}, std::false_type{}, ph + 1);
    }

    // ---- epilogue: x = acc + bias + (hi + lo); out: hi, lo [, fp32], per-row partial sums of this wave's 128 columns
    const bool interior = (tm + 1) * D_BM <= p.M && (tn + 1) * D_BN <= p.N;
    float *const cf = (float *)p.C;
    const V8 zero8 = {};
    // the residual pair of group g = (mi, pp) (16 rows x this lane's 8 columns) travels three groups ahead of its use
    // in a ring of four register pairs (a whole row block in flight does not fit beside 128 accumulators)
    V8 rh[4], rl[4];
    auto r_load = [&](int g) {
        const int m = o_m + (g >> 2) * 16, n = o_n + (g & 3) * 32;
        if (interior || (m < p.M && n < p.N)) {
            rh[g & 3] = *(const V8 *)((const T *)p.R16_hi + (int64_t)m * p.ld16 + n);
            rl[g & 3] = *(const V8 *)((const T *)p.R16_lo + (int64_t)m * p.ld16 + n);
        } else {
            rh[g & 3] = zero8;
            rl[g & 3] = zero8;
        }
    };
    r_load(0);
    r_load(1);
    r_load(2);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = o_m + mi * 16;
        float rsum = 0.f, rsq = 0.f;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int g = mi * 4 + pp;
            if (g + 3 < 16) r_load(g + 3);
            __builtin_amdgcn_sched_barrier(0);
            const int n = o_n + pp * 32;
            const f32x4 b0 = *(const f32x4 *)(bstash + pp * 32 + fq * 8);
            const f32x4 b1 = *(const f32x4 *)(bstash + pp * 32 + fq * 8 + 4);
            const V8 hh = rh[g & 3], ll = rl[g & 3];
            f32x4 x0, x1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x0[r] = duo_acc_read(acc[mi][2 * pp][r]) + b0[r] + (Op<DT>::to_f32(hh[r]) + Op<DT>::to_f32(ll[r]));
                x1[r] = duo_acc_read(acc[mi][2 * pp + 1][r]) + b1[r] +
                        (Op<DT>::to_f32(hh[4 + r]) + Op<DT>::to_f32(ll[4 + r]));
            }
            if (!interior && n >= p.N) {
                x0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                x1 = x0;
            }
            const bool ok = interior || (m < p.M && n < p.n_store);
            if (cf) {
                char *dst = ok ? (char *)(cf + (int64_t)m * p.ldc + n) : sink;
                *(f32x4 *)dst = x0;
                *(f32x4 *)(ok ? dst + 16 : dst) = x1;
            }
            V8 h, l;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rsum += x0[r] + x1[r];
                rsq += x0[r] * x0[r] + x1[r] * x1[r];
                h[r] = Op<DT>::from_f32(x0[r]);
                h[4 + r] = Op<DT>::from_f32(x1[r]);
                l[r] = Op<DT>::from_f32(x0[r] - Op<DT>::to_f32(h[r]));
                l[4 + r] = Op<DT>::from_f32(x1[r] - Op<DT>::to_f32(h[4 + r]));
            }
            *(V8 *)(ok ? (char *)((T *)p.X16 + (int64_t)m * p.ld16 + n) : sink) = h;
            *(V8 *)(ok ? (char *)((T *)p.X16_lo + (int64_t)m * p.ld16 + n) : sink) = l;
            __builtin_amdgcn_sched_barrier(0);
        }
        rsum += __shfl_xor(rsum, 16);
        rsq += __shfl_xor(rsq, 16);
        rsum += __shfl_xor(rsum, 32);
        rsq += __shfl_xor(rsq, 32);
        if (fq == 0 && m < p.M && o_n0 < p.N) {
            float *dstp = p.row_part + ((int64_t)m * (p.N >> 7) + (o_n0 >> 7)) * 2;
            dstp[0] = rsum;
            dstp[1] = rsq;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The producer form this kernel implements: fp32-grade result as a 16-bit pair, residual as a 16-bit pair, row sums
bool gemm_duo_ok(const GemmParams &p, int batch) {
    if (batch != 1 || p.npass != 1) return false;
    if (!p.X16 || !p.X16_lo || !p.row_part || !p.R16_hi || !p.R16_lo || p.R || !p.out_f32) return false;
    if (p.ln_part || p.ln_c || p.act != 0 || p.alpha != 1.0f || p.C_lo) return false;
    if (p.K < 4 * D_BK || (p.K % (2 * D_BK))) return false;
    if (p.bias_mode > 1 || (p.bias_mode == 1 && !p.bias_vec)) return false;
    if ((p.N & 127) || (p.n_store & 7) || (p.ld16 & 7)) return false;
    if (p.C && ((p.ldc & 3) || ((uintptr_t)p.C & 15))) return false;
    if ((p.lda & 7) || (p.ldb & 7) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return false;
    if (((uintptr_t)p.X16 & 15) || ((uintptr_t)p.X16_lo & 15) || ((uintptr_t)p.R16_hi & 15) ||
        ((uintptr_t)p.R16_lo & 15) || ((uintptr_t)p.row_part & 7))
        return false;
    if (((int64_t)p.M * p.lda + p.K) * 2 >= (1ll << 32) || ((int64_t)p.N * p.ldb + p.K) * 2 >= (1ll << 32)) return false;
    return true;
}

void gemm_duo_launch(const GemmParams &p, int dtype, hipStream_t s) {
    const int tiles_m = (p.M + D_BM - 1) / D_BM, tiles_n = (p.n_store + D_BN - 1) / D_BN;
    // the late workgroup of each CU starts about half a main loop behind (a phase is ~1000 cycles with two
    // workgroups on the CU); env PIO_DUO_STAGGER overrides (cycles; 0 = no stagger)
    static const int stagger_env = [] {
        const char *e = getenv("PIO_DUO_STAGGER");
        return e ? atoi(e) : -1;
    }();
    const int stagger = stagger_env >= 0 ? stagger_env : (p.K / D_BK) * 500;
    dim3 grid((unsigned)(tiles_m * tiles_n), 1, 1), block(256, 1, 1);
    if (dtype == PIO_DT_F16) hipLaunchKernelGGL((gemm_nt_duo<PIO_DT_F16>), grid, block, 0, s, p, tiles_m, tiles_n, stagger);
    else hipLaunchKernelGGL((gemm_nt_duo<PIO_DT_BF16>), grid, block, 0, s, p, tiles_m, tiles_n, stagger);
}

}  // namespace pio
