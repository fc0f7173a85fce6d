// Internal declarations shared by the HIP translation units of libpio_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/pio_hip.h"

namespace pio {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Operand-dtype traits: the 16-bit type fed to MFMA.
template <int DT> struct Op;
template <> struct Op<PIO_DT_F16> {
    typedef _Float16 T;
    typedef f16x8 V8;
    typedef f16x4 V4;
    static __device__ __forceinline__ T from_f32(float x) { return (T)x; }
    static __device__ __forceinline__ float to_f32(T x) { return (float)x; }
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Op<PIO_DT_BF16> {
    typedef __bf16 T;
    typedef bf16x8 V8;
    typedef bf16x4 V4;
    static __device__ __forceinline__ T from_f32(float x) { return (T)x; }  // v_cvt_pk_bf16_f32 (RNE, NaN kept)
    static __device__ __forceinline__ float to_f32(T x) { return (float)x; }
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int pad8(int c) { return (c + 7) & ~7; }
// Pitch (elements) of a CHANNEL axis in the 16-bit operand arrays = the K of the GEMM that reads them: a multiple of 8
// (one 16-byte DMA piece); from 256 channels on a multiple of 64, which the staged kernels (gemm_nt_wide / _stream:
// whole 64-deep K slices) require -- the multimodal decoder's 1026-channel queries run on 1088 (+5 % K) and its
// GEMMs 15-25 % faster for it (tools/mm_dec_gemm_bench.py); the flow decoder's 322 channels on 384 (round 4: its
// 182 528-row projections then run on gemm_nt_wide, 20-30 % faster: DESIGN_LOG R4.15).  Head dims and key counts keep pad8.
int padc_min();  // 256 (env PIO_PADC_MIN: experiments)
static inline int padc(int c) { return c >= padc_min() ? (c + 63) & ~63 : (c + 7) & ~7; }

// carve helper for caller-provided workspaces (256-byte aligned pieces)
struct Carver {
    char *base;
    size_t off;
    explicit Carver(void *p) : base((char *)p), off(0) {}
    void *take(size_t bytes) {
        void *r = base ? base + off : nullptr;
        off += (size_t)round_up((int64_t)bytes, 256);
        return r;
    }
};

static inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PIO_OK : PIO_E_LAUNCH;
}

// ---- optional per-launch timing for bench.py (pio_prof_begin / pio_prof_end); off by default ----------
enum {
    PROF_GEMM_LINEAR = 0,  // kernel gemm_nt_256 (large flat GEMMs that do not qualify for the streaming kernel)
    PROF_GEMM_ATTN = 1,    // kernel gemm_nt_128<.,1> (batched products)
    PROF_LAYERNORM = 2,
    PROF_SOFTMAX = 3,
    PROF_PACK = 4,
    PROF_FLASH = 5,
    PROF_GEMM_SMALL = 6,   // kernel gemm_nt_128<.,0> (flat problems too small / ragged for the 256 tile)
    PROF_GEMM_STREAM = 7,  // kernel gemm_nt_stream (persistent 256x128 tiles: the weight GEMMs of the latent stack)
    PROF_GEMM_WIDE = 8,    // kernel gemm_nt_wide (persistent 256x256 tiles, four waves: the fused q|k|v projection)
    PROF_CLASSES = 9
};
struct ProfScope {
    int idx;
    hipStream_t s;
    ProfScope(int cls, double flops, double bytes, hipStream_t stream);
    ~ProfScope();
};

// CUs a persistent kernel may size its grid for: the device's CU count, or the budget set with pio_set_cu_budget
// (launches on a CU-masked stream that owns only part of the chip)
int cu_budget();
int cu_budget_call(int v);  // per-call (thread-local) budget: v >= 0 sets it (0 = none); returns the previous value

// ---- internal launchers (defined in the .hip files) ---------------------------------------------
int gemm_nt_launch(const pio_gemm_t &g, hipStream_t s);
int gemm_kernel_override(int which);  // 0 auto, 128, 256, 1 = streaming; returns the previous choice
int layernorm_cast_launch(const pio_tensor3_t &x, const pio_layernorm_t *ln, void *y, void *y_lo, int c_pad,
                          int dtype, hipStream_t s);
// LayerNorm over the concatenation [x1 | x2] of two sources (x2 may be one batch-invariant table, B == 1)
int layernorm_cast_cat_launch(const pio_tensor3_t &x1, const pio_tensor3_t &x2, const pio_layernorm_t &ln, void *y,
                              void *y_lo, int c_pad, int dtype, hipStream_t s);
// LayerNorm fold: 16-bit cast of contiguous C-channel rows (C % 256 == 0) + their (sum, sum of squares) per slot of slot_w
// (64 / 128) columns, the form the fold's consumer GEMM reads (first layer of a stack: later layers get both from the
// producing GEMM)
int rowstats_cast_launch(const float *x, int64_t rows, int C, int slot_w, void *y16, void *y16_lo, float *part, int dtype,
                         hipStream_t s);
// VT[b][c][t] = X[b][t][c] (16-bit; columns t in [T, tkv) zero): the values of a K / V-folded cross-attend
int transpose16_launch(const void *x, int64_t ldx, int B, int T, int C, void *vt, int64_t tkv, hipStream_t s);
int ln_fold_enable(int on);   // returns the previous setting
bool ln_fold_enabled();
int softmax_rows_launch(const float *S, int64_t lds, void *P, void *P_lo, int64_t ldp, int B, int H, int Tq, int Tk,
                        float scale, const uint8_t *kv_mask, const uint8_t *q_mask, const uint8_t *full_mask,
                        const float *bias, int dtype, float *probs_out, hipStream_t s);
bool flash_supported(int dkp, int dvp);
int flash_attention_launch(int dtype, int dkp, int dvp, int dk_logical, const void *Q, const void *K, const void *VT,
                           void *O, int B, int H, int Tq, int Tk, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo,
                           int64_t sQb, int64_t sKb, int64_t sVb, int64_t sOb, bool v_rowmajor, hipStream_t s);
// fused cross-attention (pio_xattn.hip): wide single heads, dv != dk, key / query mask vectors, key splits
bool xattn_supported(int dkp, int dvp);
size_t xattn_partial_bytes(int dkp, int dvp, int B, int H, int Tq, int Tk);  // fp32 partials of the key splits (0: none)
int xattn_launch(int dtype, int dkp, int dvp, int dk_logical, const void *Q, const void *K, const void *VT, void *O,
                 void *O_lo, int B, int H, int Tq, int Tk, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo, int64_t sQb,
                 int64_t sKb, int64_t sVb, int64_t sOb, const uint8_t *kv_mask, const uint8_t *q_mask, void *partials,
                 hipStream_t s);
// fused cross-attention for a head wider than the key axis is long (pio_xtall.hip): Tk <= 512, S computed once per query
// row and kept in registers (the ImageNet decoder's 1024-wide head over 512 latents)
bool xtall_supported(int dkp, int dvp, int Tk);
size_t xtall_scratch_bytes(int B);  // key-bit words of a masked launch
int xtall_launch(int dtype, int dkp, int dvp, int dk_logical, const void *Q, const void *K, const void *VT, void *O,
                 void *O_lo, int B, int H, int Tq, int Tk, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo, int64_t sQb,
                 int64_t sKb, int64_t sVb, int64_t sOb, const uint8_t *kv_mask, const uint8_t *q_mask, void *scratch,
                 hipStream_t s);
// eval BatchNorm -> ReLU -> 3x3/2 SAME max-pool -> channels-last tokens (tail of Conv2DDownsample)
int bn_relu_pool_nhwc_launch(const float *x, const float *scale, const float *shift, float *y, int B, int C, int H, int W,
                             int pad_top, int pad_left, hipStream_t s);
int pack_linear_launch(const float *w, const float *bias, int out, int in, int64_t ldw, int row_heads,
                       int col_heads, void *dst_hi, void *dst_lo, float *dst_bias, int dst_row0, int k_pad,
                       int dtype, hipStream_t s);

}  // namespace pio
