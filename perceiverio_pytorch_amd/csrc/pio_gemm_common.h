// Shared by the two NT-GEMM kernels (pio_gemm.hip: 128x128 tile; pio_gemm256.hip: 256x256 tile).
#pragma once
#include "pio_internal.h"

namespace pio {

struct GemmParams {
    const void *A, *B;
    int64_t dA1, dB1, dA2, dB2;  // element offsets of the pass-1 / pass-2 operands relative to A / B
    int npass;                   // 1..3 K sweeps accumulating into the same registers
    int staged_epi;              // (gemm_nt_wide, LayerNorm-fold producer) 1: LDS-staged, row-coalesced epilogue
    int mf32;                    // (experiments build only) gemm_nt_wide fold GEMMs on the MFMA 32x32x16 variants
    int lo_n0;                   // (gemm_nt_wide only) B_lo exists for columns >= lo_n0 (multiple of 256); 0 = all
    int k_rev;                   // (gemm_nt_wide only) odd tiles of a workgroup sweep K backwards (see its walk)
    void *C, *C_lo;
    int M, N, K;
    int64_t lda, ldb, ldc;
    int nh;
    int64_t sAb, sAh, sBb, sBh, sCb, sCh;
    const float *bias;
    int bias_mode, act;
    float alpha;
    const float *R;
    int64_t ldr, r_stride_b;
    int r_rows;
    int out_f32, n_store;
    int tiles_n;
    int vec_ok;    // C rows are 16-byte (fp32) / 8-byte (16-bit) aligned for 4-column vectors
    int r_vec;     // residual rows are 16-byte aligned
    int bias_vec;  // bias is 16-byte aligned
    // LayerNorm fold (gemm_nt_wide only): producer outputs / consumer inputs, see pio_gemm_t
    void *X16;
    int64_t ld16;
    float *row_part;
    const float *ln_part, *ln_c;
    float ln_eps;
    void *X16_lo;
    const void *R16_hi, *R16_lo;
    int *range_flag;  // producer: set to 1 when a row statistic is not finite (the folded stack's fp16 range guard)
    // consumer: ln_part holds ln_slots (sum, sum of squares) pairs per row (K / 128 from the wide kernel's producer, K / 64
    // from the small-tile kernels'), ln_inv_k = 1 / K; producer: slot_w = columns per slot (128 wide kernel, 64 small tiles)
    int ln_slots;
    float ln_inv_k;
    int slot_w;
};


// Exact (erf) GELU, gelu(x) = x Phi(x), in eight instruction slots per element: with E = Phi(-|x|) = 0.5 erfc(|x| / sqrt 2)
//     gelu(x) = max(x, 0) - |x| E        (x >= 0: x (1 - E);  x < 0: x E)
// and log2 E is smooth enough for a degree-7 polynomial in a = min(|x|, 6) (weighted minimax fit, weight |x| E = the
// sensitivity of the result; beyond 6, |x| E < 6e-9).  One v_exp_f32, no v_rcp_f32, no sign transfer.  Max |error| in
// fp32 over [-10, 10]: 2.6e-7 absolute (the A&S 7.1.26 form it replaces -- v_rcp + v_exp + 5-term polynomial, ~20
// slots with the hazard padding of two dependent transcendentals: 4.6e-7); the epilogues that call it are bound by the
// number of instructions a single wave can issue, not by their memory traffic.  tools/gelu_fit.py makes the table.
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fminf(fabsf(x), 6.0f);
    float q = 3.152013050566893e-06f;
    q = fmaf(q, a, 2.9346412588893145e-07f);
    q = fmaf(q, a, -0.0006359288236126304f);
    q = fmaf(q, a, 0.007810706272721291f);
    q = fmaf(q, a, -0.05312380567193031f);
    q = fmaf(q, a, -0.45892834663391113f);
    q = fmaf(q, a, -1.15116286277771f);
    q = fmaf(q, a, -0.9999961256980896f);
    return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(q), fmaxf(x, 0.0f));
}
// Two elements at once with the polynomial on packed fp32 FMAs (coefficients in register pairs): 7.5 slots per element.
typedef float pio_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pio_f32x2 gelu_erf2(pio_f32x2 x) {
    const pio_f32x2 a = {fminf(fabsf(x[0]), 6.0f), fminf(fabsf(x[1]), 6.0f)};
    pio_f32x2 q = {3.152013050566893e-06f, 3.152013050566893e-06f};
    q = __builtin_elementwise_fma(q, a, (pio_f32x2){2.9346412588893145e-07f, 2.9346412588893145e-07f});
    q = __builtin_elementwise_fma(q, a, (pio_f32x2){-0.0006359288236126304f, -0.0006359288236126304f});
    q = __builtin_elementwise_fma(q, a, (pio_f32x2){0.007810706272721291f, 0.007810706272721291f});
    q = __builtin_elementwise_fma(q, a, (pio_f32x2){-0.05312380567193031f, -0.05312380567193031f});
    q = __builtin_elementwise_fma(q, a, (pio_f32x2){-0.45892834663391113f, -0.45892834663391113f});
    q = __builtin_elementwise_fma(q, a, (pio_f32x2){-1.15116286277771f, -1.15116286277771f});
    q = __builtin_elementwise_fma(q, a, (pio_f32x2){-0.9999961256980896f, -0.9999961256980896f});
    pio_f32x2 r;
    r[0] = fmaf(-fabsf(x[0]), __builtin_amdgcn_exp2f(q[0]), fmaxf(x[0], 0.0f));
    r[1] = fmaf(-fabsf(x[1]), __builtin_amdgcn_exp2f(q[1]), fmaxf(x[1], 0.0f));
    return r;
}


// bias / activation / residual / store of 4 consecutive output columns (m, n0..n0+3) held in v.
template <int DT>
__device__ __forceinline__ void epilogue_store4(const GemmParams &p, int64_t coffz, int m, int n0, f32x4 v,
                                                f32x4 bias_n) {
    typedef typename Op<DT>::T T;
    const bool nfull = n0 + 3 < p.N;
    const float bias_m = (p.bias_mode == 2) ? p.bias[m] : 0.f;
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    if (p.R) {
        const float *rrow = (p.r_rows > 0)
                                ? p.R + (int64_t)(m / p.r_rows) * p.r_stride_b + (int64_t)(m % p.r_rows) * p.ldr
                                : p.R + (int64_t)m * p.ldr;
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

template <int DT>
__device__ __forceinline__ f32x4 load_bias4(const GemmParams &p, int n0) {
    f32x4 bias_n = {0.f, 0.f, 0.f, 0.f};
    if (p.bias_mode == 1) {
        if (n0 + 3 < p.N && p.bias_vec) {
            bias_n = *(const f32x4 *)(p.bias + n0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n0 + r < p.N) bias_n[r] = p.bias[n0 + r];
        }
    }
    return bias_n;
}

// 256x256-tile kernel (pio_gemm256.hip); `attn` only selects the kernel-name tag.
void gemm256_launch(const GemmParams &p, int dtype, bool attn, int tiles_m, int tiles_n, int batch, hipStream_t s);

// persistent 256x128-tile streaming kernel (pio_gemm_stream.hip)
bool gemm_stream_ok(const GemmParams &p, int batch);
void gemm_stream_launch(const GemmParams &p, int dtype, int batch, hipStream_t s);

// persistent 256x256-tile four-wave kernel (pio_gemm_wide.hip): plain 16-bit-out projections
bool gemm_wide_ok(const GemmParams &p, int batch);
void gemm_wide_launch(const GemmParams &p, int dtype, hipStream_t s);

#ifdef PIO_EXPERIMENTS
// the LayerNorm fold's producer with two 128x256-tile workgroups per CU (tools/experiments/pio_gemm_duo.hip)
bool gemm_duo_ok(const GemmParams &p, int batch);
void gemm_duo_launch(const GemmParams &p, int dtype, hipStream_t s);
#endif

}  // namespace pio
