/*
 * pio_hip.h -- C-ABI of libpio_hip.so: the MI355X (gfx950) PerceiverIO forward hot path.
 *
 * This is the drop-in boundary for the reference's perceiver_io/transformer_primitives.py and the
 * encoder/decoder drivers of perceiver_io/perceiver.py.  The reference is pure Python and has no
 * FFI of its own; each entry point below names the reference callable
 * (file:line under /root/reference) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add to bind them.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures: `stream` is a hipStream_t passed as void*.
 *   - every pointer is a DEVICE pointer owned by the caller; compute calls allocate and free nothing and
 *     keep no state between calls => they are thread-safe per (stream, workspace).  The only process-wide
 *     mutable state is opt-in tooling, none of it thread-safe: the per-launch profiler (pio_prof_*) and the
 *     A/B switches pio_ln_fold_enable / pio_gemm_kernel_override / pio_set_cu_budget (set them before
 *     concurrent use).
 *   - scratch memory is caller-provided: query pio_*_workspace_bytes() first.
 *   - tensors at the boundary are float32, last dimension contiguous; batch / row strides are given
 *     in ELEMENTS (a batch stride of 0 is a broadcast view, e.g. the latent table of
 *     position_encoding.py:117-121).
 *   - masks are uint8 (0 = masked out), the storage of torch.bool.
 *   - return value: PIO_OK (0) or a negative PIO_E_* code; no exceptions cross the ABI.
 *   - matrix operands inside the library are fp16 (default) or bf16 with fp32 accumulation;
 *     LayerNorm, softmax, bias, GELU and the residual stream are fp32.
 */
#ifndef PIO_HIP_H
#define PIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIO_VERSION 100 /* 0.1.0 */

enum {
    PIO_OK = 0,
    PIO_E_SHAPE = -1,     /* unsupported / inconsistent shape                       */
    PIO_E_ALIGN = -2,     /* pointer or stride alignment requirement violated       */
    PIO_E_ARCH = -3,      /* device is not gfx950                                   */
    PIO_E_WORKSPACE = -4, /* workspace too small                                    */
    PIO_E_LAUNCH = -5,    /* HIP launch error (hipGetLastError)                     */
    PIO_E_ARG = -6,       /* NULL / invalid argument                                */
    PIO_E_RANGE = -7      /* values left the range of the 16-bit operand dtype (raised by the host-side guard of the
                             LayerNorm-folded stack, whose residual stream is an fp16 pair: see pio_ln_fold_t)     */
};

/* operand dtype of the MFMA matrices (accumulation is always fp32) */
enum { PIO_DT_F16 = 0, PIO_DT_BF16 = 1 };

/* --- descriptors (host structs holding device pointers) --------------------------------------- */

/* One nn.Linear packed for the kernels: `y = x W^T + b`, W is [out,in] in the reference
 * (transformer_primitives.py:73-75,86 / 201-206).  Packed image: [n_pad][k_pad] operand dtype, K
 * contiguous, zero padded; optional second image w_lo = W - float(w_hi) (two-pass weights). */
typedef struct pio_linear_t {
    const void *w_hi;
    const void *w_lo;  /* NULL => single pass */
    const float *bias; /* [n_pad] zero padded, or NULL */
    int32_t n;         /* padded output features: heads x pio_pad8(head dim); pio_padc(hidden) for MLP.fc1         */
    int32_t k;         /* padded input features: pio_padc(channels); heads x pio_pad8(dv) for Attention.final      */
    int32_t lo_row0;   /* w_lo holds rows [lo_row0, n) only (a stacked q|k|v image whose v part alone is split);
                          0 = every row.  Must be a multiple of 256; honoured by the wide GEMM kernel only.          */
} pio_linear_t;

/* nn.LayerNorm(c) (transformer_primitives.py:270-271, 365-367) */
typedef struct pio_layernorm_t {
    const float *gamma;
    const float *beta;
    int32_t c;
    float eps;
} pio_layernorm_t;

/* Attention (transformer_primitives.py:34-88): proj_q/k/v + final.  Per-head widths are padded to
 * multiples of 8 inside the packed images (dkp, dvp); logical widths are dk, dv. */
typedef struct pio_attention_t {
    pio_linear_t q, k, v, o;
    int32_t heads;
    int32_t dk, dv;   /* logical channels per head for q/k and v  */
    int32_t dkp, dvp; /* padded (multiple of 8)                    */
    int32_t q_in, k_in, v_in, out; /* logical channel counts (k_in == v_in inside Self/CrossAttention) */
    int32_t dtype;     /* PIO_DT_*                                  */
    int32_t act_split; /* 1: activations are carried as hi+lo pairs too (3 MFMA sweeps, ~fp32 products);
                          2: the same for the four projections, but the attention core (Q K^T, softmax, P V) runs
                          single-sweep on the hi halves through the fused kernel wherever one covers the shape (no
                          score matrix) and returns its output as a pair                                      */
    pio_linear_t qk;   /* optional (w_hi may be NULL): proj_q and proj_k stacked along the output rows
                          [q rows | k rows], used as ONE GEMM when inputs_q and inputs_k are the same tensor */
    pio_linear_t qkv;  /* optional (w_hi may be NULL): [q rows | k rows | v rows] for self-attention with head widths
                          the fused kernel covers: one GEMM, V consumed row-major by the fused attention kernel */
    /* Optional K / V projection fold of a SINGLE-HEAD cross-attend over many keys (both w_hi non-NULL; SURVEY.md section 7,
     * transformer_primitives.py:93-95,138,163): q (x Wk^T + bk)^T = (q Wk) x^T + const, and P (x Wv^T + bv) Wo^T + bo =
     * (P x) (Wo Wv)^T + (Wo bv + bo) because softmax rows sum to one.  kq = Wk^T packed as a linear [k_in <- qk] without
     * bias (Q' = Q Wk), vo = Wo Wv packed [out <- v_in] with bias Wo bv + bo.  The fused kernel then reads the
     * LayerNorm'd inputs themselves as K and (transposed) as V: the two [keys, C] x [C, C] projection GEMMs vanish.
     * Taken when heads == 1, dk == dv == k_in == v_in, inputs_k is inputs_v, no mask / bias / probabilities, and
     * keys outnumber query rows at least 4 : 1. */
    pio_linear_t kq, vo;
} pio_attention_t;

/* MLP (transformer_primitives.py:183-216) */
typedef struct pio_mlp_t {
    pio_linear_t fc1, fc2;
    int32_t in, hidden, out;
    int32_t dtype;
    int32_t act_split;
} pio_mlp_t;

/* Optional LayerNorm fold of a SelfAttention block (all pointers NULL = not available): the two GEMMs that
 * consume a LayerNorm output take the un-normalised 16-bit activations instead and apply mean / rstd in their
 * epilogue; the GEMMs that produce the LayerNorm input leave the row statistics (pio_gemm_t.X16 / row_part).
 * qkv: [q | k | v] rows packed from W * gamma1 with bias W beta1 + b; fc1: from W1 * gamma2, bias W1 beta2 + b1;
 * *_c[n] = sum_k of the packed 16-bit weights of row n.  Used for 1024-channel blocks under the single-sweep
 * policies when the block has at least 2048 rows and no mask / bias / probabilities are requested. */
typedef struct pio_ln_fold_t {
    pio_linear_t qkv;
    const float *qkv_c;
    pio_linear_t fc1;
    const float *fc1_c;
    /* Range guard of the folded stack (optional device word, zeroed by the caller): inside the fold the residual stream
     * is a 16-bit pair, so an activation beyond the operand range (fp16: 65504) turns into inf / NaN there.  Every GEMM
     * that produces the stream ORs 1 into *range_flag when a row statistic of its result is not finite -- no extra
     * kernel, no host synchronisation; the caller reads the word when it next synchronises anyway and re-runs the call
     * un-folded (PIO_E_RANGE if that does not help).  NULL: not reported. */
    int32_t *range_flag;
} pio_ln_fold_t;

/* SelfAttention (transformer_primitives.py:219-297) */
typedef struct pio_self_attention_t {
    pio_layernorm_t ln1, ln2;
    pio_attention_t attn;
    pio_mlp_t mlp;
    pio_ln_fold_t fold;
} pio_self_attention_t;

/* CrossAttention (transformer_primitives.py:300-406) */
typedef struct pio_cross_attention_t {
    pio_layernorm_t ln_q, ln_kv, ln2;
    pio_attention_t attn;
    pio_mlp_t mlp;
    int32_t use_query_residual;
} pio_cross_attention_t;

/* A float32 activation tensor [B, T, C] at the boundary: element strides, C contiguous. */
typedef struct pio_tensor3_t {
    const float *data;
    int64_t stride_b; /* 0 = broadcast over the batch */
    int64_t stride_t;
    int32_t B, T, C;
} pio_tensor3_t;

/* --- library / device ---------------------------------------------------------------------------- */
int pio_version(void);
/* 1 when the current HIP device is gfx950, 0 otherwise, <0 on error. */
int pio_arch_ok(void);
const char *pio_error_string(int code);

/* --- per-launch timing for benchmarks (NOT thread-safe, off by default) ------------------------ */
/* classes: 0 gemm_nt_256 (flat 256x256 tiles), 1 batched gemm_nt_128 (materialised attention products),
 *          2 layernorm / cast / row statistics, 3 softmax_rows, 4 pack, 5 fused attention (flash_attn),
 *          6 flat gemm_nt_128 (small / ragged problems), 7 gemm_nt_stream (persistent 256x128 tiles: fp32 + residual
 *          projections outside a folded stack), 8 gemm_nt_wide (persistent 256x256 four-wave kernel:
 *          every weight GEMM of the latent self-attend stack -- q|k|v, out, fc1, fc2 -- and the decoder projections) */
#define PIO_PROF_CLASSES 9
/* Start recording a HIP-event pair around every kernel launch (up to max_records launches). */
int pio_prof_begin(int32_t max_records);
/* Stop, wait for the recorded launches and sum per class: device milliseconds, ALGORITHMIC flops
 * (2*M*N*K per product, extra precision sweeps not counted), algorithmic bytes, launch count.
 * Arrays have PIO_PROF_CLASSES entries (any may be NULL).  Returns the number of records or <0. */
int pio_prof_end(double *ms, double *flops, double *bytes, int64_t *launches);

/* --- LayerNorm fold of the SelfAttention blocks (pio_ln_fold_t), for tests and A/B benchmarks ------ */
/* 0: never; 1: where it pays (default; env PIO_LN_FOLD gives the initial value) -- blocks of >= 6144 rows (env
 * PIO_LN_FOLD_MIN_ROWS), below which the un-folded block's smaller GEMM tiles fill the chip better; 2: wherever a
 * block offers it (>= 2048 rows).  Returns the previous setting. */
int pio_ln_fold_enable(int on);

/* --- kernel selection of pio_gemm_nt, for tests and A/B benchmarks ------------------------------- */
/* 0: automatic (default; env PIO_GEMM_TILE gives the initial value), 64: 64x64 tiles of the 128-tile kernel (automatic
 * for problems with fewer than 192 tiles of 128x128: small batches), 128: 128x128 tile, 256: 256x256 tile,
 * 1: persistent 256x128 streaming kernel wherever it is legal, 2: persistent 256x256 four-wave kernel wherever
 * it is legal.  Returns the previous setting.  (Experiment kernels that measured level with these -- a two-workgroups-
 * per-CU fold producer, MFMA 32x32x16 variants of the fold GEMMs, two more fused self-attention kernels -- are not in
 * this library: tools/experiments, built by `make -C perceiverio_pytorch_amd/csrc experiments`.) */
int pio_gemm_kernel_override(int which);

/* --- running on a part of the chip ----------------------------------------------------------------- */
/* A stream whose kernels run only on the CUs whose bit is set in `mask` (`words` 32-bit words; bit i = CU i / 8 of
 * XCD i % 8 on MI355X, the layout of hipExtStreamCreateWithCUMask).  Two such streams with complementary masks run
 * two independent kernel chains side by side: one chain's HBM-bound epilogues overlap the other's MFMA main loops
 * (PerceiverEncoder.forward with PIO_CU_SPLIT=1 runs the two halves of a batch that way).  Destroy with
 * pio_stream_destroy. */
int pio_stream_create_cu_mask(void **stream, const uint32_t *mask, uint32_t words);
int pio_stream_destroy(void *stream);
/* Process-wide (not thread-safe): the number of CUs the persistent GEMM kernels size their grids for; 0 (default) =
 * every CU of the device.  Set it to the population of a stream's CU mask around the calls that launch on that
 * stream.  Returns the previous value. */
int pio_set_cu_budget(int32_t n_cu);

/* --- weight packing (one-off, after load_state_dict) ------------------------------------------ */
/* Round a head dim / key count up to the packing granule (8). */
int32_t pio_pad8(int32_t c);
/* Pitch of a CHANNEL axis in the 16-bit operand arrays (= pio_linear_t.k of the layers that read it, and
 * pio_linear_t.n of MLP.fc1): a multiple of 8; from 512 channels on a multiple of 64 (322 -> 328, 1026 -> 1088). */
int32_t pio_padc(int32_t c);
/* Bytes of ONE packed image (hi or lo) for out x in with per-head padding of rows / columns:
 * rows = row_heads groups of (out/row_heads) padded to 8 each; columns likewise. */
size_t pio_packed_weight_bytes(int32_t out, int32_t in, int32_t row_heads, int32_t col_heads);
/* Pack W [out,in] fp32 (row stride ldw) into dst_hi (and dst_lo when non-NULL) starting at packed
 * row `dst_row0` of an image with `k_pad` columns.  Rows are split in `row_heads` equal groups, each
 * padded to a multiple of 8 rows; columns likewise with `col_heads`.  Padding is written as zeros.
 * bias (may be NULL) is packed the same way into dst_bias[dst_row0...]. */
int pio_pack_linear(const float *w, const float *bias, int32_t out, int32_t in, int64_t ldw,
                    int32_t row_heads, int32_t col_heads, void *dst_hi, void *dst_lo, float *dst_bias,
                    int32_t dst_row0, int32_t k_pad, int32_t dtype, void *stream);

/* --- primitive kernels (exposed for tests and for callers that compose their own blocks) ------- */
/* y[r, 0:c_pad] = operand_dtype( LN(x[r, 0:c]) ) with zero fill of [c, c_pad); ln == NULL => plain
 * cast.  Rows are (b, t) of x.  y_lo (optional) receives the rounding residual v - float(y).
 * Replaces nn.LayerNorm + the implicit cast in front of every GEMM. */
int pio_layernorm_cast(const pio_tensor3_t *x, const pio_layernorm_t *ln, void *y, void *y_lo, int32_t c_pad,
                       int32_t dtype, void *stream);

/* The same LayerNorm over the channel-wise CONCATENATION [x1 | x2] of two arrays, never materialised: x1 [B,T,C1],
 * x2 [B,T,C2] or ONE batch-invariant table [1,T,C2] (x2->B == 1); ln->c == C1 + C2; C1, C2 even, rows 8-byte aligned.
 * Replaces torch.cat([features, position_features], -1) of preprocessors.py:176-200 followed by layer_norm_kv
 * (transformer_primitives.py:379); bit-identical to pio_layernorm_cast of the concatenated array. */
int pio_layernorm_cast_cat(const pio_tensor3_t *x1, const pio_tensor3_t *x2, const pio_layernorm_t *ln, void *y,
                           void *y_lo, int32_t c_pad, int32_t dtype, void *stream);

/* Tail of Conv2DDownsample (processor_utils.py:163-180) in one pass over the conv's output x [B,C,H,W] fp32:
 * y[b, oh*OW + ow, c] = max over the 3x3 stride-2 TF-"SAME" window of relu(x * scale[c] + shift[c]) -- eval-mode
 * BatchNorm folded into (scale, shift) = (gamma / sqrt(var + eps), beta - mean * scale), or (1, 0) without one.
 * OH = ceil(H/2), OW = ceil(W/2); pad_top / pad_left = the leading SAME padding (0 or 1).  y is the channels-last token
 * array [B, OH*OW, C] the encoder consumes. */
int pio_bn_relu_maxpool_tokens(const float *x, const float *scale, const float *shift, float *y, int32_t B, int32_t C,
                               int32_t H, int32_t W, int32_t pad_top, int32_t pad_left, void *stream);

/* C = epilogue(alpha * A B^T): A [M,K], B [N,K] operand dtype, K contiguous (multiple of 8).
 * Batched over z = zb*nh + zh with element strides; bias_mode 0 none / 1 per column / 2 per row;
 * act 0 none / 1 exact-erf GELU (F.gelu, transformer_primitives.py:214); optional fp32 residual R
 * added after the activation (row m of the flattened problem reads
 * R + (m / r_rows_per_batch)*r_stride_b + (m % r_rows_per_batch)*ldr when r_rows_per_batch > 0);
 * output fp32 (out_f32=1) or operand dtype with zero fill of columns [N, n_store). */
typedef struct pio_gemm_t {
    const void *A, *B;
    const void *A_lo, *B_lo; /* optional extra K sweeps: C += A*B_lo^T, C += A_lo*B^T (split operands) */
    void *C;
    void *C_lo;              /* optional (16-bit output only): residual C - float(C16), same layout as C */
    int32_t M, N, K;
    int64_t lda, ldb, ldc;
    int32_t batch, nh;
    int64_t sAb, sAh, sBb, sBh, sCb, sCh;
    const float *bias;
    int32_t bias_mode;
    int32_t act;
    float alpha;
    const float *R;
    int64_t ldr, r_stride_b;
    int32_t r_rows_per_batch;
    int32_t out_f32;
    int32_t n_store;
    int32_t dtype;
    /* LayerNorm folded into the GEMMs around it (optional; a shape neither fold kernel takes: PIO_E_SHAPE).
     * Producer (fp32 out): X16 receives a 16-bit copy of the result (row stride ld16) and row_part, [M][N/w][2]
     * fp32, the (sum, sum of squares) of every result row over each w-column slot (w = row_slot_w below).
     * Consumer (16-bit out): ln_part is the row_part a producer wrote for this GEMM's A operand and
     * ln_c[n] = sum_k B[n,k]; the result is rstd_m * (A B^T)[m,n] - rstd_m * mean_m * ln_c[n] + bias[n] [GELU], i.e.
     * LayerNorm(x) W^T + b for B = W * gamma, bias = W beta + b (transformer_primitives.py:281-292). */
    void *X16;
    int64_t ld16;
    float *row_part;
    const float *ln_part;
    const float *ln_c;
    float ln_eps;
    /* The residual stream as a 16-bit pair (producer only): X16_lo receives result - float(X16) (same layout as X16),
     * and the residual may be given as R16_hi + R16_lo (row stride ld16, instead of the fp32 R).  With X16 and X16_lo
     * both set, C may be NULL: the fp32 result is then not written at all. */
    void *X16_lo;
    const void *R16_hi, *R16_lo;
    /* B_lo holds rows (= output columns) [b_lo_n0, N) only: the extra sweep C += A B_lo^T runs for those columns
     * alone.  0 = all rows.  Multiple of 256; kernel gemm_nt_wide only (else PIO_E_SHAPE). */
    int32_t b_lo_n0;
    /* LayerNorm-fold producer only (optional): *range_flag |= 1 when a row's (sum, sum of squares) is not finite */
    int32_t *range_flag;
    /* Slot form of the row statistics.  Producer: row_slot_w = columns per (sum, sum of squares) slot of row_part,
     * [M][N / row_slot_w][2]: 0 or 128 = the 256x256-tile kernel (gemm_nt_wide: N % 128 == 0), 64 = the 128 / 64-tile
     * kernel (stacks of fewer rows than the wide kernel takes; N % 64 == 0, residual and result as 16-bit pairs).
     * Consumer: ln_slots = slots per row of ln_part (0 = K / 128).  K / 128 slots, K % 128 == 0, K <= 1536 and
     * M >= 2048 run on gemm_nt_wide, anything else on the tile kernel. */
    int32_t ln_slots;
    int32_t row_slot_w;
} pio_gemm_t;
int pio_gemm_nt(const pio_gemm_t *g, void *stream);

/* P = softmax_j((S + bias) * scale) with masking, rows of length Tk (transformer_primitives.py:143-158,
 * 168-175): S fp32 [B,H,Tq,Tk] (row stride lds), P operand dtype [B,H,Tq,ldp] zero filled to ldp.
 * kv_mask [B,Tk], q_mask [B,Tq], full_mask [B,Tq,Tk] are optional uint8.  A row with no attendable key
 * is written as zeros (the reference's "wipe").  bias optional fp32 [B,H,Tq,Tk]; P_lo optional residual. */
int pio_softmax_rows(const float *S, int64_t lds, void *P, void *P_lo, int64_t ldp, int32_t B, int32_t H,
                     int32_t Tq, int32_t Tk, float scale, const uint8_t *kv_mask, const uint8_t *q_mask,
                     const uint8_t *full_mask, const float *bias, int32_t dtype, void *stream);

/* O = softmax(Q K^T / sqrt(dk)) V without the score matrix in HBM, un-masked (transformer_primitives.py:138-166 for the
 * latent self-attention): 16-bit Q [B][Tq][.. h*dkp ..] (row stride ldq, batch stride sQb elements), K likewise, V either
 * K-contiguous V^T [B][H*dvp][keys] (row stride ldv) or -- v_rowmajor != 0 -- row-major [B][Tk][.. h*dvp ..] (the layout ONE
 * fused q|k|v GEMM writes), O [B][Tq][.. h*dvp ..].  (dkp, dvp) in {(128,128), (64,64), (32,32), (32,160)}; dk = the logical
 * per-head width the scale uses.  The kernel the SelfAttention blocks run; exposed for tests and tools/r4_ceiling.py. */
int pio_flash_attention(int32_t dtype, int32_t dkp, int32_t dvp, int32_t dk, const void *Q, const void *K, const void *V,
                        void *O, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int64_t ldq, int64_t ldk, int64_t ldv,
                        int64_t ldo, int64_t sQb, int64_t sKb, int64_t sVb, int64_t sOb, int32_t v_rowmajor, void *stream);

/* --- blocks: the reference's nn.Module.forward calls ------------------------------------------ */
/* Attention.forward (transformer_primitives.py:90-115 + attend 117-180).  inputs_k and inputs_v must
 * have equal shapes.  out [B,Tq,out] fp32 contiguous.  probs_out (optional, fp32 [B,H,Tq,Tk]) receives
 * the attention matrix (return_matrix=True). */
size_t pio_attention_workspace_bytes(const pio_attention_t *a, int32_t B, int32_t Tq, int32_t Tk);
int pio_attention_fwd(const pio_attention_t *a, const pio_tensor3_t *inputs_q, const pio_tensor3_t *inputs_k,
                      const pio_tensor3_t *inputs_v, const uint8_t *kv_mask, const uint8_t *q_mask,
                      const uint8_t *full_mask, const float *attention_bias, float *out, float *probs_out,
                      void *workspace, size_t workspace_bytes, void *stream);

/* MLP.forward (transformer_primitives.py:212-216). out [B,T,out] fp32 contiguous. */
size_t pio_mlp_workspace_bytes(const pio_mlp_t *m, int64_t rows);
int pio_mlp_fwd(const pio_mlp_t *m, const pio_tensor3_t *x, float *out, void *workspace,
                size_t workspace_bytes, void *stream);

/* Per-call options of the *_opts entry points (NULL or all zero = the library defaults).  They replace, call by call
 * and thread by thread, the process-wide switches pio_ln_fold_enable / pio_set_cu_budget (which stay as test / A-B
 * overrides): two threads driving two encoders on two devices never see each other's choice.
 *   ln_fold:   0 = process-wide setting, 1 = LayerNorm fold off (the range guard's un-folded repeat), 2 = where it pays,
 *              3 = wherever a block offers it
 *   cu_budget: CUs the persistent GEMM kernels size their grids for (a CU-masked stream's share); 0 = process-wide */
typedef struct pio_call_opts_t {
    int32_t ln_fold;
    int32_t cu_budget;
} pio_call_opts_t;

/* SelfAttention.forward (transformer_primitives.py:275-297), no mask (perceiver.py:106 never passes one;
 * masks, attention_bias [B,H,N,N] and probs_out [B,H,N,N] (return_matrix) are accepted for interface
 * parity, all optional).  out [B,N,D] fp32 contiguous; out may alias x.data when x is contiguous. */
size_t pio_self_attention_workspace_bytes(const pio_self_attention_t *s, int32_t B, int32_t N);
int pio_self_attention_fwd(const pio_self_attention_t *s, const pio_tensor3_t *x, const uint8_t *kv_mask,
                           const uint8_t *q_mask, const uint8_t *full_mask, const float *attention_bias,
                           float *out, float *probs_out, void *workspace, size_t workspace_bytes, void *stream);

int pio_self_attention_fwd_opts(const pio_self_attention_t *s, const pio_tensor3_t *x, const uint8_t *kv_mask,
                                const uint8_t *q_mask, const uint8_t *full_mask, const float *attention_bias,
                                float *out, float *probs_out, void *workspace, size_t workspace_bytes, void *stream,
                                const pio_call_opts_t *opts);

/* CrossAttention.forward (transformer_primitives.py:371-406). out [B,Tq,q_in] fp32 contiguous. */
size_t pio_cross_attention_workspace_bytes(const pio_cross_attention_t *c, int32_t B, int32_t Tq, int32_t Tk);
int pio_cross_attention_fwd(const pio_cross_attention_t *c, const pio_tensor3_t *inputs_q,
                            const pio_tensor3_t *inputs_kv, const uint8_t *kv_mask, const uint8_t *q_mask,
                            const uint8_t *full_mask, const float *attention_bias, float *out, float *probs_out,
                            void *workspace, size_t workspace_bytes, void *stream);

/* PerceiverEncoder.forward (perceiver.py:98-107): cross-attend(latents <- inputs, key mask = input_mask)
 * then num_blocks x [layers[0..L)] weight-shared self-attends.  `latents` is the query tensor the
 * caller built (PerceiverEncoder.latents, perceiver.py:94-96: normally a stride-0 broadcast of the
 * [N,D] table).  out [B,N,D] fp32 contiguous. */
size_t pio_encoder_workspace_bytes(const pio_cross_attention_t *cross, const pio_self_attention_t *layers,
                                   int32_t L, int32_t B, int32_t M, int32_t N);
int pio_encoder_fwd(const pio_cross_attention_t *cross, const pio_self_attention_t *layers, int32_t L,
                    int32_t num_blocks, const pio_tensor3_t *inputs, const pio_tensor3_t *latents,
                    const uint8_t *input_mask, float *out, void *workspace, size_t workspace_bytes,
                    void *stream);

/* The same with the encoder input given as two channel-wise concatenated arrays [inputs | inputs_tail] (inputs_tail
 * may be NULL = pio_encoder_fwd; inputs_tail->B == 1: one batch-invariant table): the ImageNet preprocessor's 64 conv
 * features [B,3136,64] and its [3136,258] Fourier table instead of the replicated [B,3136,322] array
 * (preprocessors.py:176-200).  Workspace as pio_encoder_workspace_bytes. */
int pio_encoder_fwd_split(const pio_cross_attention_t *cross, const pio_self_attention_t *layers, int32_t L,
                          int32_t num_blocks, const pio_tensor3_t *inputs, const pio_tensor3_t *inputs_tail,
                          const pio_tensor3_t *latents, const uint8_t *input_mask, float *out, void *workspace,
                          size_t workspace_bytes, void *stream);

/* The same with ONE SET OF PACKED IMAGES PER BLOCK (per_block != 0): block b runs layers[b*L .. b*L + L).  The
 * reference shares the parameters of the L layers over the num_blocks blocks (perceiver.py:104-106); with one 16-bit
 * image per weight its rounding error then acts num_blocks times in the same direction.  Under the "fp16sd" policy
 * the binder packs block b from fp16(W + E_{b-1}), E_b = (W + E_{b-1}) - image_b (error feedback over the block index:
 * every image is a rounding of W within one ulp, their accumulated error stays below half an ulp) -- same shapes,
 * same kernels, same time.  per_block == 0: pio_encoder_fwd_split.  Workspace as pio_encoder_workspace_bytes. */
int pio_encoder_fwd_blocks(const pio_cross_attention_t *cross, const pio_self_attention_t *layers, int32_t L,
                           int32_t num_blocks, int32_t per_block, const pio_tensor3_t *inputs,
                           const pio_tensor3_t *inputs_tail, const pio_tensor3_t *latents, const uint8_t *input_mask,
                           float *out, void *workspace, size_t workspace_bytes, void *stream);

/* pio_encoder_fwd_blocks with per-call options (opts == NULL: the same call). */
int pio_encoder_fwd_opts(const pio_cross_attention_t *cross, const pio_self_attention_t *layers, int32_t L,
                         int32_t num_blocks, int32_t per_block, const pio_tensor3_t *inputs,
                         const pio_tensor3_t *inputs_tail, const pio_tensor3_t *latents, const uint8_t *input_mask,
                         float *out, void *workspace, size_t workspace_bytes, void *stream,
                         const pio_call_opts_t *opts);

/* PerceiverDecoder.forward (perceiver.py:166-180): cross-attend(query <- latents, query mask) and the
 * optional final nn.Linear (final == NULL => final_project=False).  out [B,Q,out_channels] fp32. */
size_t pio_decoder_workspace_bytes(const pio_cross_attention_t *cross, const pio_linear_t *final_layer,
                                   int32_t B, int32_t Q, int32_t N);
int pio_decoder_fwd(const pio_cross_attention_t *cross, const pio_linear_t *final_layer, int32_t final_out,
                    const pio_tensor3_t *query, const pio_tensor3_t *latents, const uint8_t *query_mask,
                    float *out, void *workspace, size_t workspace_bytes, void *stream);

/* The same with the PROJECTED QUERIES kept by the caller across calls (q16_hi == NULL: pio_decoder_fwd).  A decoder
 * whose query array depends on parameters and constants only (the multimodal model's per-chunk Fourier queries,
 * multimodal_perceiver.py:146-161) normalises and projects it once: q16_hi (and q16_lo when the attention descriptor
 * carries split activations) are caller-owned device buffers of pio_decoder_qcache_bytes(cross, Bq, Q) bytes each, Bq = 1
 * for a stride-0 (batch-invariant) query tensor, else B.  q16_valid == 0: LayerNorm_q + proj_q run and leave their result
 * there; != 0: both are skipped and the buffers are read.  Only for cross->use_query_residual == 0 (PIO_E_ARG otherwise:
 * the query rows themselves are needed then); the caller invalidates when the query array, layer_norm_q, proj_q or the
 * precision policy change. */
size_t pio_decoder_qcache_bytes(const pio_cross_attention_t *cross, int32_t Bq, int32_t Q);
int pio_decoder_fwd_qcache(const pio_cross_attention_t *cross, const pio_linear_t *final_layer, int32_t final_out,
                           const pio_tensor3_t *query, const pio_tensor3_t *latents, const uint8_t *query_mask,
                           float *out, void *workspace, size_t workspace_bytes, void *stream, void *q16_hi,
                           void *q16_lo, int32_t q16_valid);

/* PerceiverDecoder.forward (perceiver.py:166-180) with the query rows handed over as TWO arrays whose channels are
 * concatenated, [query | query_tail] (query_tail->B == 1: one batch-invariant table) -- the dense decoders whose queries ARE
 * the network's preprocessed input (flow_perceiver.py: FlowQuery = [conv features | Fourier position table]): layer_norm_q
 * runs over the virtual concatenation, nothing is concatenated in HBM.  Needs use_query_residual == 0. */
int pio_decoder_fwd_split(const pio_cross_attention_t *cross, const pio_linear_t *final_layer, int32_t final_out,
                          const pio_tensor3_t *query, const pio_tensor3_t *query_tail, const pio_tensor3_t *latents,
                          const uint8_t *query_mask, float *out, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIO_HIP_H */
