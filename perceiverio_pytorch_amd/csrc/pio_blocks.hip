// Host-side composition of the PerceiverIO hot-path blocks out of the gfx950 kernels.
// One C call per reference nn.Module.forward; every launch goes to the caller's stream, all scratch
// comes from the caller's workspace (graph-capture safe: no allocation, no synchronisation).
//
//   attention_core   <- Attention.forward / attend        transformer_primitives.py:90-180
//   mlp_core         <- MLP.forward                        transformer_primitives.py:212-216
//   self_attention   <- SelfAttention.forward              transformer_primitives.py:275-297
//   cross_attention  <- CrossAttention.forward             transformer_primitives.py:371-406
//   encoder / decoder<- PerceiverEncoder/Decoder.forward   perceiver.py:98-107, 166-180
//
// 16-bit operands travel as (hi, lo) pairs: lo == nullptr unless the descriptor says act_split
// (precision policy "*x3": every product is hi*hi + hi*lo + lo*hi, i.e. ~fp32-exact).
#include "pio_internal.h"

namespace pio {

#define PIO_TRY(expr)                  \
    do {                               \
        int _e = (expr);               \
        if (_e != PIO_OK) return _e;   \
    } while (0)

struct Pair {  // a 16-bit operand and its optional rounding residual
    void *hi = nullptr, *lo = nullptr;
};

static Pair take_pair(Carver &c, size_t elems, bool split) {
    Pair p;
    p.hi = c.take(elems * 2);
    p.lo = split ? c.take(elems * 2) : nullptr;
    return p;
}

struct Residual {
    const float *ptr = nullptr;
    int64_t ld = 0, stride_b = 0;
    int rows_per_batch = 0;
};

// Row pitch (floats) of an INTERNAL fp32 activation buffer: whole float4 groups, so that a channel count like 1026 or
// 322 does not force the GEMM epilogues and LayerNorm reads onto 4-byte accesses.  The columns [C, pitch4(C)) hold
// zeros or stale values and are never read as data.
static inline int pitch4(int c) { return (c + 3) & ~3; }

static Residual residual_of(const pio_tensor3_t &t) {
    Residual r;
    r.ptr = t.data;
    r.ld = t.stride_t;
    r.stride_b = t.stride_b;
    r.rows_per_batch = t.T;
    return r;
}

static pio_gemm_t gemm_defaults(int dtype) {
    pio_gemm_t g = {};
    g.batch = 1;
    g.nh = 1;
    g.alpha = 1.f;
    g.dtype = dtype;
    return g;
}

// LayerNorm fold plumbing of one GEMM (pio_ln_fold_t; all null = plain GEMM).  Consumer side: the 16-bit input is the
// UN-normalised activation and `in_part` its per-row partial sums -- the GEMM then uses the folded weights `w` / row
// sums `c`.  Producer side: where the (fp32 + residual) GEMM leaves the 16-bit copy and the partial sums of its result.
struct LnFold {
    const float *in_part = nullptr;
    const pio_linear_t *w = nullptr;
    const float *c = nullptr;
    float eps = 0.f;
    void *out16 = nullptr;
    int64_t ld16 = 0;
    float *out_part = nullptr;
    // ... and the residual stream as a 16-bit pair: lo half of the result, residual given as (hi, lo); with both the
    // fp32 output pointer of the GEMM may be null
    void *out16_lo = nullptr;
    const void *res16_hi = nullptr, *res16_lo = nullptr;
    int32_t *range_flag = nullptr;  // producer: the caller's range-guard word (pio_ln_fold_t.range_flag)
    // slot form of the row statistics (pio_gemm_t.ln_slots / row_slot_w): 128-column slots = the 256 x 256-tile kernel,
    // 64-column slots = the tile kernels (stacks of fewer rows)
    int in_slots = 0, slot_w = 0;
};

// y[rows, n] = x16[rows, lin.k] * W^T (+bias) (act) (+R); a 16-bit output has ldc = lin.n (padded).
static int linear_fwd(const pio_linear_t &lin_plain, int dtype, Pair x, int64_t rows, void *y, void *y_lo, bool out_f32,
                      int n_logical, int64_t ldc, int act, const Residual *res, hipStream_t s,
                      const LnFold *fold = nullptr, int n_store16 = 0) {
    const pio_linear_t &lin = (fold && fold->in_part) ? *fold->w : lin_plain;
    pio_gemm_t g = gemm_defaults(dtype);
    if (fold && fold->in_part) {
        g.ln_part = fold->in_part;
        g.ln_c = fold->c;
        g.ln_eps = fold->eps;
        g.ln_slots = fold->in_slots;
    }
    if (fold && fold->out16) {
        g.X16 = fold->out16;
        g.ld16 = fold->ld16;
        g.row_part = fold->out_part;
        g.X16_lo = fold->out16_lo;
        g.R16_hi = fold->res16_hi;
        g.R16_lo = fold->res16_lo;
        g.range_flag = fold->range_flag;
        g.row_slot_w = fold->slot_w;
    }
    g.A = x.hi;
    g.A_lo = x.lo;
    g.B = lin.w_hi;
    g.B_lo = lin.w_lo;
    g.b_lo_n0 = lin.w_lo ? lin.lo_row0 : 0;
    g.C = y;
    g.C_lo = y_lo;
    g.M = (int)rows;
    g.N = out_f32 ? n_logical : lin.n;
    // fp32 rows with a rounded-up pitch (pitch4: internal buffers): compute whole float4 groups -- the packed image has
    // zero rows / zero bias behind the logical ones -- so that every store and residual load is a 16-byte one
    if (out_f32 && (n_logical & 3) && ldc >= pitch4(n_logical) && lin.n >= pitch4(n_logical) &&
        !(res && res->ptr && res->ld < pitch4(n_logical)))
        g.N = pitch4(n_logical);
    g.K = lin.k;
    g.lda = lin.k;
    g.ldb = lin.k;
    g.ldc = ldc;
    g.bias = lin.bias;
    g.bias_mode = lin.bias ? 1 : 0;
    g.act = act;
    g.out_f32 = out_f32 ? 1 : 0;
    g.n_store = (!out_f32 && n_store16 > g.N) ? n_store16 : g.N;   // (16-bit rows zero-filled up to the channel pitch)
    if (res && res->ptr && !(fold && fold->res16_hi)) {
        g.R = res->ptr;
        g.ldr = res->ld;
        g.r_stride_b = res->stride_b;
        g.r_rows_per_batch = res->rows_per_batch;
    }
    return gemm_nt_launch(g, s);
}

static int cast_pair(const pio_tensor3_t &x, const pio_layernorm_t *ln, Pair y, int c_pad, int dtype, hipStream_t s) {
    return layernorm_cast_launch(x, ln, y.hi, y.lo, c_pad, dtype, s);
}

// ------------------------------------------------------------------------------------------------------
// attention core on already normalised / cast 16-bit inputs
// ------------------------------------------------------------------------------------------------------
// Upper bound of the materialised score matrix held at once (fp32 scores + 16-bit probabilities are carved for this many
// (batch, row) slabs): attention over more than this is run sample by sample and, inside a sample, in chunks of query
// rows.  Only the 3-sweep policies ("x3": split activations) still materialise scores for the wide cross-attends; the
// optical-flow ones (2048 x 182 528 scores: 1.5 GB fp32 + as much again for the probability pair) then run in two
// passes of <= 1 GiB.  (A 256 MiB cap was measured too: 8 passes, each P V product with only 6 output tiles of a
// 182 528-deep K loop -- 96 ms instead of 29 per forward.)
static const int64_t kScoreCapBytes = 1ll << 30;

struct ScoreChunks {
    int b_chunk;  // samples per pass (>= 1)
    int q_chunk;  // query rows per pass (== Tq unless b_chunk == 1 and one sample exceeds the cap)
};
static ScoreChunks score_chunks(int B, int H, int Tq, int Tk) {
    const int64_t per_sample = (int64_t)H * Tq * (int64_t)Tk * 4;
    ScoreChunks c{B, Tq};
    if ((int64_t)B * per_sample <= kScoreCapBytes) return c;
    int64_t bc = kScoreCapBytes / (per_sample > 0 ? per_sample : 1);
    if (bc >= 1) {
        c.b_chunk = (int)bc;
        return c;
    }
    c.b_chunk = 1;
    int64_t qc = kScoreCapBytes / ((int64_t)H * Tk * 4);
    qc = qc / 128 * 128;
    c.q_chunk = (int)(qc < 128 ? 128 : qc);
    if (c.q_chunk > Tq) c.q_chunk = Tq;
    return c;
}

// true when attention_core will take a fused (score-free) kernel for every call that passes no full mask / bias /
// probability output: the plans of the encoder / decoder (which never pass those) then carve no score buffers at all
static bool fused_capable(const pio_attention_t &a, int Tk) {
    const bool cross = xattn_supported(a.dkp, a.dvp) || xtall_supported(a.dkp, a.dvp, Tk);
    if (a.act_split == 2) return cross;  // split projections, single-sweep fused core
    return !a.act_split && (flash_supported(a.dkp, a.dvp) || cross);
}

struct AttnScratch {
    Pair q16, k16, vt16, p16, o16;
    float *scores;
    void *xpart;  // fp32 partials of the fused cross-attention's key splits
    // need_scores = false: the caller guarantees a fused kernel (fused_capable and no full mask / bias / probabilities)
    void carve(Carver &c, const pio_attention_t &a, int Bq, int B, int Tq, int Tk, bool need_scores = true) {
        const int64_t ldq = (int64_t)a.heads * a.dkp, ldo = (int64_t)a.heads * a.dvp, tkp = pad8(Tk);
        const int64_t tkv = round_up(Tk, 32);  // V^T row pitch: whole 32-key tiles, zero padded (pio_xattn.hip)
        const bool sp = a.act_split != 0;
        q16 = take_pair(c, (size_t)Bq * Tq * ldq, sp);
        // (the fused cross-attention kernel reads whole 32-key tiles: up to 31 rows behind key Tk - 1 of the last
        //  sample.  They only have to be readable -- their scores are masked by assignment -- and they are: vt16 and
        //  o16 follow in the same workspace.  No padding here: the fused q|k|v form needs q16 | k16 | vt16 adjacent.)
        k16 = take_pair(c, (size_t)B * Tk * ldq, sp);
        vt16 = take_pair(c, (size_t)B * ldo * tkv, sp);
        scores = nullptr;
        p16 = Pair();
        if (need_scores) {
            const ScoreChunks ch = score_chunks(B, a.heads, Tq, Tk);
            scores = (float *)c.take((size_t)ch.b_chunk * a.heads * ch.q_chunk * (int64_t)Tk * 4);
            p16 = take_pair(c, (size_t)ch.b_chunk * a.heads * ch.q_chunk * tkp, sp);
        }
        o16 = take_pair(c, (size_t)B * Tq * ldo, sp);
        size_t xb = 0;
        if (a.act_split != 1) {
            if (xattn_supported(a.dkp, a.dvp)) xb = xattn_partial_bytes(a.dkp, a.dvp, B, a.heads, Tq, Tk);
            else if (xtall_supported(a.dkp, a.dvp, Tk)) xb = xtall_scratch_bytes(B);
        }
        xpart = xb ? c.take(xb) : nullptr;
    }
};

static int check_attention(const pio_attention_t &a) {
    if (a.heads <= 0 || a.dk <= 0 || a.dv <= 0) return PIO_E_SHAPE;
    if (a.dkp != pad8(a.dk) || a.dvp != pad8(a.dv)) return PIO_E_SHAPE;
    if (a.q.n != a.heads * a.dkp || a.k.n != a.heads * a.dkp || a.v.n != a.heads * a.dvp) return PIO_E_SHAPE;
    if (a.o.k != a.heads * a.dvp || a.q.k != padc(a.q_in) || a.k.k != padc(a.k_in) || a.v.k != padc(a.v_in))
        return PIO_E_SHAPE;
    if (!a.q.w_hi || !a.k.w_hi || !a.v.w_hi || !a.o.w_hi) return PIO_E_ARG;
    // (split ACTIVATIONS do not require split weights: the "x2a" policies run A_hi B^T + A_lo B^T against single weights)
    return PIO_OK;
}

// K / V projection fold of the single-head cross-attends (attention_core step 0b); PIO_KV_FOLD=0 turns it off (read at
// every call: an A/B switch for tools/, not an API).
static bool kv_fold_enabled() {
    const char *e = getenv("PIO_KV_FOLD");
    return !e || atoi(e) != 0;
}

// Projected queries kept by the CALLER across calls (pio_decoder_fwd_qcache): a decoder whose query array depends on
// parameters and constants only (the multimodal model's chunk queries) normalises and projects it once, not per call.
struct QCache {
    Pair pair;    // [Bq * Tq][H * dkp] 16-bit (+ lo half when the attention carries split activations)
    bool valid;   // false: compute into `pair`; true: `pair` holds the projection -- skip LayerNorm_q and proj_q
};

static int attention_core(const pio_attention_t &a, Pair xq, bool q_bcast, Pair xk, Pair xv, int B, int Tq, int Tk,
                          const uint8_t *kv_mask, const uint8_t *q_mask, const uint8_t *full_mask,
                          const float *attention_bias, const Residual *res, float *out, float *probs_out,
                          AttnScratch &w, hipStream_t s, const LnFold *fold_in = nullptr,
                          const LnFold *fold_out = nullptr, int64_t out_ld = 0, const QCache *qc = nullptr) {
    if (!out_ld) out_ld = a.out;  // row pitch of `out` (>= a.out; an internal buffer may round it up: pitch4)
    PIO_TRY(check_attention(a));
    if (qc) {
        if (!qc->pair.hi || (a.act_split && !qc->pair.lo)) return PIO_E_ARG;
        w.q16 = qc->pair;  // (the plan's own q16 carve stays unused)
    }
    const int H = a.heads;
    const int64_t hdk = (int64_t)H * a.dkp, ldo = (int64_t)H * a.dvp, tkp = pad8(Tk), tkv = round_up(Tk, 32);
    const int Bq = q_bcast ? 1 : B;

    // 0: the fully fused self-attention form: ONE GEMM over the stacked [q | k | v] weight image, then the fused
    //    attention kernel reading V row-major (transposed LDS reads).  Needs: same input for q, k and v, single-
    //    sweep operands everywhere, head widths the fused kernel covers, nothing that wants the score matrix, and the
    //    q16 | k16 | vt16 scratch regions holding one [rows, H*(2 dk + dv)] matrix.
    {
        const int64_t ld3 = 2 * hdk + ldo;  // one row of [q | k | v]
        // (a q|k|v image whose V rows alone carry a lo half -- policies "x2s" / "x2w" -- is taken only inside the
        //  LayerNorm fold, where the wide GEMM kernel, which honours pio_linear_t.lo_row0, is guaranteed)
        const pio_linear_t &qkv_used = (fold_in && fold_in->in_part) ? *fold_in->w : a.qkv;
        const bool qkv_lo_ok = !qkv_used.w_lo || (fold_in && fold_in->in_part && qkv_used.lo_row0 == 2 * hdk);
        const bool fuse_qkv = !qc && a.qkv.w_hi && qkv_lo_ok && !a.act_split && !q_bcast && xq.hi == xk.hi &&
                              xk.hi == xv.hi && Tq == Tk && flash_supported(a.dkp, a.dvp) && a.qkv.n == ld3 &&
                              !kv_mask && !q_mask && !full_mask && !attention_bias && !probs_out &&
                              // (q16, k16, vt16 are consecutive carves: together they hold the [rows, ld3] matrix)
                              (char *)w.q16.hi < (char *)w.k16.hi && (char *)w.k16.hi < (char *)w.vt16.hi &&
                              (size_t)((char *)w.vt16.hi - (char *)w.q16.hi) + (size_t)B * ldo * tkv * 2 >=
                                  (size_t)B * Tq * ld3 * 2;
        if (fuse_qkv) {
            PIO_TRY(linear_fwd(a.qkv, a.dtype, xq, (int64_t)B * Tq, w.q16.hi, nullptr, false, 0, ld3, 0, nullptr, s,
                               fold_in));
            const char *base = (const char *)w.q16.hi;
            PIO_TRY(flash_attention_launch(a.dtype, a.dkp, a.dvp, a.dk, base, base + hdk * 2, base + 2 * hdk * 2,
                                           w.o16.hi, B, H, Tq, Tk, ld3, ld3, ld3, ldo, (int64_t)Tq * ld3,
                                           (int64_t)Tk * ld3, (int64_t)Tk * ld3, (int64_t)Tq * ldo, true, s));
            return linear_fwd(a.o, a.dtype, w.o16, (int64_t)B * Tq, out, nullptr, true, a.out, out_ld, 0, res, s,
                              fold_out);
        }
        if (fold_in || fold_out) return PIO_E_SHAPE;  // the fold is wired into the fused q|k|v form only
    }

    // 0b: K / V projection fold of a single-head cross-attend over many keys (pio_attention_t.kq / vo):
    //     scores = (Q Wk) x^T (+ a per-query constant the softmax cancels), output = (P x)(Wo Wv)^T + (Wo bv + bo).  The fused
    //     kernel reads the LayerNorm'd input array itself as K and its transpose as V^T: the two [keys, C] x [C, C]
    //     projection GEMMs and their 16-bit round trip are gone; what is left per call is Q' = Q Wk (query rows only),
    //     one 16-bit transpose of the inputs and the out projection over K = C.
    {
        const int kvp = pad8(a.k_in);
        const bool kv_fold = !qc && kv_fold_enabled() && a.kq.w_hi && a.vo.w_hi && H == 1 && a.k_in == a.v_in && xk.hi == xv.hi &&
                             a.dk == a.k_in && a.dv == a.v_in && a.dkp == kvp && a.dvp == kvp && a.act_split != 1 &&
                             // (no mask vectors: a row without an attendable key must come out as `final.bias` alone --
                             //  transformer_primitives.py:168-175 -- but the folded bias Wo bv + bo assumes sum(P) = 1)
                             !kv_mask && !q_mask && !full_mask && !attention_bias && !probs_out &&
                             (xattn_supported(kvp, kvp) || xtall_supported(kvp, kvp, Tk)) &&
                             a.kq.k == hdk && a.kq.n == kvp && a.vo.k == kvp && (int64_t)B * Tk >= 4 * (int64_t)Bq * Tq &&
                             !fold_in && !fold_out;
        if (kv_fold) {
            const bool single_core = a.act_split == 2;
            // Q = LN_q(xq) Wq^T + bq (as always), then Q' = Q Wk into the (otherwise unused) k16 scratch
            PIO_TRY(linear_fwd(a.q, a.dtype, xq, (int64_t)Bq * Tq, w.q16.hi, w.q16.lo, false, 0, hdk, 0, nullptr, s));
            PIO_TRY(linear_fwd(a.kq, a.dtype, w.q16, (int64_t)Bq * Tq, w.k16.hi, w.k16.lo, false, 0, kvp, 0, nullptr, s));
            const int64_t ldx = padc(a.k_in);      // row pitch of the LayerNorm'd inputs
            PIO_TRY(transpose16_launch(xk.hi, ldx, B, Tk, kvp, w.vt16.hi, tkv, s));
            if (xattn_supported(kvp, kvp))
                PIO_TRY(xattn_launch(a.dtype, kvp, kvp, a.dk, w.k16.hi, xk.hi, w.vt16.hi, w.o16.hi,
                                     single_core ? w.o16.lo : nullptr, B, H, Tq, Tk, kvp, ldx, tkv, kvp,
                                     q_bcast ? 0 : (int64_t)Tq * kvp, (int64_t)Tk * ldx, (int64_t)kvp * tkv,
                                     (int64_t)Tq * kvp, kv_mask, q_mask, w.xpart, s));
            else  // (a head wider than the tiled kernel covers over <= 512 keys: the ImageNet decoder)
                PIO_TRY(xtall_launch(a.dtype, kvp, kvp, a.dk, w.k16.hi, xk.hi, w.vt16.hi, w.o16.hi,
                                     single_core ? w.o16.lo : nullptr, B, H, Tq, Tk, kvp, ldx, tkv, kvp,
                                     q_bcast ? 0 : (int64_t)Tq * kvp, (int64_t)Tk * ldx, (int64_t)kvp * tkv,
                                     (int64_t)Tq * kvp, kv_mask, q_mask, w.xpart, s));
            return linear_fwd(a.vo, a.dtype, w.o16, (int64_t)B * Tq, out, nullptr, true, a.out, out_ld, 0, res, s);
        }
    }

    // 1/2: Q and K projections (transformer_primitives.py:93-94), head-padded columns.  When both read the same
    //      16-bit input (self-attention) they are ONE GEMM over the stacked [q rows | k rows] weight image: the
    //      q16 / k16 scratch regions are adjacent, the fused output uses them as one [rows, 2*H*dkp] matrix.
    const bool fuse_qk = !qc && a.qk.w_hi && !a.act_split && !q_bcast && xq.hi == xk.hi && Tq == Tk &&
                         (char *)w.q16.hi + (size_t)B * Tq * hdk * 2 <= (char *)w.k16.hi && a.qk.n == 2 * hdk;
    const int64_t ldq = fuse_qk ? 2 * hdk : hdk;
    const void *k_hi = fuse_qk ? (const void *)((const char *)w.q16.hi + hdk * 2) : w.k16.hi;
    const void *k_lo = fuse_qk ? nullptr : w.k16.lo;
    if (fuse_qk) {
        PIO_TRY(linear_fwd(a.qk, a.dtype, xq, (int64_t)B * Tq, w.q16.hi, nullptr, false, 0, ldq, 0, nullptr, s));
    } else {
        if (!(qc && qc->valid))
            PIO_TRY(linear_fwd(a.q, a.dtype, xq, (int64_t)Bq * Tq, w.q16.hi, w.q16.lo, false, 0, ldq, 0, nullptr, s));
        PIO_TRY(linear_fwd(a.k, a.dtype, xk, (int64_t)B * Tk, w.k16.hi, w.k16.lo, false, 0, ldq, 0, nullptr, s));
    }

    // 3: V^T[b] = Wv * X_v[b]^T + bv (transformer_primitives.py:95) produced directly in the K-contiguous layout
    //    the P*V product wants; the weight is the A operand (batch stride 0), bias is per output ROW.
    {
        pio_gemm_t g = gemm_defaults(a.dtype);
        g.A = a.v.w_hi;
        g.A_lo = a.v.w_lo;
        g.B = xv.hi;
        g.B_lo = xv.lo;
        g.C = w.vt16.hi;
        g.C_lo = w.vt16.lo;
        g.M = a.v.n;
        g.N = Tk;
        g.K = a.v.k;
        g.lda = a.v.k;
        g.ldb = a.v.k;
        g.ldc = tkv;
        g.batch = B;
        g.sBb = (int64_t)Tk * a.v.k;
        g.sCb = ldo * tkv;
        g.bias = a.v.bias;
        g.bias_mode = a.v.bias ? 2 : 0;
        g.n_store = (int)tkv;  // columns [Tk, tkv) are written as zeros
        PIO_TRY(gemm_nt_launch(g, s));
    }
    // 4-6 fused when nothing needs the score matrix (no full mask / bias / return_matrix) and the operands are
    //      single-sweep: the self-attention kernel (pio_flash.hip) for un-masked attention with its head widths, the
    //      cross-attention kernel (pio_xattn.hip) for key / query mask VECTORS, wide single heads, dv != dk and few
    //      query tiles (key splits).  Otherwise the materialised path below.
    // act_split == 2 ("x3f"): the projections around the core run with split operands, the core itself single-sweep
    // on the hi halves of q / k / v^T through the fused cross-attention kernel, which returns its output as a pair.
    const bool single_core = a.act_split == 2;
    const bool score_free = (!a.act_split || single_core) && !full_mask && !attention_bias && !probs_out;
    if (score_free && !single_core && !kv_mask && !q_mask && flash_supported(a.dkp, a.dvp)) {
        PIO_TRY(flash_attention_launch(a.dtype, a.dkp, a.dvp, a.dk, w.q16.hi, k_hi, w.vt16.hi, w.o16.hi, B, H, Tq,
                                       Tk, ldq, ldq, tkv, ldo, q_bcast ? 0 : (int64_t)Tq * ldq, (int64_t)Tk * ldq,
                                       ldo * tkv, (int64_t)Tq * ldo, false, s));
        return linear_fwd(a.o, a.dtype, w.o16, (int64_t)B * Tq, out, nullptr, true, a.out, out_ld, 0, res, s);
    }
    if (score_free && xattn_supported(a.dkp, a.dvp)) {
        PIO_TRY(xattn_launch(a.dtype, a.dkp, a.dvp, a.dk, w.q16.hi, k_hi, w.vt16.hi, w.o16.hi,
                             single_core ? w.o16.lo : nullptr, B, H, Tq, Tk, ldq, ldq,
                             tkv, ldo, q_bcast ? 0 : (int64_t)Tq * ldq, (int64_t)Tk * ldq, ldo * tkv, (int64_t)Tq * ldo,
                             kv_mask, q_mask, w.xpart, s));
        return linear_fwd(a.o, a.dtype, w.o16, (int64_t)B * Tq, out, nullptr, true, a.out, out_ld, 0, res, s);
    }
    if (score_free && xtall_supported(a.dkp, a.dvp, Tk)) {
        // a head wider than the tiled kernel covers, over at most 512 keys (the ImageNet decoder: 1024 channels x 512
        // latents): the score row of a query stays in registers, S is computed once
        PIO_TRY(xtall_launch(a.dtype, a.dkp, a.dvp, a.dk, w.q16.hi, k_hi, w.vt16.hi, w.o16.hi,
                             single_core ? w.o16.lo : nullptr, B, H, Tq, Tk, ldq, ldq, tkv, ldo,
                             q_bcast ? 0 : (int64_t)Tq * ldq, (int64_t)Tk * ldq, ldo * tkv, (int64_t)Tq * ldo, kv_mask,
                             q_mask, w.xpart, s));
        return linear_fwd(a.o, a.dtype, w.o16, (int64_t)B * Tq, out, nullptr, true, a.out, out_ld, 0, res, s);
    }
    if (!w.scores) return PIO_E_WORKSPACE;  // (the plan promised a fused kernel)
    // Materialised path, in passes of (b_chunk samples) x (q_chunk query rows) so that the score matrix held at once
    // stays below kScoreCapBytes.  Row chunks inside a sample only happen with b_chunk == 1 (mask pointers then move
    // with the sample and the row).
    const ScoreChunks ch = score_chunks(B, H, Tq, Tk);
    if ((full_mask || attention_bias || probs_out) && (ch.b_chunk != B || ch.q_chunk != Tq)) {
        // full masks / biases / probability outputs are [B,H,Tq,Tk]-sized themselves: callers that pass them have
        // the memory; run them in one pass per sample
        if (ch.q_chunk != Tq) return PIO_E_WORKSPACE;
    }
    for (int b0 = 0; b0 < B; b0 += ch.b_chunk) {
        const int nb = b0 + ch.b_chunk <= B ? ch.b_chunk : B - b0;
        for (int r0 = 0; r0 < Tq; r0 += ch.q_chunk) {
            const int nr = r0 + ch.q_chunk <= Tq ? ch.q_chunk : Tq - r0;
            const int64_t qoff = (q_bcast ? 0 : (int64_t)b0 * Tq * ldq) + (int64_t)r0 * ldq;
            const int64_t koff = (int64_t)b0 * Tk * ldq;
            // 4: S[b,h] = Q[b,h] K[b,h]^T (transformer_primitives.py:138), fp32 scores
            {
                pio_gemm_t g = gemm_defaults(a.dtype);
                g.A = (const char *)w.q16.hi + qoff * 2;
                g.A_lo = w.q16.lo ? (const char *)w.q16.lo + qoff * 2 : nullptr;
                g.B = (const char *)k_hi + koff * 2;
                g.B_lo = k_lo ? (const char *)k_lo + koff * 2 : nullptr;
                g.C = w.scores;
                g.M = nr;
                g.N = Tk;
                g.K = a.dkp;
                g.lda = ldq;
                g.ldb = ldq;
                g.ldc = Tk;
                g.batch = nb * H;
                g.nh = H;
                g.sAb = q_bcast ? 0 : (int64_t)Tq * ldq;
                g.sAh = a.dkp;
                g.sBb = (int64_t)Tk * ldq;
                g.sBh = a.dkp;
                g.sCb = (int64_t)H * nr * Tk;
                g.sCh = (int64_t)nr * Tk;
                g.out_f32 = 1;
                g.n_store = Tk;
                PIO_TRY(gemm_nt_launch(g, s));
            }
            // 5: bias, scale, mask, softmax, wipe (transformer_primitives.py:143-158, 168-175)
            {
                const int64_t moff = (int64_t)b0 * H * Tq * Tk;  // (full-size operands: only with nr == Tq)
                PIO_TRY(softmax_rows_launch(w.scores, Tk, w.p16.hi, w.p16.lo, tkp, nb, H, nr, Tk,
                                            1.0f / sqrtf((float)a.dk), kv_mask ? kv_mask + (int64_t)b0 * Tk : nullptr,
                                            q_mask ? q_mask + (int64_t)b0 * Tq + r0 : nullptr,
                                            full_mask ? full_mask + (int64_t)b0 * Tq * Tk : nullptr,
                                            attention_bias ? attention_bias + moff : nullptr, a.dtype,
                                            probs_out ? probs_out + moff : nullptr, s));
            }
            // 6: O[b,:,h] = P[b,h] V[b,h] (transformer_primitives.py:163-166), heads merged by the store layout
            {
                const int64_t ooff = ((int64_t)b0 * Tq + r0) * ldo, voff = (int64_t)b0 * ldo * tkv;
                pio_gemm_t g = gemm_defaults(a.dtype);
                g.A = w.p16.hi;
                g.A_lo = w.p16.lo;
                g.B = (const char *)w.vt16.hi + voff * 2;
                g.B_lo = w.vt16.lo ? (const char *)w.vt16.lo + voff * 2 : nullptr;
                g.C = (char *)w.o16.hi + ooff * 2;
                g.C_lo = w.o16.lo ? (char *)w.o16.lo + ooff * 2 : nullptr;
                g.M = nr;
                g.N = a.dvp;
                g.K = (int)tkp;
                g.lda = tkp;
                g.ldb = tkv;
                g.ldc = ldo;
                g.batch = nb * H;
                g.nh = H;
                g.sAb = (int64_t)H * nr * tkp;
                g.sAh = (int64_t)nr * tkp;
                g.sBb = ldo * tkv;
                g.sBh = (int64_t)a.dvp * tkv;
                g.sCb = (int64_t)Tq * ldo;
                g.sCh = a.dvp;
                g.n_store = a.dvp;
                PIO_TRY(gemm_nt_launch(g, s));
            }
        }
    }
    // 7: final projection (+ residual) (transformer_primitives.py:110; SelfAttention :290, CrossAttention :396-399)
    return linear_fwd(a.o, a.dtype, w.o16, (int64_t)B * Tq, out, nullptr, true, a.out, out_ld, 0, res, s);
}

// ------------------------------------------------------------------------------------------------------
// MLP core on a 16-bit input
// ------------------------------------------------------------------------------------------------------
// out16 (optional; then `out` is not written): the result leaves as the 16-bit operand (pair under split activations) of
// the GEMM that follows -- rows of padc(m.out) elements, zero-filled behind the logical columns -- instead of as fp32 rows
// that a cast pass would re-read (the decoders' y in front of their final Linear).
static int mlp_core(const pio_mlp_t &m, Pair x, int64_t rows, Pair h, const Residual *res, float *out,
                    hipStream_t s, const LnFold *fold_in = nullptr, const LnFold *fold_out = nullptr,
                    int64_t out_ld = 0, const Pair *out16 = nullptr) {
    if (!out_ld) out_ld = m.out;
    if (m.fc1.k != padc(m.in) || m.fc1.n != padc(m.hidden) || m.fc2.k != m.fc1.n) return PIO_E_SHAPE;
    PIO_TRY(linear_fwd(m.fc1, m.dtype, x, rows, h.hi, h.lo, false, 0, m.fc1.n, 1, nullptr, s, fold_in));
    if (out16)
        return linear_fwd(m.fc2, m.dtype, h, rows, out16->hi, out16->lo, false, 0, padc(m.out), 0, res, s, nullptr,
                          padc(m.out));
    return linear_fwd(m.fc2, m.dtype, h, rows, out, nullptr, true, m.out, out_ld, 0, res, s, fold_out);
}

// ======================================================================================================
// plans (workspace layouts)
// ======================================================================================================
struct AttentionPlan {
    Pair xq16, xk16, xv16;
    AttnScratch core;
    size_t carve(void *base, const pio_attention_t &a, int B, int Tq, int Tk, bool qb, bool same) {
        Carver c(base);
        const bool sp = a.act_split != 0;
        const int Bq = qb ? 1 : B;
        xq16 = take_pair(c, (size_t)Bq * Tq * padc(a.q_in), sp);
        xk16 = take_pair(c, (size_t)B * Tk * padc(a.k_in), sp);
        xv16 = same ? xk16 : take_pair(c, (size_t)B * Tk * padc(a.v_in), sp);
        core.carve(c, a, Bq, B, Tq, Tk);
        return c.off;
    }
};

static bool same_tensor(const pio_tensor3_t &a, const pio_tensor3_t &b) {
    return a.data == b.data && a.stride_b == b.stride_b && a.stride_t == b.stride_t && a.B == b.B && a.T == b.T &&
           a.C == b.C;
}

static pio_tensor3_t first_batch(const pio_tensor3_t &t) {
    pio_tensor3_t r = t;
    r.B = 1;
    return r;
}

static Pair pair_if(Pair p, bool split) {
    if (!split) p.lo = nullptr;
    return p;
}

// In-place residual stream of a folded self-attend stack (see SelfPlan::carve); PIO_FOLD_INPLACE=0 restores the two
// ping-pong pairs (read at every carve: an A/B switch for tools/, not an API).
static bool fold_inplace() {
    const char *e = getenv("PIO_FOLD_INPLACE");
    return !e || atoi(e) != 0;
}

struct SelfPlan {
    Pair x16, h16;
    float *x1;
    AttnScratch core;
    // LayerNorm fold (pio_ln_fold_t): 16-bit copy of x1 and the per-row partial sums of x (A) and x1 (B); the 16-bit
    // copy of x lives in x16
    void *x16b = nullptr, *lo_a = nullptr, *lo_b = nullptr;  // lo_*: x - x16 / x1 - x16b (the stream as a 16-bit pair)
    float *part_a = nullptr, *part_b = nullptr;
    // lean: the caller never passes a full mask / bias / probability output (the encoder stack): no score buffers
    // when a fused kernel covers the block
    size_t carve(void *base, const pio_self_attention_t &sa, int B, int N, bool lean = false) {
        Carver c(base);
        const int64_t rows = (int64_t)B * N;
        const int cmax = padc(sa.attn.q_in) > padc(sa.mlp.in) ? padc(sa.attn.q_in) : padc(sa.mlp.in);
        x16 = take_pair(c, (size_t)rows * cmax, sa.attn.act_split || sa.mlp.act_split);
        h16 = take_pair(c, (size_t)rows * padc(sa.mlp.hidden), sa.mlp.act_split != 0);
        x1 = (float *)c.take((size_t)rows * sa.attn.out * 4);
        core.carve(c, sa.attn, B, B, N, N, !(lean && fused_capable(sa.attn, N)));
        if (sa.fold.qkv.w_hi && sa.fold.fc1.w_hi) {
            // The residual GEMMs (out, fc2) read the residual pair and write the result pair element for element from
            // the same lane (load, add, store), and their A operand is another array (attention output / hidden
            // activations): the stream is updated IN PLACE -- one 16-bit pair instead of two ping-pong pairs, and the
            // hidden activations take the attention output's buffer (dead once the out projection has run).  Per layer
            // at B = 32 the arrays in flight shrink from 288 MB (beyond the 256 MB Infinity Cache) to 192 MB.
            const bool inplace = fold_inplace();
            x16b = inplace ? x16.hi : c.take((size_t)rows * cmax * 2);
            lo_a = c.take((size_t)rows * cmax * 2);
            lo_b = inplace ? lo_a : c.take((size_t)rows * cmax * 2);
            part_a = (float *)c.take((size_t)rows * (cmax / 64 + 1) * 2 * 4);  // (up to one slot per 64 columns)
            part_b = (float *)c.take((size_t)rows * (cmax / 64 + 1) * 2 * 4);
            if (inplace && !sa.mlp.act_split && !sa.attn.act_split &&
                (size_t)rows * padc(sa.mlp.hidden) <= (size_t)rows * sa.attn.heads * sa.attn.dvp)
                h16.hi = core.o16.hi;
        }
        return c.off;
    }
};

// LayerNorm fold switch (pio_ln_fold_enable; initial value from env PIO_LN_FOLD, default 1).
// 0: never, 1: where it pays, 2: wherever a block offers it.  "Where it pays": the fold's GEMMs are the 256 x 256-tile
// kernel, which needs >= 96 tiles of a 1024-wide output to beat the 128 / 64-tile kernels of the un-folded block --
// measured on the ImageNet classifier (tools/latency_probe.py, ms per forward, fold on / off): B=4 7.75 / 5.22,
// B=8 8.12 / 7.64, B=12 10.14 / 10.09, B=16 10.69 / 11.57, B=32 16.62 / 18.24 -- hence 6144 rows (env
// PIO_LN_FOLD_MIN_ROWS).
static int &ln_fold_global() {
    static int choice = [] {
        const char *e = getenv("PIO_LN_FOLD");
        const int v = e ? atoi(e) : 1;
        return v < 0 ? 0 : v > 2 ? 2 : v;
    }();
    return choice;
}
// Per-call override (pio_call_opts_t.ln_fold of the *_opts entry points: 1 = off, 2 = where it pays, 3 = wherever
// offered; 0 = the process-wide setting above): thread-local for the duration of the call, so that two threads driving
// two encoders never see each other's choice.
static thread_local int tl_ln_fold = 0;
static int ln_fold_choice() { return tl_ln_fold > 0 ? tl_ln_fold - 1 : ln_fold_global(); }
struct CallOpts {  // RAII around one C-ABI call
    int prev_fold, prev_cu;
    explicit CallOpts(const pio_call_opts_t *o) : prev_fold(tl_ln_fold), prev_cu(cu_budget_call(-1)) {
        if (o) {
            tl_ln_fold = (o->ln_fold >= 0 && o->ln_fold <= 3) ? o->ln_fold : 0;
            cu_budget_call(o->cu_budget > 0 ? o->cu_budget : 0);
        }
    }
    ~CallOpts() {
        tl_ln_fold = prev_fold;
        cu_budget_call(prev_cu);
    }
};
int ln_fold_enable(int on) {
    int &c = ln_fold_global();
    const int prev = c;
    c = on < 0 ? 0 : on > 2 ? 2 : on;
    return prev;
}
bool ln_fold_enabled() { return ln_fold_choice() != 0; }
static int64_t ln_fold_min_rows() {
    static const int64_t auto_rows = [] {
        const char *e = getenv("PIO_LN_FOLD_MIN_ROWS");
        const long long v = e ? atoll(e) : 6144;
        return (int64_t)(v < 2048 ? 2048 : v);
    }();
    return ln_fold_choice() == 2 ? 2048 : auto_rows;
}
// Below that row count the fold runs on the tile kernels (64-column statistics slots) -- from this many rows on
// (env PIO_LN_FOLD_SMALL_MIN_ROWS; mode 2: from 128 rows).
static int64_t ln_fold_small_min_rows() {
    static const int64_t auto_rows = [] {
        const char *e = getenv("PIO_LN_FOLD_SMALL_MIN_ROWS");
        const long long v = e ? atoll(e) : 512;
        return (int64_t)(v < 128 ? 128 : v);
    }();
    return ln_fold_choice() == 2 ? 128 : auto_rows;
}
static int64_t ln_fold_small_max_rows() {
    static const int64_t max_rows = [] {
        const char *e = getenv("PIO_LN_FOLD_SMALL_MAX_ROWS");
        return (int64_t)(e ? atoll(e) : 4096);
    }();
    return max_rows;
}

// Carried from one SelfAttention block to the next inside a stack: the 16-bit copy and the partial sums of the block's
// INPUT, left in the plan's (x16, part_a) buffers by the previous block's fc2 GEMM.
struct FoldCarry {
    const void *x = nullptr;  // the fp32 tensor they describe (its CONTENT is stale unless f32_valid)
    const void *x16 = nullptr;
    const float *part = nullptr;
    bool f32_valid = true;
};

static int self_attention_run(const pio_self_attention_t &sa, const pio_tensor3_t &x, const uint8_t *kv_mask,
                              const uint8_t *q_mask, const uint8_t *full_mask, const float *attention_bias,
                              float *out, float *probs_out, SelfPlan &p, hipStream_t s, FoldCarry *carry = nullptr,
                              bool need_f32_out = true) {
    const int B = x.B, N = x.T;
    const int64_t rows = (int64_t)B * N;
    if (x.C != sa.attn.q_in || sa.attn.k_in != x.C || sa.attn.v_in != x.C || sa.attn.out != x.C ||
        sa.mlp.in != x.C || sa.mlp.out != x.C)
        return PIO_E_SHAPE;  // residual adds need matching widths (the reference raises a RuntimeError)
    // LayerNorm fold: contiguous rows of 512 / 768 / 1024 / 1280 / 1536 channels, single-sweep activations, the fused
    // q|k|v form (head widths the fused attention kernel covers), nothing that needs the score matrix.  Two kernel
    // families: the 256 x 256-tile kernel with 128-column statistics slots for stacks with enough rows to fill the chip
    // (weights may then be (hi, lo) pairs -- policies "x2s" / "x2w": second K sweep against the lo image; of the stacked
    // q|k|v image only the V rows may have one), the tile kernels with 64-column slots below that (single weights).
    const bool fold_ok = ln_fold_choice() != 0 && p.x16b && x.C >= 512 && x.C <= 1536 && (x.C % 256) == 0 &&
                         x.stride_t == x.C && (B == 1 || x.stride_b == (int64_t)N * x.C) && !sa.attn.act_split &&
                         !sa.mlp.act_split && sa.attn.qkv.w_hi &&
                         (!sa.fold.qkv.w_lo || sa.fold.qkv.lo_row0 == 2 * sa.attn.heads * sa.attn.dkp) &&
                         (!sa.fold.fc1.w_lo || sa.fold.fc1.lo_row0 == 0) && flash_supported(sa.attn.dkp, sa.attn.dvp) &&
                         sa.fold.qkv.n == sa.attn.qkv.n && sa.fold.qkv.k == x.C && sa.fold.fc1.k == x.C &&
                         sa.fold.fc1.n == sa.mlp.fc1.n && sa.mlp.hidden == x.C && sa.mlp.dtype == sa.attn.dtype &&
                         !kv_mask && !q_mask && !full_mask && !attention_bias && !probs_out &&
                         (((uintptr_t)x.data) & 15) == 0;
    const bool fold_wide = fold_ok && rows >= ln_fold_min_rows();
    // (tile-kernel family: up to 4095 rows in the automatic mode -- ImageNet B = 8, 4096 rows, measures 8.33 ms folded on
    //  the tile kernels against 7.86 un-folded, whose q|k|v GEMM runs on the 256 x 256-tile kernel; B = 1 / 2 / 4:
    //  3.78 / 4.28 / 5.05 against 3.97 / 4.39 / 5.27 ms -- tools/r4_probe4.sh)
    const bool fold_small = fold_ok && !fold_wide && rows >= ln_fold_small_min_rows() &&
                            (ln_fold_choice() == 2 || rows < ln_fold_small_max_rows()) && !sa.fold.qkv.w_lo &&
                            !sa.fold.fc1.w_lo && !sa.attn.o.w_lo && !sa.mlp.fc2.w_lo;
    const bool fold = fold_wide || fold_small;
    if (fold) {
        const int slot_w = fold_wide ? 128 : 64, nslots = x.C / slot_w;
        // x16 / part_a: the block input (from the previous block's fc2, or computed here for the first block)
        // Inside the fold the residual stream is the 16-bit pair (x16, lo): 22 mantissa bits, and 64 MB less traffic
        // per residual GEMM than fp32 + copy.  The fp32 form is read here once (first block) and written when the
        // caller needs it (last block).
        if (!(carry && carry->x == x.data && carry->x16 == p.x16.hi && carry->part == p.part_a)) {
            if (carry && !carry->f32_valid) return PIO_E_ARG;  // (the stack decides the fold for all its blocks)
            PIO_TRY(rowstats_cast_launch(x.data, rows, x.C, slot_w, p.x16.hi, p.lo_a, p.part_a, sa.attn.dtype, s));
        }
        const Pair xa = {p.x16.hi, nullptr};
        LnFold f_qkv, f_out, f_fc1, f_fc2;
        f_qkv.in_part = p.part_a; f_qkv.w = &sa.fold.qkv; f_qkv.c = sa.fold.qkv_c; f_qkv.eps = sa.ln1.eps;
        f_out.out16 = p.x16b; f_out.ld16 = x.C; f_out.out_part = p.part_b;
        f_out.out16_lo = p.lo_b; f_out.res16_hi = p.x16.hi; f_out.res16_lo = p.lo_a;
        f_out.range_flag = f_fc2.range_flag = sa.fold.range_flag;
        f_qkv.in_slots = f_fc1.in_slots = nslots;
        f_out.slot_w = f_fc2.slot_w = slot_w;
        f_fc1.in_part = p.part_b; f_fc1.w = &sa.fold.fc1; f_fc1.c = sa.fold.fc1_c; f_fc1.eps = sa.ln2.eps;
        f_fc2.out16 = p.x16.hi; f_fc2.ld16 = x.C; f_fc2.out_part = p.part_a;
        f_fc2.out16_lo = p.lo_a; f_fc2.res16_hi = p.x16b; f_fc2.res16_lo = p.lo_b;
        const Residual rx = residual_of(x);
        PIO_TRY(attention_core(sa.attn, xa, false, xa, xa, B, N, N, nullptr, nullptr, nullptr, nullptr, &rx, nullptr,
                               nullptr, p.core, s, &f_qkv, &f_out));  // (x1 exists as the pair (x16b, lo_b) only)
        pio_tensor3_t t1 = {p.x1, (int64_t)N * x.C, x.C, B, N, x.C};
        const Pair xm = {p.x16b, nullptr};
        const Residual r1 = residual_of(t1);
        PIO_TRY(mlp_core(sa.mlp, xm, rows, p.h16, &r1, need_f32_out ? out : nullptr, s, &f_fc1, &f_fc2));
        if (carry) {
            carry->x = out;
            carry->x16 = p.x16.hi;
            carry->part = p.part_a;
            carry->f32_valid = need_f32_out;
        }
        return PIO_OK;
    }
    if (carry) {
        if (!carry->f32_valid) return PIO_E_ARG;  // (the stack decides the fold for all its blocks)
        *carry = FoldCarry();
    }
    // LN1 -> attention -> + x     (transformer_primitives.py:281-290)
    const Pair xa = pair_if(p.x16, sa.attn.act_split);
    PIO_TRY(cast_pair(x, &sa.ln1, xa, padc(x.C), sa.attn.dtype, s));
    const Residual rx = residual_of(x);
    PIO_TRY(attention_core(sa.attn, xa, false, xa, xa, B, N, N, kv_mask, q_mask, full_mask, attention_bias, &rx, p.x1,
                           probs_out, p.core, s));
    // LN2 -> MLP -> + x1          (transformer_primitives.py:292)
    pio_tensor3_t t1 = {p.x1, (int64_t)N * x.C, x.C, B, N, x.C};
    const Pair xm = pair_if(p.x16, sa.mlp.act_split);
    PIO_TRY(cast_pair(t1, &sa.ln2, xm, padc(x.C), sa.mlp.dtype, s));
    const Residual r1 = residual_of(t1);
    return mlp_core(sa.mlp, xm, rows, p.h16, &r1, out, s);
}

struct CrossPlan {
    Pair q16, kv16, h16;
    float *x1;
    AttnScratch core;
    bool q_bcast;
    size_t carve(void *base, const pio_cross_attention_t &ca, int B, int Tq, int Tk, bool qb, bool lean = false) {
        Carver c(base);
        q_bcast = qb;
        const int Bq = qb ? 1 : B;
        const int64_t rows = (int64_t)B * Tq;
        const bool sp = ca.attn.act_split || ca.mlp.act_split;
        // q16 is reused for LN2(x1): size it for all B*Tq rows
        q16 = take_pair(c, (size_t)rows * padc(ca.attn.q_in), sp);
        kv16 = take_pair(c, (size_t)B * Tk * padc(ca.attn.k_in), ca.attn.act_split != 0);
        h16 = take_pair(c, (size_t)rows * padc(ca.mlp.hidden), ca.mlp.act_split != 0);
        x1 = (float *)c.take((size_t)rows * pitch4(ca.attn.out) * 4);
        core.carve(c, ca.attn, Bq, B, Tq, Tk, !(lean && fused_capable(ca.attn, Tk)));
        return c.off;
    }
};

// ikv_tail (optional): the key / value input arrives as TWO arrays whose channels are concatenated, [ikv | ikv_tail]
// (ikv_tail->B == 1: one batch-invariant table, e.g. Fourier position features); layer_norm_kv runs over the virtual
// concatenation and nothing is ever concatenated in HBM.  iq_tail (optional, blocks without a query residual): the same
// for the query rows, [iq | iq_tail] under layer_norm_q (the dense decoders whose queries ARE the network's input).
static int cross_attention_run(const pio_cross_attention_t &ca, const pio_tensor3_t &iq, const pio_tensor3_t &ikv,
                               const uint8_t *kv_mask, const uint8_t *q_mask, const uint8_t *full_mask,
                               const float *attention_bias, float *out, float *probs_out, CrossPlan &p,
                               hipStream_t s, const pio_tensor3_t *ikv_tail = nullptr, int64_t out_ld = 0,
                               const QCache *qc = nullptr, const Pair *out16 = nullptr,
                               const pio_tensor3_t *iq_tail = nullptr) {
    const int B = iq.B, Tq = iq.T, Tk = ikv.T;
    const int64_t rows = (int64_t)B * Tq;
    const int kv_c = ikv.C + (ikv_tail ? ikv_tail->C : 0);
    const int q_c = iq.C + (iq_tail ? iq_tail->C : 0);
    if (q_c != ca.attn.q_in || kv_c != ca.attn.k_in || kv_c != ca.attn.v_in || ikv.B != B) return PIO_E_SHAPE;
    if (ca.attn.out != q_c || ca.mlp.in != q_c || ca.mlp.out != q_c) return PIO_E_SHAPE;
    if (iq_tail && (ca.use_query_residual || qc || p.q_bcast)) return PIO_E_ARG;  // (the rows themselves are needed then)
    // layer_norm_kv, layer_norm_q  (transformer_primitives.py:379-380)
    if (ikv_tail)
        PIO_TRY(layernorm_cast_cat_launch(ikv, *ikv_tail, ca.ln_kv, p.kv16.hi, p.kv16.lo, padc(kv_c), ca.attn.dtype, s));
    else
        PIO_TRY(cast_pair(ikv, &ca.ln_kv, p.kv16, padc(ikv.C), ca.attn.dtype, s));
    const pio_tensor3_t q1 = p.q_bcast ? first_batch(iq) : iq;
    const Pair qa = pair_if(p.q16, ca.attn.act_split);
    if (qc && ca.use_query_residual) return PIO_E_ARG;  // (the query rows themselves are needed then)
    if (iq_tail)
        PIO_TRY(layernorm_cast_cat_launch(iq, *iq_tail, ca.ln_q, qa.hi, qa.lo, padc(q_c), ca.attn.dtype, s));
    else if (!(qc && qc->valid)) PIO_TRY(cast_pair(q1, &ca.ln_q, qa, padc(q_c), ca.attn.dtype, s));
    const Residual rq = residual_of(iq);
    PIO_TRY(attention_core(ca.attn, qa, p.q_bcast, p.kv16, p.kv16, B, Tq, Tk, kv_mask, q_mask, full_mask,
                           attention_bias, ca.use_query_residual ? &rq : nullptr, p.x1, probs_out, p.core, s, nullptr,
                           nullptr, pitch4(q_c), qc));
    // x + MLP(LN2(x))  (transformer_primitives.py:401)
    pio_tensor3_t t1 = {p.x1, (int64_t)Tq * pitch4(q_c), pitch4(q_c), B, Tq, q_c};
    const Pair qm = pair_if(p.q16, ca.mlp.act_split);
    PIO_TRY(cast_pair(t1, &ca.ln2, qm, padc(q_c), ca.mlp.dtype, s));
    const Residual r1 = residual_of(t1);
    return mlp_core(ca.mlp, qm, rows, p.h16, &r1, out, s, nullptr, nullptr, out_ld, out16);
}

struct DecoderPlan {
    CrossPlan cp;
    float *y;
    Pair y16;
    size_t carve(void *base, const pio_cross_attention_t &cross, const pio_linear_t *fin, int B, int Q, int N,
                 bool qb) {
        Carver c(base);
        const int64_t rows = (int64_t)B * Q;
        y = nullptr;
        y16 = Pair();
        if (fin) {
            y = (float *)c.take((size_t)rows * pitch4(cross.attn.q_in) * 4);
            y16 = take_pair(c, (size_t)rows * padc(cross.attn.q_in), cross.mlp.act_split != 0);
        }
        const size_t inner = cp.carve(base ? (char *)base + c.off : nullptr, cross, B, Q, N, qb, true);
        return c.off + inner;
    }
};

}  // namespace pio

using namespace pio;

extern "C" {

size_t pio_attention_workspace_bytes(const pio_attention_t *a, int32_t B, int32_t Tq, int32_t Tk) {
    if (!a) return 0;
    AttentionPlan p;
    return p.carve(nullptr, *a, B, Tq, Tk, false, false);
}

int pio_attention_fwd(const pio_attention_t *a, const pio_tensor3_t *iq, const pio_tensor3_t *ik,
                      const pio_tensor3_t *iv, const uint8_t *kv_mask, const uint8_t *q_mask,
                      const uint8_t *full_mask, const float *attention_bias, float *out, float *probs_out,
                      void *workspace, size_t workspace_bytes, void *stream) {
    if (!a || !iq || !ik || !iv || !out || !workspace) return PIO_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (iq->C != a->q_in || ik->C != a->k_in || iv->C != a->v_in) return PIO_E_SHAPE;
    if (ik->B != iq->B || iv->B != iq->B || ik->T != iv->T) return PIO_E_SHAPE;
    const int B = iq->B, Tq = iq->T, Tk = ik->T;
    const bool qb = (iq->stride_b == 0 && B > 1);
    const bool same = same_tensor(*ik, *iv);
    AttentionPlan p;
    if (p.carve(workspace, *a, B, Tq, Tk, qb, same) > workspace_bytes) return PIO_E_WORKSPACE;
    const pio_tensor3_t q1 = qb ? first_batch(*iq) : *iq;
    PIO_TRY(cast_pair(q1, nullptr, p.xq16, padc(a->q_in), a->dtype, s));
    PIO_TRY(cast_pair(*ik, nullptr, p.xk16, padc(a->k_in), a->dtype, s));
    if (!same) PIO_TRY(cast_pair(*iv, nullptr, p.xv16, padc(a->v_in), a->dtype, s));
    return attention_core(*a, p.xq16, qb, p.xk16, p.xv16, B, Tq, Tk, kv_mask, q_mask, full_mask, attention_bias,
                          nullptr, out, probs_out, p.core, s);
}

// ======================================================================================================
// MLP.forward
// ======================================================================================================
size_t pio_mlp_workspace_bytes(const pio_mlp_t *m, int64_t rows) {
    if (!m) return 0;
    Carver c(nullptr);
    take_pair(c, (size_t)rows * padc(m->in), m->act_split != 0);
    take_pair(c, (size_t)rows * padc(m->hidden), m->act_split != 0);
    return c.off;
}

int pio_mlp_fwd(const pio_mlp_t *m, const pio_tensor3_t *x, float *out, void *workspace, size_t workspace_bytes,
                void *stream) {
    if (!m || !x || !out || !workspace) return PIO_E_ARG;
    if (x->C != m->in) return PIO_E_SHAPE;
    const int64_t rows = (int64_t)x->B * x->T;
    if (pio_mlp_workspace_bytes(m, rows) > workspace_bytes) return PIO_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Carver c(workspace);
    const Pair x16 = take_pair(c, (size_t)rows * padc(m->in), m->act_split != 0);
    const Pair h16 = take_pair(c, (size_t)rows * padc(m->hidden), m->act_split != 0);
    PIO_TRY(cast_pair(*x, nullptr, x16, padc(m->in), m->dtype, s));
    return mlp_core(*m, x16, rows, h16, nullptr, out, s);
}

// ======================================================================================================
// SelfAttention.forward / CrossAttention.forward
// ======================================================================================================
int pio_ln_fold_enable(int on) { return ln_fold_enable(on); }

size_t pio_self_attention_workspace_bytes(const pio_self_attention_t *sa, int32_t B, int32_t N) {
    if (!sa) return 0;
    SelfPlan p;
    return p.carve(nullptr, *sa, B, N);
}

int pio_self_attention_fwd(const pio_self_attention_t *sa, const pio_tensor3_t *x, const uint8_t *kv_mask,
                           const uint8_t *q_mask, const uint8_t *full_mask, const float *attention_bias, float *out,
                           float *probs_out, void *workspace, size_t workspace_bytes, void *stream) {
    return pio_self_attention_fwd_opts(sa, x, kv_mask, q_mask, full_mask, attention_bias, out, probs_out, workspace,
                                       workspace_bytes, stream, nullptr);
}

int pio_self_attention_fwd_opts(const pio_self_attention_t *sa, const pio_tensor3_t *x, const uint8_t *kv_mask,
                                const uint8_t *q_mask, const uint8_t *full_mask, const float *attention_bias,
                                float *out, float *probs_out, void *workspace, size_t workspace_bytes, void *stream,
                                const pio_call_opts_t *opts) {
    if (!sa || !x || !out || !workspace) return PIO_E_ARG;
    CallOpts scope(opts);
    SelfPlan p;
    if (p.carve(workspace, *sa, x->B, x->T) > workspace_bytes) return PIO_E_WORKSPACE;
    return self_attention_run(*sa, *x, kv_mask, q_mask, full_mask, attention_bias, out, probs_out, p,
                              (hipStream_t)stream);
}

size_t pio_cross_attention_workspace_bytes(const pio_cross_attention_t *ca, int32_t B, int32_t Tq, int32_t Tk) {
    if (!ca) return 0;
    CrossPlan p;
    return p.carve(nullptr, *ca, B, Tq, Tk, false);
}

int pio_cross_attention_fwd(const pio_cross_attention_t *ca, const pio_tensor3_t *iq, const pio_tensor3_t *ikv,
                            const uint8_t *kv_mask, const uint8_t *q_mask, const uint8_t *full_mask,
                            const float *attention_bias, float *out, float *probs_out, void *workspace,
                            size_t workspace_bytes, void *stream) {
    if (!ca || !iq || !ikv || !out || !workspace) return PIO_E_ARG;
    CrossPlan p;
    const bool qb = (iq->stride_b == 0 && iq->B > 1);
    if (p.carve(workspace, *ca, iq->B, iq->T, ikv->T, qb) > workspace_bytes) return PIO_E_WORKSPACE;
    return cross_attention_run(*ca, *iq, *ikv, kv_mask, q_mask, full_mask, attention_bias, out, probs_out, p,
                               (hipStream_t)stream);
}

// ======================================================================================================
// PerceiverEncoder.forward
// ======================================================================================================
size_t pio_encoder_workspace_bytes(const pio_cross_attention_t *cross, const pio_self_attention_t *layers, int32_t L,
                                   int32_t B, int32_t M, int32_t N) {
    if (!cross) return 0;
    CrossPlan cp;
    size_t need = cp.carve(nullptr, *cross, B, N, M, false, true);
    for (int l = 0; l < L; ++l) {
        SelfPlan sp;
        const size_t n = sp.carve(nullptr, layers[l], B, N, true);
        if (n > need) need = n;
    }
    return need;
}

int pio_encoder_fwd(const pio_cross_attention_t *cross, const pio_self_attention_t *layers, int32_t L,
                    int32_t num_blocks, const pio_tensor3_t *inputs, const pio_tensor3_t *latents,
                    const uint8_t *input_mask, float *out, void *workspace, size_t workspace_bytes, void *stream) {
    return pio_encoder_fwd_split(cross, layers, L, num_blocks, inputs, nullptr, latents, input_mask, out, workspace,
                                 workspace_bytes, stream);
}

int pio_encoder_fwd_split(const pio_cross_attention_t *cross, const pio_self_attention_t *layers, int32_t L,
                          int32_t num_blocks, const pio_tensor3_t *inputs, const pio_tensor3_t *inputs_tail,
                          const pio_tensor3_t *latents, const uint8_t *input_mask, float *out, void *workspace,
                          size_t workspace_bytes, void *stream) {
    return pio_encoder_fwd_blocks(cross, layers, L, num_blocks, 0, inputs, inputs_tail, latents, input_mask, out,
                                  workspace, workspace_bytes, stream);
}

int pio_encoder_fwd_blocks(const pio_cross_attention_t *cross, const pio_self_attention_t *layers, int32_t L,
                           int32_t num_blocks, int32_t per_block, const pio_tensor3_t *inputs,
                           const pio_tensor3_t *inputs_tail, const pio_tensor3_t *latents, const uint8_t *input_mask,
                           float *out, void *workspace, size_t workspace_bytes, void *stream) {
    return pio_encoder_fwd_opts(cross, layers, L, num_blocks, per_block, inputs, inputs_tail, latents, input_mask, out,
                                workspace, workspace_bytes, stream, nullptr);
}

int pio_encoder_fwd_opts(const pio_cross_attention_t *cross, const pio_self_attention_t *layers, int32_t L,
                         int32_t num_blocks, int32_t per_block, const pio_tensor3_t *inputs,
                         const pio_tensor3_t *inputs_tail, const pio_tensor3_t *latents, const uint8_t *input_mask,
                         float *out, void *workspace, size_t workspace_bytes, void *stream,
                         const pio_call_opts_t *opts) {
    if (!cross || !inputs || !latents || !out || !workspace || (L > 0 && !layers)) return PIO_E_ARG;
    CallOpts scope(opts);
    if (L < 0 || num_blocks < 0) return PIO_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int B = inputs->B, M = inputs->T, N = latents->T, D = latents->C;
    if (latents->B != B) return PIO_E_SHAPE;
    if (pio_encoder_workspace_bytes(cross, layers, L, B, M, N) > workspace_bytes) return PIO_E_WORKSPACE;
    {
        CrossPlan cp;
        const bool qb = (latents->stride_b == 0 && B > 1);
        cp.carve(workspace, *cross, B, N, M, qb, true);
        // perceiver.py:99-103: only the cross-attend is masked, with mask[b,i,j] = input_mask[b,j]
        PIO_TRY(cross_attention_run(*cross, *latents, *inputs, input_mask, nullptr, nullptr, nullptr, out, nullptr,
                                    cp, s, inputs_tail));
    }
    const pio_tensor3_t z = {out, (int64_t)N * D, D, B, N, D};
    FoldCarry carry;  // LayerNorm fold: the row statistics of z travel from one block's fc2 to the next block's q|k|v
    for (int blk = 0; blk < num_blocks; ++blk) {  // perceiver.py:104-106: weights shared across blocks
        for (int l = 0; l < L; ++l) {
            // (per_block: block blk has its own IMAGES of the shared parameters -- same shapes -- at layers[blk * L + l])
            const pio_self_attention_t &lay = layers[(per_block ? (size_t)blk * L : 0) + l];
            SelfPlan sp;
            sp.carve(workspace, lay, B, N, true);
            const bool last = blk == num_blocks - 1 && l == L - 1;
            PIO_TRY(self_attention_run(lay, z, nullptr, nullptr, nullptr, nullptr, out, nullptr, sp, s, &carry, last));
        }
    }
    return PIO_OK;
}

// ======================================================================================================
// PerceiverDecoder.forward
// ======================================================================================================
size_t pio_decoder_workspace_bytes(const pio_cross_attention_t *cross, const pio_linear_t *final_layer, int32_t B,
                                   int32_t Q, int32_t N) {
    if (!cross) return 0;
    DecoderPlan p;
    return p.carve(nullptr, *cross, final_layer, B, Q, N, false);
}

int pio_decoder_fwd(const pio_cross_attention_t *cross, const pio_linear_t *final_layer, int32_t final_out,
                    const pio_tensor3_t *query, const pio_tensor3_t *latents, const uint8_t *query_mask, float *out,
                    void *workspace, size_t workspace_bytes, void *stream) {
    return pio_decoder_fwd_qcache(cross, final_layer, final_out, query, latents, query_mask, out, workspace,
                                  workspace_bytes, stream, nullptr, nullptr, 0);
}

size_t pio_decoder_qcache_bytes(const pio_cross_attention_t *cross, int32_t Bq, int32_t Q) {
    if (!cross || Bq <= 0 || Q <= 0) return 0;
    return (size_t)round_up((int64_t)Bq * Q * cross->attn.heads * cross->attn.dkp * 2, 256);
}

static int decoder_run(const pio_cross_attention_t *cross, const pio_linear_t *final_layer, int32_t final_out,
                       const pio_tensor3_t *query, const pio_tensor3_t *query_tail, const pio_tensor3_t *latents,
                       const uint8_t *query_mask, float *out, void *workspace, size_t workspace_bytes, void *stream,
                       void *q16_hi, void *q16_lo, int32_t q16_valid) {
    if (!cross || !query || !latents || !out || !workspace) return PIO_E_ARG;
    const int q_c = query->C + (query_tail ? query_tail->C : 0);
    QCache qcache{{q16_hi, q16_lo}, q16_valid != 0};
    const QCache *qc = q16_hi ? &qcache : nullptr;
    if (qc && (((uintptr_t)q16_hi & 15) || ((uintptr_t)q16_lo & 15))) return PIO_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int B = query->B, Q = query->T, N = latents->T;
    if (latents->B != B) return PIO_E_SHAPE;
    DecoderPlan p;
    const bool qb = (query->stride_b == 0 && B > 1);
    if (p.carve(workspace, *cross, final_layer, B, Q, N, qb) > workspace_bytes) return PIO_E_WORKSPACE;
    // perceiver.py:172-177: mask[b,i,j] = query_mask[b,i]
    float *y = final_layer ? p.y : out;
    const int64_t y_ld = final_layer ? pitch4(q_c) : q_c;  // (y is internal when a final layer follows)
    // (with a final Linear the cross-attend's result is only ever its operand: fc2 writes it as 16-bit rows directly --
    //  env PIO_DEC_Y16=0: the fp32 rows + cast pass of rounds 1-3, for A/B)
    static const bool y16_direct = [] {
        const char *e = getenv("PIO_DEC_Y16");
        return !e || atoi(e) != 0;
    }();
    const bool direct = final_layer && y16_direct && final_layer->k == padc(q_c);
    PIO_TRY(cross_attention_run(*cross, *query, *latents, nullptr, query_mask, nullptr, nullptr, y, nullptr, p.cp,
                                s, nullptr, y_ld, qc, direct ? &p.y16 : nullptr, query_tail));
    if (!final_layer) return PIO_OK;
    // perceiver.py:178-179: final nn.Linear on every query row
    if (final_layer->k != padc(q_c)) return PIO_E_SHAPE;
    const pio_tensor3_t ty = {y, (int64_t)Q * y_ld, y_ld, B, Q, q_c};
    if (!direct) PIO_TRY(cast_pair(ty, nullptr, p.y16, padc(q_c), cross->attn.dtype, s));
    return linear_fwd(*final_layer, cross->attn.dtype, p.y16, (int64_t)B * Q, out, nullptr, true, final_out, final_out,
                      0, nullptr, s);
}

int pio_decoder_fwd_qcache(const pio_cross_attention_t *cross, const pio_linear_t *final_layer, int32_t final_out,
                           const pio_tensor3_t *query, const pio_tensor3_t *latents, const uint8_t *query_mask,
                           float *out, void *workspace, size_t workspace_bytes, void *stream, void *q16_hi,
                           void *q16_lo, int32_t q16_valid) {
    return decoder_run(cross, final_layer, final_out, query, nullptr, latents, query_mask, out, workspace,
                       workspace_bytes, stream, q16_hi, q16_lo, q16_valid);
}

int pio_decoder_fwd_split(const pio_cross_attention_t *cross, const pio_linear_t *final_layer, int32_t final_out,
                          const pio_tensor3_t *query, const pio_tensor3_t *query_tail, const pio_tensor3_t *latents,
                          const uint8_t *query_mask, float *out, void *workspace, size_t workspace_bytes, void *stream) {
    if (!query_tail) return PIO_E_ARG;
    return decoder_run(cross, final_layer, final_out, query, query_tail, latents, query_mask, out, workspace,
                       workspace_bytes, stream, nullptr, nullptr, 0);
}

}  // extern "C"
