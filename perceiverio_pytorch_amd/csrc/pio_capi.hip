// extern "C" surface for the library/device queries and the primitive kernels (include/pio_hip.h).
#include "pio_internal.h"

#include <string.h>

#include <vector>

using namespace pio;

namespace pio {
int padc_min() {
    static const int v = [] {
        const char *e = getenv("PIO_PADC_MIN");
        return e ? atoi(e) : 256;
    }();
    return v;
}
}  // namespace pio


// ---------------------------------------------------------------------------------------------------------
// per-launch timing (bench only).  One global recorder: NOT thread-safe, never enabled on the product path.
// ---------------------------------------------------------------------------------------------------------
namespace {
struct ProfRec {
    int cls;
    double flops, bytes;
};
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;  // 2 per record
    std::vector<ProfRec> rec;
    size_t cap = 0;
} g_prof;
}  // namespace

namespace pio {
ProfScope::ProfScope(int cls, double flops, double bytes, hipStream_t stream) : idx(-1), s(stream) {
    if (!g_prof.on || g_prof.rec.size() >= g_prof.cap) return;
    idx = (int)g_prof.rec.size();
    g_prof.rec.push_back({cls, flops, bytes});
    (void)hipEventRecord(g_prof.ev[2 * idx], s);
}
ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_prof.ev[2 * idx + 1], s);
}

// CU budget of the persistent kernels' grids (pio_set_cu_budget): 0 = every CU of the device
static int g_cu_budget = 0;
// ... and the per-call value (pio_call_opts_t.cu_budget), thread-local for the duration of a *_opts call: it wins over
// the process-wide one.  cu_budget_call(v >= 0) sets it, any call returns the previous value.
static thread_local int tl_cu_budget = 0;
int cu_budget_call(int v) {
    const int prev = tl_cu_budget;
    if (v >= 0) tl_cu_budget = v;
    return prev;
}
int cu_budget() {
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int want = tl_cu_budget > 0 ? tl_cu_budget : g_cu_budget;
    return (want > 0 && want < n_cu) ? want : n_cu;
}
}  // namespace pio

extern "C" {

int pio_set_cu_budget(int32_t n_cu) {
    const int prev = g_cu_budget;
    g_cu_budget = n_cu > 0 ? n_cu : 0;
    return prev;
}

int pio_stream_create_cu_mask(void **stream, const uint32_t *mask, uint32_t words) {
    if (!stream || !mask || words == 0) return PIO_E_ARG;
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, words, mask) != hipSuccess) return PIO_E_LAUNCH;
    *stream = (void *)s;
    return PIO_OK;
}

int pio_stream_destroy(void *stream) {
    if (!stream) return PIO_E_ARG;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? PIO_OK : PIO_E_LAUNCH;
}

int pio_prof_begin(int32_t max_records) {
    if (max_records <= 0) return PIO_E_ARG;
    while (g_prof.ev.size() < (size_t)2 * max_records) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return PIO_E_LAUNCH;
        g_prof.ev.push_back(e);
    }
    g_prof.cap = (size_t)max_records;
    g_prof.rec.clear();
    g_prof.on = true;
    return PIO_OK;
}

int pio_prof_end(double *ms, double *flops, double *bytes, int64_t *launches) {
    g_prof.on = false;
    for (int c = 0; c < PROF_CLASSES; ++c) {
        if (ms) ms[c] = 0;
        if (flops) flops[c] = 0;
        if (bytes) bytes[c] = 0;
        if (launches) launches[c] = 0;
    }
    for (size_t i = 0; i < g_prof.rec.size(); ++i) {
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return PIO_E_LAUNCH;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return PIO_E_LAUNCH;
        const ProfRec &r = g_prof.rec[i];
        if (ms) ms[r.cls] += t;
        if (flops) flops[r.cls] += r.flops;
        if (bytes) bytes[r.cls] += r.bytes;
        if (launches) launches[r.cls] += 1;
    }
    const int n = (int)g_prof.rec.size();
    g_prof.rec.clear();
    return n;
}

int pio_version(void) { return PIO_VERSION; }

int pio_arch_ok(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PIO_E_LAUNCH;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return PIO_E_LAUNCH;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

const char *pio_error_string(int code) {
    switch (code) {
        case PIO_OK: return "ok";
        case PIO_E_SHAPE: return "unsupported or inconsistent shape";
        case PIO_E_ALIGN: return "pointer/stride alignment";
        case PIO_E_ARCH: return "device is not gfx950";
        case PIO_E_WORKSPACE: return "workspace too small";
        case PIO_E_LAUNCH: return "HIP launch failed";
        case PIO_E_ARG: return "invalid argument";
        case PIO_E_RANGE: return "values outside the range of the 16-bit operand dtype";
        default: return "unknown error";
    }
}

int32_t pio_pad8(int32_t c) { return pad8(c); }
int32_t pio_padc(int32_t c) { return padc(c); }

int pio_gemm_kernel_override(int which) { return gemm_kernel_override(which); }

size_t pio_packed_weight_bytes(int32_t out, int32_t in, int32_t row_heads, int32_t col_heads) {
    if (out <= 0 || in <= 0 || row_heads <= 0 || col_heads <= 0 || out % row_heads || in % col_heads) return 0;
    return (size_t)row_heads * pad8(out / row_heads) * (size_t)col_heads * pad8(in / col_heads) * 2;
}

int pio_pack_linear(const float *w, const float *bias, int32_t out, int32_t in, int64_t ldw, int32_t row_heads,
                    int32_t col_heads, void *dst_hi, void *dst_lo, float *dst_bias, int32_t dst_row0, int32_t k_pad,
                    int32_t dtype, void *stream) {
    return pack_linear_launch(w, bias, out, in, ldw, row_heads, col_heads, dst_hi, dst_lo, dst_bias, dst_row0, k_pad,
                              dtype, (hipStream_t)stream);
}

int pio_layernorm_cast(const pio_tensor3_t *x, const pio_layernorm_t *ln, void *y, void *y_lo, int32_t c_pad,
                       int32_t dtype, void *stream) {
    if (!x) return PIO_E_ARG;
    return layernorm_cast_launch(*x, ln, y, y_lo, c_pad, dtype, (hipStream_t)stream);
}

int pio_layernorm_cast_cat(const pio_tensor3_t *x1, const pio_tensor3_t *x2, const pio_layernorm_t *ln, void *y,
                           void *y_lo, int32_t c_pad, int32_t dtype, void *stream) {
    if (!x1 || !x2 || !ln) return PIO_E_ARG;
    return layernorm_cast_cat_launch(*x1, *x2, *ln, y, y_lo, c_pad, dtype, (hipStream_t)stream);
}

int pio_bn_relu_maxpool_tokens(const float *x, const float *scale, const float *shift, float *y, int32_t B, int32_t C,
                               int32_t H, int32_t W, int32_t pad_top, int32_t pad_left, void *stream) {
    return bn_relu_pool_nhwc_launch(x, scale, shift, y, B, C, H, W, pad_top, pad_left, (hipStream_t)stream);
}

int pio_gemm_nt(const pio_gemm_t *g, void *stream) {
    if (!g) return PIO_E_ARG;
    return gemm_nt_launch(*g, (hipStream_t)stream);
}

int pio_softmax_rows(const float *S, int64_t lds, void *P, void *P_lo, int64_t ldp, int32_t B, int32_t H, int32_t Tq,
                     int32_t Tk, float scale, const uint8_t *kv_mask, const uint8_t *q_mask, const uint8_t *full_mask,
                     const float *bias, int32_t dtype, void *stream) {
    return softmax_rows_launch(S, lds, P, P_lo, ldp, B, H, Tq, Tk, scale, kv_mask, q_mask, full_mask, bias, dtype, nullptr,
                               (hipStream_t)stream);
}

int pio_flash_attention(int32_t dtype, int32_t dkp, int32_t dvp, int32_t dk, const void *Q, const void *K, const void *V,
                        void *O, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int64_t ldq, int64_t ldk, int64_t ldv,
                        int64_t ldo, int64_t sQb, int64_t sKb, int64_t sVb, int64_t sOb, int32_t v_rowmajor, void *stream) {
    if (dtype != PIO_DT_F16 && dtype != PIO_DT_BF16) return PIO_E_ARG;
    return flash_attention_launch(dtype, dkp, dvp, dk, Q, K, V, O, B, H, Tq, Tk, ldq, ldk, ldv, ldo, sQb, sKb, sVb, sOb,
                                  v_rowmajor != 0, (hipStream_t)stream);
}

}  // extern "C"
