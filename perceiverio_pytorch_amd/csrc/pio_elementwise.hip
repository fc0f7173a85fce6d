// Memory-bound helpers of the hot path (gfx950): weight packing, LayerNorm + cast to the MFMA operand
// dtype, row softmax with masking.  All arithmetic is fp32; only the stored operand is 16-bit.
#include "pio_internal.h"

namespace pio {

// =====================================================================================================
// pack_linear: W [out,in] fp32 -> packed [rows_p][k_pad] 16-bit (hi) + optional lo = W - float(hi).
// Head-padded rows / columns: group g of d logical entries occupies packed entries [g*dp, g*dp + d).
// =====================================================================================================
template <int DT>
__global__ void pack_linear_kernel(const float *__restrict__ w, const float *__restrict__ bias, int out, int in,
                                   int64_t ldw, int dr, int drp, int rows_p, int dc, int dcp, int cols_used,
                                   typename Op<DT>::T *hi, typename Op<DT>::T *lo, float *dst_bias, int dst_row0,
                                   int k_pad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)rows_p * k_pad;
    if (idx >= total) return;
    const int rp = (int)(idx / k_pad);
    const int cp = (int)(idx % k_pad);
    const int hr = rp / drp, jr = rp % drp;
    float v = 0.f;
    const bool rvalid = jr < dr;
    if (rvalid && cp < cols_used) {
        const int hc = cp / dcp, jc = cp % dcp;
        if (jc < dc) v = w[(int64_t)(hr * dr + jr) * ldw + (hc * dc + jc)];
    }
    const typename Op<DT>::T h = Op<DT>::from_f32(v);
    const int64_t o = (int64_t)(dst_row0 + rp) * k_pad + cp;
    hi[o] = h;
    if (lo) lo[o] = Op<DT>::from_f32(v - Op<DT>::to_f32(h));
    if (cp == 0 && dst_bias) dst_bias[dst_row0 + rp] = (rvalid && bias) ? bias[hr * dr + jr] : 0.f;
}

int pack_linear_launch(const float *w, const float *bias, int out, int in, int64_t ldw, int row_heads,
                       int col_heads, void *dst_hi, void *dst_lo, float *dst_bias, int dst_row0, int k_pad,
                       int dtype, hipStream_t s) {
    if (!w || !dst_hi) return PIO_E_ARG;
    if (out <= 0 || in <= 0 || row_heads <= 0 || col_heads <= 0) return PIO_E_SHAPE;
    if (out % row_heads || in % col_heads) return PIO_E_SHAPE;
    const int dr = out / row_heads, drp = pad8(dr);
    const int dc = in / col_heads, dcp = pad8(dc);
    const int rows_p = row_heads * drp;
    const int cols_used = col_heads * dcp;
    if (cols_used > k_pad || (k_pad % 8)) return PIO_E_SHAPE;
    const int64_t total = (int64_t)rows_p * k_pad;
    const int threads = 256;
    const unsigned blocks = (unsigned)((total + threads - 1) / threads);
    if (dtype == PIO_DT_F16)
        hipLaunchKernelGGL((pack_linear_kernel<PIO_DT_F16>), dim3(blocks), dim3(threads), 0, s, w, bias, out, in, ldw,
                           dr, drp, rows_p, dc, dcp, cols_used, (_Float16 *)dst_hi, (_Float16 *)dst_lo, dst_bias,
                           dst_row0, k_pad);
    else if (dtype == PIO_DT_BF16)
        hipLaunchKernelGGL((pack_linear_kernel<PIO_DT_BF16>), dim3(blocks), dim3(threads), 0, s, w, bias, out, in,
                           ldw, dr, drp, rows_p, dc, dcp, cols_used, (__bf16 *)dst_hi, (__bf16 *)dst_lo, dst_bias,
                           dst_row0, k_pad);
    else
        return PIO_E_ARG;
    return launch_status();
}

// =====================================================================================================
// layernorm_cast: one 64-lane wave per row; mean, centred variance (two-pass, like ATen), affine, cast.
// The row is re-read from L1 for the second and third pass (a row is <= a few KiB and owned by one wave),
// so HBM sees it once.  Vector path (float4 / 8-byte stores) when C % 4 == 0 and rows are 16-B aligned.
// =====================================================================================================
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int DT, bool VEC, bool NORM>
__global__ __launch_bounds__(256) void layernorm_cast_kernel(const float *__restrict__ x, int64_t stride_b,
                                                             int64_t stride_t, int T, int64_t rows, int C,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps,
                                                             typename Op<DT>::T *__restrict__ y,
                                                             typename Op<DT>::T *__restrict__ y_lo, int c_pad) {
    typedef typename Op<DT>::T OT;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (row / T) * stride_b + (row % T) * stride_t;
    OT *yr = y + row * (int64_t)c_pad;
    OT *ylr = y_lo ? y_lo + row * (int64_t)c_pad : nullptr;
    float mean = 0.f, rstd = 1.f;
    if (NORM) {
        float s = 0.f;
        if (VEC) {
            for (int i = lane * 4; i < C; i += 256) {
                const f32x4 v = *(const f32x4 *)(xr + i);
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
        } else {
            for (int i = lane; i < C; i += 64) s += xr[i];
        }
        mean = wave_sum(s) / (float)C;
        float q = 0.f;
        if (VEC) {
            for (int i = lane * 4; i < C; i += 256) {
                const f32x4 v = *(const f32x4 *)(xr + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[j] - mean;
                    q += d * d;
                }
            }
        } else {
            for (int i = lane; i < C; i += 64) {
                const float d = xr[i] - mean;
                q += d * d;
            }
        }
        rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    }
    if (VEC) {
        for (int i = lane * 4; i < c_pad; i += 256) {
            f32x4 f = {0.f, 0.f, 0.f, 0.f};
            if (i < C) {
                f = *(const f32x4 *)(xr + i);
                if (NORM) {
                    const f32x4 g = *(const f32x4 *)(gamma + i);
                    const f32x4 b = *(const f32x4 *)(beta + i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[j] = (f[j] - mean) * rstd * g[j] + b[j];
                }
            }
            typename Op<DT>::V4 o, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = Op<DT>::from_f32(f[j]);
                l[j] = Op<DT>::from_f32(f[j] - Op<DT>::to_f32(o[j]));
            }
            *(typename Op<DT>::V4 *)(yr + i) = o;
            if (ylr) *(typename Op<DT>::V4 *)(ylr + i) = l;
        }
    } else {
        for (int i = lane; i < c_pad; i += 64) {
            float v = 0.f;
            if (i < C) {
                v = xr[i];
                if (NORM) v = (v - mean) * rstd * gamma[i] + beta[i];
            }
            const OT h = Op<DT>::from_f32(v);
            yr[i] = h;
            if (ylr) ylr[i] = Op<DT>::from_f32(v - Op<DT>::to_f32(h));
        }
    }
}

// Register-resident variant for the common case (C % 4 == 0, C <= 2048, aligned): each lane keeps its <= 8 float4
// of the row in VGPRs, so the row is read from memory exactly ONCE (the 3-pass form re-reads it through L1/L2,
// which showed up as ~1.6x the algorithmic FETCH on the 64 MB latent array).
// NV = float4 per lane (4: rows up to 1024 channels, 8: up to 2048); T == 0 tells the kernel that the rows are evenly
// spaced (stride_b == T * stride_t), which spares every wave a 64-bit division.
template <int DT, bool NORM, int NV>
__global__ __launch_bounds__(256) void layernorm_cast_reg_kernel(const float *__restrict__ x, int64_t stride_b,
                                                                 int64_t stride_t, int T, int64_t rows, int C,
                                                                 const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float eps,
                                                                 typename Op<DT>::T *__restrict__ y,
                                                                 typename Op<DT>::T *__restrict__ y_lo, int c_pad) {
    typedef typename Op<DT>::T OT;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = T > 0 ? x + (row / T) * stride_b + (row % T) * stride_t : x + row * stride_t;
    OT *yr = y + row * (int64_t)c_pad;
    OT *ylr = y_lo ? y_lo + row * (int64_t)c_pad : nullptr;
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 64 + lane) * 4;
        v[j] = (i < C) ? *(const f32x4 *)(xr + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float mean = 0.f, rstd = 1.f;
    if (NORM) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = (j * 64 + lane) * 4;
            if (i < C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[j][e] - mean;
                    q += d * d;
                }
            }
        }
        rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 64 + lane) * 4;
        if (i < c_pad) {
            f32x4 f = {0.f, 0.f, 0.f, 0.f};
            if (i < C) {
                f = v[j];
                if (NORM) {
                    const f32x4 g = *(const f32x4 *)(gamma + i);
                    const f32x4 b = *(const f32x4 *)(beta + i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = (f[e] - mean) * rstd * g[e] + b[e];
                }
            }
            typename Op<DT>::V4 o, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = Op<DT>::from_f32(f[e]);
                l[e] = Op<DT>::from_f32(f[e] - Op<DT>::to_f32(o[e]));
            }
            *(typename Op<DT>::V4 *)(yr + i) = o;
            if (ylr) *(typename Op<DT>::V4 *)(ylr + i) = l;
        }
    }
}

// LayerNorm fold, first layer of a stack: y = cast(x) (NOT normalised; optionally y_lo = x - y) and, per row and slot of
// SW columns (128: the wide GEMM kernel's consumer; 64: the tile kernels'), the (sum, sum of squares) that the fold's
// consumer GEMM turns into mean / rstd.  C % 256 == 0, contiguous rows; one wave per row: float4 j of lane l covers
// columns 256 j + 4 l, i.e. slot (256 j + 4 l) / SW.
template <int DT, int SW>
__global__ __launch_bounds__(256) void rowstats_cast_kernel(const float *__restrict__ x, int64_t rows, int C,
                                                            typename Op<DT>::T *__restrict__ y,
                                                            typename Op<DT>::T *__restrict__ y_lo,
                                                            float *__restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * C;
    const int nslots = C / SW;
    constexpr int LPS = SW / 4;  // lanes per slot (32 or 16)
    for (int j = 0; j < C / 256; ++j) {
        const int i = (j * 64 + lane) * 4;
        const f32x4 v = *(const f32x4 *)(xr + i);
        float sm = (v[0] + v[1]) + (v[2] + v[3]);
        float sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
        for (int o = LPS / 2; o > 0; o >>= 1) {
            sm += __shfl_xor(sm, o, 64);
            sq += __shfl_xor(sq, o, 64);
        }
        if ((lane & (LPS - 1)) == 0) {
            float *dst = part + (row * nslots + i / SW) * 2;
            dst[0] = sm;
            dst[1] = sq;
        }
        typename Op<DT>::V4 o4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o4[e] = Op<DT>::from_f32(v[e]);
            l4[e] = Op<DT>::from_f32(v[e] - Op<DT>::to_f32(o4[e]));
        }
        *(typename Op<DT>::V4 *)(y + row * C + i) = o4;
        if (y_lo) *(typename Op<DT>::V4 *)(y_lo + row * C + i) = l4;
    }
}

int rowstats_cast_launch(const float *x, int64_t rows, int C, int slot_w, void *y16, void *y16_lo, float *part, int dtype,
                         hipStream_t s) {
    if (!x || !y16 || !part || rows <= 0) return PIO_E_ARG;
    if (C <= 0 || (C % 256) || (slot_w != 64 && slot_w != 128)) return PIO_E_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)y16 & 7) || ((uintptr_t)part & 7)) return PIO_E_ALIGN;
    if ((uintptr_t)y16_lo & 7) return PIO_E_ALIGN;
    ProfScope prof(PROF_LAYERNORM, 0.0, (double)rows * C * (y16_lo ? 8.0 : 6.0), s);
    const unsigned blocks = (unsigned)((rows + 3) / 4);
#define PIO_RS(DTV, SWV)                                                                                            \
    hipLaunchKernelGGL((rowstats_cast_kernel<DTV, SWV>), dim3(blocks), dim3(256), 0, s, x, rows, C, (Op<DTV>::T *)y16, \
                       (Op<DTV>::T *)y16_lo, part)
    if (dtype == PIO_DT_F16) {
        if (slot_w == 128) PIO_RS(PIO_DT_F16, 128);
        else PIO_RS(PIO_DT_F16, 64);
    } else if (dtype == PIO_DT_BF16) {
        if (slot_w == 128) PIO_RS(PIO_DT_BF16, 128);
        else PIO_RS(PIO_DT_BF16, 64);
    } else
        return PIO_E_ARG;
#undef PIO_RS
    return launch_status();
}

// V^T[b][c][t] = X[b][t][c] for a 16-bit array (K / V projection fold of a single-head cross-attend: the LayerNorm'd
// inputs themselves are the values, consumed K-contiguous): X [B][T][ldx] -> VT [B][C][tkv], columns t in [T, tkv) zero.
// 64 x 64 tiles through LDS, 16-byte accesses on both sides (C % 8 == 0, ldx % 8 == 0, tkv % 8 == 0).
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t *__restrict__ x, int64_t ldx, int T, int C,
                                                          uint16_t *__restrict__ vt, int64_t tkv) {
    __shared__ uint16_t tile[64][72];  // (+8: the transposed reads walk a column)
    const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const uint16_t *xb = x + (int64_t)b * T * ldx;
    typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = it * 32 + (tid >> 3), ch = (tid & 7) * 8;
        u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 + r < T && c0 + ch < C) v = *(const u16x8 *)(xb + (int64_t)(t0 + r) * ldx + c0 + ch);
        *(u16x8 *)&tile[r][ch] = v;
    }
    __syncthreads();
    uint16_t *vb = vt + (int64_t)b * C * tkv;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = it * 32 + (tid >> 3), tc = (tid & 7) * 8;
        if (c0 + c < C && t0 + tc < tkv) {
            u16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[tc + e][c];
            *(u16x8 *)(vb + (int64_t)(c0 + c) * tkv + t0 + tc) = v;
        }
    }
}

int transpose16_launch(const void *x, int64_t ldx, int B, int T, int C, void *vt, int64_t tkv, hipStream_t s) {
    if (!x || !vt || B <= 0 || T <= 0 || C <= 0) return PIO_E_ARG;
    if ((C & 7) || (ldx & 7) || (tkv & 7) || tkv < T || B > 65535) return PIO_E_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)vt & 15)) return PIO_E_ALIGN;
    ProfScope prof(PROF_LAYERNORM, 0.0, 2.0 * B * (double)T * C * 2.0, s);
    dim3 grid((unsigned)((tkv + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B);
    hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, s, (const uint16_t *)x, ldx, T, C, (uint16_t *)vt, tkv);
    return launch_status();
}

// The same with 8-byte (float2) accesses for even channel counts whose rows are only 8-byte aligned -- the 322-wide
// encoder input array [B, 3136, 322]: 129 MB at B = 32, the largest single read of the model.
template <int DT, bool NORM>
__global__ __launch_bounds__(256) void layernorm_cast_reg2_kernel(const float *__restrict__ x, int64_t stride_b,
                                                                  int64_t stride_t, int T, int64_t rows, int C,
                                                                  const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float eps,
                                                                  typename Op<DT>::T *__restrict__ y,
                                                                  typename Op<DT>::T *__restrict__ y_lo, int c_pad) {
    typedef typename Op<DT>::T OT;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef OT OT2 __attribute__((ext_vector_type(2)));
    constexpr int NV = 16;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (row / T) * stride_b + (row % T) * stride_t;
    OT *yr = y + row * (int64_t)c_pad;
    OT *ylr = y_lo ? y_lo + row * (int64_t)c_pad : nullptr;
    f32x2 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 64 + lane) * 2;
        v[j] = (i < C) ? *(const f32x2 *)(xr + i) : (f32x2){0.f, 0.f};
    }
    float mean = 0.f, rstd = 1.f;
    if (NORM) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += v[j][0] + v[j][1];
        mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = (j * 64 + lane) * 2;
            if (i < C) {
                const float d0 = v[j][0] - mean, d1 = v[j][1] - mean;
                q += d0 * d0 + d1 * d1;
            }
        }
        rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 64 + lane) * 2;
        if (i < c_pad) {
            f32x2 f = {0.f, 0.f};
            if (i < C) {
                f = v[j];
                if (NORM) {
                    const f32x2 g = *(const f32x2 *)(gamma + i);
                    const f32x2 b = *(const f32x2 *)(beta + i);
                    f[0] = (f[0] - mean) * rstd * g[0] + b[0];
                    f[1] = (f[1] - mean) * rstd * g[1] + b[1];
                }
            }
            OT2 o, l;
            o[0] = Op<DT>::from_f32(f[0]);
            o[1] = Op<DT>::from_f32(f[1]);
            l[0] = Op<DT>::from_f32(f[0] - Op<DT>::to_f32(o[0]));
            l[1] = Op<DT>::from_f32(f[1] - Op<DT>::to_f32(o[1]));
            *(OT2 *)(yr + i) = o;
            if (ylr) *(OT2 *)(ylr + i) = l;
        }
    }
}

// LayerNorm over the CONCATENATION [x1 row | x2 row] without materialising it: x1 [B, T, C1] (e.g. the 64 conv features
// of the ImageNet preprocessor), x2 [T, C2] or [B, T, C2] (the 258 Fourier position channels: ONE batch-invariant table,
// stride_b == 0).  Same lane <-> channel mapping and arithmetic as layernorm_cast_reg2_kernel on the concatenated row,
// so the result is bit-identical to LayerNorm(torch.cat([x1, x2], -1)) through that kernel; C1 and C2 even, rows 8-byte
// aligned.  Replaces preprocessors.py:176-200's concat + transformer_primitives.py:379's layer_norm_kv: the encoder's
// largest read drops from B x 3136 x 322 fp32 (129 MB at B = 32, 80 % of it the replicated table) to the 26 MB of
// features plus one 3.2 MB table that stays in cache.
template <int DT>
__global__ __launch_bounds__(256) void layernorm_cast_cat2_kernel(const float *__restrict__ x1, int64_t sb1, int64_t st1,
                                                                  int C1, const float *__restrict__ x2, int64_t sb2,
                                                                  int64_t st2, int C2, int T, int64_t rows,
                                                                  const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float eps,
                                                                  typename Op<DT>::T *__restrict__ y,
                                                                  typename Op<DT>::T *__restrict__ y_lo, int c_pad) {
    typedef typename Op<DT>::T OT;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef OT OT2 __attribute__((ext_vector_type(2)));
    constexpr int NV = 16;
    const int C = C1 + C2;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t b = row / T, t = row % T;
    const float *r1 = x1 + b * sb1 + t * st1;
    const float *r2 = x2 + b * sb2 + t * st2 - C1;   // indexed with the concatenated channel
    OT *yr = y + row * (int64_t)c_pad;
    OT *ylr = y_lo ? y_lo + row * (int64_t)c_pad : nullptr;
    f32x2 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 64 + lane) * 2;
        v[j] = (i < C1) ? *(const f32x2 *)(r1 + i) : ((i < C) ? *(const f32x2 *)(r2 + i) : (f32x2){0.f, 0.f});
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += v[j][0] + v[j][1];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 64 + lane) * 2;
        if (i < C) {
            const float d0 = v[j][0] - mean, d1 = v[j][1] - mean;
            q += d0 * d0 + d1 * d1;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 64 + lane) * 2;
        if (i < c_pad) {
            f32x2 f = {0.f, 0.f};
            if (i < C) {
                const f32x2 g = *(const f32x2 *)(gamma + i);
                const f32x2 bb = *(const f32x2 *)(beta + i);
                f[0] = (v[j][0] - mean) * rstd * g[0] + bb[0];
                f[1] = (v[j][1] - mean) * rstd * g[1] + bb[1];
            }
            OT2 o, l;
            o[0] = Op<DT>::from_f32(f[0]);
            o[1] = Op<DT>::from_f32(f[1]);
            l[0] = Op<DT>::from_f32(f[0] - Op<DT>::to_f32(o[0]));
            l[1] = Op<DT>::from_f32(f[1] - Op<DT>::to_f32(o[1]));
            *(OT2 *)(yr + i) = o;
            if (ylr) *(OT2 *)(ylr + i) = l;
        }
    }
}

int layernorm_cast_cat_launch(const pio_tensor3_t &x1, const pio_tensor3_t &x2, const pio_layernorm_t &ln, void *y,
                              void *y_lo, int c_pad, int dtype, hipStream_t s) {
    if (!x1.data || !x2.data || !y || !ln.gamma || !ln.beta) return PIO_E_ARG;
    const int C = x1.C + x2.C;
    if (x1.B <= 0 || x1.T <= 0 || x1.C <= 0 || x2.C <= 0 || x2.T != x1.T || (x2.B != x1.B && x2.B != 1) || ln.c != C ||
        c_pad < C || (c_pad % 8) || c_pad > 2048 || (x1.C & 1) || (x2.C & 1))
        return PIO_E_SHAPE;
    if (((uintptr_t)x1.data & 7) || ((uintptr_t)x2.data & 7) || (x1.stride_b & 1) || (x1.stride_t & 1) ||
        (x2.stride_b & 1) || (x2.stride_t & 1) || ((uintptr_t)y & 3) || ((uintptr_t)y_lo & 3) ||
        ((uintptr_t)ln.gamma & 7) || ((uintptr_t)ln.beta & 7))
        return PIO_E_ALIGN;
    const int64_t rows = (int64_t)x1.B * x1.T;
    const int64_t sb2 = x2.B == 1 ? 0 : x2.stride_b;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    ProfScope prof(PROF_LAYERNORM, 0.0, (double)rows * (4.0 * x1.C + (y_lo ? 4.0 : 2.0) * c_pad) + 4.0 * x2.T * x2.C, s);
    if (dtype == PIO_DT_F16)
        hipLaunchKernelGGL((layernorm_cast_cat2_kernel<PIO_DT_F16>), dim3(blocks), dim3(256), 0, s, x1.data, x1.stride_b,
                           x1.stride_t, x1.C, x2.data, sb2, x2.stride_t, x2.C, x1.T, rows, ln.gamma, ln.beta, ln.eps,
                           (Op<PIO_DT_F16>::T *)y, (Op<PIO_DT_F16>::T *)y_lo, c_pad);
    else if (dtype == PIO_DT_BF16)
        hipLaunchKernelGGL((layernorm_cast_cat2_kernel<PIO_DT_BF16>), dim3(blocks), dim3(256), 0, s, x1.data, x1.stride_b,
                           x1.stride_t, x1.C, x2.data, sb2, x2.stride_t, x2.C, x1.T, rows, ln.gamma, ln.beta, ln.eps,
                           (Op<PIO_DT_BF16>::T *)y, (Op<PIO_DT_BF16>::T *)y_lo, c_pad);
    else
        return PIO_E_ARG;
    return launch_status();
}

int layernorm_cast_launch(const pio_tensor3_t &x, const pio_layernorm_t *ln, void *y, void *y_lo, int c_pad,
                          int dtype, hipStream_t s) {
    if (!x.data || !y) return PIO_E_ARG;
    if (x.B <= 0 || x.T <= 0 || x.C <= 0 || c_pad < x.C || (c_pad % 8)) return PIO_E_SHAPE;
    if (ln && (ln->c != x.C || !ln->gamma || !ln->beta)) return PIO_E_SHAPE;
    const int64_t rows = (int64_t)x.B * x.T;
    const bool vec = (x.C % 4 == 0) && (((uintptr_t)x.data & 15) == 0) && (x.stride_b % 4 == 0) &&
                     (x.stride_t % 4 == 0) && (((uintptr_t)y & 7) == 0) && (((uintptr_t)y_lo & 7) == 0) &&
                     (!ln || ((((uintptr_t)ln->gamma) & 15) == 0 && (((uintptr_t)ln->beta) & 15) == 0));
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    ProfScope prof(PROF_LAYERNORM, 0.0, (double)rows * (4.0 * x.C + (y_lo ? 4.0 : 2.0) * c_pad), s);
    const float eps = ln ? ln->eps : 0.f;
    const float *g = ln ? ln->gamma : nullptr;
    const float *b = ln ? ln->beta : nullptr;
#define PIO_LN_LAUNCH(DTV, VECV, NORMV)                                                                          \
    hipLaunchKernelGGL((layernorm_cast_kernel<DTV, VECV, NORMV>), dim3(blocks), dim3(256), 0, s, x.data,          \
                       x.stride_b, x.stride_t, x.T, rows, x.C, g, b, eps, (typename Op<DTV>::T *)y,                \
                       (typename Op<DTV>::T *)y_lo, c_pad)
    if (vec && c_pad <= 2048 && (dtype == PIO_DT_F16 || dtype == PIO_DT_BF16)) {
    const bool even = x.B == 1 || x.stride_b == (int64_t)x.T * x.stride_t;
    const int t_arg = even ? 0 : x.T;
#define PIO_LNR_LAUNCH2(DTV, NORMV, NVV)                                                                               \
    hipLaunchKernelGGL((layernorm_cast_reg_kernel<DTV, NORMV, NVV>), dim3(blocks), dim3(256), 0, s, x.data, x.stride_b, \
                       x.stride_t, t_arg, rows, x.C, g, b, eps, (typename Op<DTV>::T *)y,                               \
                       (typename Op<DTV>::T *)y_lo, c_pad)
#define PIO_LNR_LAUNCH(DTV, NORMV)                              \
    do {                                                        \
        if (c_pad <= 1024) PIO_LNR_LAUNCH2(DTV, NORMV, 4);      \
        else PIO_LNR_LAUNCH2(DTV, NORMV, 8);                    \
    } while (0)
        if (dtype == PIO_DT_F16) { if (ln) PIO_LNR_LAUNCH(PIO_DT_F16, true); else PIO_LNR_LAUNCH(PIO_DT_F16, false); }
        else                     { if (ln) PIO_LNR_LAUNCH(PIO_DT_BF16, true); else PIO_LNR_LAUNCH(PIO_DT_BF16, false); }
#undef PIO_LNR_LAUNCH2
#undef PIO_LNR_LAUNCH
        return launch_status();
    }
    const bool vec2 = (x.C % 2 == 0) && (((uintptr_t)x.data & 7) == 0) && (x.stride_b % 2 == 0) &&
                      (x.stride_t % 2 == 0) && (((uintptr_t)y & 3) == 0) && (((uintptr_t)y_lo & 3) == 0) &&
                      (!ln || ((((uintptr_t)ln->gamma) & 7) == 0 && (((uintptr_t)ln->beta) & 7) == 0));
    if (vec2 && c_pad <= 2048 && (dtype == PIO_DT_F16 || dtype == PIO_DT_BF16)) {
#define PIO_LNR2_LAUNCH(DTV, NORMV)                                                                               \
    hipLaunchKernelGGL((layernorm_cast_reg2_kernel<DTV, NORMV>), dim3(blocks), dim3(256), 0, s, x.data, x.stride_b, \
                       x.stride_t, x.T, rows, x.C, g, b, eps, (typename Op<DTV>::T *)y,                            \
                       (typename Op<DTV>::T *)y_lo, c_pad)
        if (dtype == PIO_DT_F16) { if (ln) PIO_LNR2_LAUNCH(PIO_DT_F16, true); else PIO_LNR2_LAUNCH(PIO_DT_F16, false); }
        else                     { if (ln) PIO_LNR2_LAUNCH(PIO_DT_BF16, true); else PIO_LNR2_LAUNCH(PIO_DT_BF16, false); }
#undef PIO_LNR2_LAUNCH
        return launch_status();
    }
    if (dtype == PIO_DT_F16) {
        if (ln) { if (vec) PIO_LN_LAUNCH(PIO_DT_F16, true, true); else PIO_LN_LAUNCH(PIO_DT_F16, false, true); }
        else    { if (vec) PIO_LN_LAUNCH(PIO_DT_F16, true, false); else PIO_LN_LAUNCH(PIO_DT_F16, false, false); }
    } else if (dtype == PIO_DT_BF16) {
        if (ln) { if (vec) PIO_LN_LAUNCH(PIO_DT_BF16, true, true); else PIO_LN_LAUNCH(PIO_DT_BF16, false, true); }
        else    { if (vec) PIO_LN_LAUNCH(PIO_DT_BF16, true, false); else PIO_LN_LAUNCH(PIO_DT_BF16, false, false); }
    } else {
        return PIO_E_ARG;
    }
#undef PIO_LN_LAUNCH
    return launch_status();
}

// =====================================================================================================
// softmax_rows: P = softmax_j((S + bias) * scale) over valid keys; masked keys get exactly 0 (the
// reference's exp(-1e30 - max) underflows to 0 in fp32), rows without any valid key are zeros ("wipe",
// transformer_primitives.py:168-175).  One wave per row for short rows, one 256-thread block per row
// for long ones (M up to 182 528 in the optical-flow configuration).
// =====================================================================================================
struct SoftmaxParams {
    const float *S;
    int64_t lds;
    void *P, *P_lo;
    int64_t ldp;
    int B, H, Tq, Tk;
    float scale;
    const uint8_t *kv_mask, *q_mask, *full_mask;
    const float *bias;
    float *probs;
};

template <int DT, int WAVES_PER_ROW>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const SoftmaxParams p) {
    typedef typename Op<DT>::T OT;
    __shared__ float red[8];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t rows = (int64_t)p.B * p.H * p.Tq;
    int64_t row;
    int t0, tstep;
    if (WAVES_PER_ROW == 1) {
        row = (int64_t)blockIdx.x * 4 + wave;
        t0 = lane;
        tstep = 64;
    } else {
        row = blockIdx.x;
        t0 = threadIdx.x;
        tstep = 256;
    }
    const bool active = row < rows;
    if (WAVES_PER_ROW == 1 && !active) return;
    const int i = (int)(row % p.Tq);
    const int b = (int)(row / ((int64_t)p.H * p.Tq));
    const float *s = p.S + row * p.lds;
    const float *bi = p.bias ? p.bias + row * (int64_t)p.Tk : nullptr;
    const uint8_t *km = p.kv_mask ? p.kv_mask + (int64_t)b * p.Tk : nullptr;
    const uint8_t *fm = p.full_mask ? p.full_mask + ((int64_t)b * p.Tq + i) * p.Tk : nullptr;
    const bool qok = p.q_mask ? p.q_mask[(int64_t)b * p.Tq + i] != 0 : true;
    OT *pr = (OT *)p.P + row * p.ldp;
    OT *plr = p.P_lo ? (OT *)p.P_lo + row * p.ldp : nullptr;
    float *po = p.probs ? p.probs + row * (int64_t)p.Tk : nullptr;

    auto valid = [&](int j) -> bool { return qok && (!km || km[j]) && (!fm || fm[j]); };
    auto score = [&](int j) -> float { return (s[j] + (bi ? bi[j] : 0.f)) * p.scale; };

    float m = -INFINITY;
    for (int j = t0; j < p.Tk; j += tstep)
        if (valid(j)) m = fmaxf(m, score(j));
    m = wave_max(m);
    if (WAVES_PER_ROW > 1) {
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
    }
    const bool any = m > -INFINITY;  // wave-uniform
    float sum = 0.f;
    if (any)
        for (int j = t0; j < p.Tk; j += tstep)
            if (valid(j)) sum += __expf(score(j) - m);
    sum = wave_sum(sum);
    if (WAVES_PER_ROW > 1) {
        if (lane == 0) red[4 + wave] = sum;
        __syncthreads();
        sum = (red[4] + red[5]) + (red[6] + red[7]);
    }
    const float inv = any ? 1.f / sum : 0.f;
    const float uni = 1.f / (float)p.Tk;
    for (int j = t0; j < (int)p.ldp; j += tstep) {
        float v = 0.f;
        if (j < p.Tk && any && valid(j)) v = __expf(score(j) - m) * inv;
        const OT h = Op<DT>::from_f32(v);
        pr[j] = h;
        if (plr) plr[j] = Op<DT>::from_f32(v - Op<DT>::to_f32(h));
        if (po && j < p.Tk) po[j] = any ? v : uni;  // reference returns the un-wiped (uniform) matrix
    }
}

// Register-resident variant: one wave per row, the row (<= NV4*256 scores, Tk % 4 == 0) is read from memory ONCE
// with 16-byte loads and kept in VGPRs for the max / sum / normalise steps (the generic kernel above re-reads it three
// times through L1/L2).  Loads are unconditional from clamped addresses -- a per-element "load or constant" select
// makes hipcc branch around every load and drain vmcnt per element -- and masking is a register select afterwards.
// PLAIN = no mask / bias / lo / probability outputs (the common case): nothing but the loads, exp and stores.
template <int DT, int NV4, bool PLAIN>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const SoftmaxParams p) {
    typedef typename Op<DT>::T OT;
    typedef typename Op<DT>::V4 OV4;
    const int lane = threadIdx.x & 63;
    const int64_t rows = (int64_t)p.B * p.H * p.Tq;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int i = (int)(row % p.Tq);
    const int b = (int)(row / ((int64_t)p.H * p.Tq));
    const float *s = p.S + row * p.lds;
    OT *pr = (OT *)p.P + row * p.ldp;
    const int last4 = p.Tk - 4;  // Tk % 4 == 0: the last aligned float4 of the row

    f32x4 v[NV4];
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
        const int k = (j * 64 + lane) * 4;
        const f32x4 x = *(const f32x4 *)(s + (k <= last4 ? k : last4));
        v[j] = x;
    }
    bool qok = true;
    if (!PLAIN) {
        const float *bi = p.bias ? p.bias + row * (int64_t)p.Tk : nullptr;
        const uint8_t *km = p.kv_mask ? p.kv_mask + (int64_t)b * p.Tk : nullptr;
        const uint8_t *fm = p.full_mask ? p.full_mask + ((int64_t)b * p.Tq + i) * p.Tk : nullptr;
        qok = p.q_mask ? p.q_mask[(int64_t)b * p.Tq + i] != 0 : true;
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int k = (j * 64 + lane) * 4;
            const int kc = k <= last4 ? k : last4;
            if (bi) {
                const f32x4 bv = *(const f32x4 *)(bi + kc);
                v[j] += bv;
            }
            uint32_t mk = 0x01010101u;
            if (km) mk &= *(const uint32_t *)(km + kc);
            if (fm) {
                const uint32_t f = *(const uint32_t *)(fm + kc);
                mk = ((mk & 0xffu) && (f & 0xffu) ? 1u : 0u) | ((mk & 0xff00u) && (f & 0xff00u) ? 0x100u : 0u) |
                     ((mk & 0xff0000u) && (f & 0xff0000u) ? 0x10000u : 0u) |
                     ((mk & 0xff000000u) && (f & 0xff000000u) ? 0x1000000u : 0u);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!qok || !((mk >> (8 * e)) & 0xffu)) v[j][e] = -INFINITY;
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
        const int k = (j * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = (k < p.Tk) ? v[j][e] * p.scale : -INFINITY;
            v[j][e] = x;
            m = fmaxf(m, x);
        }
    }
    m = wave_max(m);
    const bool any = m > -INFINITY;
    const float msafe = any ? m : 0.f;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ex = __expf(v[j][e] - msafe);   // exp(-inf) = 0 for masked / out-of-range entries
            v[j][e] = ex;
            sum += ex;
        }
    sum = wave_sum(sum);
    const float inv = any ? 1.f / sum : 0.f;
    OT *plr = (!PLAIN && p.P_lo) ? (OT *)p.P_lo + row * p.ldp : nullptr;
    float *po = (!PLAIN && p.probs) ? p.probs + row * (int64_t)p.Tk : nullptr;
    const float uni = 1.f / (float)p.Tk;
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
        const int k = (j * 64 + lane) * 4;
        if (k < (int)p.ldp) {     // ldp % 4 == 0; columns [Tk, ldp) get exact zeros
            OV4 h, l;
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = v[j][e] * inv;
                h[e] = Op<DT>::from_f32(x[e]);
                l[e] = Op<DT>::from_f32(x[e] - Op<DT>::to_f32(h[e]));
            }
            *(OV4 *)(pr + k) = h;
            if (plr) *(OV4 *)(plr + k) = l;
            if (po && k < p.Tk) {
                if (!any) x = (f32x4){uni, uni, uni, uni};
                *(f32x4 *)(po + k) = x;
            }
        }
    }
}

int softmax_rows_launch(const float *S, int64_t lds, void *P, void *P_lo, int64_t ldp, int B, int H, int Tq, int Tk,
                        float scale, const uint8_t *kv_mask, const uint8_t *q_mask, const uint8_t *full_mask,
                        const float *bias, int dtype, float *probs_out, hipStream_t s) {
    if (!S || !P) return PIO_E_ARG;
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || ldp < Tk || lds < Tk) return PIO_E_SHAPE;
    if (dtype != PIO_DT_F16 && dtype != PIO_DT_BF16) return PIO_E_ARG;
    SoftmaxParams p{S, lds, P, P_lo, ldp, B, H, Tq, Tk, scale, kv_mask, q_mask, full_mask, bias, probs_out};
    const int64_t rows = (int64_t)B * H * Tq;
    ProfScope prof(PROF_SOFTMAX, 0.0, (double)rows * (4.0 * Tk + (P_lo ? 4.0 : 2.0) * ldp), s);
    const bool plain = !kv_mask && !q_mask && !full_mask && !bias && !P_lo && !probs_out;
    const bool vec4 = (Tk % 4 == 0) && (ldp % 4 == 0) && (lds % 4 == 0) && (((uintptr_t)S & 15) == 0) &&
                      (((uintptr_t)P & 7) == 0) && (((uintptr_t)P_lo & 7) == 0) && (((uintptr_t)bias & 15) == 0) &&
                      (((uintptr_t)probs_out & 15) == 0) && (((uintptr_t)kv_mask & 3) == 0) &&
                      (((uintptr_t)full_mask & 3) == 0);
    if (vec4 && ldp <= 16 * 256) {
        const unsigned blocks = (unsigned)((rows + 3) / 4);
#define PIO_SMR2(DTV, NVV)                                                                                         \
    do {                                                                                                           \
        if (plain) hipLaunchKernelGGL((softmax_rows_reg_kernel<DTV, NVV, true>), dim3(blocks), dim3(256), 0, s, p); \
        else hipLaunchKernelGGL((softmax_rows_reg_kernel<DTV, NVV, false>), dim3(blocks), dim3(256), 0, s, p);      \
    } while (0)
#define PIO_SMR(NVV)                                             \
    do {                                                         \
        if (dtype == PIO_DT_F16) PIO_SMR2(PIO_DT_F16, NVV);      \
        else PIO_SMR2(PIO_DT_BF16, NVV);                         \
    } while (0)
        if (ldp <= 2 * 256) PIO_SMR(2);
        else if (ldp <= 8 * 256) PIO_SMR(8);
        else PIO_SMR(16);
#undef PIO_SMR
#undef PIO_SMR2
        return launch_status();
    }
    if (Tk <= 2048) {
        const unsigned blocks = (unsigned)((rows + 3) / 4);
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((softmax_rows_kernel<PIO_DT_F16, 1>), dim3(blocks), dim3(256), 0, s, p);
        else if (dtype == PIO_DT_BF16) hipLaunchKernelGGL((softmax_rows_kernel<PIO_DT_BF16, 1>), dim3(blocks), dim3(256), 0, s, p);
        else return PIO_E_ARG;
    } else {
        const unsigned blocks = (unsigned)rows;
        if (dtype == PIO_DT_F16) hipLaunchKernelGGL((softmax_rows_kernel<PIO_DT_F16, 4>), dim3(blocks), dim3(256), 0, s, p);
        else if (dtype == PIO_DT_BF16) hipLaunchKernelGGL((softmax_rows_kernel<PIO_DT_BF16, 4>), dim3(blocks), dim3(256), 0, s, p);
        else return PIO_E_ARG;
    }
    return launch_status();
}

// =====================================================================================================
// Tail of Conv2DDownsample (processor_utils.py:163-180) in one pass: eval-mode BatchNorm (scale / shift per channel) ->
// ReLU -> 3x3 stride-2 max-pool with TF-"SAME" padding -> channels-last tokens.  x [B, C, H, W] (the 7x7 conv's
// output) -> y [B, OH * OW, C], the layout the encoder consumes.  Replaces four full passes over the 103 MB feature
// map (BatchNorm, ReLU, pad + max-pool, movedim + reshape copy) by one read of it and a 26 MB write.  One workgroup =
// one output row (b, oh) of 64 channels.  Reading runs along w (lane = ow: a wave's nine window taps are nine
// stride-2 spans of one input row, served by the caches), writing along c: the 64 x OW results turn through a small
// LDS tile (pitch 65: conflict-free both ways) and leave as whole 256-byte token rows.  Out-of-range window cells
// are skipped: after the ReLU every value is >= 0, so this equals the reference's zero padding.
// =====================================================================================================
__global__ __launch_bounds__(256) void bn_relu_pool_nhwc_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift, float *__restrict__ y,
                                                                int C, int H, int W, int OH, int OW, int pad_top,
                                                                int pad_left) {
    __shared__ float tile[64 * 65];  // [ow within the 64-wide chunk][c]
    const int b = blockIdx.z, oh = blockIdx.x, c0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = C - c0 < 64 ? C - c0 : 64;
    const int ih0 = 2 * oh - pad_top;
    for (int ow0 = 0; ow0 < OW; ow0 += 64) {
        const int ow = ow0 + lane;
        const int iw0 = 2 * ow - pad_left;
        if (ow < OW) {
            for (int c = wave; c < nc; c += 4) {
                const float sc = scale[c0 + c], sh = shift[c0 + c];
                const float *xc = x + ((int64_t)b * C + c0 + c) * H * W;
                float m = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int ih = ih0 + r;
                    if (ih < 0 || ih >= H) continue;
                    const float *xr = xc + (int64_t)ih * W;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int iw = iw0 + dx;
                        if (iw >= 0 && iw < W) m = fmaxf(m, fmaf(xr[iw], sc, sh));
                    }
                }
                tile[lane * 65 + c] = m;
            }
        }
        __syncthreads();
        const int now = OW - ow0 < 64 ? OW - ow0 : 64;
        for (int idx = threadIdx.x; idx < now * 64; idx += 256) {
            const int o = idx >> 6, c = idx & 63;
            if (c < nc) y[((int64_t)b * OH * OW + (int64_t)oh * OW + ow0 + o) * C + c0 + c] = tile[o * 65 + c];
        }
        __syncthreads();
    }
}

int bn_relu_pool_nhwc_launch(const float *x, const float *scale, const float *shift, float *y, int B, int C, int H, int W,
                             int pad_top, int pad_left, hipStream_t s) {
    if (!x || !scale || !shift || !y) return PIO_E_ARG;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || pad_top < 0 || pad_top > 1 || pad_left < 0 || pad_left > 1) return PIO_E_SHAPE;
    if (B > 65535 || (C + 63) / 64 > 65535) return PIO_E_SHAPE;
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    ProfScope prof(PROF_LAYERNORM, 0.0, 4.0 * B * C * ((double)H * W + (double)OH * OW), s);
    hipLaunchKernelGGL(bn_relu_pool_nhwc_kernel, dim3((unsigned)OH, (unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0,
                       s, x, scale, shift, y, C, H, W, OH, OW, pad_top, pad_left);
    return launch_status();
}

}  // namespace pio
