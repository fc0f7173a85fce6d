"""Numerics model of the HIP pipeline (CPU, numpy): the oracle's algorithm with 16-bit rounding inserted at
exactly the points where libpio_hip.so stores a 16-bit operand.  Used to choose the precision policy and to
attribute error to individual rounding points (DESIGN.md section "precision").  Dev tool, not product code."""
import math
import sys
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import perceiver_oracle as O  # noqa: E402


class Rounder:
    def __init__(self, dtype="f16", points=None, w_passes=2, fold=False):
        self.dtype = dtype
        self.points = points  # None = all
        self.w_passes = w_passes
        self.fold = fold      # self-attend LayerNorms folded into the consuming GEMM

    def r16(self, x):
        if self.dtype == "f16":
            return x.astype(np.float16).astype(np.float64)
        # bf16 RNE
        f = x.astype(np.float32)
        u = f.view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return u.astype(np.uint32).view(np.float32).astype(np.float64)

    def __call__(self, x, tag):
        if self.points is not None and tag not in self.points:
            return x
        return self.r16(x)

    def weight(self, w):
        if self.points is not None and "w" not in self.points:
            return w
        hi = self.r16(w)
        if self.w_passes == 1:
            return hi
        return hi + self.r16(w - hi)


def lin(x, p, name, rd):
    return x @ rd.weight(p[name + ".weight"].astype(np.float64)).T + p[name + ".bias"].astype(np.float64)


def attention(p, xq, xkv, H, rd, mask=None, qk3=False):
    q = rd(lin(xq, p, "proj_q", rd), "q")
    k = rd(lin(xkv, p, "proj_k", rd), "k")
    v = rd(lin(xkv, p, "proj_v", rd), "v")
    B, Tq, qk = q.shape
    Tk = k.shape[1]
    dk, dv = qk // H, v.shape[2] // H
    qh = q.reshape(B, Tq, H, dk).transpose(0, 2, 1, 3)
    kh = k.reshape(B, Tk, H, dk).transpose(0, 2, 1, 3)
    vh = v.reshape(B, Tk, H, dv).transpose(0, 2, 1, 3)
    s = qh @ kh.transpose(0, 1, 3, 2) / math.sqrt(dk)
    if mask is not None:
        s = np.where(mask[:, None], s, -1e30)
    pr = rd(O.softmax_lastdim(s), "p")
    o = rd((pr @ vh).transpose(0, 2, 1, 3).reshape(B, Tq, H * dv), "o")
    if mask is not None:
        o = np.where(np.all(mask == 0, axis=2, keepdims=True), 0.0, o)
    return lin(o, p, "final", rd)


def mlp(p, x, rd):
    h = rd(O.gelu(lin(x, p, "fc1", rd)), "h")
    return lin(h, p, "fc2", rd)


def ln(x, p, name):
    return O.layer_norm(x, p[name + ".weight"].astype(np.float64), p[name + ".bias"].astype(np.float64))


def lin_folded(x, p, ln_name, p_lin, name, rd):
    """LN(x) W^T + b with the LayerNorm folded into the GEMM: the 16-bit operand is x itself, the weight is
    gamma-scaled before rounding, and mean / rstd (exact row statistics) enter in the epilogue:
    rs * (x16 W'^T - mu * rowsum(W')) + (W beta + b)."""
    g, be = p[ln_name + ".weight"].astype(np.float64), p[ln_name + ".bias"].astype(np.float64)
    w, b = p_lin[name + ".weight"].astype(np.float64), p_lin[name + ".bias"].astype(np.float64)
    mu = x.mean(-1, keepdims=True)
    rs = 1.0 / np.sqrt(x.var(-1, keepdims=True) + 1e-5)
    wp = rd.weight(w * g[None, :])
    return rs * (rd(x, "ln") @ wp.T - mu * wp.sum(1)[None, None, :]) + (w @ be + b)


def self_attention_folded(p, x, H, rd):
    pa = O._sub(p, "attention")
    q = rd(lin_folded(x, p, "layer_norm1", pa, "proj_q", rd), "q")
    k = rd(lin_folded(x, p, "layer_norm1", pa, "proj_k", rd), "k")
    v = rd(lin_folded(x, p, "layer_norm1", pa, "proj_v", rd), "v")
    B, T, qk = q.shape
    dk, dv = qk // H, v.shape[2] // H
    qh = q.reshape(B, T, H, dk).transpose(0, 2, 1, 3)
    kh = k.reshape(B, T, H, dk).transpose(0, 2, 1, 3)
    vh = v.reshape(B, T, H, dv).transpose(0, 2, 1, 3)
    pr = rd(O.softmax_lastdim(qh @ kh.transpose(0, 1, 3, 2) / math.sqrt(dk)), "p")
    o = rd((pr @ vh).transpose(0, 2, 1, 3).reshape(B, T, H * dv), "o")
    x = x + lin(o, pa, "final", rd)
    pm = O._sub(p, "mlp")
    h = rd(O.gelu(lin_folded(x, p, "layer_norm2", pm, "fc1", rd)), "h")
    return x + lin(h, pm, "fc2", rd)


def self_attention(p, x, H, rd):
    if rd.fold:
        return self_attention_folded(p, x, H, rd)
    n = rd(ln(x, p, "layer_norm1"), "ln")
    x = x + attention(O._sub(p, "attention"), n, n, H, rd)
    return x + mlp(O._sub(p, "mlp"), rd(ln(x, p, "layer_norm2"), "ln"), rd)


def cross_attention(p, xq, xkv, H, resid, rd, mask=None):
    kvn = rd(ln(xkv, p, "layer_norm_kv"), "ln")
    qn = rd(ln(xq, p, "layer_norm_q"), "ln")
    a = attention(O._sub(p, "attention"), qn, kvn, H, rd, mask)
    x = xq + a if resid else a
    return x + mlp(O._sub(p, "mlp"), rd(ln(x, p, "layer_norm2"), "ln"), rd)


def encode_decode(p_enc, p_dec, x, qtab, rd, *, num_blocks, num_self_attends_per_block, num_cross_attend_heads=1,
                  num_self_attend_heads=8, encoder_query_residual=True, decoder_heads=1,
                  decoder_query_residual=False, final_project=True, input_mask=None, query_mask=None):
    x = x.astype(np.float64)
    B = x.shape[0]
    lat = p_enc["latent_pos_enc.pos_embs"].astype(np.float64)
    z = np.broadcast_to(lat[None], (B,) + lat.shape)
    m = None
    if input_mask is not None:
        m = O.make_cross_attention_mask(np.ones(z.shape[:2], bool), input_mask)
    z = cross_attention(O._sub(p_enc, "cross_attend"), z, x, num_cross_attend_heads, encoder_query_residual, rd, m)
    for _ in range(num_blocks):
        for l in range(num_self_attends_per_block):
            z = self_attention(O._sub(p_enc, f"self_attends.{l}"), z, num_self_attend_heads, rd)
    q = np.broadcast_to(qtab.astype(np.float64)[None], (B,) + qtab.shape)
    m = None
    if query_mask is not None:
        m = O.make_cross_attention_mask(query_mask, np.ones(z.shape[:2], bool))
    y = cross_attention(O._sub(p_dec, "decoding_cross_attn"), q, z, decoder_heads, decoder_query_residual, rd, m)
    if final_project:
        y = lin(rd(y, "y"), p_dec, "final_layer", rd)
    return y


if __name__ == "__main__":
    from cases import ENCDEC_CASES, gen_encdec_inputs, encdec_kwargs
    name = sys.argv[1] if len(sys.argv) > 1 else "encdec_tiny"
    cfg = dict(ENCDEC_CASES[name])
    if len(sys.argv) > 2:
        cfg["B"] = int(sys.argv[2])
    p_enc, p_dec, qtab, x, im, qm = gen_encdec_inputs(name, cfg, 21)
    kw = encdec_kwargs(cfg, im, qm)
    c64 = lambda d: {k: a.astype(np.float64) for k, a in d.items()}  # noqa: E731
    ref = O.encode_decode(c64(p_enc), c64(p_dec), x.astype(np.float64), qtab.astype(np.float64), **kw)
    allp = ["ln", "q", "k", "v", "p", "o", "h", "y", "w"]
    if "--fold" in sys.argv:
        for fold in (False, True):
            y = encode_decode(p_enc, p_dec, x, qtab, Rounder("f16", None, 1, fold=fold), **kw)
            print(f"{name} f16 w_passes=1 fold={fold}: relL2=%.2e max=%.2e" % O.rel_errors(y, ref), flush=True)
        sys.exit(0)
    for dt in ("f16", "bf16"):
        for wp in (2, 1):
            y = encode_decode(p_enc, p_dec, x, qtab, Rounder(dt, None, wp), **kw)
            print(f"{name} {dt} w_passes={wp}: relL2=%.2e max=%.2e" % O.rel_errors(y, ref))
    for pt in allp:
        y = encode_decode(p_enc, p_dec, x, qtab, Rounder("f16", [pt], 1), **kw)
        print(f"  only {pt:3s} rounded (f16, 1 pass): relL2=%.2e max=%.2e" % O.rel_errors(y, ref))
