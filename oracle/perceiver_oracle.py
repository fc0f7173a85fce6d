"""CPU oracle for the PerceiverIO hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a from-scratch numpy restatement of the algorithm in the reference's
``perceiver_io/transformer_primitives.py`` and of the encoder/decoder drivers in
``perceiver_io/perceiver.py``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it -- as the checker, never as the thing
that is shipped or measured.  The product path (``perceiverio_pytorch_amd``) never
imports anything under ``oracle/``.

Parity pin: ``oracle/make_goldens.py`` imports the real reference from ``/root/reference``
(build container only), checks every function below against it in float64 (<=1e-11) and
float32 (<=3e-6), and freezes the reference's float32 outputs into ``tests/golden/*.npz``,
which ``tests/test_oracle_golden.py`` replays anywhere (the GPU box has no reference).  The reference itself ships no
tests/golden vectors, so the import-pin is the only pin.

Parameters are flat dicts keyed exactly like the reference ``state_dict`` leaves
(``attention.proj_q.weight`` ...), ``W`` is ``[out, in]`` (``y = x @ W.T + b``).
All functions compute in the dtype of their inputs (float32 or float64).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
from scipy.special import erf as _erf

Params = Dict[str, np.ndarray]

LN_EPS = 1e-5  # nn.LayerNorm default, transformer_primitives.py:270-271, 365-367


def _sub(p: Params, prefix: str) -> Params:
    """Sub-dict of ``p`` under ``prefix.`` with the prefix stripped."""
    pre = prefix + "."
    return {k[len(pre):]: v for k, v in p.items() if k.startswith(pre)}


# ---------------------------------------------------------------------------------------
# elementary ops (PyTorch ATen semantics restated)
# ---------------------------------------------------------------------------------------
def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray]) -> np.ndarray:
    """``F.linear``: y = x W^T + b.  (transformer_primitives.py:93-95, 110, 213-215)"""
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float = LN_EPS) -> np.ndarray:
    """``nn.LayerNorm`` over the last dim, biased variance. (transformer_primitives.py:282, 292, 379-380)"""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * g + b


def gelu(x: np.ndarray) -> np.ndarray:
    """Exact erf GELU (``F.gelu`` default approximate='none'). (transformer_primitives.py:214)"""
    return (0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))).astype(x.dtype)


def softmax_lastdim(s: np.ndarray) -> np.ndarray:
    """``F.softmax(dim=-1)``. (transformer_primitives.py:158)"""
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(axis=-1, keepdims=True)


# ---------------------------------------------------------------------------------------
# a1  make_cross_attention_mask  (transformer_primitives.py:10-15)
# ---------------------------------------------------------------------------------------
def make_cross_attention_mask(query_mask: np.ndarray, kv_mask: np.ndarray) -> np.ndarray:
    """mask[b,i,j] = query_mask[b,i] AND kv_mask[b,j]; bool in -> bool out."""
    assert query_mask.shape[0] == kv_mask.shape[0]
    return np.logical_and(query_mask[:, :, None].astype(bool), kv_mask[:, None, :].astype(bool))


# ---------------------------------------------------------------------------------------
# a3/a4  Attention.forward + attend  (transformer_primitives.py:90-180)
# ---------------------------------------------------------------------------------------
def attend(q: np.ndarray, k: np.ndarray, v: np.ndarray,
           attention_mask: Optional[np.ndarray] = None,
           attention_bias: Optional[np.ndarray] = None,
           return_matrix: bool = False):
    """q [B,Tq,H,dk], k [B,Tk,H,dk], v [B,Tk,H,dv], mask [B,Tq,Tk] bool -> [B,Tq,H*dv].

    Follows transformer_primitives.py:117-180 step by step: scores = q k^T (:138);
    bias is added BEFORE the 1/sqrt(dk) scale (:143-147); masked scores := -1e30 (:149-156);
    softmax (:158); P v (:163); head merge (:164-166); rows whose mask is all-false are
    forced to zero (:168-175).
    """
    B, Tq, H, dk = q.shape
    dv = v.shape[-1]
    qh = q.transpose(0, 2, 1, 3)
    kh = k.transpose(0, 2, 1, 3)
    vh = v.transpose(0, 2, 1, 3)
    s = qh @ kh.transpose(0, 1, 3, 2)                       # [B,H,Tq,Tk]
    if attention_bias is not None:
        s = s + attention_bias
    s = s * s.dtype.type(1.0 / math.sqrt(dk))
    if attention_mask is not None:
        large_k = s.dtype.type(1e4 if s.dtype == np.float16 else 1e30)
        s = np.where(attention_mask[:, None, :, :], s, -large_k)
    p = softmax_lastdim(s)
    o = p @ vh                                              # [B,H,Tq,dv]
    o = o.transpose(0, 2, 1, 3).reshape(B, Tq, H * dv)
    if attention_mask is not None:
        wipe = np.all(attention_mask == 0, axis=2, keepdims=True)   # [B,Tq,1]
        o = np.where(wipe, np.zeros_like(o), o)
    if return_matrix:
        return p, o
    return o


def attention(p: Params, inputs_q: np.ndarray, inputs_k: np.ndarray, inputs_v: np.ndarray,
              num_heads: int, attention_mask: Optional[np.ndarray] = None,
              attention_bias: Optional[np.ndarray] = None, return_matrix: bool = False):
    """``Attention.forward`` (transformer_primitives.py:90-115)."""
    q = linear(inputs_q, p["proj_q.weight"], p["proj_q.bias"])
    k = linear(inputs_k, p["proj_k.weight"], p["proj_k.bias"])
    v = linear(inputs_v, p["proj_v.weight"], p["proj_v.bias"])
    B, Tq, qk = q.shape
    Tk = k.shape[1]
    vc = v.shape[2]
    if qk % num_heads or vc % num_heads:
        raise ValueError("channels must be divisible by num_heads")  # :66-71
    q = q.reshape(B, Tq, num_heads, qk // num_heads)
    k = k.reshape(B, Tk, num_heads, qk // num_heads)
    v = v.reshape(B, Tk, num_heads, vc // num_heads)
    res = attend(q, k, v, attention_mask, attention_bias, return_matrix)
    if return_matrix:
        mat, res = res
    out = linear(res, p["final.weight"], p.get("final.bias"))
    if return_matrix:
        return mat, out
    return out


# ---------------------------------------------------------------------------------------
# a5  MLP  (transformer_primitives.py:183-216)
# ---------------------------------------------------------------------------------------
def mlp(p: Params, x: np.ndarray) -> np.ndarray:
    h = gelu(linear(x, p["fc1.weight"], p["fc1.bias"]))
    return linear(h, p["fc2.weight"], p["fc2.bias"])


# ---------------------------------------------------------------------------------------
# a6  SelfAttention  (transformer_primitives.py:275-297)
# ---------------------------------------------------------------------------------------
def self_attention(p: Params, inputs: np.ndarray, num_heads: int = 8,
                   attention_mask: Optional[np.ndarray] = None) -> np.ndarray:
    x = inputs
    n = layer_norm(inputs, p["layer_norm1.weight"], p["layer_norm1.bias"])
    a = attention(_sub(p, "attention"), n, n, n, num_heads, attention_mask)
    x = x + a
    x = x + mlp(_sub(p, "mlp"), layer_norm(x, p["layer_norm2.weight"], p["layer_norm2.bias"]))
    return x


# ---------------------------------------------------------------------------------------
# a7  CrossAttention  (transformer_primitives.py:371-406)
# ---------------------------------------------------------------------------------------
def cross_attention(p: Params, inputs_q: np.ndarray, inputs_kv: np.ndarray, num_heads: int,
                    use_query_residual: bool = True,
                    attention_mask: Optional[np.ndarray] = None) -> np.ndarray:
    kvn = layer_norm(inputs_kv, p["layer_norm_kv.weight"], p["layer_norm_kv.bias"])
    qn = layer_norm(inputs_q, p["layer_norm_q.weight"], p["layer_norm_q.bias"])
    a = attention(_sub(p, "attention"), qn, kvn, kvn, num_heads, attention_mask)
    x = inputs_q + a if use_query_residual else a
    x = x + mlp(_sub(p, "mlp"), layer_norm(x, p["layer_norm2.weight"], p["layer_norm2.bias"]))
    return x


# ---------------------------------------------------------------------------------------
# a8/a10  PerceiverEncoder  (perceiver.py:94-107; position_encoding.py:117-121)
# ---------------------------------------------------------------------------------------
def encoder(p: Params, inputs: np.ndarray, *, num_blocks: int, num_self_attends_per_block: int,
            num_cross_attend_heads: int = 1, num_self_attend_heads: int = 8,
            use_query_residual: bool = True, input_mask: Optional[np.ndarray] = None,
            return_intermediates: bool = False):
    """inputs [B,M,C] -> latents [B,N,D].  ``p`` holds the ``_encoder.`` sub-state-dict."""
    B = inputs.shape[0]
    lat = p["latent_pos_enc.pos_embs"]
    z = np.broadcast_to(lat[None], (B,) + lat.shape).astype(inputs.dtype)
    mask = None
    if input_mask is not None:
        mask = make_cross_attention_mask(np.ones(z.shape[:2], dtype=bool), input_mask)
    z = cross_attention(_sub(p, "cross_attend"), z, inputs, num_cross_attend_heads,
                        use_query_residual, mask)
    inter = [z]
    for _ in range(num_blocks):
        for l in range(num_self_attends_per_block):
            z = self_attention(_sub(p, f"self_attends.{l}"), z, num_self_attend_heads)
            inter.append(z)
    if return_intermediates:
        return z, inter
    return z


# ---------------------------------------------------------------------------------------
# a9  PerceiverDecoder  (perceiver.py:166-180)
# ---------------------------------------------------------------------------------------
def decoder(p: Params, query: np.ndarray, latents: np.ndarray, *, num_heads: int = 1,
            use_query_residual: bool = False, final_project: bool = True,
            query_mask: Optional[np.ndarray] = None) -> np.ndarray:
    mask = None
    if query_mask is not None:
        mask = make_cross_attention_mask(query_mask, np.ones(latents.shape[:2], dtype=bool))
    y = cross_attention(_sub(p, "decoding_cross_attn"), query, latents, num_heads,
                        use_query_residual, mask)
    if final_project:
        y = linear(y, p["final_layer.weight"], p["final_layer.bias"])
    return y


# ---------------------------------------------------------------------------------------
# Hot-path composition used by bench/smoke: encoder -> decoder with a learned query table
# (PerceiverIO.forward, perceiver.py:302-310, for the single-modality trainable-query case)
# ---------------------------------------------------------------------------------------
def encode_decode(p_enc: Params, p_dec: Params, inputs: np.ndarray, query_table: np.ndarray, *,
                  num_blocks: int, num_self_attends_per_block: int,
                  num_cross_attend_heads: int = 1, num_self_attend_heads: int = 8,
                  encoder_query_residual: bool = True, decoder_heads: int = 1,
                  decoder_query_residual: bool = False, final_project: bool = True,
                  input_mask=None, query_mask=None) -> np.ndarray:
    B = inputs.shape[0]
    z = encoder(p_enc, inputs, num_blocks=num_blocks,
                num_self_attends_per_block=num_self_attends_per_block,
                num_cross_attend_heads=num_cross_attend_heads,
                num_self_attend_heads=num_self_attend_heads,
                use_query_residual=encoder_query_residual, input_mask=input_mask)
    q = np.broadcast_to(query_table[None], (B,) + query_table.shape).astype(inputs.dtype)
    return decoder(p_dec, q, z, num_heads=decoder_heads, use_query_residual=decoder_query_residual,
                   final_project=final_project, query_mask=query_mask)


# ---------------------------------------------------------------------------------------
# Deterministic parameter generator shared by tests / bench / smoke (NOT reference code):
# weights ~ trunc-normal fan-in like the reference init, but biases and LayerNorm affine are
# perturbed away from the reference's (0 / 1,0) init so parity tests are not blind to them
# (SURVEY.md section 4).
# ---------------------------------------------------------------------------------------
def _rng_for(name: str, seed: int) -> np.random.Generator:
    import zlib
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def gen_tensor(name: str, shape, seed: int, kind: str) -> np.ndarray:
    rng = _rng_for(name, seed)
    if kind == "weight":            # [out, in]: variance 1/fan_in, clipped at 2 sigma
        fan_in = shape[-1]
        w = np.clip(rng.standard_normal(shape), -2.0, 2.0) / 0.87962566103423978
        return (w / math.sqrt(fan_in)).astype(np.float32)
    if kind == "bias":
        return (0.02 * rng.standard_normal(shape)).astype(np.float32)
    if kind == "ln_weight":
        return (1.0 + 0.02 * rng.standard_normal(shape)).astype(np.float32)
    if kind == "table":             # learned latents / queries (init_scale 0.02 in the reference;
        return (0.5 * rng.standard_normal(shape)).astype(np.float32)   # larger here to exercise LN)
    raise ValueError(kind)


def gen_attention(prefix: str, q_in: int, kv_in: int, qk: int, v: int, out: int, seed: int) -> Params:
    p = {}
    for nm, (o, i) in {"proj_q": (qk, q_in), "proj_k": (qk, kv_in), "proj_v": (v, kv_in),
                       "final": (out, v)}.items():
        p[f"{prefix}{nm}.weight"] = gen_tensor(f"{prefix}{nm}.weight", (o, i), seed, "weight")
        p[f"{prefix}{nm}.bias"] = gen_tensor(f"{prefix}{nm}.bias", (o,), seed, "bias")
    return p


def gen_mlp(prefix: str, cin: int, widening: int, cout: int, seed: int) -> Params:
    p = {}
    p[f"{prefix}fc1.weight"] = gen_tensor(f"{prefix}fc1.weight", (widening * cin, cin), seed, "weight")
    p[f"{prefix}fc1.bias"] = gen_tensor(f"{prefix}fc1.bias", (widening * cin,), seed, "bias")
    p[f"{prefix}fc2.weight"] = gen_tensor(f"{prefix}fc2.weight", (cout, widening * cin), seed, "weight")
    p[f"{prefix}fc2.bias"] = gen_tensor(f"{prefix}fc2.bias", (cout,), seed, "bias")
    return p


def gen_ln(prefix: str, c: int, seed: int) -> Params:
    return {f"{prefix}weight": gen_tensor(f"{prefix}weight", (c,), seed, "ln_weight"),
            f"{prefix}bias": gen_tensor(f"{prefix}bias", (c,), seed, "bias")}


def gen_self_attention(prefix: str, d: int, seed: int, widening: int = 1,
                       qk: Optional[int] = None, v: Optional[int] = None) -> Params:
    qk = qk or d
    v = v or qk
    p = {}
    p.update(gen_attention(prefix + "attention.", d, d, qk, v, v, seed))
    p.update(gen_mlp(prefix + "mlp.", v, widening, v, seed))
    p.update(gen_ln(prefix + "layer_norm1.", d, seed))
    p.update(gen_ln(prefix + "layer_norm2.", v, seed))
    return p


def gen_cross_attention(prefix: str, q_in: int, kv_in: int, seed: int, widening: int = 1,
                        shape_for_attn: str = "kv", qk: Optional[int] = None,
                        v: Optional[int] = None) -> Params:
    if qk is None:
        qk = kv_in if shape_for_attn == "kv" else q_in
    v = v or qk
    p = {}
    p.update(gen_attention(prefix + "attention.", q_in, kv_in, qk, v, q_in, seed))
    p.update(gen_mlp(prefix + "mlp.", q_in, widening, q_in, seed))
    p.update(gen_ln(prefix + "layer_norm_q.", q_in, seed))
    p.update(gen_ln(prefix + "layer_norm_kv.", kv_in, seed))
    p.update(gen_ln(prefix + "layer_norm2.", q_in, seed))
    return p


def gen_encoder(c_in: int, n_lat: int, d: int, layers: int, seed: int,
                qk: Optional[int] = None, v: Optional[int] = None) -> Params:
    p = {"latent_pos_enc.pos_embs": gen_tensor("latent_pos_enc.pos_embs", (n_lat, d), seed, "table")}
    p.update(gen_cross_attention("cross_attend.", d, c_in, seed, qk=qk, v=v))
    for l in range(layers):
        p.update(gen_self_attention(f"self_attends.{l}.", d, seed, qk=qk, v=v))
    return p


def gen_decoder(q_ch: int, d: int, out_ch: Optional[int], seed: int,
                qk: Optional[int] = None, v: Optional[int] = None) -> Params:
    p = gen_cross_attention("decoding_cross_attn.", q_ch, d, seed, qk=qk, v=v)
    if out_ch is not None:
        p["final_layer.weight"] = gen_tensor("final_layer.weight", (out_ch, q_ch), seed, "weight")
        p["final_layer.bias"] = gen_tensor("final_layer.bias", (out_ch,), seed, "bias")
    return p


def rel_errors(y: np.ndarray, ref: np.ndarray):
    """(relL2, max-abs / abs-max) -- the two parity figures SURVEY.md section 8d asks for."""
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = y - ref
    rl2 = float(np.sqrt((d * d).sum()) / max(np.sqrt((ref * ref).sum()), 1e-300))
    rmax = float(np.abs(d).max() / max(np.abs(ref).max(), 1e-300))
    return rl2, rmax
