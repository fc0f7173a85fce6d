"""Opt-in CPU plumbing backend of the hot-path modules (BASELINE config 1: `example_img_classify.py` on a machine
without a GPU -- constructor surface, state_dict, preprocessing, the layer loop, postprocessing end to end).

This is NOT a fallback.  It runs only when the caller has asked for it (`perceiverio_pytorch_amd.set_backend("torch")`
or PIO_BACKEND=torch) AND the tensors are on the CPU; a CUDA tensor under this backend raises, a CPU tensor under the
default backend ("hip") raises, nothing ever switches by itself, and nothing here is imported from `oracle/` (the
checker stays test infrastructure).  Plain eager float32 torch ops written from the semantics of
perceiver_io/transformer_primitives.py:90-180, 212-216, 275-297, 371-406 and perceiver.py:94-107, 166-180; no
performance claim is attached to it (bench.py never uses it).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def attention(m, xq, xk, xv, attention_mask=None, attention_bias=None, return_matrix=False):
    """Attention.forward (:90-180): projections, per-head scaled dot product (the bias is added BEFORE the scale),
    mask to -1e30, softmax, P V, rows without an attendable key wiped to zero, final projection."""
    H = m._num_heads
    B, Tq, _ = xq.shape
    Tk = xk.shape[1]
    q = m.proj_q(xq).reshape(B, Tq, H, -1).permute(0, 2, 1, 3)
    k = m.proj_k(xk).reshape(B, Tk, H, -1).permute(0, 2, 1, 3)
    v = m.proj_v(xv).reshape(B, Tk, H, -1).permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2)
    if attention_bias is not None:
        s = s + attention_bias
    s = s * (1.0 / math.sqrt(q.shape[-1]))
    if attention_mask is not None:
        s = torch.where(attention_mask[:, None, :, :], s, torch.full_like(s, -1e30))
    p = m.dropout(torch.softmax(s, dim=-1))             # (:158-160; identity in eval mode / p = 0)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, Tq, -1)
    if attention_mask is not None:
        live = attention_mask.any(dim=2)
        o = torch.where(live[:, :, None], o, torch.zeros_like(o))
    out = m.final(o)
    return (p, out) if return_matrix else out


def mlp(m, x):
    return m.dropout(m.fc2(F.gelu(m.fc1(x))))           # exact (erf) GELU, :212-216


def self_attention(m, x, attention_mask=None, attention_bias=None, return_matrix=False):
    r = attention(m.attention, *(3 * (m.layer_norm1(x),)), attention_mask, attention_bias, return_matrix)
    probs, a = r if return_matrix else (None, r)
    x = x + m.dropout(a)                                # (:289-290)
    x = x + mlp(m.mlp, m.layer_norm2(x))
    return (probs, x) if return_matrix else x


def cross_attention(m, xq, xkv, attention_mask=None, attention_bias=None, return_matrix=False):
    kv = m.layer_norm_kv(xkv)
    r = attention(m.attention, m.layer_norm_q(xq), kv, kv, attention_mask, attention_bias, return_matrix)
    probs, a = r if return_matrix else (None, r)
    a = m.dropout(a)                                    # (:390)
    x = xq + a if m._use_query_residual else a
    x = x + mlp(m.mlp, m.layer_norm2(x))
    return (probs, x) if return_matrix else x


def _outer_mask(q_mask, kv_mask):
    return q_mask[:, :, None] & kv_mask[:, None, :]     # make_cross_attention_mask (:10-15)


def encoder(m, inputs, latents, input_mask=None):
    """PerceiverEncoder.forward (perceiver.py:98-107): only the cross-attend is masked; weights shared across blocks."""
    if isinstance(inputs, (tuple, list)):                # (features, batch-invariant table): concatenate here
        feats, table = inputs
        table = table if table.dim() == 3 else table[None]
        inputs = torch.cat([feats, table.expand(feats.shape[0], -1, -1)], dim=-1)
    mask = None
    if input_mask is not None:
        ones = torch.ones(latents.shape[:2], dtype=torch.bool, device=latents.device)
        mask = _outer_mask(ones, input_mask.bool())
    z = cross_attention(m.cross_attend, latents, inputs, mask)
    for _ in range(m._num_blocks):
        for sa in m.self_attends:
            z = self_attention(sa, z)
    return z


def decoder(m, query, latents, query_mask=None):
    """PerceiverDecoder.forward (perceiver.py:166-180)."""
    mask = None
    if query_mask is not None:
        ones = torch.ones(latents.shape[:2], dtype=torch.bool, device=latents.device)
        mask = _outer_mask(query_mask.bool(), ones)
    y = cross_attention(m.decoding_cross_attn, query, latents, mask)
    return m.final_layer(y) if m._final_project else y
