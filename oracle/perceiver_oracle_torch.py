"""CPU oracle #2: the same hot path as ``perceiver_oracle.py`` restated on torch CPU ops -- TEST INFRASTRUCTURE.

Why a second restatement: ``bench.py``'s ``cpu_baseline`` leg should time what the reference itself executes on a
host, i.e. eager float32 ATen kernels (``F.linear`` / ``matmul`` / ``F.softmax`` / ``F.layer_norm`` / erf ``F.gelu``,
the call sites listed in SURVEY.md section 8c: transformer_primitives.py:93-95, 110, 138, 158, 163, 213-215, 270-271,
365-367) with torch's intra-op thread pool -- the numpy oracle goes through a different BLAS and is ~6x slower.
Written from scratch against the algorithm description; pinned by the SAME committed goldens
as the numpy oracle (tests/test_oracle_golden.py::test_torch_oracle_*), which are outputs of the real reference.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.
Parameters: flat dicts of numpy arrays or torch tensors keyed like the reference ``state_dict`` leaves.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5


def _t(x):
    if x is None or isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(x))


def _sub(p: Dict, prefix: str) -> Dict:
    pre = prefix + "."
    return {k[len(pre):]: v for k, v in p.items() if k.startswith(pre)}


def to_torch(p: Dict) -> Dict:
    return {k: _t(v) for k, v in p.items()}


def make_cross_attention_mask(query_mask, kv_mask):
    """transformer_primitives.py:10-15"""
    return _t(query_mask)[:, :, None].bool() & _t(kv_mask)[:, None, :].bool()


def attend(q, k, v, mask=None):
    """transformer_primitives.py:117-180: q [B,Tq,H,dk], k [B,Tk,H,dk], v [B,Tk,H,dv] -> [B,Tq,H*dv]."""
    B, Tq, H, dk = q.shape
    dv = v.shape[-1]
    s = torch.matmul(q.permute(0, 2, 1, 3), k.permute(0, 2, 3, 1))          # :134-138
    s = s * (1.0 / math.sqrt(dk))                                             # :147
    if mask is not None:
        s = torch.where(mask[:, None, :, :], s, torch.full((), -1e30, dtype=s.dtype))   # :149-156
    p = F.softmax(s, dim=-1)                                                  # :158
    o = torch.matmul(p, v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, Tq, H * dv)   # :163-166
    if mask is not None:                                                      # :168-175
        wipe = ~mask.any(dim=2, keepdim=True)
        o = torch.where(wipe, torch.zeros((), dtype=o.dtype), o)
    return o


def attention(p, xq, xk, xv, num_heads, mask=None):
    """transformer_primitives.py:90-115"""
    q = F.linear(xq, p["proj_q.weight"], p["proj_q.bias"])
    k = F.linear(xk, p["proj_k.weight"], p["proj_k.bias"])
    v = F.linear(xv, p["proj_v.weight"], p["proj_v.bias"])
    B, Tq, qk = q.shape
    Tk, vc = k.shape[1], v.shape[2]
    o = attend(q.reshape(B, Tq, num_heads, qk // num_heads), k.reshape(B, Tk, num_heads, qk // num_heads),
               v.reshape(B, Tk, num_heads, vc // num_heads), mask)
    return F.linear(o, p["final.weight"], p.get("final.bias"))


def mlp(p, x):
    """transformer_primitives.py:212-216 (exact erf GELU)"""
    return F.linear(F.gelu(F.linear(x, p["fc1.weight"], p["fc1.bias"])), p["fc2.weight"], p["fc2.bias"])


def _ln(x, p, name):
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], LN_EPS)


def self_attention(p, x, num_heads=8, mask=None):
    """transformer_primitives.py:275-297"""
    n = _ln(x, p, "layer_norm1")
    x = x + attention(_sub(p, "attention"), n, n, n, num_heads, mask)
    return x + mlp(_sub(p, "mlp"), _ln(x, p, "layer_norm2"))


def cross_attention(p, xq, xkv, num_heads, use_query_residual=True, mask=None):
    """transformer_primitives.py:371-406"""
    kvn = _ln(xkv, p, "layer_norm_kv")
    qn = _ln(xq, p, "layer_norm_q")
    a = attention(_sub(p, "attention"), qn, kvn, kvn, num_heads, mask)
    x = xq + a if use_query_residual else a
    return x + mlp(_sub(p, "mlp"), _ln(x, p, "layer_norm2"))


def encoder(p, inputs, *, num_blocks, num_self_attends_per_block, num_cross_attend_heads=1,
            num_self_attend_heads=8, use_query_residual=True, input_mask=None):
    """perceiver.py:94-107"""
    B = inputs.shape[0]
    lat = p["latent_pos_enc.pos_embs"]
    z = torch.broadcast_to(lat[None], (B,) + tuple(lat.shape))
    mask = None
    if input_mask is not None:
        mask = make_cross_attention_mask(torch.ones(z.shape[:2], dtype=torch.bool), input_mask)
    z = cross_attention(_sub(p, "cross_attend"), z, inputs, num_cross_attend_heads, use_query_residual, mask)
    layers = [_sub(p, f"self_attends.{l}") for l in range(num_self_attends_per_block)]
    for _ in range(num_blocks):
        for lp in layers:
            z = self_attention(lp, z, num_self_attend_heads)
    return z


def decoder(p, query, latents, *, num_heads=1, use_query_residual=False, final_project=True, query_mask=None):
    """perceiver.py:166-180"""
    mask = None
    if query_mask is not None:
        mask = make_cross_attention_mask(query_mask, torch.ones(latents.shape[:2], dtype=torch.bool))
    y = cross_attention(_sub(p, "decoding_cross_attn"), query, latents, num_heads, use_query_residual, mask)
    if final_project:
        y = F.linear(y, p["final_layer.weight"], p["final_layer.bias"])
    return y


def encode_decode(p_enc, p_dec, inputs, query_table, *, num_blocks, num_self_attends_per_block,
                  num_cross_attend_heads=1, num_self_attend_heads=8, encoder_query_residual=True, decoder_heads=1,
                  decoder_query_residual=False, final_project=True, input_mask=None, query_mask=None):
    """Same signature as perceiver_oracle.encode_decode; numpy in, numpy out."""
    p_enc, p_dec = to_torch(p_enc), to_torch(p_dec)
    x, qt = _t(inputs), _t(query_table)
    with torch.inference_mode():
        z = encoder(p_enc, x, num_blocks=num_blocks, num_self_attends_per_block=num_self_attends_per_block,
                    num_cross_attend_heads=num_cross_attend_heads, num_self_attend_heads=num_self_attend_heads,
                    use_query_residual=encoder_query_residual, input_mask=input_mask)
        q = torch.broadcast_to(qt[None], (x.shape[0],) + tuple(qt.shape))
        y = decoder(p_dec, q, z, num_heads=decoder_heads, use_query_residual=decoder_query_residual,
                    final_project=final_project, query_mask=query_mask)
    return y.numpy()
