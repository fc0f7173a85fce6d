"""MI355X-native counterparts of the reference's ``perceiver_io/transformer_primitives.py``.

Same class names, constructor arguments, sub-module names (=> identical ``state_dict`` keys) and
``forward`` signatures as the reference; the arithmetic runs in ``libpio_hip.so`` (hand-written gfx950
kernels behind the C-ABI of ``include/pio_hip.h``).  Forward / inference only: there is no autograd
through the HIP path and no CPU fallback -- a CPU tensor raises.

Reference lines (under /root/reference/perceiver_io/transformer_primitives.py):
  make_cross_attention_mask 10-15 | Attention 18-180 | MLP 183-216 | SelfAttention 219-297 |
  CrossAttention 300-406
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import runtime as R


# --------------------------------------------------------------------------------------------------
# init helpers (the reference takes these from timm; only the distribution matters, and only at init)
# --------------------------------------------------------------------------------------------------
def variance_scaling_(tensor: torch.Tensor, scale: float = 1.0, mode: str = "fan_in",
                      distribution: str = "truncated_normal") -> torch.Tensor:
    fan_out, fan_in = tensor.shape[0], int(math.prod(tensor.shape[1:]))
    denom = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2}[mode]
    variance = scale / denom
    with torch.no_grad():
        if distribution == "truncated_normal":
            std = math.sqrt(variance) / 0.87962566103423978
            nn.init.trunc_normal_(tensor, std=std, a=-2 * std, b=2 * std)
        elif distribution == "normal":
            tensor.normal_(std=math.sqrt(variance))
        elif distribution == "uniform":
            bound = math.sqrt(3 * variance)
            tensor.uniform_(-bound, bound)
        else:
            raise ValueError(f"invalid distribution {distribution}")
    return tensor


def lecun_normal_(tensor: torch.Tensor) -> torch.Tensor:
    return variance_scaling_(tensor, mode="fan_in", distribution="truncated_normal")


def make_cross_attention_mask(query_mask: torch.Tensor, kv_mask: torch.Tensor) -> torch.Tensor:
    """mask[b,i,j] = query_mask[b,i] & kv_mask[b,j] (reference :10-15).  Pure index bookkeeping; the
    encoder/decoder never materialise it -- they hand the two vectors to the kernels instead."""
    batch_size, query_len = query_mask.shape
    _, key_len = kv_mask.shape
    mask = query_mask[:, :, None] & kv_mask[:, None, :] if query_mask.dtype == torch.bool and \
        kv_mask.dtype == torch.bool else torch.einsum("bi,bj->bij", (query_mask, kv_mask))
    assert mask.shape == (batch_size, query_len, key_len)
    return mask


# per-head (qk, v) widths, padded to 8, that the fused self-attention kernel covers with V read row-major (pio_flash.hip):
# for these q | k | v come out of ONE GEMM over a stacked weight image
FUSED_SELF_HEADS = {(128, 128), (64, 64), (32, 32), (32, 160)}


def _no_training_dropout(mod: nn.Module, *probs: float) -> None:
    if mod.training and any(p > 0.0 for p in probs):
        raise NotImplementedError("the HIP path is forward/inference only: call .eval() or use dropout_prob=0")


class _HipModule(nn.Module):
    """Shared packed-weight cache handling."""

    def __init__(self):
        super().__init__()
        self._pio_cache = None      # (key, descriptor, keep-alive list)
        self._pio_block_cache = None    # (key, [descriptor per block], keep-alive list): policy "fp16sd"

    def _cached(self, key, builder):
        c = self._pio_cache
        if c is None or c[0] != key:
            desc, keep = builder()
            self._pio_cache = c = (key, desc, keep)
        return c[1]

    def _apply(self, fn, *a, **k):       # .to()/.cuda()/.float() invalidate packed images
        self._pio_cache = self._pio_block_cache = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):   # load_state_dict copies through .data on some paths: repack
        self._pio_cache = self._pio_block_cache = None
        return super()._load_from_state_dict(*a, **k)


# ==================================================================================================
# Attention (reference :18-180)
# ==================================================================================================
class Attention(_HipModule):
    """Multi-headed {cross, self}-attention: proj_q/k/v -> softmax(q k^T / sqrt(d)) v -> final."""

    def __init__(self, q_in_channels: int, k_in_channels: int = None, v_in_channels: int = None,
                 num_heads: int = 8, init_scale: float = 1.0, with_final_bias: bool = True,
                 final_init_scale_multiplier: float = 1., dropout_prob: float = 0.0,
                 qk_out_channels: int = None, v_out_channels: int = None, output_channels: int = None):
        super().__init__()
        self._num_heads = num_heads
        final_init_scale = final_init_scale_multiplier * init_scale
        if qk_out_channels is None:
            qk_out_channels = q_in_channels
        if v_out_channels is None:
            v_out_channels = qk_out_channels
        if output_channels is None:
            output_channels = v_out_channels
        self._qk_channels_per_head = qk_out_channels // num_heads
        self._v_channels_per_head = v_out_channels // num_heads
        if qk_out_channels % num_heads != 0:
            raise ValueError(f"qk_out_channels ({qk_out_channels}) must be divisible by"
                             f" num_heads ({num_heads}).")
        if v_out_channels % num_heads != 0:
            raise ValueError(f"v_channels ({v_out_channels}) must be divisible by"
                             f" num_heads ({num_heads}).")
        self.proj_q = nn.Linear(q_in_channels, qk_out_channels, bias=True)
        self.proj_k = nn.Linear(k_in_channels, qk_out_channels, bias=True)
        self.proj_v = nn.Linear(v_in_channels, v_out_channels, bias=True)
        for lin in (self.proj_q, self.proj_k, self.proj_v):
            variance_scaling_(lin.weight, scale=init_scale, mode="fan_in", distribution="truncated_normal")
            nn.init.constant_(lin.bias, 0)
        self.dropout = nn.Dropout(dropout_prob)
        self.final = nn.Linear(v_out_channels, output_channels, bias=with_final_bias)
        variance_scaling_(self.final.weight, scale=final_init_scale, mode="fan_in",
                          distribution="truncated_normal")
        if self.final.bias is not None:
            nn.init.constant_(self.final.bias, 0)

    # ---- packed descriptor -------------------------------------------------------------------
    def _build_desc(self, ov=None):
        """`ov` (policy "fp16sd"): {linear: fp32 weight image to pack instead of its parameter} for one block."""
        dtype, wlevel, split = R.policy_dtype()
        H = self._num_heads
        two = wlevel >= 3          # proj_q / proj_k: split only under the all-split (x3) policies
        wt = (lambda lin: ov[lin]) if ov else (lambda lin: lin.weight)
        q = R.PackedLinear(wt(self.proj_q), self.proj_q.bias, H, 1, dtype, two)
        k = R.PackedLinear(wt(self.proj_k), self.proj_k.bias, H, 1, dtype, two)
        fine = R.policy_fine_split()
        v_two, o_two = wlevel >= 1 or "proj_v" in fine, wlevel >= 1 or "final" in fine
        v = R.PackedLinear(wt(self.proj_v), self.proj_v.bias, H, 1, dtype, v_two)
        o = R.PackedLinear(wt(self.final), self.final.bias, 1, H, dtype, o_two, k_channels=False)
        dk, dv = self._qk_channels_per_head, self._v_channels_per_head
        d = L.Attention(q.desc, k.desc, v.desc, o.desc, H, dk, dv, R.pad8(dk), R.pad8(dv),
                        self.proj_q.in_features, self.proj_k.in_features, self.proj_v.in_features,
                        self.final.out_features, dtype, (2 if R.policy_core_single() else 1) if split else 0)
        keep = [q, k, v, o]
        # K / V projection fold of a single-head cross-attend (pio_attention_t.kq / vo; SURVEY.md section 7 "legal algebraic
        # restructurings", reference :93-95,138,163): q (x Wk^T + bk)^T = (q Wk) x^T + const and
        # P (x Wv^T + bv) Wo^T + bo = (P x)(Wo Wv)^T + (Wo bv + bo).  The folded weights are formed in float64 and packed
        # like any other; the library takes them when keys outnumber query rows 4 : 1 (the encoders' cross-attends).
        # Offered under the single-weight policies only ("fp16", "fp16sd", "fp16x2af"): there it replaces single-sweep K / V
        # projections with exact (float64-folded, pair-packed) weights -- on the eight classifier goldens "fp16sd" moves from
        # 8.3e-4 / 1.72e-3 (fails the trained-like one) to 7.7e-4 / 9.0e-4, the multimodal model under "fp16/fp16x3f" from
        # 9.7e-4 / 1.06e-3 to 5.4e-4 / 5.6e-4, the flow forward from 5.66 to 5.42 ms.  Under the split-weight policies the
        # un-folded projections (K rounded AFTER an exact product) stay: 5.5e-4 / 5.8e-4 against 6.0e-4 / 8.4e-4 folded --
        # Q Wk concentrates on the outlier columns of a heavy-tailed Wk and multiplies the rounding of x there.
        # The library also takes it for the ImageNet DECODER (1024-wide head over 512 latents, batch-invariant queries:
        # xattn_tall_kernel reads the LayerNorm'd latents) -- under "fp16x2af" its K / V projections (2 x 34 GFLOP x 2
        # sweeps at B = 32) are gone: 6.5e-4 / 7.0e-4 -> 6.05e-4 / 5.6e-4 on the eight goldens, 16.49 -> 16.26 ms against
        # the split-operand ("x3f") decoder.
        kin = self.proj_k.in_features
        if H == 1 and kin == self.proj_v.in_features and dk == kin and dv == kin and (
                wlevel == 0 or os.environ.get("PIO_KV_FOLD_ANY") == "1"):   # (env: A/B experiments, tools/)
            with torch.no_grad():
                Wk, Wv, Wo = wt(self.proj_k).double(), wt(self.proj_v).double(), wt(self.final).double()
                bv = self.proj_v.bias.double() if self.proj_v.bias is not None else torch.zeros(kin, device=Wv.device)
                bo = self.final.bias.double() if self.final.bias is not None else 0.0
                kq = R.PackedLinear(Wk.t().float().contiguous(), None, 1, 1, dtype, True, k_channels=False)
                # (always a pair: a single image under "fp16x2af" saves 0.2 % of the step and costs 6.05e-4 -> 6.35e-4)
                vo = R.PackedLinear((Wo @ Wv).float().contiguous(), (Wo @ bv + bo).float().contiguous(), 1, 1, dtype,
                                    True, k_channels=False)
            d.kq, d.vo = kq.desc, vo.desc
            keep += [kq, vo]
        if self.proj_q.in_features == self.proj_k.in_features and not split:
            qk = R.PackedStack([(wt(self.proj_q), self.proj_q.bias), (wt(self.proj_k), self.proj_k.bias)],
                               H, dtype, two)
            d.qk = qk.desc
            keep.append(qk)
            # q | k | v in one image when the fused attention kernel can read V row-major (the head widths it covers).
            # Under "x2s" / "x2w" the V rows alone carry a lo image (pio_linear_t.lo_row0): the wide GEMM kernel runs
            # its second K sweep for those columns only -- the library takes such an image inside the LayerNorm fold.
            if (wlevel <= 2 and not two and self.proj_v.in_features == self.proj_q.in_features
                    and (R.pad8(dk), R.pad8(dv)) in FUSED_SELF_HEADS
                    and (not v_two or (2 * H * R.pad8(dk)) % 256 == 0)):
                qkv = R.PackedStack([(wt(self.proj_q), self.proj_q.bias), (wt(self.proj_k), self.proj_k.bias),
                                     (wt(self.proj_v), self.proj_v.bias)], H, dtype, [False, False, v_two])
                d.qkv = qkv.desc
                keep.append(qkv)
        return d, keep

    def _params(self):
        return (self.proj_q.weight, self.proj_q.bias, self.proj_k.weight, self.proj_k.bias, self.proj_v.weight,
                self.proj_v.bias, self.final.weight, self.final.bias)

    def _desc(self) -> L.Attention:
        return self._cached(R.param_key(*self._params()), self._build_desc)

    def forward(self, inputs_q, inputs_k, inputs_v, attention_mask=None, attention_bias=None,
                return_matrix=False):
        if R.cpu_plumbing(inputs_q, "Attention.forward"):
            from . import cpu_plumbing as CP
            return CP.attention(self, inputs_q, inputs_k, inputs_v, attention_mask, attention_bias, return_matrix)
        R.require_device(inputs_q, "Attention.forward")
        _no_training_dropout(self, self.dropout.p)
        lib = L.lib()
        xq, xk, xv = R.as_f32_3d(inputs_q), R.as_f32_3d(inputs_k), R.as_f32_3d(inputs_v)
        if inputs_v is inputs_k:
            xv = xk
        B, Tq, _ = xq.shape
        Tk = xk.shape[1]
        dev = xq.device
        d = self._desc()
        H = self._num_heads
        fm, fm_ptr = R.mask_u8(attention_mask, (B, Tq, Tk), dev)
        bias_t = None
        if attention_bias is not None:
            bias_t = torch.broadcast_to(attention_bias.to(device=dev, dtype=torch.float32),
                                        (B, H, Tq, Tk)).contiguous()
        out = torch.empty((B, Tq, self.final.out_features), dtype=torch.float32, device=dev)
        probs = torch.empty((B, H, Tq, Tk), dtype=torch.float32, device=dev) if return_matrix else None
        nbytes = lib.pio_attention_workspace_bytes(d, B, Tq, Tk)
        ws = R.workspace(dev, nbytes)
        tq, tk, tv = R.tensor3(xq), R.tensor3(xk), R.tensor3(xv)
        with R.on_device(dev):
            L.check(lib.pio_attention_fwd(d, tq, tk, tv, None, None, fm_ptr,
                                          bias_t.data_ptr() if bias_t is not None else None, out.data_ptr(),
                                          probs.data_ptr() if probs is not None else None, ws.data_ptr(),
                                          ws.numel(), R.stream_ptr(dev)), "pio_attention_fwd")
        out = R.forward_only(out, inputs_q, inputs_k, inputs_v, *self._params())
        if return_matrix:
            return probs, out
        return out


# ==================================================================================================
# MLP (reference :183-216)
# ==================================================================================================
class MLP(_HipModule):
    """Transformer-style dense module: fc2(gelu(fc1(x))), exact (erf) GELU."""

    def __init__(self, in_channels: int, out_channels: int = None, widening_factor: int = 4,
                 dropout_prob: float = 0.0, init_scale: float = 1.):
        super().__init__()
        out_channels = out_channels or in_channels
        self.fc1 = nn.Linear(in_channels, widening_factor * in_channels)
        variance_scaling_(self.fc1.weight, scale=init_scale, mode="fan_in", distribution="truncated_normal")
        nn.init.constant_(self.fc1.bias, 0)
        self.fc2 = nn.Linear(widening_factor * in_channels, out_channels)
        variance_scaling_(self.fc2.weight, scale=init_scale, mode="fan_in", distribution="truncated_normal")
        nn.init.constant_(self.fc2.bias, 0)
        self.dropout = nn.Dropout(dropout_prob)

    def _build_desc(self, ov=None):
        dtype, wlevel, split = R.policy_dtype()
        two = wlevel >= 2
        fine = R.policy_fine_split()
        wt = (lambda lin: ov[lin]) if ov else (lambda lin: lin.weight)
        f1 = R.PackedLinear(wt(self.fc1), self.fc1.bias, 1, 1, dtype, two or "fc1" in fine, n_channels=True)
        f2 = R.PackedLinear(wt(self.fc2), self.fc2.bias, 1, 1, dtype, two or "fc2" in fine)
        d = L.Mlp(f1.desc, f2.desc, self.fc1.in_features, self.fc1.out_features, self.fc2.out_features, dtype,
                  int(split))
        return d, [f1, f2]

    def _params(self):
        return (self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)

    def _desc(self) -> L.Mlp:
        return self._cached(R.param_key(*self._params()), self._build_desc)

    def forward(self, x):
        if R.cpu_plumbing(x, "MLP.forward"):
            from . import cpu_plumbing as CP
            return CP.mlp(self, x)
        R.require_device(x, "MLP.forward")
        _no_training_dropout(self, self.dropout.p)
        lib = L.lib()
        shape = x.shape
        x3 = R.as_f32_3d(x.reshape(-1, shape[-2], shape[-1]) if x.dim() != 3 else x)
        d = self._desc()
        dev = x3.device
        out = torch.empty(x3.shape[:2] + (self.fc2.out_features,), dtype=torch.float32, device=dev)
        rows = x3.shape[0] * x3.shape[1]
        ws = R.workspace(dev, lib.pio_mlp_workspace_bytes(d, rows))
        with R.on_device(dev):
            L.check(lib.pio_mlp_fwd(d, R.tensor3(x3), out.data_ptr(), ws.data_ptr(), ws.numel(), R.stream_ptr(dev)),
                    "pio_mlp_fwd")
        out = R.forward_only(out, x, *self._params())
        return out.reshape(shape[:-1] + (self.fc2.out_features,))


# ==================================================================================================
# SelfAttention (reference :219-297)
# ==================================================================================================
class SelfAttention(_HipModule):
    """Pre-LN block: x + Attn(LN1(x)); x + MLP(LN2(x))."""

    def __init__(self, in_channels: int, widening_factor: int = 4, dropout_prob: float = 0.0,
                 dropout_attn_prob: float = 0.0, num_heads: int = 8, att_init_scale: float = 1.0,
                 dense_init_scale: float = 1.0, qk_channels: int = None, v_channels: int = None):
        super().__init__()
        if qk_channels is None:
            qk_channels = in_channels
        if v_channels is None:
            v_channels = qk_channels
        self.mlp = MLP(in_channels=v_channels, widening_factor=widening_factor, dropout_prob=dropout_prob,
                       init_scale=dense_init_scale)
        self.attention = Attention(q_in_channels=in_channels, k_in_channels=in_channels,
                                   v_in_channels=in_channels, num_heads=num_heads, init_scale=att_init_scale,
                                   qk_out_channels=qk_channels, v_out_channels=v_channels,
                                   dropout_prob=dropout_attn_prob)
        self.layer_norm1 = nn.LayerNorm(in_channels)
        self.layer_norm2 = nn.LayerNorm(v_channels)
        self.dropout = nn.Dropout(dropout_prob)
        self._in_channels = in_channels
        self._v_channels = v_channels

    def _build_desc(self):
        keep = []
        a = self.attention._desc()
        m = self.mlp._desc()
        d = L.SelfAttention(R.layernorm_desc(self.layer_norm1, keep), R.layernorm_desc(self.layer_norm2, keep), a, m)
        self._build_ln_fold(d, a, keep)
        return d, keep + [self.attention._pio_cache, self.mlp._pio_cache]

    def _desc_blocks(self, nblk: int):
        """Policy "fp16sd": one descriptor per block of a weight-shared stack, block b packed from the b-th
        error-feedback rounding of every weight (runtime.feedback_images) -- of W for the plain images, of W * gamma
        for the images of the LayerNorm fold (the feedback runs over what the GEMM actually multiplies with)."""
        key = R.param_key(*self._params()) + (nblk,)
        c = self._pio_block_cache
        if c is None or c[0] != key:
            dtype, _wlevel, _split = R.policy_dtype()
            att, mlp = self.attention, self.mlp
            lins = (att.proj_q, att.proj_k, att.proj_v, att.final, mlp.fc1, mlp.fc2)
            descs, keep = [], []
            with torch.no_grad():
                g1, g2 = self.layer_norm1.weight.float(), self.layer_norm2.weight.float()
                plain = {lin: R.feedback_images(lin.weight, nblk, dtype) for lin in lins}
                fold = {lin: R.feedback_images(lin.weight.float() * g[None, :], nblk, dtype)
                        for lin, g in ((att.proj_q, g1), (att.proj_k, g1), (att.proj_v, g1), (mlp.fc1, g2))}
                for b in range(nblk):
                    a, ka = att._build_desc({lin: plain[lin][b] for lin in lins[:4]})
                    m, km = mlp._build_desc({lin: plain[lin][b] for lin in lins[4:]})
                    d = L.SelfAttention(R.layernorm_desc(self.layer_norm1, keep), R.layernorm_desc(self.layer_norm2, keep),
                                        a, m)
                    self._build_ln_fold(d, a, keep, {lin: imgs[b] for lin, imgs in fold.items()})
                    descs.append(d)
                    keep.extend([ka, km])
            self._pio_block_cache = c = (key, descs, keep)
        return c[1]

    def _build_ln_fold(self, d, a, keep, fold_ov=None):
        """LayerNorm folded into the consuming GEMMs (pio_ln_fold_t): LN(x) W^T + b = rstd (x W'^T - mean c) + b' with
        W' = W * gamma, c = rowsum(W' as packed), b' = W beta + b (reference :281-292 computes LN then Linear).
        Offered for blocks of 512 / 768 / 1024 / 1280 / 1536 channels (widening factor 1) under the single-sweep policies;
        the library decides per call."""
        dtype, wlevel, split = R.policy_dtype()
        att, mlp = self.attention, self.mlp
        C_ = self._in_channels
        if (wlevel > 2 or split or C_ % 256 or not 512 <= C_ <= 1536 or not a.qkv.w_hi
                or mlp.fc1.in_features != C_ or mlp.fc1.out_features != C_ or att.proj_q.in_features != C_):
            return
        with torch.no_grad():
            g1, b1 = self.layer_norm1.weight.float(), self.layer_norm1.bias.float()
            g2, b2 = self.layer_norm2.weight.float(), self.layer_norm2.bias.float()

            def folded(lin, g, b):
                w = lin.weight.float()
                bias = lin.bias.float() if lin.bias is not None else torch.zeros(w.shape[0], device=w.device)
                # (fold_ov: this block's image of W * gamma; the bias is that of the parameters)
                wg = fold_ov[lin] if fold_ov else w * g[None, :]
                return wg.contiguous(), (w @ b + bias).contiguous()

            H = att._num_heads
            # (the split levels of the un-folded descriptors: proj_v from "x2s" on, fc1 from "x2w" on)
            qkv = R.PackedStack([folded(att.proj_q, g1, b1), folded(att.proj_k, g1, b1), folded(att.proj_v, g1, b1)],
                                H, dtype, [False, False, wlevel >= 1 or "proj_v" in R.policy_fine_split()])
            fc1 = R.PackedLinear(*folded(mlp.fc1, g2, b2), 1, 1, dtype, wlevel >= 2, n_channels=True)
            # c[n] = sum_k of the packed weights the GEMM actually multiplies with: hi (+ lo where there is one)
            qkv_c = (qkv.hi.float() + (qkv.lo.float() if qkv.lo is not None else 0)).sum(1).contiguous()
            fc1_c = (fc1.hi.float() + (fc1.lo.float() if fc1.lo is not None else 0)).sum(1).contiguous()
        d.fold = L.LnFold(qkv.desc, qkv_c.data_ptr(), fc1.desc, fc1_c.data_ptr())
        keep.extend([qkv, fc1, qkv_c, fc1_c])

    def _params(self):
        return self.attention._params() + self.mlp._params() + (self.layer_norm1.weight, self.layer_norm1.bias,
                                                                 self.layer_norm2.weight, self.layer_norm2.bias)

    def _desc(self) -> L.SelfAttention:
        return self._cached(R.param_key(*self._params()), self._build_desc)

    def forward(self, inputs, *, attention_mask=None, attention_bias=None, return_matrix: bool = False):
        if R.cpu_plumbing(inputs, "SelfAttention.forward"):
            from . import cpu_plumbing as CP
            if self._v_channels != self._in_channels:
                raise RuntimeError(f"The size of tensor a ({self._in_channels}) must match the size of tensor b "
                                   f"({self._v_channels}) at non-singleton dimension 2")
            return CP.self_attention(self, inputs, attention_mask, attention_bias, return_matrix)
        R.require_device(inputs, "SelfAttention.forward")
        _no_training_dropout(self, self.dropout.p, self.attention.dropout.p, self.mlp.dropout.p)
        if self._v_channels != self._in_channels:
            # the residual add of the reference (:290) raises for mismatched widths
            raise RuntimeError(f"The size of tensor a ({self._in_channels}) must match the size of tensor b "
                               f"({self._v_channels}) at non-singleton dimension 2")
        lib = L.lib()
        x = R.as_f32_3d(inputs)
        B, N, D = x.shape
        dev = x.device
        H = self.attention._num_heads
        d = self._desc()
        fm, fm_ptr = R.mask_u8(attention_mask, (B, N, N), dev)
        bias_t = None
        if attention_bias is not None:
            bias_t = torch.broadcast_to(attention_bias.to(device=dev, dtype=torch.float32), (B, H, N, N)).contiguous()
        out = torch.empty((B, N, D), dtype=torch.float32, device=dev)
        probs = torch.empty((B, H, N, N), dtype=torch.float32, device=dev) if return_matrix else None
        ws = R.workspace(dev, lib.pio_self_attention_workspace_bytes(d, B, N))
        with R.on_device(dev):
            L.check(lib.pio_self_attention_fwd(d, R.tensor3(x), None, None, fm_ptr,
                                               bias_t.data_ptr() if bias_t is not None else None, out.data_ptr(),
                                               probs.data_ptr() if probs is not None else None, ws.data_ptr(),
                                               ws.numel(), R.stream_ptr(dev)), "pio_self_attention_fwd")
        out = R.forward_only(out, inputs, *self._params())
        if return_matrix:
            return probs, out
        return out


# ==================================================================================================
# CrossAttention (reference :300-406)
# ==================================================================================================
class CrossAttention(_HipModule):
    """Pre-LN cross-attention block with optional query residual, followed by an MLP."""

    def __init__(self, q_in_channels: int, kv_in_channels: int, widening_factor: int = 1,
                 dropout_prob: float = 0.0, dropout_attn_prob: float = 0.0, num_heads: int = 8,
                 attn_init_scale: float = 1.0, mlp_init_scale: float = 1.0, shape_for_attn: str = "kv",
                 use_query_residual: bool = True, qk_channels: int = None, v_channels: int = None):
        super().__init__()
        self._use_query_residual = use_query_residual
        output_channels = q_in_channels
        if qk_channels is None:
            if shape_for_attn == "q":
                qk_channels = q_in_channels
            elif shape_for_attn == "kv":
                qk_channels = kv_in_channels
            else:
                raise ValueError(f"Unknown value {shape_for_attn} for "
                                 "shape_for_attention.")
        if v_channels is None:
            v_channels = qk_channels
        self.attention = Attention(q_in_channels=q_in_channels, k_in_channels=kv_in_channels,
                                   v_in_channels=kv_in_channels, num_heads=num_heads, init_scale=attn_init_scale,
                                   dropout_prob=dropout_attn_prob, qk_out_channels=qk_channels,
                                   v_out_channels=v_channels, output_channels=output_channels)
        self.mlp = MLP(in_channels=output_channels, widening_factor=widening_factor, dropout_prob=dropout_prob,
                       init_scale=mlp_init_scale)
        self.layer_norm_q = nn.LayerNorm(q_in_channels)
        self.layer_norm_kv = nn.LayerNorm(kv_in_channels)
        self.layer_norm2 = nn.LayerNorm(output_channels)
        self.dropout = nn.Dropout(dropout_prob)

    def _build_desc(self):
        keep = []
        d = L.CrossAttention(R.layernorm_desc(self.layer_norm_q, keep), R.layernorm_desc(self.layer_norm_kv, keep),
                             R.layernorm_desc(self.layer_norm2, keep), self.attention._desc(), self.mlp._desc(),
                             1 if self._use_query_residual else 0)
        return d, keep + [self.attention._pio_cache, self.mlp._pio_cache]

    def _params(self):
        return self.attention._params() + self.mlp._params() + (
            self.layer_norm_q.weight, self.layer_norm_q.bias, self.layer_norm_kv.weight, self.layer_norm_kv.bias,
            self.layer_norm2.weight, self.layer_norm2.bias)

    def _desc(self) -> L.CrossAttention:
        return self._cached(R.param_key(*self._params()), self._build_desc)

    def forward(self, inputs_q, inputs_kv, *, attention_mask=None, attention_bias=None,
                return_matrix: bool = False):
        if R.cpu_plumbing(inputs_q, "CrossAttention.forward"):
            from . import cpu_plumbing as CP
            return CP.cross_attention(self, inputs_q, inputs_kv, attention_mask, attention_bias, return_matrix)
        R.require_device(inputs_q, "CrossAttention.forward")
        _no_training_dropout(self, self.dropout.p, self.attention.dropout.p, self.mlp.dropout.p)
        lib = L.lib()
        xq, xkv = R.as_f32_3d(inputs_q), R.as_f32_3d(inputs_kv)
        B, Tq, Cq = xq.shape
        Tk = xkv.shape[1]
        dev = xq.device
        H = self.attention._num_heads
        d = self._desc()
        fm, fm_ptr = R.mask_u8(attention_mask, (B, Tq, Tk), dev)
        bias_t = None
        if attention_bias is not None:
            bias_t = torch.broadcast_to(attention_bias.to(device=dev, dtype=torch.float32),
                                        (B, H, Tq, Tk)).contiguous()
        out = torch.empty((B, Tq, Cq), dtype=torch.float32, device=dev)
        probs = torch.empty((B, H, Tq, Tk), dtype=torch.float32, device=dev) if return_matrix else None
        ws = R.workspace(dev, lib.pio_cross_attention_workspace_bytes(d, B, Tq, Tk))
        with R.on_device(dev):
            L.check(lib.pio_cross_attention_fwd(d, R.tensor3(xq), R.tensor3(xkv), None, None, fm_ptr,
                                                bias_t.data_ptr() if bias_t is not None else None, out.data_ptr(),
                                                probs.data_ptr() if probs is not None else None, ws.data_ptr(),
                                                ws.numel(), R.stream_ptr(dev)), "pio_cross_attention_fwd")
        out = R.forward_only(out, inputs_q, inputs_kv, *self._params())
        if return_matrix:
            return probs, out
        return out
