"""MI355X-native PerceiverEncoder / PerceiverDecoder (reference: perceiver_io/perceiver.py:13-180).

Same constructor arguments, sub-module names and forward signatures as the reference.  ``forward``
is ONE C call each (``pio_encoder_fwd`` / ``pio_decoder_fwd``): the whole cross-attend + num_blocks x
num_self_attends_per_block stack is enqueued on the current HIP stream without returning to Python.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from . import runtime as R
from .position_encoding import TrainablePositionEncoding
from .transformer_primitives import CrossAttention, SelfAttention, lecun_normal_, make_cross_attention_mask  # noqa: F401


class PerceiverEncoder(nn.Module):
    """Cross-attend inputs into the latent array, then num_blocks x (weight-shared) self-attend block."""

    def __init__(self, num_input_channels: int, num_self_attends_per_block: int = 6, num_blocks: int = 8,
                 num_latents: int = 512, num_latent_channels: int = 1024, qk_channels: int = None,
                 v_channels: int = None, num_cross_attend_heads: int = 1, num_self_attend_heads: int = 8,
                 cross_attend_widening_factor: int = 1, self_attend_widening_factor: int = 1,
                 dropout_prob: float = 0.0, latent_pos_enc_init_scale: float = 0.02,
                 cross_attention_shape_for_attn: str = "kv", use_query_residual: bool = True):
        super().__init__()
        if num_latent_channels % num_self_attend_heads != 0:
            raise ValueError(f"num_z_channels ({num_latent_channels}) must be divisible by"
                             f" num_self_attend_heads ({num_self_attend_heads}).")
        if num_latent_channels % num_cross_attend_heads != 0:
            raise ValueError(f"num_z_channels ({num_latent_channels}) must be divisible by"
                             f" num_cross_attend_heads ({num_cross_attend_heads}).")
        self._num_blocks = num_blocks
        self.latent_pos_enc = TrainablePositionEncoding(index_dim=num_latents, num_channels=num_latent_channels,
                                                        init_scale=latent_pos_enc_init_scale)
        self.cross_attend = CrossAttention(q_in_channels=num_latent_channels, kv_in_channels=num_input_channels,
                                           dropout_prob=dropout_prob, num_heads=num_cross_attend_heads,
                                           widening_factor=cross_attend_widening_factor,
                                           shape_for_attn=cross_attention_shape_for_attn, qk_channels=qk_channels,
                                           v_channels=v_channels, use_query_residual=use_query_residual)
        self.self_attends = nn.ModuleList()
        for _ in range(num_self_attends_per_block):
            self.self_attends.append(SelfAttention(in_channels=num_latent_channels, num_heads=num_self_attend_heads,
                                                   dropout_prob=dropout_prob, qk_channels=qk_channels,
                                                   v_channels=v_channels,
                                                   widening_factor=self_attend_widening_factor))

    def latents(self, inputs):
        return self.latent_pos_enc(batch_size=inputs.shape[0])

    def forward(self, inputs, latents, *, input_mask=None):
        R.require_device(inputs, "PerceiverEncoder.forward")
        if self.training and any(m.dropout.p > 0 for m in self.self_attends):
            raise NotImplementedError("the HIP path is forward/inference only")
        lib = L.lib()
        x, z0 = R.as_f32_3d(inputs), R.as_f32_3d(latents)
        B, M, _ = x.shape
        N, D = z0.shape[1], z0.shape[2]
        dev = x.device
        cross = self.cross_attend._desc()
        Lyr = len(self.self_attends)
        layers = (L.SelfAttention * max(Lyr, 1))()
        for i, sa in enumerate(self.self_attends):
            layers[i] = sa._desc()
        im, im_ptr = R.mask_u8(input_mask, (B, M), dev)
        out = torch.empty((B, N, D), dtype=torch.float32, device=dev)
        ws = R.workspace(dev, lib.pio_encoder_workspace_bytes(cross, layers, Lyr, B, M, N))
        L.check(lib.pio_encoder_fwd(cross, layers, Lyr, self._num_blocks, R.tensor3(x), R.tensor3(z0), im_ptr,
                                    out.data_ptr(), ws.data_ptr(), ws.numel(), R.stream_ptr(dev)),
                "pio_encoder_fwd")
        return out


class PerceiverDecoder(nn.Module):
    """Cross-attention decoder (queries attend to latents) with an optional final nn.Linear."""

    def __init__(self, query_channels: int, final_project_out_channels: int, num_latent_channels: int = 1024,
                 qk_channels: int = None, v_channels: int = None, use_query_residual: bool = False,
                 output_w_init: str = "lecun_normal", num_heads: int = 1, final_project: bool = True):
        super().__init__()
        self._output_num_channels = final_project_out_channels
        self._output_w_init = output_w_init
        self._use_query_residual = use_query_residual
        self._qk_channels = qk_channels
        self._v_channels = v_channels
        self._final_project = final_project
        self._num_heads = num_heads
        self.query_channels = query_channels
        self.decoding_cross_attn = CrossAttention(q_in_channels=query_channels, kv_in_channels=num_latent_channels,
                                                  dropout_prob=0.0, num_heads=self._num_heads, widening_factor=1,
                                                  shape_for_attn="kv", qk_channels=self._qk_channels,
                                                  v_channels=self._v_channels,
                                                  use_query_residual=self._use_query_residual)
        self._final_cache = None
        if self._final_project:
            self.final_layer = nn.Linear(query_channels, self._output_num_channels)
            if self._output_w_init == "lecun_normal":
                lecun_normal_(self.final_layer.weight)
            elif self._output_w_init == "zeros":
                nn.init.constant_(self.final_layer.weight, 0)
            else:
                raise ValueError(f"{self._output_w_init} not supported as output_w_init")
            nn.init.constant_(self.final_layer.bias, 0)

    def _apply(self, fn, *a, **k):
        self._final_cache = None
        return super()._apply(fn, *a, **k)

    def _final_desc(self):
        key = R.param_key(self.final_layer.weight, self.final_layer.bias)
        if self._final_cache is None or self._final_cache[0] != key:
            dtype, two, _split = R.policy_dtype()
            self._final_cache = (key, R.PackedLinear(self.final_layer.weight, self.final_layer.bias, 1, 1, dtype, two))
        return self._final_cache[1]

    def forward(self, query, latents, *, query_mask=None):
        R.require_device(query, "PerceiverDecoder.forward")
        lib = L.lib()
        q, z = R.as_f32_3d(query), R.as_f32_3d(latents)
        B, Q, _ = q.shape
        N = z.shape[1]
        dev = q.device
        cross = self.decoding_cross_attn._desc()
        fin = self._final_desc() if self._final_project else None
        out_ch = self._output_num_channels if self._final_project else self.query_channels
        qm, qm_ptr = R.mask_u8(query_mask, (B, Q), dev)
        out = torch.empty((B, Q, out_ch), dtype=torch.float32, device=dev)
        fin_ptr = C.byref(fin.desc) if fin is not None else None
        ws = R.workspace(dev, lib.pio_decoder_workspace_bytes(cross, fin_ptr, B, Q, N))
        L.check(lib.pio_decoder_fwd(cross, fin_ptr, out_ch, R.tensor3(q), R.tensor3(z), qm_ptr, out.data_ptr(),
                                    ws.data_ptr(), ws.numel(), R.stream_ptr(dev)), "pio_decoder_fwd")
        return out
