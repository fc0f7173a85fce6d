"""Decoder query builders -- the [B, Q, Cq] array the decoder cross-attend consumes ("next" row of SURVEY.md 8f;
reference behaviour: perceiver_io/output_queries.py).  Plain PyTorch plumbing.

A query module answers two calls: ``n_query_channels()`` and
``forward(inputs, inputs_without_pos=None, subsampled_points=None) -> [B, ..., Cq]``.  Learned and Fourier tables are
batch-invariant stride-0 views (the decoder kernels project them once); sub-sampled Fourier queries are built on the
device of ``inputs``.

Shapes produced for the shipped models (B = batch):

=====================  ==========================================  =======================================
model                  query module                                decoder query array
=====================  ==========================================  =======================================
ImageNet classifier    TrainableQuery(1000, num_channels=1024)     [B, 1000, 1024]  (stride-0 over B)
language (MLM)         TrainableQuery(2048, num_channels=768)      [B, 2048, 768]   (stride-0 over B)
optical flow           FlowQuery(322, (368, 496))                  [B, 182528, 322] = the encoder input itself
multimodal autoencode  FourierQuery video (16,224,224), 32 bands   [B, n_sub, 195] -> padded to 1026 channels
                       FourierQuery audio (1920,), 192 bands       [B, n_sub, 385] -> padded to 1026
                       TrainableQuery label (1,), 1024 channels    [B, 1, 1024]    -> padded to 1026
=====================  ==========================================  =======================================

``PerceiverIO.decoder_query`` flattens everything between the batch and channel axes, appends the per-modality
padding embedding up to the common query width and concatenates modalities in sorted-name order.  Sub-sampling
(``subsampled_points``: flat indices into the output index space, one LongTensor per modality) is what lets the
multimodal model decode its 800k-point output in chunks against a single set of latents.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import position_encoding as PE
from .io_processors import unravel_index
from .position_encoding import PosEncodingType


def _has_encoding(kind) -> bool:
    return kind is not None and kind != PosEncodingType.NONE


class BasicQuery(nn.Module):
    """A position encoding over the output index space, optionally concatenated behind the (position-free)
    preprocessed inputs; with ``PosEncodingType.NONE`` the preprocessed inputs themselves are the queries."""

    def __init__(self, output_index_dims=None, concat_preprocessed_input: bool = False,
                 preprocessed_input_channels: int = None,
                 position_encoding_type: PosEncodingType = PosEncodingType.TRAINABLE, **position_encoding_kwargs):
        super().__init__()
        self._output_index_dim = output_index_dims
        self._concat_preprocessed_input = concat_preprocessed_input
        self._position_encoding_type = position_encoding_type
        width = 0
        self._position_encoding = None
        if _has_encoding(position_encoding_type):
            self._position_encoding = PE.build_position_encoding(position_encoding_type, index_dims=output_index_dims,
                                                                 **position_encoding_kwargs)
            width = self._position_encoding.n_output_channels()
        elif not concat_preprocessed_input:
            raise AssertionError("a query without a position encoding must concatenate the preprocessed input")
        if concat_preprocessed_input:
            if preprocessed_input_channels is None:
                raise AssertionError("preprocessed_input_channels is required with concat_preprocessed_input")
            width += preprocessed_input_channels
        self._n_query_channels = width

    def n_query_channels(self):
        return self._n_query_channels

    def _encode(self, batch: int, device, subsampled_points):
        """[B, Q, C] encoding of every output position, or of the flat indices in ``subsampled_points``
        (mapped to [-1, 1) as index / size, the convention of the reference's sub-sampling path)."""
        enc_mod = self._position_encoding
        if subsampled_points is None:
            return enc_mod(batch_size=batch, device=device).to(device)
        sizes = torch.tensor(self._output_index_dim, device=device)
        grid_pos = unravel_index(subsampled_points.to(device), self._output_index_dim)
        unit = -1 + 2 * grid_pos / sizes[None, :]
        unit = torch.broadcast_to(unit[None], (batch,) + tuple(unit.shape))
        enc = enc_mod(batch_size=batch, pos=unit, device=device)
        return enc.reshape(batch, -1, enc.shape[-1]).to(device)

    def forward(self, inputs, inputs_without_pos=None, subsampled_points=None):
        enc = None
        if self._position_encoding is not None:
            enc = self._encode(inputs.shape[0], inputs.device, subsampled_points)
        if not self._concat_preprocessed_input:
            return enc
        if inputs_without_pos is None:
            raise ValueError("Value is required for inputs_without_pos if concat_preprocessed_input is True")
        return inputs if enc is None else torch.cat([inputs_without_pos, enc], dim=-1)


class TrainableQuery(BasicQuery):
    """A learned table [output_index_dims, num_channels]."""

    def __init__(self, output_index_dims: int = None, concat_preprocessed_input: bool = False,
                 preprocessed_input_channels: int = None, num_channels: int = 128, init_scale: float = 0.02):
        table = {"num_channels": num_channels, "init_scale": init_scale}
        super().__init__(output_index_dims, concat_preprocessed_input, preprocessed_input_channels,
                         PosEncodingType.TRAINABLE, trainable_position_encoding_kwargs=table)


class FourierQuery(BasicQuery):
    """Fourier features of the output index grid."""

    def __init__(self, output_index_dims=None, concat_preprocessed_input: bool = False,
                 preprocessed_input_channels: int = None, num_bands=64, concat_pos=True, max_resolution=None,
                 sine_only=False):
        fourier = {"num_bands": num_bands, "concat_pos": concat_pos, "max_resolution": max_resolution,
                   "sine_only": sine_only}
        super().__init__(output_index_dims, concat_preprocessed_input, preprocessed_input_channels,
                         PosEncodingType.FOURIER, fourier_position_encoding_kwargs=fourier)


class FlowQuery(BasicQuery):
    """Dense per-pixel queries: the preprocessed inputs are the queries."""

    def __init__(self, preprocessed_input_channels: int, output_img_size, output_num_channels: int = 2):
        super().__init__(tuple(output_img_size) + (output_num_channels,), True, preprocessed_input_channels,
                         PosEncodingType.NONE)
