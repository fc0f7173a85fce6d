"""Decoder query builders (reference: perceiver_io/output_queries.py) -- plain PyTorch plumbing that produces the
[B, Q, Cq] array the decoder cross-attend consumes; "next" row of SURVEY.md section 8f.

Protocol: ``forward(inputs, inputs_without_pos=None, subsampled_points=None) -> [B, ..., Cq]`` and
``n_query_channels()``.  Learned / Fourier tables are batch-invariant (stride-0 broadcast), which the decoder
kernels exploit; sub-sampled Fourier queries are computed on the device of ``inputs``."""
from __future__ import annotations

from typing import Sequence, Union

import torch
import torch.nn as nn

from . import position_encoding as PE
from .io_processors import unravel_index
from .position_encoding import PosEncodingType


class BasicQuery(nn.Module):
    """Position-encoding query, optionally concatenated with the preprocessed inputs (reference :11-81)."""

    def __init__(self, output_index_dims: Union[int, Sequence[int]] = None, concat_preprocessed_input: bool = False,
                 preprocessed_input_channels: int = None,
                 position_encoding_type: PosEncodingType = PosEncodingType.TRAINABLE, **position_encoding_kwargs):
        super().__init__()
        self._output_index_dim = output_index_dims
        self._concat_preprocessed_input = concat_preprocessed_input
        self._position_encoding_type = position_encoding_type
        if position_encoding_type != PosEncodingType.NONE and position_encoding_type is not None:
            self._position_encoding = PE.build_position_encoding(position_encoding_type,
                                                                 index_dims=output_index_dims,
                                                                 **position_encoding_kwargs)
            self._n_query_channels = self._position_encoding.n_output_channels()
        else:
            self._position_encoding = None
            assert concat_preprocessed_input is True, \
                "concat_preprocessed_input must be True if position_encoding_type is None"
            self._n_query_channels = 0
        if concat_preprocessed_input:
            assert preprocessed_input_channels is not None, \
                "preprocessed_input_channels must be set if concat_preprocessed_input is True"
            self._n_query_channels += preprocessed_input_channels

    def n_query_channels(self):
        return self._n_query_channels

    def forward(self, inputs, inputs_without_pos=None, subsampled_points=None):
        batch = inputs.shape[0]
        enc = None
        if self._position_encoding is not None:
            if subsampled_points is not None:
                # flat output indices -> coordinates -> [-1, 1) with the i/n convention of the reference (:58)
                dims = torch.tensor(self._output_index_dim, device=inputs.device)
                coords = unravel_index(subsampled_points.to(inputs.device), self._output_index_dim)
                pos = -1 + 2 * coords / dims[None, :]
                pos = torch.broadcast_to(pos[None], (batch,) + tuple(pos.shape))
                enc = self._position_encoding(batch_size=batch, pos=pos, device=inputs.device)
                enc = enc.reshape(batch, -1, enc.shape[-1])
            else:
                enc = self._position_encoding(batch_size=batch, device=inputs.device)
            enc = enc.to(inputs.device)
        if self._concat_preprocessed_input:
            if inputs_without_pos is None:
                raise ValueError("Value is required for inputs_without_pos if concat_preprocessed_input is True")
            enc = inputs if enc is None else torch.cat([inputs_without_pos, enc], dim=-1)
        return enc


class TrainableQuery(BasicQuery):
    """Learned query table (reference :84-102)."""

    def __init__(self, output_index_dims: int = None, concat_preprocessed_input: bool = False,
                 preprocessed_input_channels: int = None, num_channels: int = 128, init_scale: float = 0.02):
        super().__init__(output_index_dims=output_index_dims, concat_preprocessed_input=concat_preprocessed_input,
                         preprocessed_input_channels=preprocessed_input_channels,
                         position_encoding_type=PosEncodingType.TRAINABLE,
                         trainable_position_encoding_kwargs=dict(num_channels=num_channels, init_scale=init_scale))


class FourierQuery(BasicQuery):
    """Fourier-feature query over an output index grid (reference :105-126)."""

    def __init__(self, output_index_dims: Union[int, Sequence[int]] = None, concat_preprocessed_input: bool = False,
                 preprocessed_input_channels: int = None, num_bands=64, concat_pos=True, max_resolution=None,
                 sine_only=False):
        super().__init__(output_index_dims=output_index_dims, concat_preprocessed_input=concat_preprocessed_input,
                         preprocessed_input_channels=preprocessed_input_channels,
                         position_encoding_type=PosEncodingType.FOURIER,
                         fourier_position_encoding_kwargs=dict(num_bands=num_bands, max_resolution=max_resolution,
                                                               sine_only=sine_only, concat_pos=concat_pos))


class FlowQuery(BasicQuery):
    """The preprocessed inputs themselves are the queries (reference :129-139)."""

    def __init__(self, preprocessed_input_channels: int, output_img_size: Sequence[int], output_num_channels: int = 2):
        super().__init__(output_index_dims=tuple(output_img_size) + (output_num_channels,),
                         concat_preprocessed_input=True, preprocessed_input_channels=preprocessed_input_channels,
                         position_encoding_type=PosEncodingType.NONE)
