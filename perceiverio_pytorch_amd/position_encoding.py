"""Position encodings (reference: perceiver_io/position_encoding.py).

Only ``TrainablePositionEncoding`` sits on the hot path (the latent array and learned output queries): it is a
stride-0 broadcast VIEW of its parameter, which the kernels exploit -- batch-invariant rows are normalised and
projected once instead of B times.  The Fourier encodings are batch-invariant tables built with stock torch ops
in float32 (reference :19-89, 151-187); they feed the preprocessors / output queries ("next" rows of SURVEY.md 8f).
"""
from __future__ import annotations

import math
from enum import Enum
from typing import Optional, Sequence

import torch
import torch.nn as nn


class PosEncodingType(Enum):
    FOURIER = 1
    TRAINABLE = 2
    NONE = 3


# ---------------------------------------------------------------------------------------------------
# functional pieces
# ---------------------------------------------------------------------------------------------------
def build_linear_positions(index_dims: Sequence[int], output_range=(-1.0, 1.0)) -> torch.Tensor:
    """Regular grid over ``index_dims`` with every axis spanning ``output_range`` inclusively:
    shape [*index_dims, len(index_dims)], float32 (reference :70-89)."""
    lo, hi = output_range
    axes = [torch.linspace(lo, hi, steps=int(n), dtype=torch.float32) for n in index_dims]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1)


def generate_fourier_features(pos: torch.Tensor, num_bands: int, max_resolution=(224, 224), concat_pos: bool = True,
                              sine_only: bool = False) -> torch.Tensor:
    """Fourier features of n points in d dimensions (``pos`` [n, d]) with num_bands linearly spaced frequencies
    from 1 to the Nyquist frequency res/2 per dimension.  Channel order (reference :19-67):
    [pos (d) | sin(pi f x) grouped by dimension (d*K) | cos(pi f x) grouped by dimension (d*K)]."""
    bands = torch.stack([torch.linspace(1.0, res / 2, steps=num_bands) for res in max_resolution], dim=0)  # [d, K]
    phase = (pos[:, :, None] * bands[None].to(pos.device)).reshape(pos.shape[0], -1) * math.pi
    feats = [torch.sin(phase)] if sine_only else [torch.sin(phase), torch.cos(phase)]
    if concat_pos:
        feats = [pos] + feats
    return torch.cat(feats, dim=-1)


# ---------------------------------------------------------------------------------------------------
# modules
# ---------------------------------------------------------------------------------------------------
class TrainablePositionEncoding(nn.Module):
    """Trainable position encoding: ``pos_embs`` [index_dim, num_channels] (reference :104-124)."""

    def __init__(self, index_dim, num_channels: int = 128, init_scale: float = 0.02):
        super().__init__()
        self.pos_embs = nn.Parameter(torch.zeros((index_dim, num_channels)))
        if num_channels > 0:
            nn.init.trunc_normal_(self.pos_embs, std=init_scale, a=-2.0, b=2.0)
        self._output_channels = num_channels

    def forward(self, batch_size, pos=None, device=None):
        del pos, device
        if batch_size is not None:
            pos_embs = torch.broadcast_to(self.pos_embs[None, :, :], (batch_size,) + self.pos_embs.shape)
        return pos_embs  # batch_size=None raises UnboundLocalError, like the reference (:119-121)

    def n_output_channels(self):
        return self._output_channels


class FourierPositionEncoding(nn.Module):
    """Fourier (sinusoidal) encoding of a regular grid or of caller-supplied positions (reference :151-187).
    Batch-invariant: only ``pos[0]`` is featurised, then broadcast."""

    def __init__(self, index_dims, num_bands, concat_pos=True, max_resolution=None, sine_only=False):
        super().__init__()
        self._index_dims = index_dims
        self._num_bands = num_bands
        self._concat_pos = concat_pos
        self._sine_only = sine_only
        self._max_resolution = max_resolution or index_dims
        d = len(self._max_resolution)
        self._output_channels = d * num_bands * (1 if sine_only else 2) + (d if concat_pos else 0)
        self._grid_tables = {}   # device -> [prod(index_dims), C] table of the regular grid (input-independent)

    def forward(self, batch_size, pos=None, device=None):
        """``device`` (optional) asks for the result on that device; the regular-grid table is then built once
        (on the host, with the reference's exact float32 arithmetic) and cached there instead of being recomputed
        and copied on every forward as the reference does."""
        if pos is None:
            key = str(device)
            enc = self._grid_tables.get(key)
            if enc is None:
                grid = build_linear_positions(self._index_dims)
                enc = generate_fourier_features(grid.reshape(-1, grid.shape[-1]), num_bands=self._num_bands,
                                                max_resolution=self._max_resolution, concat_pos=self._concat_pos,
                                                sine_only=self._sine_only)
                if device is not None:
                    enc = enc.to(device)
                self._grid_tables[key] = enc
        else:
            assert pos.shape[-1] == len(self._index_dims)
            enc = generate_fourier_features(pos[0], num_bands=self._num_bands, max_resolution=self._max_resolution,
                                            concat_pos=self._concat_pos, sine_only=self._sine_only)
        if batch_size is not None:
            enc = torch.broadcast_to(enc[None], (batch_size,) + enc.shape)
        return enc

    def n_output_channels(self):
        return self._output_channels


class PositionEncodingProjector(nn.Module):
    """A position encoding followed by a learned linear map to ``output_size`` channels (reference :190-207)."""

    def __init__(self, input_size, output_size, base_position_encoding):
        super().__init__()
        self._base_position_encoding = base_position_encoding
        self._projector = nn.Linear(input_size, output_size)
        self._output_channels = output_size
        from .transformer_primitives import lecun_normal_
        lecun_normal_(self._projector.weight)
        nn.init.constant_(self._projector.bias, 0)

    def forward(self, batch_size, pos=None, device=None):
        return self._projector(self._base_position_encoding(batch_size, pos, device=device))

    def n_output_channels(self):
        return self._output_channels


def build_position_encoding(position_encoding_type, index_dims, project_pos_dim=-1,
                            trainable_position_encoding_kwargs=None, fourier_position_encoding_kwargs=None):
    """Factory used by preprocessors and queries (reference :210-239)."""
    if position_encoding_type == PosEncodingType.TRAINABLE:
        assert trainable_position_encoding_kwargs is not None
        enc = TrainablePositionEncoding(index_dim=int(math.prod(index_dims)) if not isinstance(index_dims, int)
                                        else index_dims, **trainable_position_encoding_kwargs)
    elif position_encoding_type == PosEncodingType.FOURIER:
        assert fourier_position_encoding_kwargs is not None
        enc = FourierPositionEncoding(index_dims=index_dims, **fourier_position_encoding_kwargs)
    else:
        raise ValueError(f"Unknown position encoding: {position_encoding_type}.")
    if project_pos_dim > 0:
        enc = PositionEncodingProjector(input_size=enc.n_output_channels(), output_size=project_pos_dim,
                                        base_position_encoding=enc)
    return enc
