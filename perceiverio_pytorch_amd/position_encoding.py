"""Position encodings on the hot path.  Reference: perceiver_io/position_encoding.py:104-124.

Only ``TrainablePositionEncoding`` sits on the hot path (the latent array and learned output queries);
it is a stride-0 broadcast VIEW of its parameter, which the kernels exploit: batch-invariant rows are
normalised and projected once instead of B times.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class TrainablePositionEncoding(nn.Module):
    """Trainable position encoding: ``pos_embs`` [index_dim, num_channels]."""

    def __init__(self, index_dim, num_channels: int = 128, init_scale: float = 0.02):
        super().__init__()
        self.pos_embs = nn.Parameter(torch.zeros((index_dim, num_channels)))
        if num_channels > 0:
            nn.init.trunc_normal_(self.pos_embs, std=init_scale, a=-2.0, b=2.0)
        self._output_channels = num_channels

    def forward(self, batch_size, pos=None):
        del pos
        if batch_size is not None:
            pos_embs = torch.broadcast_to(self.pos_embs[None, :, :], (batch_size,) + self.pos_embs.shape)
        return pos_embs  # batch_size=None raises UnboundLocalError, like the reference (:119-121)

    def n_output_channels(self):
        return self._output_channels
