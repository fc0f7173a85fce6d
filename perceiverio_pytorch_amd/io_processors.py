"""Input preprocessors and output postprocessors around the hot path -- plain PyTorch (MIOpen / rocBLAS through
ATen) plumbing, "next" rows of SURVEY.md section 8f.  They produce the [B, M, C] array the encoder kernels consume
and turn decoder outputs into task outputs.  Same class names, constructor arguments, parameter names and
(inputs_with_pos, inputs_without_pos) protocol as the reference's
perceiver_io/io_processors/{preprocessors,postprocessors,processor_utils}.py, so reference checkpoints load.
"""
from __future__ import annotations

import math
from typing import Mapping, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import position_encoding as PE
from .position_encoding import PosEncodingType, TrainablePositionEncoding
from .transformer_primitives import lecun_normal_


# ---------------------------------------------------------------------------------------------------
# shape helpers
# ---------------------------------------------------------------------------------------------------
def same_padding(input_size: Sequence[int], kernel_size, stride=1, dims: int = 2):
    """TensorFlow-"SAME" padding for the last ``dims`` axes as an F.pad list (last axis first); when the total is
    odd the extra cell goes to the right / bottom (utils/utils.py:72-101)."""
    ks = [kernel_size] * dims if isinstance(kernel_size, int) else list(kernel_size)
    st = [stride] * dims if isinstance(stride, int) else list(stride)
    lead = len(input_size) - dims
    pads = []
    for d in reversed(range(dims)):
        rem = input_size[lead + d] % st[d]
        total = ks[d] - (st[d] if rem == 0 else rem)
        pads += [total // 2, total - total // 2]
    return pads


def unravel_index(indices: torch.Tensor, shape) -> torch.Tensor:
    """Flat indices -> coordinates in ``shape`` ([..., len(shape)], row-major), vectorised (utils/utils.py:41-69
    does it with a Python loop over dimensions on the CPU; the arithmetic is identical)."""
    total = int(math.prod(shape))
    idx = indices.to(torch.long) % total
    coords = []
    for dim in reversed(tuple(shape)):
        coords.append(idx % dim)
        idx = idx // dim
    return torch.stack(coords[::-1], dim=-1)


def space_to_depth(frames: torch.Tensor, temporal_block_size: int = 1, spatial_block_size: int = 1) -> torch.Tensor:
    """Channels-last block stacking: [B,(T),H,W,C] -> [B,(T/dt),H/dh,W/dw, dt*dh*dw*C] with (dt, dh, dw, c) order
    inside the new channel axis (processor_utils.py:21-38)."""
    s = spatial_block_size
    if frames.dim() == 4:
        b, h, w, c = frames.shape
        x = frames.reshape(b, h // s, s, w // s, s, c).permute(0, 1, 3, 2, 4, 5)
        return x.reshape(b, h // s, w // s, s * s * c)
    if frames.dim() == 5:
        t_ = temporal_block_size
        b, t, h, w, c = frames.shape
        x = frames.reshape(b, t // t_, t_, h // s, s, w // s, s, c).permute(0, 1, 3, 5, 2, 4, 6, 7)
        return x.reshape(b, t // t_, h // s, w // s, t_ * s * s * c)
    raise ValueError("Frames should be of rank 4 (batch, height, width, channels)"
                     " or rank 5 (batch, time, height, width, channels)")


def reverse_space_to_depth(frames: torch.Tensor, temporal_block_size: int = 1, spatial_block_size: int = 1):
    """Inverse of :func:`space_to_depth` (processor_utils.py:41-56)."""
    s = spatial_block_size
    if frames.dim() == 4:
        b, h, w, cc = frames.shape
        c = cc // (s * s)
        return frames.reshape(b, h, w, s, s, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h * s, w * s, c)
    if frames.dim() == 5:
        t_ = temporal_block_size
        b, t, h, w, cc = frames.shape
        c = cc // (t_ * s * s)
        x = frames.reshape(b, t, h, w, t_, s, s, c).permute(0, 1, 4, 2, 5, 3, 6, 7)
        return x.reshape(b, t * t_, h * s, w * s, c)
    raise ValueError("Frames should be of rank 4 (batch, height, width, channels)"
                     " or rank 5 (batch, time, height, width, channels)")


def extract_patches(images: torch.Tensor, size, stride=1, dilation=1, padding: str = "VALID") -> torch.Tensor:
    """[B,C,H,W] -> [B, rows, cols, ph*pw*C] sliding patches, channel fastest inside a patch position
    (processor_utils.py:59-95)."""
    if padding != "VALID":
        raise ValueError(f"Only valid padding is supported. Got {padding}")
    if images.ndim != 4:
        raise ValueError(f"Rank of images must be 4 (got tensor of shape {images.shape})")
    n, c, h, w = images.shape
    ph, pw = size
    st = (stride, stride) if isinstance(stride, int) else tuple(stride)
    dl = (dilation, dilation) if isinstance(dilation, int) else tuple(dilation)
    oh = (h - dl[0] * (ph - 1) - 1) // st[0] + 1
    ow = (w - dl[1] * (pw - 1) - 1) // st[1] + 1
    cols = F.unfold(images, (ph, pw), dilation=dl, padding=0, stride=st)          # [n, c*ph*pw, oh*ow]
    cols = cols.reshape(n, c, ph, pw, oh, ow).permute(0, 4, 5, 2, 3, 1)
    return cols.reshape(n, oh, ow, ph * pw * c)


def patches_for_flow(inputs: torch.Tensor) -> torch.Tensor:
    """[N,2,C,H,W] frame pairs -> [N,2,H,W,9*C]: the zero-padded 3x3 neighbourhood of every pixel
    (processor_utils.py:98-116)."""
    n, t = inputs.shape[:2]
    x = F.pad(inputs.reshape((n * t,) + inputs.shape[2:]), [1, 1, 1, 1])
    p = extract_patches(x, size=[3, 3])
    return p.reshape((n, t) + p.shape[1:])


class Conv2DDownsample(nn.Module):
    """num_layers x [7x7 stride-2 conv (TF SAME padding) -> BatchNorm -> ReLU -> 3x3 stride-2 max-pool (SAME)]:
    4x spatial downsampling per layer (processor_utils.py:124-180).  Parameters: convs.<l>, norms.<l>."""

    def __init__(self, num_layers: int = 1, in_channels: int = 3, num_channels: int = 64, use_batchnorm: bool = True):
        super().__init__()
        self._num_layers = num_layers
        self.norms = nn.ModuleList() if use_batchnorm else None
        self.convs = nn.ModuleList()
        for _ in range(num_layers):
            conv = nn.Conv2d(in_channels, num_channels, kernel_size=7, stride=2, bias=False)
            nn.init.trunc_normal_(conv.weight, mean=0.0, std=0.01, a=-2.0, b=2.0)
            self.convs.append(conv)
            in_channels = num_channels
            if use_batchnorm:
                self.norms.append(nn.BatchNorm2d(num_channels))

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        x = inputs
        for layer, conv in enumerate(self.convs):
            x = conv(F.pad(x, same_padding(x.shape[1:], conv.kernel_size, conv.stride)))
            if self.norms is not None:
                x = self.norms[layer](x)
            x = F.relu(x)
            x = F.max_pool2d(F.pad(x, same_padding(x.shape[1:], 3, 2)), kernel_size=3, stride=2)
        return x

    def forward_tokens(self, inputs: torch.Tensor):
        """The same network with its LAST layer's BatchNorm -> ReLU -> max-pool -> channels-last reshape done by ONE
        HIP kernel (pio_bn_relu_maxpool_tokens): returns the token array [B, OH*OW, C] directly, or None when this
        configuration is not covered (CPU tensor, training-mode BatchNorm)."""
        if not inputs.is_cuda or inputs.dim() != 4 or (self.norms is not None and self.training):
            return None
        from . import _lib as L
        from . import runtime as R
        x = inputs
        last = len(self.convs) - 1
        for layer, conv in enumerate(self.convs):
            x = conv(F.pad(x, same_padding(x.shape[1:], conv.kernel_size, conv.stride)))
            if layer < last:
                if self.norms is not None:
                    x = self.norms[layer](x)
                x = F.max_pool2d(F.pad(F.relu(x), same_padding(x.shape[1:], 3, 2)), kernel_size=3, stride=2)
        x = x.float().contiguous()
        b, c, h, w = x.shape
        if self.norms is not None:
            bn = self.norms[last]
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float()
            shift = (bn.bias - bn.running_mean * scale).float()
        else:
            scale = torch.ones(c, device=x.device)
            shift = torch.zeros(c, device=x.device)
        pads = same_padding(x.shape[1:], 3, 2)          # [left, right, top, bottom]
        y = torch.empty((b, ((h + 1) // 2) * ((w + 1) // 2), c), dtype=torch.float32, device=x.device)
        with R.on_device(x.device):
            L.check(L.lib().pio_bn_relu_maxpool_tokens(x.data_ptr(), scale.contiguous().data_ptr(),
                                                       shift.contiguous().data_ptr(), y.data_ptr(), b, c, h, w,
                                                       pads[2], pads[0], R.stream_ptr(x.device)),
                    "pio_bn_relu_maxpool_tokens")
        return R.forward_only(y, inputs, *self.parameters())


# ---------------------------------------------------------------------------------------------------
# preprocessors: forward(inputs, *, pos=None) -> (inputs_with_pos [B,M,C], inputs_without_pos); n_output_channels()
# ---------------------------------------------------------------------------------------------------
class _PosMixin:
    """Shared tail of the image / audio preprocessors: optional extra MLPs on the position encoding, then
    concat or add (preprocessors.py:176-200, 333-353)."""

    def _init_pos(self, position_encoding_type, index_dims, n_extra_pos_mlp, concat_or_add_pos, kwargs):
        if concat_or_add_pos not in ["concat", "add"]:
            raise ValueError(f"Invalid value {concat_or_add_pos} for concat_or_add_pos.")
        self._concat_or_add_pos = concat_or_add_pos
        self._positional_encoding = PE.build_position_encoding(position_encoding_type=position_encoding_type,
                                                               index_dims=index_dims, **kwargs)
        self._n_extra_pos_mlp = n_extra_pos_mlp
        if n_extra_pos_mlp > 0:
            self._extra_pos_mlps = nn.ModuleList()
            width = self._positional_encoding.n_output_channels()
            for _ in range(n_extra_pos_mlp):
                lin = nn.Linear(width, width)
                lecun_normal_(lin.weight)
                nn.init.constant_(lin.bias, 0)
                self._extra_pos_mlps.append(lin)

    def _attach_pos(self, feats: torch.Tensor, pos):
        enc = self._positional_encoding(batch_size=feats.shape[0], pos=pos, device=feats.device).to(feats.device)
        for i in range(self._n_extra_pos_mlp):
            enc = enc + self._extra_pos_mlps[i](enc)
            if i < self._n_extra_pos_mlp - 1:
                enc = F.relu(enc)
        with_pos = torch.cat([feats, enc], dim=-1) if self._concat_or_add_pos == "concat" else feats + enc
        return with_pos, feats

    def _split_pos(self, feats: torch.Tensor):
        """(features [B,M,C1], position table [M,C2]) when the network input is their concatenation and the position
        features are ONE batch-invariant table -- the form pio_encoder_fwd_split consumes without ever building the
        replicated [B,M,C1+C2] array; None otherwise."""
        if self._concat_or_add_pos != "concat" or self._n_extra_pos_mlp > 0:
            return None
        enc = self._positional_encoding(batch_size=feats.shape[0], pos=None, device=feats.device).to(feats.device)
        if enc.dim() != 3 or (enc.shape[0] > 1 and enc.stride(0) != 0) or enc.shape[1] != feats.shape[1]:
            return None
        if (feats.shape[2] | enc.shape[2]) & 1:
            return None
        return feats, enc[0]


class EmbeddingPreprocessor(nn.Module):
    """Token embedding + learned position embedding (preprocessors.py:18-54)."""

    def __init__(self, vocab_size: int, max_seq_len: int, embedding_dims: int):
        super().__init__()
        self.input_pos_encoding = TrainablePositionEncoding(index_dim=max_seq_len, num_channels=embedding_dims)
        self.embed = nn.Embedding(num_embeddings=vocab_size, embedding_dim=embedding_dims)
        self._output_channels = embedding_dims

    def n_output_channels(self):
        return self._output_channels

    def forward(self, inputs: torch.Tensor, *, pos: Optional[torch.Tensor] = None):
        tokens = self.embed(inputs)
        return tokens + self.input_pos_encoding(inputs.shape[0]), tokens


class ImagePreprocessor(nn.Module, _PosMixin):
    """Images / videos -> [B, M, C]: "conv" (Conv2DDownsample), "conv1x1", "patches" (space-to-depth) or "pixels",
    then the position encoding is concatenated or added (preprocessors.py:57-258)."""

    def __init__(self, img_size: Sequence[int], num_frames: int = 1, input_channels: int = 3, prep_type: str = "conv",
                 spatial_downsample: int = 4, temporal_downsample: int = 1,
                 position_encoding_type: PosEncodingType = PosEncodingType.FOURIER, n_extra_pos_mlp: int = 0,
                 num_channels: int = 64, conv_after_patching: bool = False, conv2d_use_batchnorm: bool = True,
                 concat_or_add_pos: str = "concat", **position_encoding_kwargs):
        super().__init__()
        if prep_type not in ("conv", "patches", "pixels", "conv1x1"):
            raise ValueError("Invalid prep_type!")
        if concat_or_add_pos not in ["concat", "add"]:
            raise ValueError(f"Invalid value {concat_or_add_pos} for concat_or_add_pos.")
        self._prep_type = prep_type
        self._spatial_downsample = spatial_downsample
        self._temporal_downsample = temporal_downsample
        self._conv_after_patching = conv_after_patching
        self._position_encoding_type = position_encoding_type
        if prep_type == "conv":
            layers = math.log(spatial_downsample, 4)
            if layers != round(layers) or temporal_downsample != 1:
                raise ValueError("Only powers of 4 expected for spatial and 1 expected for temporal "
                                 "downsampling with conv.")
            self.convnet = Conv2DDownsample(in_channels=input_channels, num_layers=int(layers),
                                            num_channels=num_channels, use_batchnorm=conv2d_use_batchnorm)
        elif prep_type == "conv1x1":
            assert temporal_downsample == 1, "conv1x1 does not downsample in time."
            self.convnet_1x1 = nn.Conv2d(input_channels, num_channels, kernel_size=1,
                                         stride=(spatial_downsample, spatial_downsample))
            nn.init.trunc_normal_(self.convnet_1x1.weight, mean=0.0, std=0.01, a=-2.0, b=2.0)
            nn.init.constant_(self.convnet_1x1.bias, 0)
        self.index_dims = [d // spatial_downsample for d in img_size]
        if num_frames > 1:
            self.index_dims = [num_frames // temporal_downsample] + self.index_dims
        self._init_pos(position_encoding_type, self.index_dims, n_extra_pos_mlp, concat_or_add_pos,
                       position_encoding_kwargs)
        if conv_after_patching:
            self._conv_after_patch_layer = nn.Linear(input_channels * spatial_downsample * temporal_downsample,
                                                     num_channels)
            lecun_normal_(self._conv_after_patch_layer.weight)
            nn.init.constant_(self._conv_after_patch_layer.bias, 0)
        if prep_type == "pixels":
            self.output_channels = input_channels
        elif prep_type == "patches":
            self.output_channels = num_channels if conv_after_patching else \
                input_channels * spatial_downsample ** 2 * temporal_downsample
        else:
            self.output_channels = num_channels
        if concat_or_add_pos == "concat":
            self.output_channels += self._positional_encoding.n_output_channels()

    def n_output_channels(self):
        return self.output_channels

    def _features(self, inputs: torch.Tensor) -> torch.Tensor:
        x = inputs
        if self._prep_type == "conv" and x.dim() == 4:
            tokens = self.convnet.forward_tokens(x)      # conv (MIOpen) + one fused HIP pass, channels-last tokens
            if tokens is not None:
                return tokens
        if self._prep_type in ("conv", "conv1x1"):
            video = x.dim() == 5
            if video:
                b, t = x.shape[:2]
                x = x.reshape((b * t,) + x.shape[2:])
            x = self.convnet(x) if self._prep_type == "conv" else self.convnet_1x1(x)
            x = x.movedim(-3, -1)
            if video:
                x = x.reshape((b, t) + x.shape[1:])
        elif self._prep_type == "patches":
            x = space_to_depth(x.movedim(-3, -1), temporal_block_size=self._temporal_downsample,
                               spatial_block_size=self._spatial_downsample)
            if x.ndim == 5 and x.shape[1] == 1:     # optical flow: the frame pair collapsed into channels
                x = x.squeeze(1)
            if self._conv_after_patching:
                x = self._conv_after_patch_layer(x)
        else:  # "pixels": crude strided subsampling
            x = x.movedim(-3, -1)
            s, t_ = self._spatial_downsample, self._temporal_downsample
            if x.ndim == 4:
                x = x[:, ::s, ::s]
            elif x.ndim == 5:
                x = x[:, ::t_, ::s, ::s]
            else:
                raise ValueError("Unsupported data format for pixels.")
        if x.dim() > 3:
            x = x.reshape(x.shape[0], int(math.prod(self.index_dims)), -1)
        return x

    def forward(self, inputs: torch.Tensor, *, pos=None):
        """inputs: [..., channel, height, width] (PyTorch image layout)."""
        return self._attach_pos(self._features(inputs), pos)

    def forward_split(self, inputs: torch.Tensor):
        """The same network input as two arrays: ("split", features [B,M,C1], position table [M,C2]); when this
        configuration does not concatenate a batch-invariant position table (see _PosMixin._split_pos) the ordinary
        hand-off ("full", inputs_with_pos, inputs_without_pos) built from the SAME features -- the preprocessing network
        runs once either way (in train mode a second pass would also update the BatchNorm statistics twice)."""
        feats = self._features(inputs)
        sp = self._split_pos(feats)
        if sp is not None:
            return ("split",) + tuple(sp)
        return ("full",) + tuple(self._attach_pos(feats, None))


class OneHotPreprocessor(nn.Module):
    """[B, C] one-hot / dense vector -> one token [B, 1, C] (preprocessors.py:261-282)."""

    def __init__(self, input_channels: int):
        super().__init__()
        self.input_channels = input_channels

    def n_output_channels(self):
        return self.input_channels

    def forward(self, inputs: torch.Tensor, *, pos: Optional[torch.Tensor] = None):
        tok = inputs[:, None, :]
        return tok, tok


class AudioPreprocessor(nn.Module, _PosMixin):
    """Raw audio [B, samples, 1] -> patches of ``samples_per_patch`` + position encoding (preprocessors.py:285-364)."""

    def __init__(self, samples_per_batch: int, prep_type: str = "patches", samples_per_patch: int = 96,
                 position_encoding_type: PosEncodingType = PosEncodingType.FOURIER, n_extra_pos_mlp: int = 0,
                 concat_or_add_pos: str = "concat", **position_encoding_kwargs):
        super().__init__()
        if prep_type not in ("patches",):
            raise ValueError("Invalid prep_type!")
        self._samples_per_patch = samples_per_patch
        self.index_dims = [samples_per_batch // samples_per_patch]
        self._init_pos(position_encoding_type, self.index_dims, n_extra_pos_mlp, concat_or_add_pos,
                       position_encoding_kwargs)
        self.output_channels = samples_per_patch
        if concat_or_add_pos == "concat":
            self.output_channels += self._positional_encoding.n_output_channels()

    def n_output_channels(self):
        return self.output_channels

    def forward(self, inputs: torch.Tensor, *, pos: Optional[torch.Tensor] = None):
        return self._attach_pos(inputs.reshape(inputs.shape[0], -1, self._samples_per_patch), pos)


# ---------------------------------------------------------------------------------------------------
# postprocessors: forward(inputs, *, pos=None, modality_sizes=None)
# ---------------------------------------------------------------------------------------------------
def _linear(mod: nn.Linear, x: torch.Tensor, cache: dict) -> torch.Tensor:
    """The nn.Linear heads right behind the decoder run through libpio_hip.so (runtime.hip_linear: pio_gemm_nt under
    the current -- decoder -- precision policy) on the GPU; on CPU tensors the stock torch path (plumbing tests)."""
    if x.is_cuda:
        from .runtime import hip_linear
        return hip_linear(x, mod.weight, mod.bias, cache)
    return mod(x)



class EmbeddingPostprocessor(nn.Module):
    """Logits against the (tied) token embedding + bias (postprocessors.py:12-34)."""

    def __init__(self, embedding: nn.Embedding):
        super().__init__()
        self._embedding = embedding
        self._vocab_size, self._d_model = embedding.weight.shape
        self.bias = nn.Parameter(torch.zeros(self._vocab_size))
        self._pio_linear = {}       # packed image of the tied weight (runtime.hip_linear)

    def forward(self, inputs: torch.Tensor, *, pos=None, modality_sizes=None) -> torch.Tensor:
        b, t, _ = inputs.shape
        if inputs.is_cuda:
            # the step right behind the decoder: [B*T, d_model] x [vocab, d_model]^T through libpio_hip.so
            # (pio_gemm_nt) under the model's precision policy, not a torch matmul
            from .runtime import hip_linear
            return hip_linear(inputs, self._embedding.weight, self.bias, self._pio_linear)
        return (inputs.reshape(-1, self._d_model) @ self._embedding.weight.T + self.bias).reshape(b, t,
                                                                                                 self._vocab_size)


class ImagePostprocessor(nn.Module):
    """"pixels" (identity) and "patches" (reverse space-to-depth); the conv variants are unimplemented in the
    reference as well (postprocessors.py:37-122)."""

    def __init__(self, img_size: Sequence[int], input_channels: int = 3, postproc_type: str = "pixels",
                 spatial_upsample: int = 1, temporal_upsample: int = 1, n_outputs: int = -1,
                 input_reshape_size: Optional[Sequence[int]] = None):
        super().__init__()
        if postproc_type not in ("conv", "patches", "pixels", "raft", "conv1x1"):
            raise ValueError("Invalid postproc_type!")
        if postproc_type == "pixels" and (temporal_upsample != 1 or spatial_upsample != 1):
            raise ValueError("Pixels postprocessing should not currently upsample.")
        if postproc_type in ("conv", "conv1x1", "raft"):
            if postproc_type == "conv" and n_outputs == -1:
                raise ValueError("Expected value for n_outputs")
            raise NotImplementedError
        self._postproc_type = postproc_type
        self._temporal_upsample = temporal_upsample
        self._spatial_upsample = spatial_upsample
        self._input_reshape_size = input_reshape_size

    def forward(self, inputs: torch.Tensor, *, pos=None, modality_sizes=None) -> torch.Tensor:
        if self._input_reshape_size is not None:
            inputs = inputs.reshape([inputs.shape[0]] + list(self._input_reshape_size) + [inputs.shape[-1]])
        if self._postproc_type == "patches":
            inputs = reverse_space_to_depth(inputs, self._temporal_upsample, self._spatial_upsample)
        return inputs


class AudioPostprocessor(nn.Module):
    """Linear to ``samples_per_patch`` then flatten to a waveform (postprocessors.py:125-149)."""

    def __init__(self, postproc_type: str = "patches", in_channels: int = 1024, samples_per_patch: int = 96):
        super().__init__()
        if postproc_type not in ("patches",):
            raise ValueError("Invalid postproc_type!")
        self._postproc_type = postproc_type
        self.linear = nn.Linear(in_channels, samples_per_patch)
        lecun_normal_(self.linear.weight)
        nn.init.constant_(self.linear.bias, 0)
        self._pio_linear = {}

    def forward(self, inputs: torch.Tensor, *, pos=None, modality_sizes=None) -> torch.Tensor:
        return _linear(self.linear, inputs, self._pio_linear).reshape(inputs.shape[0], -1)


class IdentityPostprocessor(nn.Module):
    def forward(self, inputs: torch.Tensor, *, pos=None, modality_sizes=None) -> torch.Tensor:
        return inputs


class ClassificationPostprocessor(nn.Module):
    """Optional linear head, then the logits of query row 0 (postprocessors.py:164-187)."""

    def __init__(self, num_input_channels: int, num_classes: int, project: bool = True):
        super().__init__()
        self._num_classes = num_classes
        self._project = project
        self._pio_linear = {}
        if project:
            self.linear = nn.Linear(num_input_channels, num_classes)
            lecun_normal_(self.linear.weight)
            nn.init.constant_(self.linear.bias, 0)

    def forward(self, inputs: torch.Tensor, *, pos=None, modality_sizes=None) -> torch.Tensor:
        # (rows are independent: the head is applied to the one row that is kept)
        row0 = inputs[:, 0:1, :]
        return (_linear(self.linear, row0, self._pio_linear) if self._project else row0)[:, 0, :]


class ProjectionPostprocessor(nn.Module):
    """A linear projection of every output row (postprocessors.py:190-208)."""

    def __init__(self, num_inputs: int, num_outputs: int):
        super().__init__()
        self._num_outputs = num_outputs
        self.projection = nn.Linear(num_inputs, num_outputs)
        lecun_normal_(self.projection.weight)
        nn.init.constant_(self.projection.bias, 0)
        self._pio_linear = {}

    def forward(self, inputs: torch.Tensor, *, pos=None, modality_sizes=None) -> torch.Tensor:
        return _linear(self.projection, inputs, self._pio_linear)


class FlowPostprocessor(nn.Module):
    """Scale and reshape [B, H*W, 2] -> [B, 2, H, W] (postprocessors.py:211-230)."""

    def __init__(self, img_size: Sequence[int], flow_scale_factor: float = 1.0):
        super().__init__()
        self.flow_scale_factor = flow_scale_factor
        self.img_size = img_size

    def forward(self, inputs: torch.Tensor, *, pos=None, modality_sizes: Optional[Mapping[str, int]] = None):
        return (inputs * self.flow_scale_factor).reshape([inputs.shape[0], *self.img_size, 2]).permute(0, 3, 1, 2)
