"""Host-side plumbing between torch tensors and the C-ABI: precision policy, packed-weight images,
descriptors, workspaces.  PyTorch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L

# ------------------------------------------------------------------------------------------------
# precision policy: operand dtype of the MFMA matrices x number of MFMA sweeps per product.
#   fp16x3   every operand is a (hi, lo) fp16 pair, products = hi*hi + hi*lo + lo*hi: ~fp32-exact
#            results (1e-5 of the reference) at 1/3 of the MFMA rate.  DEFAULT: a drop-in must be safe.
#   fp16x2w  weights are (hi, lo) pairs, activations single fp16: removes the systematic weight-rounding
#            error that the 8x weight sharing amplifies; 3e-4 on the ImageNet config.
#   fp16     one sweep, plain fp16 operands: 7e-4 on the ImageNet config (meets 1e-3, thin margin).
#   bf16*    the same with bf16 operands (wider range, 8x coarser mantissa; only bf16x3 meets 1e-3).
# (numbers: tools/numerics_model.py and tests/test_parity_gpu.py; discussion in DESIGN.md)
# ------------------------------------------------------------------------------------------------
# (operand dtype, weight split level, activations split)
#   weight split level: 0 none | 1 proj_v + final ("x2s": the two projections whose weight rounding matters most,
#   4.5e-4 on the ImageNet model at 1.2x the fp16 cost) | 2 + fc1, fc2, decoder final_layer ("x2w": 3.1e-4) |
#   3 every weight incl. proj_q / proj_k ("x3").  proj_q / proj_k stay single below level 3: their rounding is
#   not measurable in the output (3.13e-4 with or without it, tools/numerics_model.py attribution).
_POLICIES = {"fp16": (L.PIO_DT_F16, 0, False), "fp16x2s": (L.PIO_DT_F16, 1, False),
             "fp16x2w": (L.PIO_DT_F16, 2, False), "fp16x3": (L.PIO_DT_F16, 3, True),
             "bf16": (L.PIO_DT_BF16, 0, False), "bf16x2w": (L.PIO_DT_BF16, 2, False),
             "bf16x3": (L.PIO_DT_BF16, 3, True),
             # x3f: every GEMM with split operands as x3, but the attention core (Q K^T, softmax, P V) single-sweep on
             # the fused kernels (q / k / v / p rounded once; the core's output leaves as a pair): no score matrix
             "fp16x3f": (L.PIO_DT_F16, 3, True), "bf16x3f": (L.PIO_DT_BF16, 3, True),
             # x2af: split ACTIVATIONS against single weights (A_hi B^T + A_lo B^T: two sweeps) around a single-sweep
             # fused attention core -- for decoders whose error is dominated by the rounding of wide-range inputs
             # (Fourier / position features of dense outputs), at 2/3 of the x3f cost
             "fp16x2af": (L.PIO_DT_F16, 0, True),
             # fp16sd: "fp16" with ERROR-FEEDBACK rounding of the weights an encoder shares over its blocks: block b
             # multiplies with fp16(W + E_{b-1}), E_b = (W + E_{b-1}) - that image (first-order sigma-delta over the
             # block index).  Every image is a rounding of W within one ulp; the error the 8 applications of a shared
             # weight accumulate stays below half an ulp instead of growing 8-fold.  Same kernels, same time, 8 x the
             # packed images (604 MB for the ImageNet stack); -18..-21 % error on the ImageNet goldens.  Everything
             # that is not a weight-shared block stack runs as under "fp16".
             "fp16sd": (L.PIO_DT_F16, 0, False),
             # finer weight splits of a self-attend stack (round 4): only the out projection ("final"), or only proj_v, of
             # the two that "x2s" splits -- a second K sweep costs its GEMM's time again
             "fp16x2o": (L.PIO_DT_F16, 0, False), "fp16x2v": (L.PIO_DT_F16, 0, False),
             # x2afo: "x2af" (split activations around the single-sweep fused core) + split WEIGHTS on the output side of a
             # decoder only -- the attention's out projection, the decoder's final_layer, the heads behind it -- where a
             # dense-output model has nothing left to average a weight's rounding away (multimodal, second seed: x2af
             # 1.01e-3 / 1.14e-3, x2afo 4.0e-4 / 5.3e-4 at +4 % time; x3f 0.6e-4 / 2.7e-4 at +17 %)
             "fp16x2afo": (L.PIO_DT_F16, 0, True)}
# weights split under the fine policies (beside the level of _POLICIES): Attention: proj_v, final; MLP: fc1, fc2; decoder:
# final_layer; heads behind the decoder (hip_linear): post
_FINE_SPLIT = {"fp16x2o": {"final"}, "fp16x2v": {"proj_v"}, "fp16x2afo": {"final", "final_layer", "post"}}
_FUSED_CORE = {"fp16x3f", "bf16x3f", "fp16x2af", "fp16x2afo"}
_BLOCK_FEEDBACK = {"fp16sd"}
_policy = os.environ.get("PIO_PRECISION", "fp16x3")
if _policy not in _POLICIES:
    raise ValueError(f"PIO_PRECISION={_policy!r} not in {sorted(_POLICIES)}")


def set_precision_policy(name: str) -> None:
    global _policy
    if name not in _POLICIES:
        raise ValueError(f"unknown precision policy {name!r}; choose from {sorted(_POLICIES)}")
    _policy = name


def get_precision_policy() -> str:
    return _policy


class precision:
    """Context manager: run a block under another precision policy (None = leave the current one)."""

    def __init__(self, name: Optional[str]):
        if name is not None and name not in _POLICIES:
            raise ValueError(f"unknown precision policy {name!r}; choose from {sorted(_POLICIES)}")
        self._name = name
        self._saved = None

    def __enter__(self):
        global _policy
        self._saved = _policy
        if self._name is not None:
            _policy = self._name
        return self

    def __exit__(self, *exc):
        global _policy
        _policy = self._saved
        return False


def policy_dtype(name: Optional[str] = None) -> Tuple[int, bool, bool]:
    """(operand dtype, weights split, activations split)"""
    return _POLICIES[name or _policy]


# ------------------------------------------------------------------------------------------------
# range guard of the LayerNorm-folded stack.  Inside it the residual stream is an fp16 PAIR (hi + lo) and the GEMMs
# read the un-normalised hi half: a latent beyond +-65504 becomes inf (the un-folded path keeps the stream in fp32 and
# only ever rounds LayerNorm OUTPUTS, which are O(1)).  The guard lives ON THE DEVICE: every GEMM that produces the
# folded stream already reduces each result row's (sum, sum of squares) and ORs 1 into a device word when one of them
# is not finite (pio_ln_fold_t.range_flag) -- no extra kernel, no host synchronisation inside the forward.  With the
# guard on (default; PIO_RANGE_CHECK=0 or set_range_check(False) turns it off) the word is read
#   * by PerceiverIO.forward ONCE, after the decoder and the postprocessors have been enqueued (the point where the
#     caller is about to consume the result), or
#   * by PerceiverEncoder.forward itself when it is called on its own;
# a set word re-runs the call with the fold off and raises PIO_E_RANGE if the un-folded result is not finite either
# (then the model needs a wider operand type: precision policy "bf16x3").  During stream capture nothing is read:
# the word stays in `last_range_flag()` for the owner of the graph to look at after a replay.
# ------------------------------------------------------------------------------------------------
_range_check = os.environ.get("PIO_RANGE_CHECK", "1") != "0"
_range_flags: Dict[tuple, torch.Tensor] = {}
_range_deferred: Optional[list] = None      # not None: PerceiverIO.forward collects the encoder's pending checks


def range_check() -> bool:
    return _range_check


def set_range_check(on: bool) -> None:
    global _range_check
    _range_check = bool(on)


def range_flag(device: torch.device, owner=None) -> torch.Tensor:
    """The device word a folded stack reports into (int32[1]): ONE PER OWNER (a PerceiverEncoder instance) and device, so
    that two models running on one device never clear or read each other's word; `last_range_flag(device)` is the word of
    the forward that ran last there (what the owner of a captured graph reads after a replay)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, id(owner) if owner is not None else 0)
    t = _range_flags.get(key)
    if t is None:
        with torch.inference_mode(False):     # (a normal tensor: zeroed in place from inside and outside inference mode)
            t = torch.zeros(1, dtype=torch.int32, device=device)
        _range_flags[key] = t
        if owner is not None:
            import weakref
            weakref.finalize(owner, _range_flags.pop, key, None)
    _range_flags[(idx, "last")] = t
    return t


def last_range_flag(device: Optional[torch.device] = None) -> Optional[torch.Tensor]:
    idx = (device.index if device is not None and device.index is not None else torch.cuda.current_device())
    return _range_flags.get((idx, "last"))


class defer_range_checks:
    """PerceiverIO.forward: the encoder registers its pending check here instead of synchronising mid-forward."""

    def __enter__(self):
        global _range_deferred
        self._saved = _range_deferred
        _range_deferred = []
        self.pending = _range_deferred
        return self

    def __exit__(self, *exc):
        global _range_deferred
        _range_deferred = self._saved
        return False


def range_deferred() -> Optional[list]:
    return _range_deferred


def capturing(device: torch.device) -> bool:
    try:
        return bool(torch.cuda.is_current_stream_capturing())
    except Exception:  # noqa: BLE001
        return False


def policy_fine_split(name: Optional[str] = None) -> set:
    """Names of the linears whose weights are (hi, lo) pairs beyond what the policy's level says (Attention: proj_v, final;
    MLP: fc1, fc2; decoder: final_layer; heads behind the decoder: post)."""
    extra = set(os.environ.get("PIO_EXTRA_SPLIT", "").split(",")) - {""}      # (A/B experiments: tools/)
    return _FINE_SPLIT.get(name or _policy, set()) | extra


def policy_core_single(name: Optional[str] = None) -> bool:
    """True for the "x3f" policies: split operands in the projections, single-sweep fused attention core."""
    return (name or _policy) in _FUSED_CORE


def policy_block_feedback(name: Optional[str] = None) -> bool:
    """True for the policies that pack one image set per block of a weight-shared stack ("fp16sd")."""
    return (name or _policy) in _BLOCK_FEEDBACK


def feedback_images(w: torch.Tensor, nblk: int, dtype: int):
    """The nblk error-feedback roundings of the fp32 weight `w` to the operand dtype, as fp32 tensors that are exactly
    representable in it (packing them is exact): image_b = round(w + E_{b-1}), E_b = (w + E_{b-1}) - image_b, E_{-1} = 0.
    |w - image_b| < 1 ulp for every b and |sum_{b<=k} (w - image_b)| = |E_k| <= ulp / 2 for every k."""
    tdt = torch.float16 if dtype == L.PIO_DT_F16 else torch.bfloat16
    w = w.detach().float()
    out, e = [], torch.zeros_like(w)
    for _ in range(nblk):
        t = w + e
        hi = t.to(tdt).float()
        e = t - hi
        out.append(hi)
    return out


def pad8(c: int) -> int:
    return (c + 7) & ~7


def padc(c: int) -> int:
    """Pitch of a CHANNEL axis in the 16-bit operand arrays (csrc/pio_internal.h padc): a multiple of 8, from 256
    channels on a multiple of 64 (the staged GEMM kernels read whole 64-deep K slices)."""
    return int(L.lib().pio_padc(int(c)))


# ------------------------------------------------------------------------------------------------
# backend.  "hip" (default): the hot path exists only as gfx950 kernels -- CPU tensors raise.  "torch": the opt-in CPU
# PLUMBING backend (cpu_plumbing.py) for machines without a GPU (BASELINE config 1) -- CUDA tensors raise.  Nothing ever
# switches by itself; there is no fallback in either direction.
# ------------------------------------------------------------------------------------------------
_backend = os.environ.get("PIO_BACKEND", "hip")
if _backend not in ("hip", "torch"):
    raise ValueError(f"PIO_BACKEND={_backend!r}: 'hip' or 'torch'")


def set_backend(name: str) -> None:
    global _backend
    if name not in ("hip", "torch"):
        raise ValueError(f"unknown backend {name!r}: 'hip' (MI355X kernels) or 'torch' (opt-in CPU plumbing)")
    _backend = name


def get_backend() -> str:
    return _backend


def cpu_plumbing(t: torch.Tensor, what: str) -> bool:
    """True when this call runs on the opt-in CPU plumbing backend.  A CUDA tensor under that backend raises: GPU
    tensors only ever run on the HIP kernels."""
    if _backend != "torch":
        return False
    if t.is_cuda:
        raise L.PioError(f"{what}: backend 'torch' is the CPU plumbing path and got a {t.device} tensor; GPU tensors run "
                         "on the HIP kernels only -- set_backend('hip')")
    return True


def require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise L.PioError(f"{what}: perceiverio_pytorch_amd computes on an MI355X (HIP) device only; got a "
                         f"{t.device} tensor. There is no CPU or eager fallback (an explicit CPU plumbing backend "
                         "exists for machines without a GPU: perceiverio_pytorch_amd.set_backend('torch')).")


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# ------------------------------------------------------------------------------------------------
# tensors at the boundary
# ------------------------------------------------------------------------------------------------
def as_f32_3d(t: torch.Tensor) -> torch.Tensor:
    """float32, last dim contiguous, 3-D; broadcast (stride-0) batch views are kept as views."""
    if t.dim() != 3:
        raise ValueError(f"expected a [B, T, C] tensor, got {tuple(t.shape)}")
    if t.dtype != torch.float32:
        t = t.float()
    if t.stride(2) != 1 and t.shape[2] != 1:
        t = t.contiguous()
    if t.stride(1) < t.shape[2] and t.shape[1] > 1:    # exotic row-overlapping views
        t = t.contiguous()
    return t


def tensor3(t: torch.Tensor) -> L.Tensor3:
    B, T, Cc = t.shape
    sb = t.stride(0) if B > 1 else T * Cc
    st = t.stride(1) if T > 1 else Cc
    return L.Tensor3(t.data_ptr(), sb, st, B, T, Cc)


def mask_u8(m: Optional[torch.Tensor], shape, device) -> Tuple[Optional[torch.Tensor], Optional[int]]:
    """bool/any mask -> contiguous uint8 tensor on `device` (kept alive by the caller) + pointer."""
    if m is None:
        return None, None
    if tuple(m.shape) != tuple(shape):
        raise ValueError(f"mask shape {tuple(m.shape)} != {tuple(shape)}")
    mm = m.to(device=device)
    mm = (mm != 0) if mm.dtype != torch.bool else mm
    mm = mm.contiguous().view(torch.uint8)
    return mm, mm.data_ptr()


# ------------------------------------------------------------------------------------------------
# workspaces: one grow-only byte buffer per (device, stream); launches on a stream are ordered, so
# consecutive blocks can share it.
# ------------------------------------------------------------------------------------------------
_workspaces: Dict[Tuple[int, int], torch.Tensor] = {}


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream_ptr(device))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _workspaces.pop(key, None)
        ws = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def release_workspaces() -> None:
    _workspaces.clear()


# ------------------------------------------------------------------------------------------------
# optional batch-slice streams for the encoder stack (PIO_STREAMS=k, default 1 = off)
# ------------------------------------------------------------------------------------------------
_nstreams = max(1, int(os.environ.get("PIO_STREAMS", "1")))
_side_streams: Dict[Tuple[int, int], list] = {}


def batch_streams() -> int:
    return _nstreams


def set_batch_streams(n: int) -> None:
    global _nstreams
    _nstreams = max(1, int(n))


_cu_split = os.environ.get("PIO_CU_SPLIT", "0") == "1"


def cu_split() -> bool:
    return _cu_split


def set_cu_split(on: bool) -> None:
    """Batch-slice streams that each own a disjoint share of every XCD's CUs (pio_stream_create_cu_mask) instead of
    plain streams: the slices' kernel chains then truly run side by side."""
    global _cu_split
    _cu_split = bool(on)
    release_side_streams()


_masked_cache: Dict[Tuple[int, int], list] = {}    # (device index, n) -> CU-masked streams, created ONCE and reused


def release_side_streams() -> None:
    """Forget the cached side streams.  CU-masked streams (raw HIP streams from pio_stream_create_cu_mask) are NOT
    destroyed here: tensors that were record_stream()-ed on them keep the handle alive inside PyTorch's caching
    allocator, and destroying it under the allocator crashes the process.  They are created once per (device, split) and
    reused by every later toggle instead -- a bounded set, no leak."""
    _side_streams.clear()


def cu_share(n: int) -> int:
    """CUs per slice when the chip is split `n` ways, from the device's own CU count (MI355X: 256 = 8 XCDs x 32)."""
    try:
        total = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    except Exception:  # noqa: BLE001
        total = 256
    return total // n


def _masked_streams(device: torch.device, n: int) -> list:
    """`n` streams with complementary CU masks: slice i owns CUs [i*32/n, (i+1)*32/n) of EVERY XCD (mask bit j is CU
    j // 8 of XCD j % 8), so each slice keeps the round-robin dispatch over all eight L2s."""
    import ctypes as C
    lib = L.lib()
    per = 32 // n
    out = []
    for i in range(n):
        words = (C.c_uint32 * 8)()
        for j in range(256):
            if i * per <= j // 8 < (i + 1) * per:
                words[j // 32] |= 1 << (j % 32)
        h = C.c_void_p()
        with torch.cuda.device(device):
            L.check(lib.pio_stream_create_cu_mask(C.byref(h), words, 8), "pio_stream_create_cu_mask")
        out.append(torch.cuda.ExternalStream(h.value, device=device))
    return out


def side_streams(device: torch.device, n: int) -> list:
    key = (device.index if device.index is not None else torch.cuda.current_device(), n, _cu_split)
    if key not in _side_streams:
        if _cu_split and 32 % n == 0:
            mk = (key[0], n)
            if mk not in _masked_cache:
                _masked_cache[mk] = _masked_streams(device, n)
            _side_streams[key] = _masked_cache[mk]
        else:
            _side_streams[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _side_streams[key]


# ------------------------------------------------------------------------------------------------
# packed weights
# ------------------------------------------------------------------------------------------------
class PackedLinear:
    """Device image of one nn.Linear in kernel layout (hi [+ lo] + padded bias) and its descriptor."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], row_heads: int, col_heads: int,
                 dtype: int, two_pass: bool, *, k_channels: bool = True, n_channels: bool = False):
        """`k_channels`: the input axis is a channel axis (pitch padc) -- everything but Attention.final, whose input
        is heads x padded head dim; `n_channels`: so is the OUTPUT (MLP.fc1: its 16-bit output is fc2's operand)."""
        require_device(weight, "pack_linear")
        lib = L.lib()
        out, inn = weight.shape
        if out % row_heads or inn % col_heads:
            raise ValueError("channels must be divisible by the head count")
        rows_packed = row_heads * pad8(out // row_heads)
        self.n = padc(out) if (n_channels and row_heads == 1) else rows_packed
        self.k = padc(inn) if (k_channels and col_heads == 1) else col_heads * pad8(inn // col_heads)
        dev = weight.device
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        tdt = torch.float16 if dtype == L.PIO_DT_F16 else torch.bfloat16
        alloc = torch.zeros if self.n > rows_packed else torch.empty     # (rows behind the packed ones stay zero)
        self.hi = alloc((self.n, self.k), dtype=tdt, device=dev)
        self.lo = alloc((self.n, self.k), dtype=tdt, device=dev) if two_pass else None
        self.bias = alloc((self.n,), dtype=torch.float32, device=dev)
        b = None
        if bias is not None:
            b = bias.detach()
            if b.dtype != torch.float32 or not b.is_contiguous():
                b = b.float().contiguous()
        L.check(lib.pio_pack_linear(w.data_ptr(), b.data_ptr() if b is not None else None, out, inn, inn,
                                    row_heads, col_heads, self.hi.data_ptr(),
                                    self.lo.data_ptr() if two_pass else None, self.bias.data_ptr(), 0, self.k,
                                    dtype, stream_ptr(dev)), "pio_pack_linear")
        self.desc = L.Linear(self.hi.data_ptr(), self.lo.data_ptr() if two_pass else None, self.bias.data_ptr(),
                             self.n, self.k)


class PackedStack:
    """Several nn.Linear with equal in_features stacked along the output rows in ONE packed image
    (e.g. proj_q | proj_k, consumed by a single GEMM when both read the same input).  `two_pass` is a bool (every part)
    or one bool per part: only a SUFFIX of the parts may carry a lo image (q | k | v with v alone split); the
    descriptor's lo_row0 then names the first row that has one."""

    def __init__(self, pairs, row_heads: int, dtype: int, two_pass):
        lib = L.lib()
        flags = [bool(two_pass)] * len(pairs) if isinstance(two_pass, bool) else [bool(f) for f in two_pass]
        if any(a and not b for a, b in zip(flags, flags[1:])):
            raise ValueError("only a suffix of the stacked linears may be split")
        two_pass = any(flags)
        w0 = pairs[0][0]
        require_device(w0, "pack_linear")
        dev = w0.device
        inn = w0.shape[1]
        self.k = padc(inn)
        rows = [row_heads * pad8(w.shape[0] // row_heads) for w, _ in pairs]
        self.n = sum(rows)
        tdt = torch.float16 if dtype == L.PIO_DT_F16 else torch.bfloat16
        self.hi = torch.empty((self.n, self.k), dtype=tdt, device=dev)
        self.lo = torch.empty((self.n, self.k), dtype=tdt, device=dev) if two_pass else None
        self.bias = torch.empty((self.n,), dtype=torch.float32, device=dev)
        if self.lo is not None and not all(flags):
            self.lo.zero_()          # rows without a lo half are never read; keep them defined
        r0 = 0
        lo_row0 = 0
        for (w, b), nr, fl in zip(pairs, rows, flags):
            if w.shape[1] != inn:
                raise ValueError("stacked linears need equal in_features")
            wf = w.detach().float().contiguous()
            bf = b.detach().float().contiguous() if b is not None else None
            L.check(lib.pio_pack_linear(wf.data_ptr(), bf.data_ptr() if bf is not None else None, w.shape[0], inn,
                                        inn, row_heads, 1, self.hi.data_ptr(),
                                        self.lo.data_ptr() if fl else None, self.bias.data_ptr(), r0, self.k,
                                        dtype, stream_ptr(dev)), "pio_pack_linear")
            if not fl:
                lo_row0 = r0 + nr
            r0 += nr
        self.lo_row0 = lo_row0 if two_pass else 0
        self.desc = L.Linear(self.hi.data_ptr(), self.lo.data_ptr() if two_pass else None, self.bias.data_ptr(),
                             self.n, self.k, self.lo_row0)


def _version_of(p: torch.Tensor) -> int:
    # (tensors created or moved under torch.inference_mode() do not track a version counter)
    return 0 if p.is_inference() else p._version


def param_key(*params) -> tuple:
    """Cache key that changes whenever a parameter is rebound, moved or modified in place through autograd-visible
    ops.  Writes through ``p.data`` (``p.data.copy_()``, some checkpoint loaders) do NOT bump the version counter:
    ``load_state_dict`` is covered by a hook of the modules; after any other raw write call
    ``invalidate_packed_weights(module)``."""
    return tuple((p.data_ptr(), _version_of(p), str(p.device)) if p is not None else None for p in params) + (_policy,)


def invalidate_packed_weights(module: torch.nn.Module) -> None:
    """Drop every packed / folded weight image cached under ``module`` (they are rebuilt on the next forward)."""
    for m in module.modules():
        if hasattr(m, "_pio_cache"):
            m._pio_cache = None
        if hasattr(m, "_pio_block_cache"):
            m._pio_block_cache = None
        if hasattr(m, "_final_cache"):
            m._final_cache = None


class _ForwardOnly(torch.autograd.Function):
    """Identity whose backward raises: the HIP path has no autograd.  Outputs of the HIP modules are fresh buffers
    without a grad_fn, so without this a ``loss.backward()`` would silently skip every encoder / decoder parameter."""

    @staticmethod
    def forward(ctx, out, *deps):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("perceiverio_pytorch_amd is a forward / inference path: there is no backward "
                                  "through the HIP kernels (run under torch.inference_mode() or torch.no_grad())")


def forward_only(out: torch.Tensor, *deps) -> torch.Tensor:
    """Tie `out` to the tensors it was computed from when autograd is recording, with a backward that raises."""
    if torch.is_grad_enabled():
        live = [t for t in deps if isinstance(t, torch.Tensor) and t.requires_grad]
        if live:
            return _ForwardOnly.apply(out, *live)
    return out


class on_device:
    """Make the tensor's GPU the current HIP device around a C call: the library launches on the given stream, but
    hipGetLastError, the architecture check and the cached CU count consult the CURRENT device."""

    def __init__(self, device: torch.device):
        self._ctx = torch.cuda.device(device)

    def __enter__(self):
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)


def layernorm_desc(ln: torch.nn.LayerNorm, keep: list) -> L.LayerNorm:
    w, b = ln.weight.detach(), ln.bias.detach()
    if w.dtype != torch.float32:
        w, b = w.float(), b.float()
    w, b = w.contiguous(), b.contiguous()
    keep += [w, b]
    return L.LayerNorm(w.data_ptr(), b.data_ptr(), w.numel(), float(ln.eps))


# ------------------------------------------------------------------------------------------------
# a plain nn.Linear through the C-ABI primitives (pio_layernorm_cast as the operand cast + pio_gemm_nt): used by the
# I/O plumbing right behind the decoder (tied-embedding vocabulary projection of the language model)
# ------------------------------------------------------------------------------------------------
def hip_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], cache: dict) -> torch.Tensor:
    """y[..., out] = x[..., in] W^T + b on the MI355X under the current precision policy (weights split from level
    "x2w" on, activations under "x3").  `cache` keeps the packed weight image between calls."""
    import ctypes as C
    require_device(x, "hip_linear")
    lib = L.lib()
    dev = x.device
    dtype, wlevel, split = policy_dtype()
    key = param_key(weight, bias)
    pk = cache.get("packed")
    if pk is None or pk[0] != key:
        pk = (key, PackedLinear(weight, bias, 1, 1, dtype, wlevel >= 2 or "post" in policy_fine_split()))
        cache["packed"] = pk
    lin = pk[1]
    lead = x.shape[:-1]
    x3 = as_f32_3d(x.reshape(1, -1, x.shape[-1]))
    rows, n_out = x3.shape[1], weight.shape[0]
    tdt = torch.float16 if dtype == L.PIO_DT_F16 else torch.bfloat16
    x16 = torch.empty((rows, lin.k), dtype=tdt, device=dev)
    x16lo = torch.empty((rows, lin.k), dtype=tdt, device=dev) if split else None
    out = torch.empty((rows, n_out), dtype=torch.float32, device=dev)
    with on_device(dev):
        L.check(lib.pio_layernorm_cast(tensor3(x3), None, x16.data_ptr(), x16lo.data_ptr() if split else None, lin.k,
                                       dtype, stream_ptr(dev)), "pio_layernorm_cast")
        g = L.Gemm()
        g.A, g.B = x16.data_ptr(), lin.hi.data_ptr()
        g.A_lo = x16lo.data_ptr() if split else None
        g.B_lo = lin.lo.data_ptr() if lin.lo is not None else None
        g.C = out.data_ptr()
        g.M, g.N, g.K = rows, n_out, lin.k
        g.lda = g.ldb = lin.k
        g.ldc = n_out
        g.batch = g.nh = 1
        g.bias = lin.bias.data_ptr()
        g.bias_mode = 1
        g.alpha = 1.0
        g.out_f32 = 1
        g.n_store = n_out
        g.dtype = dtype
        L.check(lib.pio_gemm_nt(C.byref(g), stream_ptr(dev)), "pio_gemm_nt")
    return forward_only(out.reshape(lead + (n_out,)), x, weight, bias)
