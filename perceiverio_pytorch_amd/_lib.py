"""ctypes binding of libpio_hip.so (C-ABI declared in include/pio_hip.h).

The library is built in-tree (``perceiverio_pytorch_amd/libpio_hip.so``) by ``__graft_entry__.build()``
or ``make -C perceiverio_pytorch_amd/csrc``.  There is NO fallback: if the shared object is missing or
fails to load, importing the compute modules raises -- the hot path only exists as HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIO_LIB_PATH", os.path.join(_HERE, "libpio_hip.so"))  # override: A/B builds only
CSRC = os.path.join(_HERE, "csrc")

PIO_DT_F16 = 0
PIO_DT_BF16 = 1

PIO_OK = 0
_ERRORS = {-1: "PIO_E_SHAPE", -2: "PIO_E_ALIGN", -3: "PIO_E_ARCH", -4: "PIO_E_WORKSPACE", -5: "PIO_E_LAUNCH",
           -6: "PIO_E_ARG", -7: "PIO_E_RANGE"}


class PioError(RuntimeError):
    pass


class Linear(C.Structure):
    _fields_ = [("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("bias", C.c_void_p), ("n", C.c_int32), ("k", C.c_int32),
                ("lo_row0", C.c_int32)]


class LayerNorm(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("c", C.c_int32), ("eps", C.c_float)]


class Attention(C.Structure):
    _fields_ = [("q", Linear), ("k", Linear), ("v", Linear), ("o", Linear), ("heads", C.c_int32),
                ("dk", C.c_int32), ("dv", C.c_int32), ("dkp", C.c_int32), ("dvp", C.c_int32),
                ("q_in", C.c_int32), ("k_in", C.c_int32), ("v_in", C.c_int32), ("out", C.c_int32), ("dtype", C.c_int32),
                ("act_split", C.c_int32), ("qk", Linear), ("qkv", Linear), ("kq", Linear), ("vo", Linear)]


class Mlp(C.Structure):
    _fields_ = [("fc1", Linear), ("fc2", Linear), ("in_", C.c_int32), ("hidden", C.c_int32), ("out", C.c_int32),
                ("dtype", C.c_int32), ("act_split", C.c_int32)]


class LnFold(C.Structure):
    _fields_ = [("qkv", Linear), ("qkv_c", C.c_void_p), ("fc1", Linear), ("fc1_c", C.c_void_p),
                ("range_flag", C.c_void_p)]


class SelfAttention(C.Structure):
    _fields_ = [("ln1", LayerNorm), ("ln2", LayerNorm), ("attn", Attention), ("mlp", Mlp), ("fold", LnFold)]


class CrossAttention(C.Structure):
    _fields_ = [("ln_q", LayerNorm), ("ln_kv", LayerNorm), ("ln2", LayerNorm), ("attn", Attention), ("mlp", Mlp),
                ("use_query_residual", C.c_int32)]


class Tensor3(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride_b", C.c_int64), ("stride_t", C.c_int64), ("B", C.c_int32),
                ("T", C.c_int32), ("C", C.c_int32)]


class CallOpts(C.Structure):
    """pio_call_opts_t: per-call (thread-local inside the library) LayerNorm-fold mode and CU budget."""
    _fields_ = [("ln_fold", C.c_int32), ("cu_budget", C.c_int32)]


class Gemm(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("A_lo", C.c_void_p), ("B_lo", C.c_void_p), ("C", C.c_void_p),
                ("C_lo", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
                ("batch", C.c_int32), ("nh", C.c_int32),
                ("sAb", C.c_int64), ("sAh", C.c_int64), ("sBb", C.c_int64), ("sBh", C.c_int64),
                ("sCb", C.c_int64), ("sCh", C.c_int64),
                ("bias", C.c_void_p), ("bias_mode", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float),
                ("R", C.c_void_p), ("ldr", C.c_int64), ("r_stride_b", C.c_int64),
                ("r_rows_per_batch", C.c_int32), ("out_f32", C.c_int32), ("n_store", C.c_int32),
                ("dtype", C.c_int32),
                ("X16", C.c_void_p), ("ld16", C.c_int64), ("row_part", C.c_void_p), ("ln_part", C.c_void_p),
                ("ln_c", C.c_void_p), ("ln_eps", C.c_float),
                ("X16_lo", C.c_void_p), ("R16_hi", C.c_void_p), ("R16_lo", C.c_void_p), ("b_lo_n0", C.c_int32),
                ("range_flag", C.c_void_p), ("ln_slots", C.c_int32), ("row_slot_w", C.c_int32)]


# name -> (restype, argtypes); must list EVERY function declared in include/pio_hip.h
# (tests/test_capi_symbols.py parses the header and checks this table and the .so against it).
P = C.POINTER
_vp, _i32, _i64, _sz, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float
SIGNATURES = {
    "pio_version": (C.c_int, []),
    "pio_arch_ok": (C.c_int, []),
    "pio_error_string": (C.c_char_p, [C.c_int]),
    "pio_prof_begin": (C.c_int, [_i32]),
    "pio_prof_end": (C.c_int, [P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_int64)]),
    "pio_pad8": (_i32, [_i32]),
    "pio_padc": (_i32, [_i32]),
    "pio_gemm_kernel_override": (C.c_int, [C.c_int]),
    "pio_ln_fold_enable": (C.c_int, [C.c_int]),
    "pio_stream_create_cu_mask": (C.c_int, [P(_vp), P(C.c_uint32), C.c_uint32]),
    "pio_stream_destroy": (C.c_int, [_vp]),
    "pio_set_cu_budget": (C.c_int, [_i32]),
    "pio_packed_weight_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "pio_pack_linear": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "pio_layernorm_cast": (C.c_int, [P(Tensor3), P(LayerNorm), _vp, _vp, _i32, _i32, _vp]),
    "pio_layernorm_cast_cat": (C.c_int, [P(Tensor3), P(Tensor3), P(LayerNorm), _vp, _vp, _i32, _i32, _vp]),
    "pio_bn_relu_maxpool_tokens": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pio_gemm_nt": (C.c_int, [P(Gemm), _vp]),
    "pio_softmax_rows": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f, _vp, _vp, _vp, _vp, _i32, _vp]),
    "pio_flash_attention": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64,
                                      _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "pio_attention_workspace_bytes": (_sz, [P(Attention), _i32, _i32, _i32]),
    "pio_attention_fwd": (C.c_int, [P(Attention), P(Tensor3), P(Tensor3), P(Tensor3), _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _sz, _vp]),
    "pio_mlp_workspace_bytes": (_sz, [P(Mlp), _i64]),
    "pio_mlp_fwd": (C.c_int, [P(Mlp), P(Tensor3), _vp, _vp, _sz, _vp]),
    "pio_self_attention_workspace_bytes": (_sz, [P(SelfAttention), _i32, _i32]),
    "pio_self_attention_fwd": (C.c_int, [P(SelfAttention), P(Tensor3), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pio_self_attention_fwd_opts": (C.c_int, [P(SelfAttention), P(Tensor3), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp,
                                              P(CallOpts)]),
    "pio_cross_attention_workspace_bytes": (_sz, [P(CrossAttention), _i32, _i32, _i32]),
    "pio_cross_attention_fwd": (C.c_int, [P(CrossAttention), P(Tensor3), P(Tensor3), _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _sz, _vp]),
    "pio_encoder_workspace_bytes": (_sz, [P(CrossAttention), P(SelfAttention), _i32, _i32, _i32, _i32]),
    "pio_encoder_fwd": (C.c_int, [P(CrossAttention), P(SelfAttention), _i32, _i32, P(Tensor3), P(Tensor3), _vp, _vp,
                                  _vp, _sz, _vp]),
    "pio_encoder_fwd_split": (C.c_int, [P(CrossAttention), P(SelfAttention), _i32, _i32, P(Tensor3), P(Tensor3),
                                        P(Tensor3), _vp, _vp, _vp, _sz, _vp]),
    "pio_encoder_fwd_blocks": (C.c_int, [P(CrossAttention), P(SelfAttention), _i32, _i32, _i32, P(Tensor3), P(Tensor3),
                                         P(Tensor3), _vp, _vp, _vp, _sz, _vp]),
    "pio_encoder_fwd_opts": (C.c_int, [P(CrossAttention), P(SelfAttention), _i32, _i32, _i32, P(Tensor3), P(Tensor3),
                                       P(Tensor3), _vp, _vp, _vp, _sz, _vp, P(CallOpts)]),
    "pio_decoder_workspace_bytes": (_sz, [P(CrossAttention), P(Linear), _i32, _i32, _i32]),
    "pio_decoder_fwd": (C.c_int, [P(CrossAttention), P(Linear), _i32, P(Tensor3), P(Tensor3), _vp, _vp, _vp, _sz,
                                  _vp]),
    "pio_decoder_qcache_bytes": (_sz, [P(CrossAttention), _i32, _i32]),
    "pio_decoder_fwd_qcache": (C.c_int, [P(CrossAttention), P(Linear), _i32, P(Tensor3), P(Tensor3), _vp, _vp, _vp, _sz,
                                         _vp, _vp, _vp, _i32]),
    "pio_decoder_fwd_split": (C.c_int, [P(CrossAttention), P(Linear), _i32, P(Tensor3), P(Tensor3), P(Tensor3), _vp, _vp, _vp,
                                        _sz, _vp]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile libpio_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or out.returncode:
        print(out.stdout[-4000:])
        print(out.stderr[-4000:])
    if out.returncode:
        raise PioError("building libpio_hip.so failed")
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library with argtypes set.  Raises loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PioError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU / eager fallback for the PerceiverIO hot path)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)      # AttributeError => stale .so: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code: int, what: str = "") -> None:
    if code != PIO_OK:
        raise PioError(f"{what}: {_ERRORS.get(code, code)}")
