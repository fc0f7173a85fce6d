"""GPU tests of the EXPERIMENT kernels (not part of libpio_hip.so): the flash_attn_pipe / flash_attn_stag variants of the
fused self-attention kernel and the MFMA 32x32x16 variants of the LayerNorm-fold GEMMs, against the shipped kernels.

    make -C perceiverio_pytorch_amd/csrc experiments
    PIO_LIB_PATH=tools/_abl/libpio_hip_exp.so python -m pytest tools/experiments/test_experiments.py -q   (GPU box)
"""
import ctypes as C
import os
import sys

import pytest
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT]
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import perceiverio_pytorch_amd as P
    lib = P.lib()
    if not hasattr(lib, "pio_debug_flash_variant"):
        pytest.skip("needs the experiments build: PIO_LIB_PATH=tools/_abl/libpio_hip_exp.so")
    lib.pio_debug_flash_variant.argtypes = [C.c_int]
    lib.pio_debug_flash_variant.restype = C.c_int
    return torch.device("cuda:0")


def _policy(name):
    import perceiverio_pytorch_amd as P
    P.set_precision_policy(name)


@pytest.mark.parametrize("policy", ["fp16", "bf16"])
@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("shape", [(2, 256), (3, 512), (2, 1024), (2, 777), (16, 640)])
def test_self_attention_kernel_variants_agree(dev, variant, shape, policy):
    """pio_debug_flash_variant: the one-wave-per-SIMD pipelined kernel (1) and the staggered-groups kernel (2) against
    the default lock-step kernel (0) on the hot shape (1024-channel SelfAttention block, 8 heads of 128, V row-major):
    same operands, same P rounding (16 bit), different summation order / softmax reference point -> a few 1e-4 of the
    output scale.  Shapes a variant does not take (odd tile counts, ragged key tails) fall back to variant 0: equal."""
    import perceiverio_pytorch_amd as P
    from perceiverio_pytorch_amd import _lib as L
    from perceiverio_pytorch_amd.transformer_primitives import SelfAttention
    lib = L.lib()
    _policy(policy)
    B, T = shape
    torch.manual_seed(T)
    m = SelfAttention(1024, widening_factor=1, num_heads=8).to(dev).eval()
    x = torch.randn(B, T, 1024, device=dev) * 1.5 + 0.2
    prev = lib.pio_debug_flash_variant(0)
    try:
        y0 = m(x).double()
        assert lib.pio_debug_flash_variant(variant) == 0
        y1 = m(x).double()
    finally:
        lib.pio_debug_flash_variant(prev)
    assert lib.pio_debug_flash_variant(7) == prev and lib.pio_debug_flash_variant(-2) == prev  # others only read
    scale = y0.abs().max()
    k = 1.0 if policy == "fp16" else 8.0       # bf16 P: 8 mantissa bits instead of 11
    assert ((y1 - y0).abs().max() / scale).item() <= 3e-4 * k, f"variant {variant} {shape}"
    assert ((y1 - y0).norm() / y0.norm()).item() <= 2e-4 * k


@pytest.mark.parametrize("policy", ["fp16", "fp16x2w"])
def test_fold_gemms_on_32x32_mfma_agree(dev, policy):
    """pio_gemm_kernel_override(4): the LayerNorm fold's consumer (q|k|v, fc1 + GELU) and staged producer (out, fc2) on
    the MFMA 32x32x16 variants of gemm_nt_wide against the default 16x16x32 ones -- same operands, same K order per
    accumulator, different accumulator / epilogue lane mapping: agreement to fp32 rounding of the epilogues."""
    from perceiverio_pytorch_amd import _lib as L
    from perceiverio_pytorch_amd.transformer_primitives import SelfAttention
    lib = L.lib()
    _policy(policy)
    torch.manual_seed(9)
    m = SelfAttention(1024, widening_factor=1, num_heads=8).to(dev).eval()
    x = torch.randn(8, 512, 1024, device=dev) * 1.5 + 0.2
    prev = lib.pio_gemm_kernel_override(0)
    try:
        y0 = m(x).double()
        lib.pio_gemm_kernel_override(4)
        y1 = m(x).double()
    finally:
        lib.pio_gemm_kernel_override(prev)
    assert not torch.equal(y0, y1), "override 4 did not change the kernels"
    assert ((y1 - y0).abs().max() / y0.abs().max()).item() <= 2e-4
    assert ((y1 - y0).norm() / y0.norm()).item() <= 5e-5
