"""CPU: libpio_hip.so loads and exports EVERY function include/pio_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pio_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(pio_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_t")))


def test_header_declares_something():
    fns = declared_functions()
    assert len(fns) >= 20 and "pio_encoder_fwd" in fns and "pio_gemm_nt" in fns


def test_library_exports_every_declared_symbol():
    from perceiverio_pytorch_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        L.build()
    lib = ctypes.CDLL(L.LIB_PATH)
    missing = [f for f in declared_functions() if not hasattr(lib, f)]
    assert not missing, f"declared in pio_hip.h but not exported: {missing}"


def test_ctypes_table_matches_header():
    from perceiverio_pytorch_amd import _lib as L
    assert sorted(L.SIGNATURES) == declared_functions()


def test_host_side_queries_need_no_gpu():
    from perceiverio_pytorch_amd import _lib as L
    lib = L.lib()
    assert lib.pio_version() == 100
    assert lib.pio_pad8(322) == 328 and lib.pio_pad8(8) == 8
    assert [lib.pio_padc(c) for c in (96, 250, 322, 504, 512, 1026, 1280)] == [96, 256, 384, 512, 512, 1088, 1280]
    assert lib.pio_packed_weight_bytes(1024, 1024, 8, 1) == 1024 * 1024 * 2
    assert lib.pio_packed_weight_bytes(322, 322, 1, 1) == 328 * 328 * 2
    assert lib.pio_packed_weight_bytes(10, 10, 3, 1) == 0          # not divisible by the head count
    assert lib.pio_error_string(-4) == b"workspace too small"


def test_struct_sizes_match_the_c_layout():
    """sizeof() of the ctypes mirrors against the values a C compiler gives for include/pio_hip.h."""
    import subprocess
    import tempfile
    from perceiverio_pytorch_amd import _lib as L
    prog = r'''
#include <stdio.h>
#include "pio_hip.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(pio_linear_t), sizeof(pio_layernorm_t),
 sizeof(pio_attention_t), sizeof(pio_mlp_t), sizeof(pio_self_attention_t), sizeof(pio_cross_attention_t),
 sizeof(pio_tensor3_t), sizeof(pio_gemm_t)); return 0;}
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    mirrors = [L.Linear, L.LayerNorm, L.Attention, L.Mlp, L.SelfAttention, L.CrossAttention, L.Tensor3, L.Gemm]
    assert sizes == [ctypes.sizeof(m) for m in mirrors]
