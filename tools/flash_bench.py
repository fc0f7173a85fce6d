"""Per-kernel timing of one latent self-attend layer (SelfAttention module, ImageNet shape 32 x 512 x 1024, 8 heads); prints the per-class profiler numbers (dev tool; A/B two builds with PIO_LIB_PATH)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import perceiverio_pytorch_amd as P
from perceiverio_pytorch_amd import _lib as L
from perceiverio_pytorch_amd.transformer_primitives import SelfAttention

lib = L.lib()
dev = torch.device("cuda:0")
B, T, D, H = 32, 512, 1024, 8
m = SelfAttention(D, num_heads=H, widening_factor=1).to(dev).eval()
x = torch.randn(B, T, D, device=dev)
P.set_precision_policy("fp16")
with torch.inference_mode():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    L.check(lib.pio_prof_begin(4096))
    for _ in range(20):
        m(x)
    ms = (C.c_double * 9)(); fl = (C.c_double * 9)(); by = (C.c_double * 9)(); ln = (C.c_int64 * 9)()
    lib.pio_prof_end(ms, fl, by, ln)
names = ["gemm256", "gemm128b", "ln", "softmax", "pack", "flash", "gemm128", "stream"]
print(" ".join(f"{n}={ms[i] / ln[i] * 1e3:.1f}us(x{ln[i]})" for i, n in enumerate(names) if ln[i]), flush=True)
