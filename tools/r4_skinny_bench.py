"""Dev tool (GPU): the flow decoder's final Linear (182 528 x 328 -> 2, split activations) on the row-per-wave kernel
against the 128 x 128 tile (PIO_GEMM_SKINNY=0 in a second process)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from perceiverio_pytorch_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")


def run(M, N, K, a_lo, iters=50):
    A = torch.randn(M, K, device=dev).half()
    Al = (torch.randn(M, K, device=dev) * 1e-3).half()
    B = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    bias = torch.randn(N, device=dev)
    Cc = torch.empty(M, N, device=dev, dtype=torch.float32)
    g = L.Gemm()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    g.A_lo = Al.data_ptr() if a_lo else None
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = K, K, N
    g.batch, g.nh = 1, 1
    g.bias, g.bias_mode, g.act, g.alpha = bias.data_ptr(), 1, 0, 1.0
    g.out_f32, g.n_store, g.dtype = 1, N, L.PIO_DT_F16
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.check(lib.pio_gemm_nt(C.byref(g), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.pio_gemm_nt(C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    gb = M * K * 2 * (2 if a_lo else 1) / us / 1e3
    print(f"M={M} N={N} K={K} a_lo={a_lo}: {us:8.1f} us  ({gb:7.1f} GB/s of A)  PIO_GEMM_SKINNY={os.environ.get('PIO_GEMM_SKINNY', '1')}", flush=True)


run(182528, 2, 328, True)
run(182528, 2, 328, False)
run(100352, 3, 512, True)
