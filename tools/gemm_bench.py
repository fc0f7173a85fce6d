"""Micro-benchmark of pio_gemm_nt on the shapes of the ImageNet self-attend layer (dev tool)."""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from perceiverio_pytorch_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")


def run(M, N, K, out_f32, resid, act, lo=False, iters=30, batch=1):
    A = torch.randn(batch, M, K, device=dev).half()
    B = (torch.randn(batch, N, K, device=dev) / K ** 0.5).half()
    Bl = (torch.randn(batch, N, K, device=dev) * 1e-4).half()
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev)
    Cc = torch.empty(batch, M, N, device=dev, dtype=torch.float32 if out_f32 else torch.float16)
    g = L.Gemm()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    g.B_lo = Bl.data_ptr() if lo else None
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = K, K, N
    g.batch, g.nh = batch, 1
    g.sAb, g.sBb, g.sCb = M * K, N * K, M * N
    g.bias, g.bias_mode, g.act, g.alpha = bias.data_ptr(), 1, act, 1.0
    if resid:
        g.R, g.ldr = R.data_ptr(), N
    g.out_f32, g.n_store, g.dtype = int(out_f32), N, L.PIO_DT_F16
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
    tf = 2.0 * M * N * K * batch / us / 1e6
    print(f"M={M} N={N} K={K} b={batch} f32={out_f32} R={resid} act={act} lo={lo}: {us:8.1f} us  {tf:7.1f} TF/s", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1:
    M, N, K = (int(v) for v in sys.argv[1:4])
    run(M, N, K, False, False, 0, iters=int(sys.argv[4]) if len(sys.argv) > 4 else 10)
elif __name__ == "__main__":
    run(16384, 1024, 1024, False, False, 0)
    run(16384, 1024, 1024, True, True, 0)
    run(16384, 1024, 1024, False, False, 1)
    run(16384, 1024, 1024, True, False, 0)
    run(16384, 1024, 1024, False, False, 0, lo=True)
    run(16384, 3072, 1024, False, False, 0)
    run(4096, 4096, 4096, False, False, 0)
    run(8192, 8192, 8192, False, False, 0, iters=5)
    run(1024, 512, 1024, False, False, 0, batch=32)
