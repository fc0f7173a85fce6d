"""Does a power-of-two leading dimension (row stride 2 KiB) throttle the operand DMA stream?  A/B of pio_gemm_nt
with lda = ldb = K + pad (dev tool)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from perceiverio_pytorch_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")


def run(M, N, K, pad, kernel, out_f32=False, resid=False, act=0, iters=30):
    A = torch.randn(M, K + pad, device=dev).half()
    B = (torch.randn(N, K + pad, device=dev) / K ** 0.5).half()
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev)
    Cc = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else torch.float16)
    g = L.Gemm()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = K + pad, K + pad, N
    g.batch, g.nh = 1, 1
    g.bias, g.bias_mode, g.act, g.alpha = bias.data_ptr(), 1, act, 1.0
    if resid:
        g.R, g.ldr = R.data_ptr(), N
    g.out_f32, g.n_store, g.dtype = int(out_f32), N, L.PIO_DT_F16
    st = torch.cuda.current_stream().cuda_stream
    prev = lib.pio_gemm_kernel_override(kernel)
    try:
        for _ in range(3):
            L.check(lib.pio_gemm_nt(C.byref(g), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.pio_gemm_nt(C.byref(g), st)
        e1.record()
        torch.cuda.synchronize()
    finally:
        lib.pio_gemm_kernel_override(prev)
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"kernel={kernel:3d} M={M} N={N} K={K} pad={pad:3d} f32={int(out_f32)} R={int(resid)} act={act}: {us:8.1f} us "
          f"{2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__" and "--wide" in sys.argv:
    for kernel in (1, 2, 1, 2):
        run(16384, 3072, 1024, 0, kernel)
        run(16384, 1024, 1024, 0, kernel)
        run(16384, 1024, 1024, 0, kernel, act=1)
        run(16384, 1024, 1024, 0, kernel, out_f32=True, resid=True)
        run(16384, 1024, 1024, 0, kernel, out_f32=True)
        run(8192, 8192, 8192, 0, kernel, iters=5)
        run(16384, 1024, 4096, 0, kernel)
    sys.exit(0)

if __name__ == "__main__":
    for kernel in (256, 1):
        pads = (0, 64) if "--pads" in sys.argv else (0,)
        for pad in pads:
            run(16384, 1024, 1024, pad, kernel)
            run(16384, 1024, 1024, pad, kernel, act=1)
            run(16384, 3072, 1024, pad, kernel)
            run(16384, 1024, 1024, pad, kernel, out_f32=True, resid=True)
            run(16384, 1024, 1024, pad, kernel, out_f32=True)
            run(4096, 4096, 4096, pad, kernel)
            run(16384, 1024, 4096, pad, kernel)
