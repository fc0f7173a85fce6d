"""Dev tool: the LayerNorm fold's producer GEMM (16384 x 1024 x 1024, residual and result as 16-bit pairs + row sums) on
gemm_nt_wide (override 0) against gemm_nt_duo (override 3; env PIO_DUO_STAGGER = cycles, 0 = no stagger)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from perceiverio_pytorch_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")
M, N, K = 16384, 1024, int(os.environ.get('PB_K', '1024'))
A = torch.randn(M, K, device=dev).half()
W = (torch.randn(N, K, device=dev) / K ** 0.5).half()
b = torch.randn(N, device=dev)
R = torch.randn(M, N, device=dev)
Rh = R.half(); Rl = (R - Rh.float()).half()
Xh = torch.empty(M, N, device=dev, dtype=torch.float16); Xl = torch.empty_like(Xh)
part = torch.empty(M, N // 128, 2, device=dev)
g = L.Gemm()
g.A, g.B, g.C = A.data_ptr(), W.data_ptr(), None
g.M, g.N, g.K = M, N, K
g.lda, g.ldb, g.ldc = K, K, N
g.batch, g.nh = 1, 1
g.bias, g.bias_mode, g.act, g.alpha = b.data_ptr(), 1, 0, 1.0
g.out_f32, g.n_store, g.dtype = 1, N, L.PIO_DT_F16
g.X16, g.ld16, g.row_part = Xh.data_ptr(), N, part.data_ptr()
g.X16_lo, g.R16_hi, g.R16_lo = Xl.data_ptr(), Rh.data_ptr(), Rl.data_ptr()
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for kernel in (0, 3):
        prev = lib.pio_gemm_kernel_override(kernel)
        for _ in range(3):
            L.check(lib.pio_gemm_nt(C.byref(g), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.pio_gemm_nt(C.byref(g), st)
        e1.record()
        torch.cuda.synchronize()
        lib.pio_gemm_kernel_override(prev)
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"kernel={kernel} stagger={os.environ.get('PIO_DUO_STAGGER', 'default')}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s",
              flush=True)
