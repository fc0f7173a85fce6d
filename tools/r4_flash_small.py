"""Dev tool (GPU): the fused self-attention kernel on small launches (few workgroups), against the key count: slope = one
64-key tile, intercept = launch + prologue + merge + epilogue.  Q | K | V as the fused q|k|v GEMM leaves them (row-major V)."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
from perceiverio_pytorch_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")


def run(B, H, Tq, Tk, d, iters=200):
    T = max(Tq, Tk)
    qkv = torch.randn(B, T, 3 * H * d, device=dev).half()
    out = torch.empty(B, Tq, H * d, device=dev, dtype=torch.float16)
    ld = 3 * H * d
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 2 * H * d, qkv.data_ptr() + 4 * H * d
    st = torch.cuda.current_stream().cuda_stream
    args = (L.PIO_DT_F16, d, d, d, q, k, v, out.data_ptr(), B, H, Tq, Tk, ld, ld, ld, H * d, T * ld, T * ld, T * ld,
            Tq * H * d, 1, st)
    for _ in range(3):
        L.check(lib.pio_flash_attention(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.pio_flash_attention(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"B={B} H={H} Tq={Tq} Tk={Tk} d={d}: {us:7.1f} us", flush=True)


print({k: v for k, v in os.environ.items() if k.startswith("PIO_")})
for Tk in (128, 256, 512, 1024, 2048):
    run(1, 8, 512, Tk, 128)
for B in (2, 4, 8):
    run(B, 8, 512, 512, 128)
run(1, 16, 2048, 2048, 32)
run(1, 16, 2048, 1024, 32)
run(1, 8, 1024, 1024, 64)
