"""Dev tool (GPU): device time of the fused SELF-attention kernel on the hot shape (B x 512 latents x 8 heads of 128,
V row-major out of the fused q|k|v GEMM).  PIO_FLASH_PIPE=0 selects the lock-step 8-wave kernel for comparison."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import perceiverio_pytorch_amd as P  # noqa: E402
from perceiverio_pytorch_amd import _lib as L  # noqa: E402
from perceiverio_pytorch_amd.transformer_primitives import SelfAttention  # noqa: E402


def main():
    lib = P.lib()
    dev = torch.device("cuda:0")
    P.set_precision_policy("fp16")
    for B, T in ((32, 512), (32, 1024), (4, 512)):
        m = SelfAttention(1024, widening_factor=1, num_heads=8).to(dev).eval()
        x = torch.randn(B, T, 1024, device=dev)
        with torch.no_grad():
            for _ in range(3):
                y = m(x)
            torch.cuda.synchronize()
            n = 20
            L.check(lib.pio_prof_begin(4096))
            for _ in range(n):
                m(x)
            ms = (C.c_double * 9)(); fl = (C.c_double * 9)(); by = (C.c_double * 9)(); ln = (C.c_int64 * 9)()
            lib.pio_prof_end(ms, fl, by, ln)
        us = ms[5] / n * 1e3
        print("   per class us per forward:", {k: round(ms[k] / n * 1e3, 1) for k in range(9) if ms[k] > 0}, flush=True)
        print(f"B={B} T={T}: fused self-attention {us:8.1f} us ({fl[5] / n / (us * 1e-6) / 1e12:7.1f} algorithmic "
              f"TFLOP/s, {ln[5] // n} launches) PIO_FLASH_PIPE={os.environ.get('PIO_FLASH_PIPE', '1')} "
              f"checksum {float(y.double().abs().mean()):.6f}", flush=True)


if __name__ == "__main__":
    main()
