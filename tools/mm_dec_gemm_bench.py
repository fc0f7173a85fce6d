"""Dev tool (GPU): the five weight GEMMs of one multimodal decoder call (25 149 query rows x 1026 channels, split
activations: policy fp16x2af) on each eligible kernel, with the channel pitch as it is (1032) and rounded to 64 (1088)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from perceiverio_pytorch_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")


def run(tag, M, N, K, out_f32, resid, act, n_store, ldc, override, iters=20):
    A = torch.randn(M, K, device=dev).half()
    Al = (torch.randn(M, K, device=dev) * 1e-4).half()
    B = (torch.randn(n_store, K, device=dev) / K ** 0.5).half()
    bias = torch.randn(n_store, device=dev)
    R = torch.randn(M, ldc, device=dev)
    Cc = torch.empty(M, ldc, device=dev, dtype=torch.float32 if out_f32 else torch.float16)
    g = L.Gemm()
    g.A, g.B, g.C, g.A_lo = A.data_ptr(), B.data_ptr(), Cc.data_ptr(), Al.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = K, K, ldc
    g.batch, g.nh = 1, 1
    g.bias, g.bias_mode, g.act, g.alpha = bias.data_ptr(), 1, act, 1.0
    if resid:
        g.R, g.ldr = R.data_ptr(), ldc
    g.out_f32, g.n_store, g.dtype = int(out_f32), n_store, L.PIO_DT_F16
    st = torch.cuda.current_stream().cuda_stream
    prev = lib.pio_gemm_kernel_override(override)
    try:
        rc = lib.pio_gemm_nt(C.byref(g), st)
        if rc != 0:
            print(f"{tag}: override {override}: rc={rc}", flush=True)
            return
        for _ in range(2):
            lib.pio_gemm_nt(C.byref(g), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.pio_gemm_nt(C.byref(g), st)
        e1.record()
        torch.cuda.synchronize()
    finally:
        lib.pio_gemm_kernel_override(prev)
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{tag:34s} N={N:5d} n_store={n_store:5d} K={K:5d} f32={int(out_f32)} R={int(resid)} act={act} override={override:3d}: "
          f"{us:7.1f} us  {4.0 * M * N * K / us / 1e6:7.1f} TF/s (both sweeps)", flush=True)


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 25149
    for ov in ([0] if len(sys.argv) > 1 else [0, 256, 1]):
        for K1, tag in ((1032, "pitch 1032"), (1088, "pitch 1088")):
            run(f"q proj [{tag}]", M, 512, K1, False, False, 0, 512, 512, ov)
            run(f"fc1 GELU [{tag}]", M, K1 if K1 == 1088 else 1032, K1, False, False, 1, K1, K1, ov)
            run(f"fc2 + residual [{tag}]", M, 1026, K1, True, True, 0, 1026, 1026, ov)
        run("attention out", M, 1026, 512, True, False, 0, 1026, 1026, ov)
        run("attention out, ldc 1028", M, 1028, 512, True, False, 0, 1028, 1028, ov)
        run("fc2 + residual, ldc 1028, K 1088", M, 1028, 1088, True, True, 0, 1028, 1028, ov)
        run("fc2 + residual, ldc 1088, K 1088", M, 1088, 1088, True, True, 0, 1088, 1088, ov)


if __name__ == "__main__":
    main()
