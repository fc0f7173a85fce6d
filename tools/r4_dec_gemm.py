"""Dev tool (GPU): the flow decoder's big-M / shallow-K projections (182 528 rows, K = 328, split activations) under each
tile family (pio_gemm_kernel_override: 0 automatic, 64, 128, 256)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from perceiverio_pytorch_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")


def run(M, N, K, act, pair_out, resid, iters=20):
    A = torch.randn(M, K, device=dev).half()
    Al = (torch.randn(M, K, device=dev) * 1e-3).half()
    B = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    bias = torch.randn(N, device=dev)
    ldc = (N + 7) // 8 * 8
    Cc = torch.empty(M, ldc, device=dev, dtype=torch.float16)
    Cl = torch.empty(M, ldc, device=dev, dtype=torch.float16)
    R = torch.randn(M, N, device=dev) if resid else None
    g = L.Gemm()
    g.A, g.B, g.C, g.A_lo = A.data_ptr(), B.data_ptr(), Cc.data_ptr(), Al.data_ptr()
    g.C_lo = Cl.data_ptr() if pair_out else None
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = K, K, ldc
    g.batch, g.nh = 1, 1
    g.bias, g.bias_mode, g.act, g.alpha = bias.data_ptr(), 1, act, 1.0
    if resid:
        g.R, g.ldr = R.data_ptr(), N
    g.out_f32, g.n_store, g.dtype = 0, ldc, L.PIO_DT_F16
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for ov in (0, 1, 2, 256):   # automatic, stream, wide, 256 x 256
        lib.pio_gemm_kernel_override(ov)
        try:
            for _ in range(2):
                L.check(lib.pio_gemm_nt(C.byref(g), st))
        except Exception as e:  # noqa: BLE001
            res.append(f"{ov}: {type(e).__name__}")
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.pio_gemm_nt(C.byref(g), st)
        e1.record()
        torch.cuda.synchronize()
        res.append(f"{ov}: {e0.elapsed_time(e1) / iters * 1e3:7.1f} us")
    lib.pio_gemm_kernel_override(0)
    print(f"M={M} N={N} K={K} act={act} pair_out={pair_out} resid={resid}:  " + "   ".join(res), flush=True)


if "--flow384" in sys.argv:
    run(182528, 384, 384, 1, True, False)      # fc1 with 322 padded to 384 (PIO_PADC_MIN=256)
    run(182528, 384, 384, 0, True, True)       # fc2
    run(182528, 512, 384, 0, False, False)     # proj_q
    sys.exit(0)
if "--flow" in sys.argv:
    run(182528, 512, 328, 0, False, False)     # proj_q
    run(182528, 328, 328, 1, True, False)      # fc1 (GELU, pair out)
    run(182528, 328, 328, 0, True, True)       # fc2 (residual, pair out)
run(104192, 1028, 1088, 1, True, False)    # multimodal decoder fc1 (GELU, pair out)
run(104192, 1028, 512, 0, False, False)    # ... a K = 512 projection, one 16-bit result
run(32000, 1024, 1024, 1, True, False)     # ImageNet decoder fc1
run(25600, 1280, 1280, 1, True, False)     # language-sized
