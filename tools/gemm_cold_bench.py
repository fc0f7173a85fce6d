"""Dev tool: the same GEMM back to back on ONE set of buffers (operands stay in L2 / MALL) against a rotation over
enough buffer sets to exceed the 256 MB Infinity Cache -- how much of the in-model slowdown of a kernel is cold operands?"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from perceiverio_pytorch_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")


def make(M, N, K, out_f32, resid, act):
    A = torch.randn(M, K, device=dev).half()
    B = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if resid else None
    Cc = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else torch.float16)
    g = L.Gemm()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = K, K, N
    g.batch, g.nh = 1, 1
    g.bias, g.bias_mode, g.act, g.alpha = bias.data_ptr(), 1, act, 1.0
    if resid:
        g.R, g.ldr = R.data_ptr(), N
    g.out_f32, g.n_store, g.dtype = int(out_f32), N, L.PIO_DT_F16
    return g, (A, B, bias, R, Cc)


def run(M, N, K, kernel, nsets, out_f32=False, resid=False, act=0, iters=48):
    sets = [make(M, N, K, out_f32, resid, act) for _ in range(nsets)]
    st = torch.cuda.current_stream().cuda_stream
    prev = lib.pio_gemm_kernel_override(kernel)
    try:
        for i in range(nsets):
            L.check(lib.pio_gemm_nt(C.byref(sets[i][0]), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            lib.pio_gemm_nt(C.byref(sets[i % nsets][0]), st)
        e1.record()
        torch.cuda.synchronize()
    finally:
        lib.pio_gemm_kernel_override(prev)
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"kernel={kernel} sets={nsets:2d} M={M} N={N} K={K} f32={int(out_f32)} R={int(resid)} act={act}: {us:7.1f} us "
          f"{2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__" and "--after-ln" not in sys.argv:
    for kernel in (1, 2):
        for nsets in (1, 8):
            run(16384, 3072, 1024, kernel, nsets)
            run(16384, 1024, 1024, kernel, nsets, act=1)
            run(16384, 1024, 1024, kernel, nsets, out_f32=True, resid=True)


def run_after_ln(M, N, K, kernel, act, iters=40):
    """The GEMM timed alone (events around each launch) when an unrelated kernel (LayerNorm + cast producing its A
    operand) runs before every launch -- as in the model -- against the same GEMM back to back."""
    from perceiverio_pytorch_amd import runtime as Rt
    g, keep = make(M, N, K, False, False, act)
    x = torch.randn(1, M, K, device=dev)
    gamma = torch.ones(K, device=dev)
    beta = torch.zeros(K, device=dev)
    ln = L.LayerNorm(gamma.data_ptr(), beta.data_ptr(), K, 1e-5)
    st = torch.cuda.current_stream().cuda_stream
    prev = lib.pio_gemm_kernel_override(kernel)
    try:
        for mode in ("back to back", "after LayerNorm"):
            tot = 0.0
            for i in range(iters + 3):
                if mode == "after LayerNorm":
                    L.check(lib.pio_layernorm_cast(Rt.tensor3(x), C.byref(ln), keep[0].data_ptr(), None, K, L.PIO_DT_F16, st))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.pio_gemm_nt(C.byref(g), st)
                e1.record()
                torch.cuda.synchronize()
                if i >= 3:
                    tot += e0.elapsed_time(e1)
            print(f"kernel={kernel} M={M} N={N} K={K} act={act} [{mode}]: {tot / iters * 1e3:7.1f} us", flush=True)
    finally:
        lib.pio_gemm_kernel_override(prev)


if __name__ == "__main__" and "--after-ln" in sys.argv:
    for kernel in (1, 2):
        run_after_ln(16384, 1024, 1024, kernel, 1)
        run_after_ln(16384, 1024, 1024, kernel, 0)
        run_after_ln(16384, 3072, 1024, kernel, 0)
