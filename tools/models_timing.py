"""Forward time and per-kernel-class breakdown of the four task models at their example sizes (dev tool, GPU)."""
import ctypes as C
import os
import sys
import time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."),
                os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")]
import torch
import perceiverio_pytorch_amd as P
from perceiverio_pytorch_amd import _lib as L
from perceiverio_pytorch_amd import models as M

lib = L.lib()
dev = torch.device("cuda:0")
NAMES = ["gemm256", "gemm128b", "ln", "softmax", "pack", "flash", "gemm128", "stream"]


def run(name, model, args, kwargs=None, n=3):
    kwargs = kwargs or {}
    with torch.inference_mode():
        for _ in range(2):
            model(*args, **kwargs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(*args, **kwargs)
        torch.cuda.synchronize()
        ms_tot = (time.perf_counter() - t0) / n * 1e3
        L.check(lib.pio_prof_begin(20000))
        model(*args, **kwargs)
        ms = (C.c_double * 9)(); fl = (C.c_double * 9)(); by = (C.c_double * 9)(); ln = (C.c_int64 * 9)()
        lib.pio_prof_end(ms, fl, by, ln)
    parts = " ".join(f"{nm}={ms[i]:.2f}ms/{ln[i]}" + (f"({fl[i] / ms[i] / 1e9:.0f}TF)" if fl[i] else "")
                     for i, nm in enumerate(NAMES) if ln[i])
    print(f"{name}: {ms_tot:.2f} ms [policy {model.precision_policy}] | {parts}", flush=True)


if __name__ == "__main__":
    m = M.LanguagePerceiver().to(dev).eval()
    ids = torch.randint(0, 262, (8, 2048), device=dev)
    run("language B=8 x 2048 tokens", m, (ids, torch.ones(8, 2048, dtype=torch.bool, device=dev)))
    m = M.FlowPerceiver(img_size=(368, 496)).to(dev).eval()
    im1, im2 = torch.randn(1, 3, 368, 496, device=dev), torch.randn(1, 3, 368, 496, device=dev)
    run("flow B=1 368x496", m, (im1, im2))
    m = M.ClassificationPerceiver().to(dev).eval()
    run("classify B=32 (fp16x2w default)", m, (torch.randn(32, 3, 224, 224, device=dev),))
