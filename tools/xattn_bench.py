"""Dev tool (GPU): device time of the fused cross-attention kernel alone (profiler class 5) at the shipped shapes."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import perceiverio_pytorch_amd as P  # noqa: E402
from perceiverio_pytorch_amd import _lib as L, runtime as R  # noqa: E402
from perceiverio_pytorch_amd.transformer_primitives import Attention  # noqa: E402

SHAPES = {  # name: heads, dk, dv, B, Tq, Tk, q_in, kv_in, broadcast q
    "imagenet_enc": (1, 322, 322, 32, 512, 3136, 1024, 322, True),
    "flow_enc": (1, 322, 322, 1, 2048, 182528, 512, 322, True),
    "flow_dec": (1, 512, 512, 1, 182528, 2048, 322, 512, False),
    "imagenet_dec": (1, 1024, 1024, 32, 1000, 512, 1024, 1024, True),    # xattn_tall_kernel (pio_xtall.hip)
    "multimodal_enc": (1, 704, 704, 1, 784, 52097, 512, 704, True),
    "multimodal_dec": (1, 512, 512, 1, 6288, 784, 1026, 512, False),
    "language_enc": (8, 32, 160, 32, 256, 2048, 1280, 768, True),
    "language_dec": (8, 32, 96, 32, 2048, 256, 768, 1280, True),
    # the latent self-attend shape of the ImageNet model through the CROSS-attention kernel (an all-ones key mask routes
    # it there) -- for comparison with flash_attn_kernel on the same shape ("sa_flash")
    "sa_xattn": (8, 128, 128, 32, 512, 512, 1024, 1024, False),
    "sa_flash": (8, 128, 128, 32, 512, 512, 1024, 1024, False),
}


def main():
    lib = P.lib()
    dev = torch.device("cuda:0")
    P.set_precision_policy("fp16")
    for name in (sys.argv[1:] or SHAPES):
        H, dk, dv, B, Tq, Tk, q_in, kv_in, bc = SHAPES[name]
        m = Attention(q_in, kv_in, kv_in, num_heads=H, qk_out_channels=H * dk, v_out_channels=H * dv,
                      output_channels=q_in).to(dev).eval()
        xq = torch.randn(1 if bc else B, Tq, q_in, device=dev)
        if bc:
            xq = torch.broadcast_to(xq, (B, Tq, q_in))
        xkv = torch.randn(B, Tk, kv_in, device=dev)
        d = m._desc()
        out = torch.empty((B, Tq, q_in), device=dev)
        ws = R.workspace(dev, lib.pio_attention_workspace_bytes(d, B, Tq, Tk))

        km = torch.ones(B, Tk, dtype=torch.uint8, device=dev) if name == "sa_xattn" else None

        def call():
            L.check(lib.pio_attention_fwd(d, R.tensor3(xq), R.tensor3(xkv), R.tensor3(xkv),
                                          km.data_ptr() if km is not None else None, None, None, None,
                                          out.data_ptr(), None, ws.data_ptr(), ws.numel(), R.stream_ptr(dev)), "fwd")
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        n = 5
        L.check(lib.pio_prof_begin(4096))
        for _ in range(n):
            call()
        ms = (C.c_double * 9)(); fl = (C.c_double * 9)(); by = (C.c_double * 9)(); ln = (C.c_int64 * 9)()
        lib.pio_prof_end(ms, fl, by, ln)
        us = ms[5] / n * 1e3
        print(f"{name:16s} H={H} dk={dk} dv={dv} B={B} Tq={Tq} Tk={Tk}: fused attention {us:9.1f} us "
              f"({fl[5] / n / (us * 1e-6) / 1e12:7.1f} algorithmic TFLOP/s, {ln[5] // n} launches); all kernels "
              f"{sum(ms) / n * 1e3:9.1f} us", flush=True)


if __name__ == "__main__":
    main()
