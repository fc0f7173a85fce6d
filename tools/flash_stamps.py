"""Dev tool: in-kernel stamps, held clock and timing-only ablations of the pipelined self-attention kernel
(flash_attn_pipe_kernel, pio_flash.hip).

    tools/flash_stamps.py --build [masks]    builds tools/_abl/libpio_flash_<n>.so (bit 0 no exponentials, 1 no LDS
                                             fragment reads, 2 no MFMAs, 3 no LDS-DMA)
    PIO_LIB_PATH=tools/_abl/libpio_flash_<n>.so tools/flash_stamps.py     (on the GPU box, one process per library)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
MASKS = tuple(int(a) for a in sys.argv[2:]) if len(sys.argv) > 2 else (0, 1, 2, 4, 8, 7, 15)

if "--build" in sys.argv:
    src = os.path.join(ROOT, "perceiverio_pytorch_amd", "csrc")
    out = os.path.join(ROOT, "tools", "_abl")
    os.makedirs(out, exist_ok=True)
    objs = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith(".o") and f != "pio_flash.o"]
    for n in MASKS:
        o = os.path.join(out, f"flash_{n}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                               "-DPIO_FLASH_STAMPS", "-DPIO_EXPERIMENTS", f"-DPIO_FLASH_ABL={n}", "-c", os.path.join(src, "pio_flash.hip"),
                               "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(out, f"libpio_flash_{n}.so"), o] + objs)
    sys.exit(0)

import torch  # noqa: E402
import perceiverio_pytorch_amd as P  # noqa: E402
from perceiverio_pytorch_amd import _lib as L  # noqa: E402
from perceiverio_pytorch_amd.transformer_primitives import SelfAttention  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")
P.set_precision_policy("fp16")
B, T = 32, 512
m = SelfAttention(1024, widening_factor=1, num_heads=8).to(dev).eval()
x = torch.randn(B, T, 1024, device=dev)
with torch.no_grad():
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    n = 50
    L.check(lib.pio_prof_begin(4096))
    for _ in range(n):
        m(x)
    ms = (C.c_double * 9)(); fl = (C.c_double * 9)(); by = (C.c_double * 9)(); ln = (C.c_int64 * 9)()
    lib.pio_prof_end(ms, fl, by, ln)
us = ms[5] / n * 1e3
s = (C.c_ulonglong * 24)()
lib.pio_debug_flash_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.pio_debug_flash_stamps(s) == 0
t = [int(v) for v in s]
mode = lib.pio_debug_flash_mode()
ghz = (t[6] - t[4]) / max(1, t[7] - t[5]) * 0.1
print(f"flash pipe B={B} T={T} [abl {mode}]: {us:7.1f} us, {ghz:.2f} GHz | tile 2 of workgroup 0: W1 {t[1] - t[0]}, "
      f"W2 {t[2] - t[1]}, W3 {t[3] - t[2]} cycles | whole kernel (workgroup 0) {t[6] - t[4]} cycles: Q fragments "
      f"{t[8] - t[4]}, addresses {t[9] - t[8]}, first tiles landed {t[10] - t[9]}, S(0) {t[11] - t[10]}, tiles 0..{T // 64 - 3} "
      f"{t[12] - t[11]}, last two tiles {t[13] - t[12]}, epilogue {t[6] - t[13]} | top of tile 2: DMA wait {t[17] - t[16]}, "
      f"barrier {t[18] - t[17]}, stage() {t[19] - t[18]}, to W1 {t[0] - t[19]}", flush=True)
