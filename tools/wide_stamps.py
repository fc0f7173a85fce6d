"""Dev tool: phase stamps, held clock and timing-only ablations of the 256x256 four-wave GEMM (pio_gemm_wide.hip).

    tools/wide_stamps.py --build     builds tools/_abl/libpio_wide_<n>.so for the ablation masks n (compile time:
                                     bit 0 no LDS fragment reads, bit 1 no DMA pieces, bit 2 no barrier)
    PIO_LIB_PATH=tools/_abl/libpio_wide_<n>.so tools/wide_stamps.py      (on the GPU box, one process per library)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
MASKS = tuple(int(a) for a in sys.argv[2:]) if len(sys.argv) > 2 else (0, 1, 2, 3, 4, 7)

if "--build" in sys.argv:
    src = os.path.join(ROOT, "perceiverio_pytorch_amd", "csrc")
    out = os.path.join(ROOT, "tools", "_abl")
    os.makedirs(out, exist_ok=True)
    objs = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith(".o") and f != "pio_gemm_wide.o"]
    for n in MASKS:
        o = os.path.join(out, f"wide_{n}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                               "-DPIO_GEMM_STAMPS", f"-DPIO_WIDE_ABL={n}", "-c", os.path.join(src, "pio_gemm_wide.hip"),
                               "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(out, f"libpio_wide_{n}.so"), o] + objs)
    sys.exit(0)

import torch  # noqa: E402
from perceiverio_pytorch_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")


def run(M, N, K, seconds=1.0):
    A = torch.randn(M, K, device=dev).half()
    B = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    bias = torch.randn(N, device=dev)
    Cc = torch.empty(M, N, device=dev, dtype=torch.float16)
    g = L.Gemm()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = K, K, N
    g.batch, g.nh = 1, 1
    g.bias, g.bias_mode, g.act, g.alpha = bias.data_ptr(), 1, 0, 1.0
    g.out_f32, g.n_store, g.dtype = 0, N, L.PIO_DT_F16
    st = torch.cuda.current_stream().cuda_stream
    prev = lib.pio_gemm_kernel_override(2)
    try:
        us = 50.0
        for rep in range(2):
            iters = 20 if rep == 0 else max(20, int(seconds * 1e6 / us))
            for _ in range(3):
                L.check(lib.pio_gemm_nt(C.byref(g), st))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                lib.pio_gemm_nt(C.byref(g), st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
    finally:
        lib.pio_gemm_kernel_override(prev)
    s = (C.c_ulonglong * 12)()
    lib.pio_debug_wide_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    assert lib.pio_debug_wide_stamps(s) == 0
    t = [int(v) for v in s[:8]]
    c = [int(v) for v in s[8:12]]
    ghz = (c[2] - c[0]) / max(1, c[3] - c[1]) * 0.1
    mode = lib.pio_debug_wide_mode()
    what = {0: "full", 1: "no LDS reads", 2: "no DMA", 3: "no LDS reads, no DMA", 4: "no barrier", 7: "MFMAs only"}.get(mode, str(mode))
    names = ["8 MFMA + bookkeeping", "wait vm", "wait lgkm", "barrier", "56 MFMA + 16 LDS reads + 8 DMA pieces"]
    print(f"wide M={M} N={N} K={K} [{what}]: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s, {ghz:.2f} GHz | " +
          ", ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names)) + f" | phase {t[5] - t[0]} | last epilogue (issue) {t[7] - t[6]}, whole kernel {c[2] - c[0]} cycles = "
          f"{(c[2] - c[0]) / (K // 32 * max(1, (M // 256) * (N // 256) // 256)):.0f} per phase (epilogues included)", flush=True)


if __name__ == "__main__":
    run(16384, 3072, 1024)
    run(8192, 8192, 8192)
