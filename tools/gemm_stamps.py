"""Phase breakdown of one gemm_nt_256 tile by in-kernel s_memtime stamps + epilogue ablations (dev tool).

Build the instrumented private copy of the library first (never the shipped one):
    tools/gemm_stamps.py --build        (hipcc -DPIO_GEMM_STAMPS ... -o tools/_abl/libpio_hip_stamps.so)
then on the GPU box:
    PIO_LIB_PATH=tools/_abl/libpio_hip_stamps.so python tools/gemm_stamps.py
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

if "--build" in sys.argv:
    src = os.path.join(ROOT, "perceiverio_pytorch_amd", "csrc")
    out = os.path.join(ROOT, "tools", "_abl")
    os.makedirs(out, exist_ok=True)
    files = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith(".hip")]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DPIO_GEMM_STAMPS",
           "-shared", "-o", os.path.join(out, "libpio_hip_stamps.so")] + files
    print(" ".join(cmd))
    sys.exit(subprocess.call(cmd))

import torch
from perceiverio_pytorch_amd import _lib as L

lib = L.lib()
dbg = lib.pio_debug_gemm_stamps
dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device("cuda:0")
NAMES = {0: "full", 1: "no global stores", 2: "regs->global (no LDS pass)", 3: "no epilogue"}


def run(M, N, K, out_f32, resid, act, mode, iters=20, quiet=False):
    A = torch.randn(M, K, device=dev).half()
    B = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev)
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
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    assert dbg(None, mode) == 0
    for _ in range(3):
        L.check(lib.pio_gemm_nt(C.byref(g), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.pio_gemm_nt(C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    if quiet:
        return us
    s = (C.c_ulonglong * 8)()
    assert dbg(s, -1) == 0
    t = [int(v) for v in s]
    d = lambda a, b: t[b] - t[a]
    tail = "" if mode == 3 else f" epilogue(issue) {d(2, 3)} stores-landed +{d(3, 4)}"
    print(f"M={M} N={N} K={K} f32={int(out_f32)} R={int(resid)} act={act} [{NAMES[mode]}]: {us:7.1f} us" +
          (f" | prologue {d(0, 1)} mainloop {d(1, 2)}{tail}" if t[2] > t[0] > 0 else ""), flush=True)


def run_stream(M, N, K, out_f32, resid, act, iters=10, smode=0):
    """Phase stamps of the streaming kernel: one unrolled step with epilogue work, one late step of a long K.
    smode: ablation (1 = no fragment reads / MFMAs, 2 = no DMA after the prologue; results are then garbage)."""
    sdbg = lib.pio_debug_stream_stamps
    sdbg.argtypes = [C.POINTER(C.c_ulonglong)]
    torch.cuda.synchronize()
    assert lib.pio_debug_stream_mode(smode) == 0
    if smode:
        print(f" -- ablation mode {smode}: " + {1: "DMA only", 2: "LDS reads + MFMA only", 3: "barriers only", 4: "residual from a cache-resident region", 8: "no residual loads", 16: "MFMAs without fragment reads", 18: "MFMAs only (no LDS reads, no DMA)"}.get(smode, str(smode)))
    prev = lib.pio_gemm_kernel_override(1)
    try:
        run(M, N, K, out_f32, resid, act, 0, iters=iters)
    finally:
        lib.pio_gemm_kernel_override(prev)
    s = (C.c_ulonglong * 16)()
    assert sdbg(s) == 0
    names = ["LDS fragment reads + DMA issue", "-", "waits (lgkm, team 1: vm)", "barrier",
             "32 MFMA + epilogue / residual fillers", "wait vm (team 0)", "barrier"]
    for row, what in ((0, "step 9 (with epilogue work)"), (1, "late step (K > 1024)")):
        t = [int(v) for v in s[row * 8:row * 8 + 8]]
        if t[7] <= t[0] or t[7] - t[0] > 100000:
            continue
        print(f"   stream {what}: " + ", ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names) if n != "-") +
              f" | total {t[7] - t[0]}", flush=True)


def run_clock(M, N, K, out_f32, resid, act, seconds=2.5):
    """In-kernel clock of the streaming GEMM after `seconds` of back-to-back launches on random data:
    d(s_memtime) / d(s_memrealtime) x 100 MHz around the last launch (workgroup 0)."""
    cdbg = lib.pio_debug_stream_clock
    cdbg.argtypes = [C.POINTER(C.c_ulonglong)]
    assert lib.pio_debug_stream_mode(0) == 0
    prev = lib.pio_gemm_kernel_override(1)
    try:
        us = run(M, N, K, out_f32, resid, act, 0, iters=20, quiet=True)
        us = run(M, N, K, out_f32, resid, act, 0, iters=max(20, int(seconds * 1e6 / us)), quiet=True)
    finally:
        lib.pio_gemm_kernel_override(prev)
    c = (C.c_ulonglong * 4)()
    assert cdbg(c) == 0
    cyc, ref = int(c[2]) - int(c[0]), int(c[3]) - int(c[1])
    ghz = cyc / ref * 0.1
    tf = 2.0 * M * N * K / us / 1e6
    print(f"clock M={M} N={N} K={K} f32={int(out_f32)} R={int(resid)} act={act}: {us:7.1f} us {tf:7.1f} TF/s | kernel "
          f"{cyc} cycles / {ref * 10} ns = {ghz:.3f} GHz held => MFMA peak at that clock {2500 * ghz / 2.4:6.0f} TF/s, "
          f"achieved {tf / (2500 * ghz / 2.4) * 100:.1f} % of it", flush=True)


if __name__ == "__main__" and "--clock" in sys.argv:
    run_clock(16384, 3072, 1024, False, False, 0)
    run_clock(16384, 1024, 1024, True, True, 0)
    run_clock(16384, 1024, 1024, False, False, 1)
    run_clock(8192, 8192, 8192, False, False, 0)
    sys.exit(0)

if __name__ == "__main__" and "--stream" in sys.argv:
    run_stream(16384, 1024, 1024, False, False, 0)
    run_stream(16384, 1024, 1024, True, True, 0)
    run_stream(16384, 1024, 1024, False, False, 1)
    run_stream(8192, 8192, 8192, False, False, 0, iters=3)
    for sm in (1, 2, 3):
        run_stream(16384, 1024, 1024, False, False, 0, smode=sm)
        run_stream(16384, 3072, 1024, False, False, 0, smode=sm)
    run_stream(16384, 3072, 1024, False, False, 0, smode=0)
    run_stream(16384, 1024, 1024, False, False, 0, smode=18)
    run_stream(16384, 1024, 1024, False, False, 0, smode=16)
    run_stream(16384, 1024, 1024, True, True, 0, smode=4)
    run_stream(16384, 1024, 1024, True, True, 0, smode=8)
    run_stream(16384, 1024, 1024, True, True, 0, smode=0)
    run_stream(16384, 1024, 1024, True, False, 0, smode=0)
elif __name__ == "__main__":
    for shape in ((16384, 1024, 1024), (16384, 3072, 1024)):
        for (f32, r, act) in ((False, False, 0), (True, True, 0), (False, False, 1)):
            for mode in (0, 1, 2, 3):
                run(*shape, f32, r, act, mode)
