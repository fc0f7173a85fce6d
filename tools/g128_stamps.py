"""Dev tool: per-K-step s_memtime stamps of one gemm_nt_128 workgroup (thread 0): loop top, after the counted wait, after
the barrier, after the DMA issue; kernel edges.
    python tools/g128_stamps.py --build      (private copy: tools/_abl/libpio_hip_g128.so, -DPIO_G128_STAMPS)
    PIO_LIB_PATH=tools/_abl/libpio_hip_g128.so python tools/g128_stamps.py M N K [block]
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
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DPIO_G128_STAMPS",
           "-shared", "-o", os.path.join(out, "libpio_hip_g128.so")] + files
    print(" ".join(cmd))
    sys.exit(subprocess.call(cmd))

import torch  # noqa: E402
from perceiverio_pytorch_amd import _lib as L  # noqa: E402

lib = L.lib()
dbg = lib.pio_debug_g128_stamps
dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device("cuda:0")


def run(M, N, K, block):
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
    assert dbg(None, block) == 0
    for _ in range(20):
        L.check(lib.pio_gemm_nt(C.byref(g), st))
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 128)()
    assert dbg(out, -1) == 0
    v = list(out)
    t0 = v[120]
    print(f"M={M} N={N} K={K} block {block}: start->loop {v[121]-t0}, loop end {v[122]-t0}, drained+barrier {v[123]-t0}, "
          f"epilogue done {v[124]-t0}  (s_memtime ticks)")
    nk = min(30, (K + 63) // 64)
    print("  kt: top  wait+barrier  frags  dma-issue  body(to next top)")
    for kt in range(nk):
        top, bar, iss, wt = v[4 * kt], v[4 * kt + 1], v[4 * kt + 2], v[4 * kt + 3]
        nxt = v[4 * kt + 4] if kt + 1 < nk else v[122]
        print(f"  {kt:2d}: {top-t0:6d} {bar-top:5d} {wt-bar:5d} {iss-wt:5d} {nxt-iss:5d}")


if __name__ == "__main__":
    M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 1024, 1024)
    blk = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    run(M, N, K, blk)
    run(1024, 1024, 1024, blk)
    run(2048, 1024, 1024, blk)
