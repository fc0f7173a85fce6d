"""Dev tool (GPU, private build with -DPIO_GEMM_STAMPS): s_memtime stamps inside the LDS-staged epilogue of the
LayerNorm fold's producer (gemm_nt_wide<.,0,2,2,1>), wave 0 of workgroup 0, on a 16384 x 1024 x 1024 launch.

    tools/wide_epi_stamps.py --build      -> tools/_abl/libpio_wide_epi.so
    PIO_LIB_PATH=tools/_abl/libpio_wide_epi.so tools/wide_epi_stamps.py
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
    objs = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith(".o") and f != "pio_gemm_wide.o"]
    o = os.path.join(out, "wide_epi.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DPIO_GEMM_STAMPS",
                           "-c", os.path.join(src, "pio_gemm_wide.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                           os.path.join(out, "libpio_wide_epi.so"), o] + objs)
    sys.exit(0)

import torch  # noqa: E402
from perceiverio_pytorch_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")
M, N, K = 16384, 1024, 1024
A = torch.randn(M, K, device=dev).half()
B = (torch.randn(N, K, device=dev) / K ** 0.5).half()
bias = torch.randn(N, device=dev)
rh = torch.randn(M, N, device=dev).half()
rl = (torch.randn(M, N, device=dev) * 1e-3).half()
xh, xl = torch.empty_like(rh), torch.empty_like(rl)
part = torch.empty(M, N // 128, 2, device=dev)
g = L.Gemm()
g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), None
g.M, g.N, g.K = M, N, K
g.lda, g.ldb, g.ldc = K, K, N
g.batch, g.nh = 1, 1
g.bias, g.bias_mode, g.act, g.alpha = bias.data_ptr(), 1, 0, 1.0
g.out_f32, g.n_store, g.dtype = 1, N, L.PIO_DT_F16
g.X16, g.X16_lo, g.ld16 = xh.data_ptr(), xl.data_ptr(), N
g.R16_hi, g.R16_lo = rh.data_ptr(), rl.data_ptr()
g.row_part = part.data_ptr()
st = torch.cuda.current_stream().cuda_stream
prev = lib.pio_gemm_kernel_override(2)
try:
    for _ in range(5):
        L.check(lib.pio_gemm_nt(C.byref(g), st), "producer")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 50
    for _ in range(n):
        lib.pio_gemm_nt(C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
finally:
    lib.pio_gemm_kernel_override(prev)
ref = (A.float() @ B.float().T + bias + rh.float() + rl.float())
got = xh.float() + xl.float()
err = ((got - ref).abs().max() / ref.abs().max()).item()
s = (C.c_ulonglong * 16)()
lib.pio_debug_wide_epi_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.pio_debug_wide_epi_stamps(s) == 0
t = [int(v) for v in s]
w = (C.c_ulonglong * 12)()
lib.pio_debug_wide_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
lib.pio_debug_wide_stamps(w)
c = [int(v) for v in w[8:12]]
ghz = (c[2] - c[0]) / max(1, c[3] - c[1]) * 0.1
print(f"producer {M}x{N}x{K}: {us:6.1f} us back to back, max err {err:.1e}, {ghz:.2f} GHz; whole kernel (workgroup 0) "
      f"{c[2] - c[0]} cycles | epilogue: barrier {t[1] - t[0]}, row blocks " + " ".join(str(t[3 + i] - t[2 + i]) for i in range(7)) +
      f" (first incl. prologue loads {t[2] - t[1]}), stores drained {t[10] - t[9]}, total {t[10] - t[0]}", flush=True)
