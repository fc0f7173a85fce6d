"""Per-kernel-class device time of one hot-path stage (dev tool, GPU): python tools/stage_profile.py [cross|sa|dec]"""
import ctypes as C, os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")]
import torch
import perceiverio_pytorch_amd as P
from perceiverio_pytorch_amd import _lib as L
from perceiverio_pytorch_amd.models import ClassificationPerceiver
which = sys.argv[1] if len(sys.argv) > 1 else "cross"
pol = sys.argv[2] if len(sys.argv) > 2 else "fp16"
dev = torch.device("cuda:0")
m = ClassificationPerceiver(precision_policy=pol).to(dev).eval()
lib = P.lib()
B = 32
x = torch.randn(B, 3136, 322, device=dev)
pio = m.perceiver
names = ["gemm_nt_256", "gemm_nt_128_batched", "layernorm_cast", "softmax", "pack", "flash_attn", "gemm_nt_128_flat", "gemm_nt_stream",
         "gemm_nt_wide"]
with torch.inference_mode(), P.runtime.precision(pol):
    lat0 = pio._encoder.latents(x)
    z = pio._encoder.cross_attend(lat0, x)
    qt = pio._output_queries["__default"]._position_encoding.pos_embs
    qv = torch.broadcast_to(qt[None], (B,) + qt.shape)
    fn = {"cross": lambda: pio._encoder.cross_attend(lat0, x), "sa": lambda: pio._encoder.self_attends[0](z),
          "dec": lambda: pio._decoder(qv, z)}[which]
    fn(); torch.cuda.synchronize()
    L.check(lib.pio_prof_begin(4096))
    for _ in range(3): fn()
    ms = (C.c_double * 9)(); fl = (C.c_double * 9)(); by = (C.c_double * 9)(); ln = (C.c_int64 * 9)()
    lib.pio_prof_end(ms, fl, by, ln)
for i, n in enumerate(names):
    if ln[i]:
        print(f"{n:22s} launches {ln[i]//3:3d}  {ms[i]/3*1e3:8.1f} us/step  avg {ms[i]/ln[i]*1e3:7.1f} us"
              + (f"  {fl[i]/(ms[i]*1e-3)/1e12:6.1f} TF/s" if fl[i] else f"  {by[i]/(ms[i]*1e-3)/1e9:6.0f} GB/s"))
