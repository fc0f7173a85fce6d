"""Dev tool (GPU): localise errors of xattn_tall_kernel -- one-hot probabilities: which V row does a key select?"""
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import perceiverio_pytorch_amd as P  # noqa: E402
from perceiverio_pytorch_amd.transformer_primitives import Attention  # noqa: E402

dev = torch.device("cuda:0")
P.set_precision_policy("fp16")
H, dk, dv, B, Tq, Tk = 1, 1024, 1024, 1, 128, 512
q_in, kv_in = 64, 96
torch.manual_seed(0)
m = Attention(q_in, kv_in, kv_in, num_heads=H, qk_out_channels=H * dk, v_out_channels=H * dv, output_channels=dv).to(dev).eval()
CH = int(os.environ.get("XT_CH", "0"))      # which dk channel carries the signal
with torch.no_grad():
    m.final.weight.copy_(torch.eye(dv)); m.final.bias.zero_()
    m.proj_q.weight.zero_(); m.proj_q.bias.zero_(); m.proj_q.bias[CH] = 64.0
    m.proj_k.weight.zero_(); m.proj_k.bias.zero_(); m.proj_k.weight[CH, 0] = 64.0
    m.proj_v.bias.zero_(); m.proj_v.weight[:, 0] = 0
res = {}
bad = []
for j0 in range(0, Tk, int(os.environ.get('XT_STEP', '1'))):
    xkv = torch.randn(B, Tk, kv_in, device=dev)
    xkv[:, :, 0] = 0
    xkv[:, j0, 0] = 1.0
    xq = torch.randn(B, Tq, q_in, device=dev)
    with torch.no_grad():
        y = m(xq, xkv, xkv)[0]                       # [Tq, dv]
        v = torch.nn.functional.linear(xkv[0], m.proj_v.weight).half().float()   # [Tk, dv]
    dist = torch.cdist(y[:1], v)[0]                   # query 0 against every V row
    jm = int(dist.argmin())
    dq = torch.cdist(y, v[jm:jm + 1])[:, 0]
    if jm != j0 or float(dist[jm]) > 0.1:
        bad.append(j0)
    if jm != j0 or float(dist[jm]) > 0.1: print(f"j0={j0:4d} -> matched key {jm:4d}  (dist {float(dist[jm]):.3e}, next best {float(dist.topk(2, largest=False).values[1]):.3e}); "
          f"rows agreeing with query 0: {int((dq < 1e-2).sum())}/{Tq}; |y| {float(y[0].norm()):.3f}", flush=True)
print("bad keys:", bad)
