"""Dev tool (GPU): the three variants of the fused self-attention kernel on the same inputs (SelfAttention block, fp16)."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import perceiverio_pytorch_amd as P  # noqa: E402
from perceiverio_pytorch_amd import _lib as L  # noqa: E402
from perceiverio_pytorch_amd.transformer_primitives import SelfAttention  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")
P.set_precision_policy("fp16")
torch.manual_seed(3)
for B, T in [(2, int(t)) for t in sys.argv[1:]] or ((4, 512), (2, 777), (32, 512)):
    m = SelfAttention(1024, widening_factor=1, num_heads=8).to(dev).eval()
    x = (torch.randn(B, T, 1024, device=dev) * 1.5 + 0.2)
    ys = []
    with torch.no_grad():
        for v in (0, 1):
            prev = lib.pio_debug_flash_variant(v)
            ys.append(m(x).double())
            lib.pio_debug_flash_variant(prev)
    for v in (1,):
        d = (ys[v] - ys[0]).abs()
        print(f"B={B} T={T}: variant {v} vs 0: max |diff| {d.max().item():.3e} (|y| max {ys[0].abs().max().item():.3f}), "
              f"rel L2 {(d.norm() / ys[0].norm()).item():.3e}, rows with diff > 1e-2: "
              f"{int((d.amax(-1) > 1e-2).sum())} of {B * T}", flush=True)
        if d.max() > 1e-2:
            bad = (d.amax(-1) > 1e-2).nonzero()
            print("   first bad (batch, row):", bad[:8].tolist(), flush=True)
