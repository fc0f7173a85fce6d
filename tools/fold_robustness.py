import sys, os, torch
sys.path.insert(0, "/root/repo")
import bench as Bn
from perceiverio_pytorch_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda:0")
for pol in ("fp16", "bf16", "fp16x2w"):
    model, _, _ = Bn.build_model(dev, pol)
    for B in (1, 3, 4, 8, 16, 32):
        x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(B)).to(dev)
        with torch.inference_mode():
            lib.pio_ln_fold_enable(2)
            y1 = model(x).double()
            lib.pio_ln_fold_enable(0)
            y0 = model(x).double()
            model.precision_policy = "fp16x3"
            yr = model(x).double()
            model.precision_policy = pol
            lib.pio_ln_fold_enable(2)
        rel = lambda a, b: ((a - b).norm() / b.norm()).item()
        print(f"policy {pol:8s} B={B:2d}: fold {rel(y1, yr):.2e}  plain {rel(y0, yr):.2e}  fold-taken={not torch.equal(y1, y0)}  finite={bool(torch.isfinite(y1).all())}", flush=True)
