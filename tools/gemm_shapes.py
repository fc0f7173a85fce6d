"""Dev tool (GPU): which GEMM kernel each launch of one model forward goes to (PIO_GEMM_LOG=1 lines, counted).
usage: PIO_GEMM_LOG=1 python tools/gemm_shapes.py <imagenet|language|flow|multimodal> 2> log; sort log | uniq -c"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import bench as Bn  # noqa: E402

name = sys.argv[1]
dev = torch.device("cuda:0")
cfg = Bn.CONFIGS[name]
model, params = Bn.build_model(name, dev, cfg["policy"])
ins = Bn.make_inputs(name, int(os.environ.get("PIO_BATCH", cfg["batch"])), 0, dev)
with torch.inference_mode():
    print("=== forward begins", file=sys.stderr, flush=True)
    y = model(*ins)
    torch.cuda.synchronize()
