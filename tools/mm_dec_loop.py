"""Dev tool (GPU): the multimodal decoder alone, called in a loop on one query array -- looking for the ~90 ms queue
stalls seen with decode_chunks_per_call >= 6."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import bench as Bn  # noqa: E402
from perceiverio_pytorch_amd.runtime import precision  # noqa: E402

dev = torch.device("cuda:0")
cfg = Bn.CONFIGS["multimodal"]
model, params = Bn.build_model("multimodal", dev, cfg["policy"])
P = model.perceiver
dec = P._decoder
lat = torch.randn(1, 784, 512, device=dev)
mode = sys.argv[2] if len(sys.argv) > 2 else "sync"
for Q in [int(v) for v in sys.argv[1].split(",")]:
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    qs = [torch.randn(1, Q, 1026, device=dev) for _ in range(nq)]
    ts = []
    with torch.inference_mode(), precision(P.decoder_policy):
        for i in range(60):
            if mode == "sync":
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = dec(qs[i % nq], lat)
            if mode == "sync":
                torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    ts = ts[5:]
    print(f"Q={Q} [{mode}]: median {sorted(ts)[len(ts) // 2]:.2f} ms, max {max(ts):.2f} ms, slow (>10 ms) at "
          f"{[i for i, t in enumerate(ts) if t > 10]}", flush=True)
