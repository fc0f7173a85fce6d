"""Dev tool (GPU): small-batch latency of the ImageNet classifier forward -- eager launches against one HIP graph replay
(torch.cuda.CUDAGraph around the same forward: the C-ABI allocates nothing and synchronises nothing, so it captures)."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import bench as Bn  # noqa: E402
import perceiverio_pytorch_amd as P  # noqa: E402
from perceiverio_pytorch_amd import runtime as R  # noqa: E402

dev = torch.device("cuda:0")
model, _ = Bn.build_model("imagenet", dev, "fp16")
for B in [int(b) for b in os.environ.get("PIO_PROBE_BATCHES", "1,2,4,8").split(",")]:
    x = torch.randn(B, 3, 224, 224, device=dev)
    with torch.inference_mode():
        for _ in range(3):
            y = model(x)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            y = model(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e3
        g = torch.cuda.CUDAGraph()
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    model(x)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                yg = model(x)
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / n * 1e3
            same = bool(torch.equal(yg, y))
            print(f"B={B}: eager {eager:7.3f} ms, graph replay {graph:7.3f} ms, identical logits {same}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"B={B}: eager {eager:7.3f} ms, graph capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
