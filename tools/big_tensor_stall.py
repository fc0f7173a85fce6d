"""Dev tool (GPU): does reading from many cached > 128 MiB tensors stall the queue?  (multimodal decode_chunks_per_call >= 6)"""
import sys
import time
import torch

dev = torch.device("cuda:0")
for mib in [int(v) for v in (sys.argv[1:] or ["120", "150"])]:
    n = mib * 2**20 // 4
    xs = [torch.randn(n, device=dev) for _ in range(22)]
    worst, tot = 0.0, 0.0
    for rep in range(4):
        for x in xs:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = x[: n // 2] * 2.0
            z = torch.empty(n // 2, device=dev)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if rep:
                worst = max(worst, dt)
                tot += dt
    print(f"{mib} MiB x 22 tensors: worst step {worst:.2f} ms, total of 3 sweeps {tot:.1f} ms", flush=True)
    del xs
    torch.cuda.empty_cache()
