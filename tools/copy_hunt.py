"""Dev tool (GPU): which Python lines issue device-to-device copies during one model forward (torch profiler, stacks)."""
import os
import sys
from collections import Counter

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
import bench as Bn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "flow"
dev = torch.device("cuda:0")
cfg = Bn.CONFIGS[name]
model, params = Bn.build_model(name, dev, cfg["policy"])
ins = Bn.make_inputs(name, cfg["batch"], 0, dev)
with torch.inference_mode():
    for _ in range(2):
        model(*ins)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model(*ins)
        torch.cuda.synchronize()
cnt = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::to", "aten::_to_copy"):
        st = [s for s in ev.stack if "perceiverio_pytorch_amd" in s or "bench.py" in s][:2]
        cnt[(ev.name, " <- ".join(s.split("/")[-1] for s in st))] += 1
for (nm, st), n in cnt.most_common(25):
    print(f"{n:4d} {nm:18s} {st}")
