"""Dev tool (GPU): the six B = 4 ImageNet goldens (three copies per batch, as bench.py's gate runs them) under "fp16"
and under "fp16sd" (error-feedback rounding of the shared weights over the 8 blocks)."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cases import model_inputs, model_seed  # noqa: E402
from _golden import load  # noqa: E402
import test_models as TM  # noqa: E402
import bench as Bn  # noqa: E402

dev = torch.device("cuda:0")
pols = sys.argv[1:] or ["fp16", "fp16sd", "fp16x2s"]
worst = {p: [0.0, 0.0] for p in pols}
for name in Bn.CONFIGS["imagenet"]["parity_goldens"]:
    g = load(name)
    model = TM._load_generated(TM.build(name), g, dev, model_seed(name))
    x = torch.from_numpy(model_inputs(name)[0]).to(dev).repeat(3, 1, 1, 1)
    line = f"{name:28s}"
    for pol in pols:
        model.precision_policy = pol                 # "name", "encoder/decoder" or "cross/stack/decoder"
        with torch.inference_mode():
            y = model(x).cpu().numpy().reshape(3, *g["out"].shape)
        e = [max(v) for v in zip(*(Bn.rel_errors(yc, g["out"]) for yc in y))]
        worst[pol] = [max(a, b) for a, b in zip(worst[pol], e)]
        line += f" | {pol}: {e[0]:.2e} / {e[1]:.2e}"
    print(line, flush=True)
print("worst:", {p: [f"{v:.2e}" for v in w] for p, w in worst.items()})
