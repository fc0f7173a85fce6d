"""Dev tool (GPU): worst-case parity of candidate precision policies over the widened golden sets (round 4: trained-like
parameter statistics for the classifier, second seeds for the dense-output models).

    python tools/r4_policy_table.py imagenet fp16sd fp16x3f/fp16sd/fp16x3f ...
    python tools/r4_policy_table.py multimodal fp16x2w/fp16x2af fp16x2w/fp16x3f ...
    python tools/r4_policy_table.py flow fp16/fp16x2af ...
"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import bench as Bn  # noqa: E402

name = sys.argv[1]
dev = torch.device("cuda:0")
for pol in sys.argv[2:]:
    model, params = Bn.build_model(name, dev, pol)
    r = Bn.parity_check(name, model, params, dev, pol)
    per = r.get("per_golden", {})
    cells = "  ".join(f"{k.replace('model_', '').replace('classify_b4_', '')}:{v['relL2']*1e4:.1f}/{v['max_abs_over_absmax']*1e4:.1f}"
                      for k, v in per.items())
    ms = ""
    if os.environ.get("PIO_TABLE_TIME", "1") != "0":
        inputs = Bn.make_inputs(name, Bn.CONFIGS[name]["batch"], 0, dev)
        with torch.inference_mode():
            for _ in range(3):
                model(*inputs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10 if name in ("imagenet", "language") else 3
            e0.record()
            for _ in range(n):
                model(*inputs)
            e1.record()
            torch.cuda.synchronize()
            ms = f" {e0.elapsed_time(e1) / n:7.3f} ms eager"
        del inputs
    kl = r.get("known_limit")
    if kl:
        cells += f"  | trained-like (known limit): {kl['relL2']*1e4:.1f}/{kl['max_abs_over_absmax']*1e4:.1f}"
    print(f"{pol:28s}{ms} worst relL2 {r['relL2']:.2e} max {r['max_abs_over_absmax']:.2e} ok={r['ok']} | x1e-4: {cells}", flush=True)
    del model
    torch.cuda.empty_cache()
