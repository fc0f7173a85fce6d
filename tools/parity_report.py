"""Dev tool (GPU): relL2 / max-abs-over-abs-max and forward time of the whole-model goldens per precision policy.
    python tools/parity_report.py [case ...] [--policies fp16,fp16x2w,fp16x3]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cases import MODEL_CASES, gen_state_dict, model_inputs, model_seed  # noqa: E402
from _golden import load  # noqa: E402
import test_models as TM  # noqa: E402


def errs(y, ref, absmax=None):
    y = y.detach().float().cpu().numpy().astype(np.float64)
    d = y - ref.astype(np.float64)
    am = float(absmax) if absmax is not None else np.abs(ref).max()
    return np.sqrt((d * d).sum()) / np.sqrt((ref.astype(np.float64) ** 2).sum()), np.abs(d).max() / am


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pol = "fp16,fp16x2s,fp16x2w,fp16x3"
    for a in sys.argv[1:]:
        if a.startswith("--policies"):
            pol = a.split("=", 1)[1]
    names = args or ["model_classify_conv", "model_language", "model_flow_full", "model_multimodal_full"]
    dev = torch.device("cuda:0")
    from perceiverio_pytorch_amd.runtime import precision
    for name in names:
        g = load(name)
        c = MODEL_CASES[name]
        model = TM._load_generated(TM.build(name), g, dev, model_seed(name))
        ins = [torch.from_numpy(a).to(dev) for a in model_inputs(name)]
        for policy in pol.split(","):
            model.precision_policy = policy
            with torch.inference_mode():
                def run():
                    if name == "model_multimodal_full":
                        images, audio = ins
                        b, t, ch, h, w = images.shape
                        k = c["chunks"][0]
                        ics = t * h * w // c["n_chunks"]
                        acs = audio.shape[1] // model.audio_samples_per_patch // c["n_chunks"]
                        sub = {"image": torch.arange(ics * k, ics * (k + 1)),
                               "audio": torch.arange(acs * k, acs * (k + 1)), "label": None}
                        from perceiverio_pytorch_amd.models import _policy_scope
                        with _policy_scope(model):
                            return model.perceiver({"image": images, "audio": audio,
                                                    "label": torch.zeros((b, model.num_classes), device=dev)},
                                                   subsampled_output_points=sub)
                    return model(*ins[:2]) if len(ins) > 1 else model(ins[0])
                y = run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                y = run()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
            if name == "model_flow_full":
                e = errs(y[:, :, ::8, ::8], g["out_sub"], g["out_absmax"])
            elif name == "model_multimodal_full":
                e = errs(y["image"], g[f"out_image_{c['chunks'][0]}"])
            elif name == "model_language":
                e = errs(y[:, :96], g["out"], g["out_absmax"])
            else:
                e = errs(y, g["out"])
            print(f"{name:28s} {policy:8s} relL2={e[0]:.3e} max/absmax={e[1]:.3e}  {ms:8.2f} ms  "
                  f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
            torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
