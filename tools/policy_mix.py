"""Dev tool (GPU): which part of a dense-output model needs which precision?  Runs the full-size flow model with a
separate precision policy for (encoder cross-attend, latent self-attend stack, decoder) and prints the error against
the reference golden for every combination asked for."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cases import model_inputs, model_seed  # noqa: E402
from _golden import load  # noqa: E402
import test_models as TM  # noqa: E402
from perceiverio_pytorch_amd.runtime import precision  # noqa: E402
from perceiverio_pytorch_amd.io_processors import patches_for_flow  # noqa: E402


def run(model, ins, pe, ps, pd):
    P = model.perceiver
    pair = torch.stack([ins[0].contiguous(), ins[1].contiguous()], dim=1)
    inp = {"__default": patches_for_flow(pair).movedim(-1, -3)}
    x, sizes, without_pos = P._multi_preprocessor(inp, pos=None)
    enc = P._encoder
    with precision(pe):
        z = enc.cross_attend(enc.latents(x), x)
    with precision(ps):
        for _ in range(enc._num_blocks):
            for sa in enc.self_attends:
                z = sa(z)
    query, qsizes = P.decoder_query(x, sizes, without_pos)
    with precision(pd):
        y = P._decoder(query, z)
    return P._output_postprocessors["__default"](y, pos=None, modality_sizes=None)


def main():
    name = "model_flow_full"
    dev = torch.device("cuda:0")
    g = load(name)
    model = TM._load_generated(TM.build(name), g, dev, model_seed(name))
    ins = [torch.from_numpy(a).to(dev) for a in model_inputs(name)]
    combos = [c.split("/") for c in (sys.argv[1:] or ["fp16x2w/fp16x2w/fp16x2w", "fp16x2w/fp16x3/fp16x2w",
                                                      "fp16x3/fp16x2w/fp16x2w", "fp16x2w/fp16x2w/fp16x3",
                                                      "fp16x3/fp16x3/fp16x2w", "fp16x2w/fp16x3/fp16x3",
                                                      "fp16/fp16x3/fp16"])]
    ref = g["out_sub"].astype(np.float64)
    for pe, ps, pd in combos:
        with torch.inference_mode():
            y = run(model, ins, pe, ps, pd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = run(model, ins, pe, ps, pd)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
        d = y[:, :, ::8, ::8].double().cpu().numpy() - ref
        rl2 = np.sqrt((d * d).sum()) / np.sqrt((ref * ref).sum())
        rmax = np.abs(d).max() / float(g["out_absmax"])
        print(f"enc={pe:8s} stack={ps:8s} dec={pd:8s}: relL2={rl2:.3e} max/absmax={rmax:.3e}  {ms:7.2f} ms", flush=True)


if __name__ == "__main__":
    main()
