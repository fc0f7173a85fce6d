"""Dev tool (GPU): does ERROR-FEEDBACK rounding of the shared weights across the 8 blocks cut the fp16 policy's error?
The ImageNet encoder applies the same 6 layers 8 times; with one fp16 image per weight its rounding error acts 8 times
in the same direction.  Here block b runs with hi_b = fp16(W + E_{b-1}), E_b = (W + E_{b-1}) - hi_b (first-order
sigma-delta over the block index): every block still multiplies with a valid fp16 rounding of W (within one ulp), but the
accumulated error of the 8 images stays below half an ulp instead of growing 8-fold.  Prototype: the encoder is driven
layer by layer from Python with the layer's weights swapped per block."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cases import model_inputs, model_seed  # noqa: E402
from _golden import load  # noqa: E402
import test_models as TM  # noqa: E402
from perceiverio_pytorch_amd.runtime import precision  # noqa: E402


def errs(y, ref):
    d = y.double().cpu().numpy() - ref.astype(np.float64)
    return np.sqrt((d * d).sum()) / np.sqrt((ref.astype(np.float64) ** 2).sum()), np.abs(d).max() / np.abs(ref).max()


def variants(w, scale, nblk, mode):
    """Effective fp32 weights per block such that fp16(w_eff * scale) follows the chosen rounding sequence of w * scale."""
    ws = w * scale
    out, e = [], torch.zeros_like(ws)
    for b in range(nblk):
        if mode == "plain":
            t = ws
        elif mode == "sd":
            t = ws + e
        hi = t.half().float()
        e = t - hi
        out.append(hi / scale)
    return out


def run_encoder(model, x, nblk, sets):
    P = model.perceiver
    enc = P._encoder
    xin = P._multi_preprocessor({"__default": x})[0]
    z = enc.cross_attend(enc.latents(xin), xin)
    for b in range(nblk):
        for l, sa in enumerate(enc.self_attends):
            if sets is not None:
                for lin, ws_ in sets[l]:
                    lin.weight.data = ws_[b]
            z = sa(z)
    return z


def main():
    dev = torch.device("cuda:0")
    names = sys.argv[1:] or ["model_classify_b4_s31", "model_classify_b4_s32", "model_classify_b4_natural"]
    for name in names:
        g = load(name)
        model = TM._load_generated(TM.build(name), g, dev, model_seed(name))
        model.precision_policy = "fp16"
        x = torch.from_numpy(model_inputs(name)[0]).to(dev).repeat(3, 1, 1, 1)
        P = model.perceiver
        enc = P._encoder
        nblk = enc._num_blocks
        with torch.inference_mode(), precision("fp16"):
            y_ref = model(x)
            print(f"{name}: module forward            relL2=%.3e max=%.3e" % errs(y_ref[:4], g["out"]), flush=True)
            orig = {}
            for mode in ("plain", "sd"):
                sets = []
                for sa in enc.self_attends:
                    g1 = sa.layer_norm1.weight.detach().float()
                    g2 = sa.layer_norm2.weight.detach().float()
                    one = torch.ones_like(g1)
                    per = []
                    for lin, sc in ((sa.attention.proj_q, g1), (sa.attention.proj_k, g1), (sa.attention.proj_v, g1),
                                    (sa.attention.final, one), (sa.mlp.fc1, g2), (sa.mlp.fc2, one)):
                        w = orig.setdefault(id(lin), lin.weight.detach().float().clone())
                        per.append((lin, variants(w, sc[None, :], nblk, mode)))
                    sets.append(per)
                z = run_encoder(model, x, nblk, sets)
                qtab = P._output_queries["__default"]._position_encoding.pos_embs
                yd = P._decoder(torch.broadcast_to(qtab[None], (x.shape[0],) + qtab.shape), z)[:, 0, :]
                print(f"{name}: layer by layer, {mode:5s}      relL2=%.3e max=%.3e" % errs(yd[:4], g["out"]), flush=True)
            for sa in enc.self_attends:
                for lin in (sa.attention.proj_q, sa.attention.proj_k, sa.attention.proj_v, sa.attention.final,
                            sa.mlp.fc1, sa.mlp.fc2):
                    lin.weight.data = orig[id(lin)]


if __name__ == "__main__":
    main()
