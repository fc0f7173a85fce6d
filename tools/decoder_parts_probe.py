"""Dev tool (GPU): which part of the ImageNet decoder needs split operands?  Encoder as the class default
(cross-attend fp16x3f, stack fp16sd); the decoder's attention (q / k / v / final), its MLP and the final Linear each under
"fp16" or "fp16x3f"; worst of the six B = 4 goldens (three copies per batch) and the decoder's time at B = 32."""
import itertools
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from cases import model_inputs, model_seed  # noqa: E402
from _golden import load  # noqa: E402
import test_models as TM  # noqa: E402
import bench as Bn  # noqa: E402
from perceiverio_pytorch_amd import runtime as R  # noqa: E402

dev = torch.device("cuda:0")
names = Bn.CONFIGS["imagenet"]["parity_goldens"]
models = {}
for name in names:
    g = load(name)
    m = TM._load_generated(TM.build(name), g, dev, model_seed(name))
    m.precision_policy = "fp16x3f/fp16sd/fp16"
    models[name] = (m, g, torch.from_numpy(model_inputs(name)[0]).to(dev).repeat(3, 1, 1, 1))


def patch(model, p_attn, p_mlp, p_fin):
    dec = model.perceiver._decoder
    ca = dec.decoding_cross_attn
    for mod, pol in ((ca.attention, p_attn), (ca.mlp, p_mlp)):
        cls_desc = type(mod)._desc

        def f(mod=mod, pol=pol, cls_desc=cls_desc):
            with R.precision(pol):
                return cls_desc(mod)
        mod._desc = f
    cls_fin = type(dec)._final_desc

    def ff(dec=dec, pol=p_fin, cls_fin=cls_fin):
        with R.precision(pol):
            return cls_fin(dec)
    dec._final_desc = ff
    R.invalidate_packed_weights(model.perceiver._decoder)


combos = list(itertools.product(["fp16", "fp16x3f"], repeat=3))
for p_attn, p_mlp, p_fin in combos:
    worst = [0.0, 0.0]
    for name, (m, g, x) in models.items():
        patch(m, p_attn, p_mlp, p_fin)
        with torch.inference_mode():
            y = m(x).cpu().numpy().reshape(3, *g["out"].shape)
        e = [max(v) for v in zip(*(Bn.rel_errors(yc, g["out"]) for yc in y))]
        worst = [max(a, b) for a, b in zip(worst, e)]
    # decoder time at B = 32 on the first model
    m, g, x = models[names[0]]
    pio = m.perceiver
    with torch.inference_mode(), R.precision("fp16"):
        z = torch.randn(32, 512, 1024, device=dev)
        qtab = pio._output_queries["__default"]._position_encoding.pos_embs
        qv = torch.broadcast_to(qtab[None], (32,) + qtab.shape)
        for _ in range(2):
            pio._decoder(qv, z)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            pio._decoder(qv, z)
        e1.record()
        torch.cuda.synchronize()
    print(f"attention={p_attn:8s} mlp={p_mlp:8s} final={p_fin:8s}: worst {worst[0]:.2e} / {worst[1]:.2e}, decoder "
          f"{e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
