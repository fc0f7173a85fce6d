"""Seeded case registry shared by oracle/make_goldens.py and tests/ -- TEST INFRASTRUCTURE.

Holds only shapes, seeds and the recipe that regenerates inputs/parameters through
``perceiver_oracle.gen_*``; it never touches the reference, so it also runs on the GPU box.
"""
from __future__ import annotations

import numpy as np

import perceiver_oracle as O


def _rand(name, shape, seed, scale=1.0):
    return (scale * O._rng_for(name, seed).standard_normal(shape)).astype(np.float32)


ENCDEC_CASES = {
    # tiny, full tensors stored
    "encdec_tiny": dict(B=2, M=50, C=20, N=16, D=32, L=2, blocks=2, xh=1, sh=4, enc_resid=True,
                        Q=10, Dq=24, out=7, dh=1, dec_resid=False, masks=False, store="full"),
    "encdec_tiny_masked": dict(B=3, M=40, C=24, N=16, D=32, L=2, blocks=2, xh=2, sh=4, enc_resid=True,
                               Q=40, Dq=24, out=None, dh=2, dec_resid=False, masks=True, store="full",
                               qk=16, v=32, dqk=16, dv=24),
    "encdec_tiny_decresid": dict(B=2, M=30, C=322, N=24, D=64, L=3, blocks=1, xh=1, sh=8, enc_resid=True,
                                 Q=12, Dq=64, out=10, dh=1, dec_resid=True, masks=False, store="full"),
    # mid size: a few seconds of oracle time; sub-sampled outputs stored, weights regenerated from seed
    "encdec_mid": dict(B=2, M=784, C=322, N=128, D=256, L=2, blocks=3, xh=1, sh=8, enc_resid=True,
                       Q=100, Dq=256, out=100, dh=1, dec_resid=True, masks=False, store="sub"),
    "encdec_lang_like": dict(B=2, M=512, C=192, N=64, D=320, L=3, blocks=1, xh=8, sh=8, enc_resid=True,
                             qk=64, v=320, Q=512, Dq=192, out=None, dh=8, dec_resid=False, dqk=64, dv=192,
                             masks=True, store="sub"),
    # the headline config at B=2 (ImageNet-224 conv preprocessing: M=3136, C=322; 512x1024 latents; 8x6 SA;
    # 1000 learned queries x 1024; final Linear 1024->1000): regenerated weights, sub-sampled outputs
    "encdec_imagenet_b2": dict(B=2, M=3136, C=322, N=512, D=1024, L=6, blocks=8, xh=1, sh=8, enc_resid=True,
                               Q=1000, Dq=1024, out=1000, dh=1, dec_resid=True, masks=False, store="sub"),
}


def gen_encdec_inputs(name, cfg, seed):
    """(p_enc, p_dec, query_table, x, input_mask, query_mask) for a registry entry."""
    p_enc = O.gen_encoder(cfg["C"], cfg["N"], cfg["D"], cfg["L"], seed, qk=cfg.get("qk"), v=cfg.get("v"))
    p_dec = O.gen_decoder(cfg["Dq"], cfg["D"], cfg["out"], seed + 1, qk=cfg.get("dqk"), v=cfg.get("dv"))
    qtab = O.gen_tensor("query_table", (cfg["Q"], cfg["Dq"]), seed, "table")
    x = _rand(name + "x", (cfg["B"], cfg["M"], cfg["C"]), seed)
    im = qm = None
    if cfg["masks"]:
        rng = np.random.default_rng(seed)
        im = np.zeros((cfg["B"], cfg["M"]), dtype=bool)
        qm = np.zeros((cfg["B"], cfg["Q"]), dtype=bool)
        for b in range(cfg["B"]):
            im[b, : int(rng.integers(cfg["M"] // 4, cfg["M"]))] = True
            qm[b, : int(rng.integers(cfg["Q"] // 4, cfg["Q"]))] = True
    return p_enc, p_dec, qtab, x, im, qm


def encdec_kwargs(cfg, im=None, qm=None):
    """Keyword arguments of ``perceiver_oracle.encode_decode`` for a registry entry."""
    return dict(num_blocks=cfg["blocks"], num_self_attends_per_block=cfg["L"],
                num_cross_attend_heads=cfg["xh"], num_self_attend_heads=cfg["sh"],
                encoder_query_residual=cfg["enc_resid"], decoder_heads=cfg["dh"],
                decoder_query_residual=cfg["dec_resid"], final_project=cfg["out"] is not None,
                input_mask=im, query_mask=qm)


# ---------------------------------------------------------------------------------------------------
# whole-model cases ("next" rows of SURVEY.md 8f): parameters for an arbitrary state_dict spec
# ---------------------------------------------------------------------------------------------------
def gen_state_dict(spec, seed, stats=None):
    """spec: iterable of (name, shape).  Deterministic, non-trivial values for every entry kind that occurs in the
    reference's models (weights ~ fan-in scaled, biases / LayerNorm / BatchNorm statistics perturbed).

    stats="trained": parameter statistics of a TRAINED checkpoint instead of an initialiser's -- the rows of every
    nn.Linear weight scaled by a log-normal factor spanning about a decade (sigma = ln(10) / 4, mean square kept), three
    outlier input channels per wide Linear (x 6), LayerNorm gains log-uniform in [0.2, 5] with biases N(0, 0.3).  Heavy-
    tailed rows and large LayerNorm gains are what the fp16 operand rounding and the LayerNorm fold (weights x gamma) are
    most sensitive to; a truncated-normal initialisation has neither."""
    out = {}
    for name, shape in spec:
        shape = tuple(int(s) for s in shape)
        leaf = name.rsplit(".", 1)[-1]
        rng = O._rng_for(name, seed)
        if stats == "trained" and "layer_norm" in name and len(shape) == 1:
            if leaf == "weight":
                out[name] = np.exp(rng.uniform(np.log(0.2), np.log(5.0), shape)).astype(np.float32)
            else:
                out[name] = (0.3 * rng.standard_normal(shape)).astype(np.float32)
            continue
        if stats == "trained" and leaf == "weight" and len(shape) == 2 and ".embed." not in name \
                and not name.endswith("_embedding.weight"):
            w = np.clip(rng.standard_normal(shape), -2.0, 2.0) / 0.87962566103423978 / np.sqrt(shape[1])
            srow = np.exp((np.log(10.0) / 4.0) * rng.standard_normal(shape[0]))
            w = w * (srow / np.sqrt(np.mean(srow * srow)))[:, None]
            if shape[1] >= 256:
                w[:, rng.choice(shape[1], 3, replace=False)] *= 6.0
            out[name] = w.astype(np.float32)
            continue
        if leaf == "num_batches_tracked":
            out[name] = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_mean":
            out[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "running_var":
            out[name] = (1.0 + 0.2 * np.abs(rng.standard_normal(shape))).astype(np.float32)
        elif leaf == "pos_embs":
            out[name] = O.gen_tensor(name, shape, seed, "table") if int(np.prod(shape)) else np.zeros(shape, np.float32)
        elif leaf == "bias":
            out[name] = O.gen_tensor(name, shape, seed, "bias")
        elif leaf == "weight" and len(shape) == 1:
            out[name] = O.gen_tensor(name, shape, seed, "ln_weight")
        elif leaf == "weight" and ".embed." in name or name.endswith("_embedding.weight"):
            out[name] = (0.3 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            w = np.clip(rng.standard_normal(shape), -2.0, 2.0) / 0.87962566103423978
            out[name] = (w / np.sqrt(fan_in)).astype(np.float32)
        else:
            raise ValueError(f"don't know how to generate {name} {shape}")
    return out


MODEL_CASES = {
    # name: (class, ctor kwargs, input recipe)
    "model_classify_conv": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_CONVNET"), batch=2),
    "model_classify_1x1": dict(cls="ClassificationPerceiver", kw=dict(prep="LEARNED_POS_1X1CONV"), batch=1),
    "model_classify_pixel": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_PIXEL"), batch=1),
    "model_language": dict(cls="LanguagePerceiver", kw=dict(), batch=2),
    # two more parameter / token seeds with other ragged lengths (late round 3: the class default moved to a three-part
    # policy with a thinner stack policy; one golden was not enough to call its margin)
    "model_language_s32": dict(cls="LanguagePerceiver", kw=dict(), batch=2, pseed=32, lengths=(2048, 333)),
    "model_language_s33": dict(cls="LanguagePerceiver", kw=dict(), batch=2, pseed=33, lengths=(1, 1290)),
    "model_flow_small": dict(cls="FlowPerceiver", kw=dict(img_size=(48, 64), num_latents=128, num_latent_channels=128,
                                                          num_self_attends_per_block=2), batch=1),
    # full-size optical-flow configuration: M = Q = 368*496 = 182 528 tokens, 2048 x 512 latents, 24 self-attends
    "model_flow_full": dict(cls="FlowPerceiver", kw=dict(), batch=1),
    "model_multimodal_small": dict(cls="MultiModalPerceiver",
                                   kw=dict(img_size=(16, 16), num_frames=2, num_classes=10,
                                           audio_samples_per_frame=32, audio_samples_per_patch=16,
                                           num_self_attends_per_block=1, num_latents=32, num_latent_channels=512),
                                   batch=2),
    # the BENCHMARKED code path (>= 2048 latent rows: LayerNorm fold + 16-bit-pair residual stream) pinned to the
    # reference directly: B=4, three parameter / input seeds
    "model_classify_b4_s31": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_CONVNET"), batch=4, pseed=31),
    "model_classify_b4_s32": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_CONVNET"), batch=4, pseed=32),
    "model_classify_b4_s33": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_CONVNET"), batch=4, pseed=33),
    "model_classify_b4_s34": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_CONVNET"), batch=4, pseed=34),
    "model_classify_b4_s35": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_CONVNET"), batch=4, pseed=35),
    # the same path on NON-Gaussian inputs: synthetic images with natural-image statistics (1/f amplitude spectrum,
    # piecewise-constant regions with sharp edges, sparse saturated highlights / shadows -- heavy-tailed after the
    # ImageNet mean / std normalisation of example_img_classify.py:58-60), parameter seed 36
    "model_classify_b4_natural": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_CONVNET"), batch=4,
                                      pseed=36, inputs="natural"),
    # full-size multimodal auto-encoder (BASELINE config 5): 16 x 224 x 224 video + 30720 audio samples + label,
    # M = 52 097 x 704, 784 x 512 latents; output chunks 0 and 127 of n_chunks = 128 (6272 + 15 + 1 queries each)
    "model_multimodal_full": dict(cls="MultiModalPerceiver", kw=dict(), batch=1, chunks=(0, 127), n_chunks=128),
    # round 4: what the headline margin is measured on, widened --
    # two B = 4 classifier goldens with TRAINED-LIKE parameter statistics (gen_state_dict(stats="trained"): log-normal row
    # scales over a decade, LayerNorm gains in [0.2, 5], outlier channels), one on N(0,1) pixels, one on natural-image inputs
    "model_classify_b4_trained1": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_CONVNET"), batch=4,
                                       pseed=41, stats="trained"),
    "model_classify_b4_trained2": dict(cls="ClassificationPerceiver", kw=dict(prep="FOURIER_POS_CONVNET"), batch=4,
                                       pseed=42, stats="trained", inputs="natural"),
    # ... a language model with the same trained-like statistics (two ragged lengths)
    "model_language_trained": dict(cls="LanguagePerceiver", kw=dict(), batch=2, pseed=43, stats="trained",
                                   lengths=(1700, 420)),
    # ... and a second parameter / input seed for each dense-output model at full size
    "model_flow_full_s32": dict(cls="FlowPerceiver", kw=dict(), batch=1, pseed=32),
    "model_multimodal_full_s32": dict(cls="MultiModalPerceiver", kw=dict(), batch=1, chunks=(3, 77), n_chunks=128,
                                      pseed=32),
}


def natural_images(name, B, seed, hw=(224, 224)):
    """Synthetic [B,3,H,W] images with natural-image statistics, ImageNet-normalised: a 1/f field per channel (strongly
    correlated across channels) squashed to [0,1], overlaid with piecewise-constant rectangles (sharp edges) and a few
    saturated highlights / shadows, then (x - mean) / std with the ImageNet constants -- a heavy-tailed, spatially
    correlated, non-zero-mean input, unlike the N(0,1) pixels of the other cases."""
    rng = O._rng_for(name + "natural", seed)
    H, W = hw
    fy = np.fft.fftfreq(H)[:, None]
    fx = np.fft.fftfreq(W)[None, :]
    amp = 1.0 / np.maximum(np.sqrt(fx * fx + fy * fy), 1.0 / max(H, W))
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float64)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float64)
    out = np.empty((B, 3, H, W), dtype=np.float32)
    for b in range(B):
        base = np.fft.ifft2(amp * np.fft.fft2(rng.standard_normal((H, W)))).real
        img = np.empty((3, H, W))
        for ch in range(3):
            own = np.fft.ifft2(amp * np.fft.fft2(rng.standard_normal((H, W)))).real
            f = 0.8 * base + 0.2 * own
            f = (f - f.mean()) / (f.std() + 1e-12)
            img[ch] = 1.0 / (1.0 + np.exp(-1.5 * f + rng.normal(0, 0.5)))            # [0,1], exposure varies
        for _ in range(int(rng.integers(4, 10))):                                      # objects: flat regions, sharp edges
            y0, x0 = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
            h, w = int(rng.integers(8, H // 2)), int(rng.integers(8, W // 2))
            img[:, y0:y0 + h, x0:x0 + w] = 0.6 * img[:, y0:y0 + h, x0:x0 + w] + 0.4 * rng.random(3)[:, None, None]
        for _ in range(int(rng.integers(2, 6))):                                       # saturated highlights / shadows
            y0, x0 = int(rng.integers(0, H - 4)), int(rng.integers(0, W - 4))
            h, w = int(rng.integers(2, 24)), int(rng.integers(2, 24))
            img[:, y0:y0 + h, x0:x0 + w] = float(rng.integers(0, 2))
        img = np.clip(img + 0.02 * rng.standard_normal(img.shape), 0.0, 1.0)           # sensor noise, clipped
        out[b] = ((img - mean[:, None, None]) / std[:, None, None]).astype(np.float32)
    return out


def model_seed(name):
    return MODEL_CASES[name].get("pseed", 31)


def model_stats(name):
    """gen_state_dict's `stats` argument of a MODEL_CASES entry (None = initialiser-like parameters)."""
    return MODEL_CASES[name].get("stats")


def model_inputs(name, seed=None):
    """Seeded inputs of a MODEL_CASES entry as numpy arrays (dict of forward kwargs / positional list)."""
    c = MODEL_CASES[name]
    B = c["batch"]
    seed = model_seed(name) if seed is None else seed
    if c["cls"] == "ClassificationPerceiver":
        if c.get("inputs") == "natural":
            return [natural_images(name, B, seed)]
        return [_rand(name + "img", (B, 3, 224, 224), seed)]
    if c["cls"] == "LanguagePerceiver":
        rng = np.random.default_rng(seed)
        tok = rng.integers(6, 262, size=(B, 2048)).astype(np.int64)
        mask = np.zeros((B, 2048), dtype=bool)
        for b, n in zip(range(B), c.get("lengths", (60, 700))):
            mask[b, :n] = True
        tok[~mask] = 0
        return [tok, mask]
    if c["cls"] == "FlowPerceiver":
        hw = (368, 496) if name.startswith("model_flow_full") else (60, 80)
        return [_rand(name + "i1", (B, 3) + hw, seed), _rand(name + "i2", (B, 3) + hw, seed)]
    if c["cls"] == "MultiModalPerceiver":
        kw = dict(num_frames=16, img_size=(224, 224), audio_samples_per_frame=48000 // 25)
        kw.update(c["kw"])
        return [np.abs(_rand(name + "v", (B, kw["num_frames"], 3) + tuple(kw["img_size"]), seed)),
                _rand(name + "a", (B, kw["num_frames"] * kw["audio_samples_per_frame"], 1), seed)]
    raise ValueError(name)
