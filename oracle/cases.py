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
