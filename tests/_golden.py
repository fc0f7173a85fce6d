"""Helpers to load committed golden fixtures (reference float32 outputs frozen by oracle/make_goldens.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def params(g, prefix="p."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def meta_dict(g):
    out = {}
    for item in g["meta"]:
        k, v = str(item).split("=")
        out[k] = None if v == "-1" else int(v)
    return out


ATTN = ["attn_h1_nomask", "attn_h2_qkv_differ", "attn_h8_keymask", "attn_h8_querymask",
        "attn_h4_fullmask_row", "attn_h1_dim322", "attn_h8_lang_dims"]
MLP = ["mlp_w1", "mlp_w4"]
SA = ["sa_small", "sa_w4_h2", "sa_mid_512x256_h8"]
CA = ["ca_resid_kv", "ca_noresid_q", "ca_keymask", "ca_querymask_noresid"]
ENCDEC_FULL = ["encdec_tiny", "encdec_tiny_masked", "encdec_tiny_decresid"]
ENCDEC_SUB = ["encdec_mid", "encdec_lang_like"]      # + encdec_imagenet_b2 (slow on CPU: oracle ~20 s)
