"""CPU: the oracle restatement must reproduce every committed golden (reference float32 outputs)."""
import numpy as np
import pytest

import perceiver_oracle as O
from cases import ENCDEC_CASES, gen_encdec_inputs, encdec_kwargs
from _golden import load, params, ATTN, MLP, SA, CA, ENCDEC_FULL, ENCDEC_SUB

TOL = 5e-6   # float32 oracle vs float32 reference (different BLAS summation order only)


def _mask(g):
    if "query_mask" in g:
        return O.make_cross_attention_mask(g["query_mask"], g["kv_mask"])
    return None


def test_mask_golden():
    g = load("mask")
    m = O.make_cross_attention_mask(g["query_mask"], g["kv_mask"])
    assert m.dtype == np.bool_ and (m == g["mask"]).all()


@pytest.mark.parametrize("name", ATTN)
def test_attention_golden(name):
    g = load(name)
    H = int(g["meta"][5])
    y = O.attention(params(g), g["xq"], g["xkv"], g["xkv"], H, _mask(g))
    assert O.rel_errors(y, g["out"])[1] <= TOL


def test_fully_masked_rows_give_final_bias():
    g = load("attn_h4_fullmask_row")
    p = params(g)
    y = O.attention(p, g["xq"], g["xkv"], g["xkv"], 4, _mask(g))
    # sample 1 has every key masked -> attend output wiped -> out == final.bias (transformer_primitives.py:168-175)
    assert np.allclose(y[1], np.broadcast_to(p["final.bias"], y[1].shape), atol=0, rtol=0)
    assert np.array_equal(g["out"][1], np.broadcast_to(p["final.bias"], y[1].shape))


@pytest.mark.parametrize("name", MLP)
def test_mlp_golden(name):
    g = load(name)
    assert O.rel_errors(O.mlp(params(g), g["x"]), g["out"])[1] <= TOL


@pytest.mark.parametrize("name", SA)
def test_self_attention_golden(name):
    g = load(name)
    H = int(g["meta"][3])
    assert O.rel_errors(O.self_attention(params(g), g["x"], H), g["out"])[1] <= TOL


@pytest.mark.parametrize("name", CA)
def test_cross_attention_golden(name):
    g = load(name)
    H, resid = int(g["meta"][5]), bool(g["meta"][6])
    y = O.cross_attention(params(g), g["xq"], g["xkv"], H, resid, _mask(g))
    assert O.rel_errors(y, g["out"])[1] <= TOL


@pytest.mark.parametrize("name", ENCDEC_FULL)
def test_encdec_full_golden(name):
    g = load(name)
    cfg = ENCDEC_CASES[name]
    p_enc, p_dec = params(g, "enc."), params(g, "dec.")
    # the generator must still reproduce the stored parameters bit for bit (numpy PCG64 is stable)
    ge, gd, qtab, x, im, qm = gen_encdec_inputs(name, cfg, int(g["seed"]))
    for k in p_enc:
        assert np.array_equal(ge[k], p_enc[k]), k
    assert np.array_equal(x, g["x"]) and np.array_equal(qtab, g["qtab"])
    y = O.encode_decode(p_enc, p_dec, g["x"], g["qtab"], **encdec_kwargs(cfg, g.get("input_mask"), g.get("query_mask")))
    assert O.rel_errors(y, g["out"])[1] <= TOL


@pytest.mark.parametrize("name", ENCDEC_SUB)
def test_encdec_subsampled_golden(name):
    g = load(name)
    cfg = ENCDEC_CASES[name]
    p_enc, p_dec, qtab, x, im, qm = gen_encdec_inputs(name, cfg, int(g["seed"]))
    y = O.encode_decode(p_enc, p_dec, x, qtab, **encdec_kwargs(cfg, im, qm))
    sub = y[:, g["out_rows"], :]
    assert O.rel_errors(sub, g["out"])[1] <= TOL
    assert O.rel_errors(sub, g["out64"])[1] <= 2e-5


# ---- the torch restatement (oracle/perceiver_oracle_torch.py: what bench.py's cpu_baseline leg times) is pinned by the
# ---- same reference goldens
def _tt(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("name", ATTN)
def test_torch_oracle_attention_golden(name):
    import perceiver_oracle_torch as OT
    g = load(name)
    H = int(g["meta"][5])
    m = _mask(g)
    y = OT.attention(OT.to_torch(params(g)), _tt(g["xq"]), _tt(g["xkv"]), _tt(g["xkv"]), H,
                     _tt(m) if m is not None else None).numpy()
    assert O.rel_errors(y, g["out"])[1] <= TOL


@pytest.mark.parametrize("name", CA + SA)
def test_torch_oracle_blocks_golden(name):
    import perceiver_oracle_torch as OT
    g = load(name)
    p = OT.to_torch(params(g))
    if name in SA:
        y = OT.self_attention(p, _tt(g["x"]), int(g["meta"][3])).numpy()
    else:
        m = _mask(g)
        y = OT.cross_attention(p, _tt(g["xq"]), _tt(g["xkv"]), int(g["meta"][5]), bool(g["meta"][6]),
                               _tt(m) if m is not None else None).numpy()
    assert O.rel_errors(y, g["out"])[1] <= TOL


@pytest.mark.parametrize("name", ENCDEC_FULL + [n for n in ENCDEC_SUB if n != "encdec_imagenet_b2"])
def test_torch_oracle_encdec_golden(name):
    import perceiver_oracle_torch as OT
    g = load(name)
    cfg = ENCDEC_CASES[name]
    p_enc, p_dec, qtab, x, im, qm = gen_encdec_inputs(name, cfg, int(g["seed"]))
    y = OT.encode_decode(p_enc, p_dec, x, qtab, **encdec_kwargs(cfg, im, qm))
    if name in ENCDEC_SUB:
        y = y[:, g["out_rows"], :]
    assert O.rel_errors(y, g["out"])[1] <= TOL
