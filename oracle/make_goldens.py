"""Pin the oracle against the real reference and freeze golden vectors.  BUILD CONTAINER ONLY.

Imports the reference read-only from /root/reference (plus the init-only timm stand-in in
oracle/_standin), loads parameters produced by ``perceiver_oracle.gen_*`` into the reference
modules with ``load_state_dict(strict=True)`` (which also pins every state_dict name/shape),
runs the reference in float64 and float32, checks ``perceiver_oracle`` against both, and
writes inputs + reference float32 outputs to ``tests/golden/*.npz``.

Nothing from the reference's source is copied: fixtures hold tensors only.  On a machine
without /root/reference (the GPU box) this script exits 0 without doing anything.

    python oracle/make_goldens.py            # regenerate + pin (a few minutes on 8 cores)
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

if not os.path.isdir(REF):
    print("no /root/reference here: nothing to do")
    sys.exit(0)

sys.path[:0] = [os.path.join(HERE, "_standin"), REF, HERE]

import torch  # noqa: E402

import perceiver_oracle as O  # noqa: E402
from cases import ENCDEC_CASES, gen_encdec_inputs, _rand  # noqa: E402
from perceiver_io import transformer_primitives as RT  # noqa: E402  (the reference)
from perceiver_io import perceiver as RP  # noqa: E402

torch.manual_seed(0)
F64_TOL = 1e-11
F32_TOL = 3e-6
report = []


def _load(mod, params, dtype):
    sd = {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in params.items()}
    mod.load_state_dict(sd, strict=True)       # pins names + shapes
    return mod.to(dtype).eval()


def _t(x, dtype):
    if x is None:
        return None
    if x.dtype == bool:
        return torch.from_numpy(x)
    return torch.from_numpy(x).to(dtype)


def _check(name, got64, ref64, got32, ref32):
    e64 = O.rel_errors(got64, ref64)
    e32 = O.rel_errors(got32, ref32)
    e3264 = O.rel_errors(ref32, ref64)
    report.append((name, e64, e32, e3264))
    assert e64[1] <= F64_TOL, (name, "f64", e64)
    assert e32[1] <= F32_TOL, (name, "f32", e32)
    print(f"{name:42s} f64 {e64[1]:.2e}  f32 {e32[1]:.2e}  (ref32 vs ref64 {e3264[1]:.2e})")


def _save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrays)


def _pack(prefix, d):
    return {prefix + k: v for k, v in d.items()}


# ------------------------------------------------------------------------------------
# a1 mask
# ------------------------------------------------------------------------------------
def case_mask():
    rng = np.random.default_rng(1)
    qm = rng.random((3, 5)) > 0.4
    km = rng.random((3, 7)) > 0.4
    ref = RT.make_cross_attention_mask(torch.from_numpy(qm), torch.from_numpy(km)).numpy()
    got = O.make_cross_attention_mask(qm, km)
    assert ref.dtype == np.bool_ and got.dtype == np.bool_ and (ref == got).all()
    _save("mask", query_mask=qm, kv_mask=km, mask=ref)
    print("make_cross_attention_mask                  exact")


# ------------------------------------------------------------------------------------
# a3/a4 Attention
# ------------------------------------------------------------------------------------
ATTN_CASES = [
    # name, B, Tq, Tk, q_in, kv_in, heads, qk, v, out, mask kind
    ("attn_h1_nomask", 2, 9, 13, 24, 20, 1, 20, 20, 24, None),
    ("attn_h2_qkv_differ", 2, 16, 33, 32, 48, 2, 16, 40, 32, None),
    ("attn_h8_keymask", 3, 17, 70, 64, 64, 8, 64, 64, 64, "key"),
    ("attn_h8_querymask", 2, 40, 12, 48, 64, 8, 32, 96, 48, "query"),
    ("attn_h4_fullmask_row", 3, 10, 21, 32, 32, 4, 32, 32, 32, "key_allfalse_b1"),
    ("attn_h1_dim322", 1, 12, 50, 64, 322, 1, 322, 322, 64, None),
    ("attn_h8_lang_dims", 1, 8, 40, 96, 48, 8, 256, 160 * 8 // 8 * 1, 96, "key"),
]


def _mask_for(kind, B, Tq, Tk, seed):
    if kind is None:
        return None, None, None
    rng = np.random.default_rng(seed)
    qm = np.ones((B, Tq), dtype=bool)
    km = np.ones((B, Tk), dtype=bool)
    if kind == "key":
        km = rng.random((B, Tk)) > 0.3
        km[:, 0] = True
    elif kind == "query":
        qm = rng.random((B, Tq)) > 0.3
    elif kind == "key_allfalse_b1":
        km = rng.random((B, Tk)) > 0.3
        km[:, 0] = True
        km[1, :] = False
    return O.make_cross_attention_mask(qm, km), qm, km


def case_attention():
    for (name, B, Tq, Tk, q_in, kv_in, H, qk, v, out, mk) in ATTN_CASES:
        seed = 11
        p = O.gen_attention("", q_in, kv_in, qk, v, out, seed)
        xq = _rand(name + "xq", (B, Tq, q_in), seed)
        xkv = _rand(name + "xkv", (B, Tk, kv_in), seed)
        mask, qm, km = _mask_for(mk, B, Tq, Tk, seed)
        outs = {}
        for dt, npdt in ((torch.float64, np.float64), (torch.float32, np.float32)):
            m = RT.Attention(q_in_channels=q_in, k_in_channels=kv_in, v_in_channels=kv_in, num_heads=H,
                             qk_out_channels=qk, v_out_channels=v, output_channels=out)
            _load(m, p, dt)
            with torch.inference_mode():
                ref = m(_t(xq, dt), _t(xkv, dt), _t(xkv, dt), attention_mask=_t(mask, dt)).numpy()
            pp = {k: a.astype(npdt) for k, a in p.items()}
            got = O.attention(pp, xq.astype(npdt), xkv.astype(npdt), xkv.astype(npdt), H, mask)
            outs[npdt] = (got, ref)
        _check(name, *outs[np.float64], *outs[np.float32])
        extra = {}
        if mask is not None:
            extra = dict(query_mask=qm, kv_mask=km)
        _save(name, xq=xq, xkv=xkv, out=outs[np.float32][1],
              meta=np.array([B, Tq, Tk, q_in, kv_in, H, qk, v, out]), seed=np.array(seed),
              **_pack("p.", p), **extra)


# ------------------------------------------------------------------------------------
# a5 MLP
# ------------------------------------------------------------------------------------
def case_mlp():
    for name, cin, w in (("mlp_w1", 48, 1), ("mlp_w4", 40, 4)):
        seed = 12
        p = O.gen_mlp("", cin, w, cin, seed)
        x = _rand(name + "x", (2, 11, cin), seed, 2.0)
        outs = {}
        for dt, npdt in ((torch.float64, np.float64), (torch.float32, np.float32)):
            m = _load(RT.MLP(cin, widening_factor=w), p, dt)
            with torch.inference_mode():
                ref = m(_t(x, dt)).numpy()
            got = O.mlp({k: a.astype(npdt) for k, a in p.items()}, x.astype(npdt))
            outs[npdt] = (got, ref)
        _check(name, *outs[np.float64], *outs[np.float32])
        _save(name, x=x, out=outs[np.float32][1], meta=np.array([cin, w]), **_pack("p.", p))


# ------------------------------------------------------------------------------------
# a6 SelfAttention
# ------------------------------------------------------------------------------------
def case_self_attention():
    for name, B, N, D, H, w in (("sa_small", 2, 24, 64, 8, 1), ("sa_w4_h2", 1, 33, 32, 2, 4),
                                ("sa_mid_512x256_h8", 1, 512, 256, 8, 1)):
        seed = 13
        p = O.gen_self_attention("", D, seed, widening=w)
        x = _rand(name + "x", (B, N, D), seed, 1.5)
        outs = {}
        for dt, npdt in ((torch.float64, np.float64), (torch.float32, np.float32)):
            m = _load(RT.SelfAttention(D, widening_factor=w, num_heads=H), p, dt)
            with torch.inference_mode():
                ref = m(_t(x, dt)).numpy()
            got = O.self_attention({k: a.astype(npdt) for k, a in p.items()}, x.astype(npdt), H)
            outs[npdt] = (got, ref)
        _check(name, *outs[np.float64], *outs[np.float32])
        _save(name, x=x, out=outs[np.float32][1], meta=np.array([B, N, D, H, w]), **_pack("p.", p))


# ------------------------------------------------------------------------------------
# a7 CrossAttention
# ------------------------------------------------------------------------------------
def case_cross_attention():
    cases = (("ca_resid_kv", 2, 12, 30, 48, 20, 1, True, "kv", None),
             ("ca_noresid_q", 2, 12, 30, 32, 20, 4, False, "q", None),
             ("ca_keymask", 2, 16, 45, 64, 322, 1, True, "kv", "key"),
             ("ca_querymask_noresid", 2, 20, 16, 48, 64, 1, False, "kv", "query"))
    for (name, B, Tq, Tk, q_in, kv_in, H, resid, sfa, mk) in cases:
        seed = 14
        p = O.gen_cross_attention("", q_in, kv_in, seed, shape_for_attn=sfa)
        xq = _rand(name + "xq", (B, Tq, q_in), seed)
        xkv = _rand(name + "xkv", (B, Tk, kv_in), seed)
        mask, qm, km = _mask_for(mk, B, Tq, Tk, seed)
        outs = {}
        for dt, npdt in ((torch.float64, np.float64), (torch.float32, np.float32)):
            m = _load(RT.CrossAttention(q_in, kv_in, num_heads=H, shape_for_attn=sfa,
                                        use_query_residual=resid), p, dt)
            with torch.inference_mode():
                ref = m(_t(xq, dt), _t(xkv, dt), attention_mask=_t(mask, dt)).numpy()
            got = O.cross_attention({k: a.astype(npdt) for k, a in p.items()}, xq.astype(npdt),
                                    xkv.astype(npdt), H, resid, mask)
            outs[npdt] = (got, ref)
        _check(name, *outs[np.float64], *outs[np.float32])
        extra = dict(query_mask=qm, kv_mask=km) if mask is not None else {}
        _save(name, xq=xq, xkv=xkv, out=outs[np.float32][1],
              meta=np.array([B, Tq, Tk, q_in, kv_in, H, int(resid), int(sfa == "kv")]),
              **_pack("p.", p), **extra)


# ------------------------------------------------------------------------------------
# a8/a9/a10 Encoder + Decoder
# ------------------------------------------------------------------------------------
def _ref_encdec(cfg, p_enc, p_dec, qtab, x, dt, input_mask, query_mask):
    enc = RP.PerceiverEncoder(num_input_channels=cfg["C"], num_self_attends_per_block=cfg["L"],
                              num_blocks=cfg["blocks"], num_latents=cfg["N"], num_latent_channels=cfg["D"],
                              qk_channels=cfg.get("qk"), v_channels=cfg.get("v"),
                              num_cross_attend_heads=cfg["xh"], num_self_attend_heads=cfg["sh"],
                              use_query_residual=cfg["enc_resid"])
    dec = RP.PerceiverDecoder(query_channels=cfg["Dq"], final_project_out_channels=cfg["out"] or cfg["Dq"],
                              num_latent_channels=cfg["D"], qk_channels=cfg.get("dqk"), v_channels=cfg.get("dv"),
                              use_query_residual=cfg["dec_resid"], num_heads=cfg["dh"],
                              final_project=cfg["out"] is not None)
    _load(enc, p_enc, dt)
    _load(dec, p_dec, dt)
    with torch.inference_mode():
        xt = _t(x, dt)
        lat0 = enc.latents(xt)
        assert lat0.stride(0) == 0          # broadcast view of the parameter (position_encoding.py:120)
        z = enc(xt, lat0, input_mask=_t(input_mask, dt))
        q = torch.broadcast_to(_t(qtab, dt)[None], (x.shape[0],) + qtab.shape)
        y = dec(q, z, query_mask=_t(query_mask, dt))
    return z.numpy(), y.numpy()


def case_encdec(only=None):
    for name, cfg in ENCDEC_CASES.items():
        if only and name not in only:
            continue
        seed = 21
        p_enc, p_dec, qtab, x, im, qm = gen_encdec_inputs(name, cfg, seed)
        kw = dict(num_blocks=cfg["blocks"], num_self_attends_per_block=cfg["L"],
                  num_cross_attend_heads=cfg["xh"], num_self_attend_heads=cfg["sh"],
                  encoder_query_residual=cfg["enc_resid"], decoder_heads=cfg["dh"],
                  decoder_query_residual=cfg["dec_resid"], final_project=cfg["out"] is not None,
                  input_mask=im, query_mask=qm)
        big = name == "encdec_imagenet_b2"
        outs = {}
        for dt, npdt in ((torch.float64, np.float64), (torch.float32, np.float32)):
            zr, yr = _ref_encdec(cfg, p_enc, p_dec, qtab, x, dt, im, qm)
            cast = lambda d: {k: a.astype(npdt) for k, a in d.items()}  # noqa: E731
            z = O.encoder(cast(p_enc), x.astype(npdt), num_blocks=cfg["blocks"],
                          num_self_attends_per_block=cfg["L"], num_cross_attend_heads=cfg["xh"],
                          num_self_attend_heads=cfg["sh"], use_query_residual=cfg["enc_resid"], input_mask=im)
            y = O.encode_decode(cast(p_enc), cast(p_dec), x.astype(npdt), qtab.astype(npdt), **kw)
            outs[npdt] = (y, yr, z, zr)
        tol = 1e-9 if big else F64_TOL
        e64 = O.rel_errors(outs[np.float64][0], outs[np.float64][1])
        e32 = O.rel_errors(outs[np.float32][0], outs[np.float32][1])
        ez32 = O.rel_errors(outs[np.float32][2], outs[np.float32][3])
        e3264 = O.rel_errors(outs[np.float32][1], outs[np.float64][1])
        print(f"{name:42s} f64 {e64[1]:.2e}  f32 {e32[1]:.2e} (latents {ez32[1]:.2e})  ref32-vs-ref64 {e3264[1]:.2e}")
        assert e64[1] <= tol, (name, e64)
        assert e32[1] <= (2e-5 if big else 5e-6), (name, e32)
        report.append((name, e64, e32, e3264))
        y32, z32 = outs[np.float32][1], outs[np.float32][3]
        y64, z64 = outs[np.float64][1], outs[np.float64][3]
        meta = {k: (-1 if v is None else int(v)) for k, v in cfg.items() if k != "store"}
        meta_arr = np.array([f"{k}={v}" for k, v in meta.items()])
        if cfg["store"] == "full":
            extra = dict(input_mask=im, query_mask=qm) if cfg["masks"] else {}
            _save(name, x=x, qtab=qtab, out=y32, latents=z32, meta=meta_arr, seed=np.array(seed),
                  **_pack("enc.", p_enc), **_pack("dec.", p_dec), **extra)
        else:
            # sub-sampled outputs; float64 reference kept too (the "true" answer for reduced-precision kernels)
            rs = slice(0, None, max(1, cfg["Q"] // 8))
            _save(name, out_rows=np.arange(cfg["Q"])[rs], out=y32[:, rs, :], out64=y64[:, rs, :],
                  latents_sub=z32[:, ::8, ::8], latents_sub64=z64[:, ::8, ::8],
                  out_absmax=np.array(np.abs(y64).max()), out_l2=np.array(np.sqrt((y64 ** 2).sum())),
                  meta=meta_arr, seed=np.array(seed))


if __name__ == "__main__" and not (sys.argv[1:] and all(a.startswith("model_") for a in sys.argv[1:])):
    only = set(a for a in sys.argv[1:] if not a.startswith("model_"))
    if not only:
        case_mask()
        case_attention()
        case_mlp()
        case_self_attention()
        case_cross_attention()
    case_encdec(only or None)
    print("oracle pinned against the reference; goldens written to", GOLD)


# ------------------------------------------------------------------------------------
# whole models ("next" rows): reference task models with generated parameters -> outputs
# ------------------------------------------------------------------------------------
def case_models(only=None):
    from cases import MODEL_CASES, gen_state_dict, model_inputs, model_seed, model_stats
    from perceiver_io.classification_perceiver import ClassificationPerceiver, PrepType
    from perceiver_io.flow_perceiver import FlowPerceiver
    from perceiver_io.language_perceiver import LanguagePerceiver
    from perceiver_io.multimodal_perceiver import MultiModalPerceiver
    for name, c in MODEL_CASES.items():
        if only and name not in only:
            continue
        kw = dict(c["kw"])
        if c["cls"] == "ClassificationPerceiver":
            model = ClassificationPerceiver(prep_type=PrepType[kw.pop("prep")])
        else:
            model = {"LanguagePerceiver": LanguagePerceiver, "FlowPerceiver": FlowPerceiver,
                     "MultiModalPerceiver": MultiModalPerceiver}[c["cls"]](**kw)
        sd = model.state_dict()
        spec = [(k, tuple(v.shape)) for k, v in sd.items()]
        params = gen_state_dict(spec, model_seed(name), model_stats(name))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        model.eval()
        ins = model_inputs(name)
        tin = [torch.from_numpy(a) for a in ins]
        save = dict(spec_names=np.array([k for k, _ in spec]),
                    spec_shapes=np.array([",".join(str(d) for d in s) for _, s in spec]))
        with torch.inference_mode():
            if name.startswith("model_flow_full"):
                out = model(tin[0], tin[1]).numpy()                      # [1, 2, 368, 496]
                save.update(out_sub=out[:, :, ::8, ::8], out_absmax=np.array(np.abs(out).max()),
                            out_l2=np.array(np.sqrt((out.astype(np.float64) ** 2).sum())))
                print(f"{name:42s} flow {out.shape} absmax {np.abs(out).max():.3f}")
            elif c["cls"] == "FlowPerceiver":
                out_train = model(tin[0][..., :48, :64], tin[1][..., :48, :64]).numpy()
                out_test = model(tin[0], tin[1], test_mode=True, min_overlap=10).numpy()
                save.update(out_train=out_train, out_test=out_test)
                print(f"{name:42s} train {out_train.shape} test {out_test.shape}")
            elif name.startswith("model_multimodal_full"):
                # single output chunks of the reference's loop (multimodal_perceiver.py:146-157): the perceiver call
                # the reference makes for chunk k, nothing else
                images, audio = tin
                b, t, ch, h, w = images.shape
                n_chunks = c["n_chunks"]
                ics = t * h * w // n_chunks
                acs = audio.shape[1] // model.audio_samples_per_patch // n_chunks
                for k in c["chunks"]:
                    sub = {"image": torch.arange(ics * k, ics * (k + 1)), "audio": torch.arange(acs * k, acs * (k + 1)),
                           "label": None}
                    out = model.perceiver({"image": images, "audio": audio,
                                           "label": torch.zeros((b, model.num_classes))},
                                          subsampled_output_points=sub)
                    save.update({f"out_image_{k}": out["image"].numpy(), f"out_audio_{k}": out["audio"].numpy(),
                                 f"out_label_{k}": out["label"].numpy()})
                    print(f"{name:42s} chunk {k}: image {tuple(out['image'].shape)} audio {tuple(out['audio'].shape)} "
                          f"label {tuple(out['label'].shape)} absmax {out['image'].abs().max():.3f}")
            elif c["cls"] == "MultiModalPerceiver":
                out = model(tin[0], tin[1], n_chunks=2)
                save.update(out_image=out["image"].numpy(), out_audio=out["audio"].numpy(),
                            out_label=out["label"].numpy())
                print(f"{name:42s} image {tuple(out['image'].shape)} audio {tuple(out['audio'].shape)}")
            elif c["cls"] == "LanguagePerceiver":
                out = model(tin[0], tin[1]).numpy()
                save.update(out=out[:, :96], out_tail=out[:, 640:704], out_absmax=np.array(np.abs(out).max()))
                print(f"{name:42s} logits {out.shape}")
            else:
                out = model(tin[0]).numpy()
                save.update(out=out)
                print(f"{name:42s} logits {out.shape} absmax {np.abs(out).max():.3f}")
        _save(name, **save)


if __name__ == "__main__" and (not sys.argv[1:] or any(a.startswith("model_") for a in sys.argv[1:])):
    case_models(set(a for a in sys.argv[1:] if a.startswith("model_")) or None)
