"""CPU: the nn.Module mirror keeps the reference's constructor surface, state_dict names and error behaviour;
the product path refuses to compute without the GPU (no fallback)."""
import numpy as np
import pytest
import torch

import perceiver_oracle as O
from cases import ENCDEC_CASES, gen_encdec_inputs
from _golden import load, params


def test_state_dict_names_and_shapes_match_reference_goldens():
    from perceiverio_pytorch_amd.transformer_primitives import Attention, CrossAttention, MLP, SelfAttention
    g = load("attn_h8_lang_dims")
    B, Tq, Tk, q_in, kv_in, H, qk, v, out = (int(x) for x in g["meta"])
    m = Attention(q_in, kv_in, kv_in, num_heads=H, qk_out_channels=qk, v_out_channels=v, output_channels=out)
    m.load_state_dict({k: torch.from_numpy(a) for k, a in params(g).items()}, strict=True)
    g = load("sa_w4_h2")
    B, N, D, H, w = (int(x) for x in g["meta"])
    SelfAttention(D, widening_factor=w, num_heads=H).load_state_dict(
        {k: torch.from_numpy(a) for k, a in params(g).items()}, strict=True)
    g = load("ca_noresid_q")
    B, Tq, Tk, q_in, kv_in, H, resid, kv = (int(x) for x in g["meta"])
    CrossAttention(q_in, kv_in, num_heads=H, shape_for_attn="q", use_query_residual=False).load_state_dict(
        {k: torch.from_numpy(a) for k, a in params(g).items()}, strict=True)
    g = load("mlp_w4")
    MLP(int(g["meta"][0]), widening_factor=int(g["meta"][1])).load_state_dict(
        {k: torch.from_numpy(a) for k, a in params(g).items()}, strict=True)


def test_encoder_decoder_state_dict_matches_generator_names():
    from perceiverio_pytorch_amd.perceiver import PerceiverDecoder, PerceiverEncoder
    cfg = ENCDEC_CASES["encdec_tiny_masked"]
    p_enc, p_dec, *_ = gen_encdec_inputs("x", cfg, 1)
    enc = PerceiverEncoder(cfg["C"], cfg["L"], cfg["blocks"], cfg["N"], cfg["D"], qk_channels=cfg["qk"],
                           v_channels=cfg["v"], num_cross_attend_heads=cfg["xh"], num_self_attend_heads=cfg["sh"])
    dec = PerceiverDecoder(cfg["Dq"], cfg["Dq"], cfg["D"], qk_channels=cfg["dqk"], v_channels=cfg["dv"],
                           num_heads=cfg["dh"], final_project=False)
    assert sorted(enc.state_dict()) == sorted(p_enc)
    assert sorted(dec.state_dict()) == sorted(p_dec)
    for k, v in enc.state_dict().items():
        assert tuple(v.shape) == p_enc[k].shape, k
    lat = enc.latents(torch.zeros(5, 3, cfg["C"]))
    assert lat.shape == (5, cfg["N"], cfg["D"]) and lat.stride(0) == 0      # broadcast view, like the reference


def test_constructor_errors_follow_the_reference():
    from perceiverio_pytorch_amd.perceiver import PerceiverDecoder, PerceiverEncoder
    from perceiverio_pytorch_amd.transformer_primitives import Attention, CrossAttention
    with pytest.raises(ValueError, match="qk_out_channels"):
        Attention(30, 30, 30, num_heads=8)                      # transformer_primitives.py:66-68
    with pytest.raises(ValueError, match="v_channels"):
        Attention(32, 32, 32, num_heads=8, v_out_channels=36)   # :69-71
    with pytest.raises(TypeError):
        Attention(8)                                            # k_in_channels=None reaches nn.Linear
    with pytest.raises(ValueError, match="shape_for_attention"):
        CrossAttention(8, 8, shape_for_attn="x")                # :342
    with pytest.raises(ValueError, match="num_self_attend_heads"):
        PerceiverEncoder(8, num_latent_channels=30)             # perceiver.py:54-56
    with pytest.raises(ValueError, match="output_w_init"):
        PerceiverDecoder(8, 8, output_w_init="ones")            # perceiver.py:163


def test_no_cpu_fallback():
    from perceiverio_pytorch_amd import PioError
    from perceiverio_pytorch_amd.transformer_primitives import MLP, SelfAttention
    with pytest.raises(PioError, match="no CPU or eager fallback"):
        SelfAttention(16, num_heads=2)(torch.zeros(1, 4, 16))
    with pytest.raises(PioError):
        MLP(16)(torch.zeros(1, 4, 16))


def test_product_package_never_imports_the_oracle():
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "perceiverio_pytorch_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "perceiver_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_precision_policy_switch():
    import perceiverio_pytorch_amd as P
    old = P.get_precision_policy()
    P.set_precision_policy("fp16x2w")
    assert P.get_precision_policy() == "fp16x2w"
    with pytest.raises(ValueError):
        P.set_precision_policy("fp8")
    P.set_precision_policy(old)


def test_mask_helper_matches_oracle():
    from perceiverio_pytorch_amd.transformer_primitives import make_cross_attention_mask
    g = load("mask")
    m = make_cross_attention_mask(torch.from_numpy(g["query_mask"]), torch.from_numpy(g["kv_mask"]))
    assert m.dtype == torch.bool and np.array_equal(m.numpy(), g["mask"])


def test_reference_import_paths_resolve():
    """The reference's example scripts import these names; the shims must keep them alive."""
    from perceiver_io.classification_perceiver import ClassificationPerceiver, PrepType            # noqa: F401
    from perceiver_io.multimodal_perceiver import MultiModalPerceiver                             # noqa: F401
    from perceiver_io.language_perceiver import LanguagePerceiver                                 # noqa: F401
    from perceiver_io.flow_perceiver import FlowPerceiver                                         # noqa: F401
    from perceiver_io.perceiver import PerceiverIO, PerceiverEncoder, PerceiverDecoder            # noqa: F401
    from perceiver_io.transformer_primitives import Attention, CrossAttention, MLP, SelfAttention  # noqa: F401
    from perceiver_io.io_processors.preprocessors import ImagePreprocessor                        # noqa: F401
    from perceiver_io.io_processors.postprocessors import ClassificationPostprocessor             # noqa: F401
    from perceiver_io.output_queries import TrainableQuery, FourierQuery, FlowQuery               # noqa: F401
    from utils.bytes_tokenizer import BytesTokenizer
    from utils.flow_utils import flow_to_image
    from utils.utils import load_image, show_animation                                            # noqa: F401
    from utils.imagenet_labels import IMAGENET_LABELS
    from utils.kinetics_700_classes import KINETICS_CLASSES
    t = BytesTokenizer()
    ids = t.to_int("Perceiver IO")
    assert t.to_string(ids) == "Perceiver IO" and int(ids.min()) >= 6 and t.vocab_size == 262 and t.mask_token == 3
    assert flow_to_image(np.zeros((3, 4, 2))).shape == (3, 4, 3)
    assert len(IMAGENET_LABELS) == 1000 and len(KINETICS_CLASSES) == 700


def test_perceiver_io_constructor_surface():
    """The five north_star constructor names + the raw (no pre/post-processor) usage of SURVEY appendix C."""
    from perceiver_io.output_queries import TrainableQuery
    from perceiver_io.perceiver import PerceiverIO
    m = PerceiverIO(num_blocks=2, num_self_attends_per_block=2, num_latents=8, num_latent_channels=16,
                    input_channels=24, output_queries=TrainableQuery(output_index_dims=5, num_channels=12),
                    perceiver_encoder_kwargs=dict(num_self_attend_heads=4))
    keys = set(m.state_dict())
    assert "padding_embeddings.__default.pos_embs" in keys and "_encoder.latent_pos_enc.pos_embs" in keys
    assert "_output_queries.__default._position_encoding.pos_embs" in keys
    assert m.state_dict()["padding_embeddings.__default.pos_embs"].shape == (1, 0)
    assert len(keys) == 73


def test_backward_through_the_hip_path_raises():
    """Outputs of the HIP modules are tied to their inputs / parameters with a backward that raises, so a training
    step cannot silently skip the encoder / decoder weights (runtime.forward_only)."""
    from perceiverio_pytorch_amd import runtime as R
    w = torch.nn.Parameter(torch.ones(3))
    x = torch.zeros(2, 3)
    out = torch.empty(2, 3).fill_(1.0)              # what a HIP forward returns: a fresh buffer, no grad_fn
    y = R.forward_only(out, x, w)
    assert y.requires_grad and torch.equal(y, out)
    with pytest.raises(NotImplementedError, match="forward / inference"):
        y.sum().backward()
    with torch.no_grad():
        assert R.forward_only(out, x, w) is out     # nothing recorded when autograd is off
    with torch.inference_mode():
        assert R.forward_only(out, x, w) is out
    assert R.forward_only(out, x, w.detach()) is out


def test_packed_weight_cache_invalidation_hooks():
    from perceiverio_pytorch_amd import invalidate_packed_weights, runtime as R
    from perceiverio_pytorch_amd.perceiver import PerceiverDecoder
    from perceiverio_pytorch_amd.transformer_primitives import MLP
    m = MLP(8, widening_factor=1)
    m._pio_cache = ("stale", None, None)
    m.load_state_dict(m.state_dict())               # load_state_dict drops the packed images
    assert m._pio_cache is None
    d = PerceiverDecoder(8, 4, num_latent_channels=8)
    d._final_cache = ("stale", None)
    d.decoding_cross_attn.mlp._pio_cache = ("stale", None, None)
    invalidate_packed_weights(d)
    assert d._final_cache is None and d.decoding_cross_attn.mlp._pio_cache is None
    with torch.inference_mode():
        p = torch.nn.Parameter(torch.ones(2))
        assert R.param_key(p)[0][1] == 0            # no version counter under inference_mode: must not raise


def test_flow_mixed_precision_flag_is_honoured():
    from perceiverio_pytorch_amd import models as M
    kw = dict(img_size=(16, 16), num_latents=8, num_latent_channels=32, num_self_attends_per_block=1)
    assert M.FlowPerceiver(**kw).precision_policy == M.DEFAULT_POLICY["FlowPerceiver"]
    assert M.FlowPerceiver(mixed_precision=True, **kw).precision_policy == "fp16"   # autocast(fp16) analogue
    assert M.FlowPerceiver(mixed_precision=True, precision_policy="fp16x2w", **kw).precision_policy == "fp16x2w"


def test_asm_hazard_checker_flags_copies_behind_inline_asm(tmp_path):
    import os
    """tools/check_asm_hazards.py (run by the csrc Makefile over the inline-assembly attention kernels): a register copy
    between an inline-asm LDS read and its wait, or right behind an inline-asm MFMA, is a finding; the same code with
    the wait / enough distance is clean; findings do not leak across basic blocks."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    tool = os.path.join(root, "tools", "check_asm_hazards.py")

    def run(body):
        f = tmp_path / "k.s"
        f.write_text("_ZN3pio4testEv:\n" + body + "\ts_endpgm\n")
        r = subprocess.run([sys.executable, tool, str(f), "test"], capture_output=True, text=True)
        return r.returncode, r.stdout

    asm = lambda s: "\t;;#ASMSTART\n\t" + s + "\n\t;;#ASMEND\n"  # noqa: E731
    rc, out = run(asm("ds_read_b128 v[4:7], v1 offset:0") + "\tv_mov_b32_e32 v9, v5\n" + asm("s_waitcnt lgkmcnt(0)"))
    assert rc == 1 and "LDS read" in out
    rc, out = run(asm("ds_read_b128 v[4:7], v1 offset:0") + asm("s_waitcnt lgkmcnt(0)") + "\tv_mov_b32_e32 v9, v5\n")
    assert rc == 0, out
    rc, out = run(asm("ds_read_b128 v[4:7], v1 offset:0") + asm("ds_read_b128 v[8:11], v1 offset:64") +
                  asm("s_waitcnt lgkmcnt(1)") + "\tv_mov_b32_e32 v20, v5\n")
    assert rc == 0, out                       # in-order returns: one read may stay in flight
    rc, out = run(asm("ds_read_b128 v[4:7], v1 offset:0") + asm("ds_read_b128 v[8:11], v1 offset:64") +
                  asm("s_waitcnt lgkmcnt(1)") + "\tv_mov_b32_e32 v20, v9\n")
    assert rc == 1                            # ... but not the one that is read
    rc, out = run(asm("v_mfma_f32_32x32x16_f16 a[0:15], v[2:5], v[6:9], a[0:15]") + "\tv_accvgpr_read_b32 v30, a3\n")
    assert rc == 1 and "MFMA" in out
    rc, out = run(asm("v_mfma_f32_32x32x16_f16 a[0:15], v[2:5], v[6:9], a[0:15]") +
                  asm("s_nop 15\n\ts_nop 15\n\ts_nop 15") + "\tv_accvgpr_read_b32 v30, a3\n")
    assert rc == 0, out
    rc, out = run(asm("ds_read_b128 v[4:7], v1 offset:0") + "\ts_cbranch_scc1 .LBB0_2\n.LBB0_2:\n\tv_mov_b32_e32 v9, v5\n")
    assert rc == 0, out                       # per basic block only


# ---- BASELINE config 1: example_img_classify.py's model on a machine without a GPU -- the OPT-IN CPU plumbing backend
def test_config1_cpu_plumbing_backend_matches_reference_golden():
    """ClassificationPerceiver() with generated parameters on CPU tensors under set_backend("torch") against the
    reference's float32 logits (tests/golden/model_classify_conv.npz, B = 2): 1e-5.  The backend is explicit: the default
    one still refuses CPU tensors (test_no_cpu_fallback) and nothing from oracle/ is involved in the product path."""
    import numpy as np
    import perceiverio_pytorch_amd as P
    from cases import gen_state_dict, model_inputs
    from _golden import load
    from perceiverio_pytorch_amd import models as M
    g = load("model_classify_conv")
    spec = [(str(n), tuple(int(d) for d in str(s).split(",") if d != "")) for n, s in zip(g["spec_names"], g["spec_shapes"])]
    model = M.ClassificationPerceiver()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in gen_state_dict(spec, 31).items()}, strict=True)
    model.eval()
    x = torch.from_numpy(model_inputs("model_classify_conv")[0])
    assert P.get_backend() == "hip"
    with pytest.raises(P.PioError, match="set_backend"):
        with torch.inference_mode():
            model(x[:1])
    P.set_backend("torch")
    try:
        with torch.inference_mode():
            y = model(x).numpy()
    finally:
        P.set_backend("hip")
    ref = g["out"]
    assert y.shape == ref.shape == (2, 1000)
    d = y.astype(np.float64) - ref.astype(np.float64)
    rl2 = np.sqrt((d * d).sum()) / np.sqrt((ref.astype(np.float64) ** 2).sum())
    rmax = np.abs(d).max() / np.abs(ref).max()
    assert rl2 <= 1e-5 and rmax <= 1e-5, (rl2, rmax)


def test_cpu_plumbing_is_not_reachable_from_the_default_backend():
    """source-level guarantee: the product package never imports oracle/, and cpu_plumbing is only entered through
    runtime.cpu_plumbing(), which is False unless the backend was switched explicitly."""
    import os
    import re
    import perceiverio_pytorch_amd as P
    from perceiverio_pytorch_amd import runtime as R
    assert P.get_backend() == "hip" and R.cpu_plumbing(torch.zeros(1), "x") is False
    root = os.path.dirname(P.__file__)
    for fn in os.listdir(root):
        if fn.endswith(".py"):
            src = open(os.path.join(root, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+(oracle|perceiver_oracle|cases)\b", src, re.M), fn


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_feedback_images_are_roundings_whose_errors_cancel(dt):
    """runtime.feedback_images (policy "fp16sd"): image_b = round(w + E_{b-1}).  Every image is exactly representable in
    the operand dtype, lies within one ulp of w (the first IS round-to-nearest), and the error the first k images
    accumulate stays within half an ulp for every k -- against k/2 ulp for k copies of the round-to-nearest image."""
    import torch
    from perceiverio_pytorch_amd import runtime as R, _lib as L
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    torch.manual_seed(3)
    w = torch.randn(257, 130) * 0.05
    w[0, :4] = torch.tensor([0.0, 1.0, -2.5e-3, 6.0e4 if dt == "f16" else 3.0e38])
    imgs = R.feedback_images(w, 8, L.PIO_DT_F16 if dt == "f16" else L.PIO_DT_BF16)
    assert len(imgs) == 8 and torch.equal(imgs[0], w.to(tdt).float())
    # ulp of w in the operand dtype (normal range): 2^(floor(log2 |w|) - mantissa bits)
    mant = 10 if dt == "f16" else 7
    ulp = torch.where(w == 0, torch.zeros_like(w), torch.exp2(torch.floor(torch.log2(w.abs().clamp_min(1e-30))) - mant))
    if dt == "f16":
        ulp = ulp.clamp_min(2.0 ** -24)                      # (subnormal spacing of fp16)
    cum = torch.zeros_like(w, dtype=torch.float64)
    for img in imgs:
        assert torch.equal(img.to(tdt).float(), img), "an image must be exactly representable"
        assert bool(((w - img).abs() <= ulp * 1.0000001).all()), "an image must be a rounding of w within one ulp"
        cum += (w.double() - img.double())
        assert bool((cum.abs() <= 0.5000001 * ulp.double() * 2).all())      # (ulp doubles across a binade edge)
    plain = 8 * (w.double() - imgs[0].double()).abs()
    assert cum.abs().sum() < 0.3 * plain.sum(), "the accumulated error must be far below that of 8 equal roundings"


def test_per_call_options_replace_the_process_wide_switches():
    """compute calls are thread-safe per (stream, workspace).  The LayerNorm-fold switch and the CU budget
    travel with the call (pio_call_opts_t -> thread-local inside the library for the duration of the call); the nn.Module
    layer no longer flips pio_ln_fold_enable / pio_set_cu_budget around its calls.  CPU side: (a) the module sources hold
    no call of the process-wide setters, (b) two threads building their per-call option blocks and descriptor arrays
    concurrently get independent objects with the values they asked for."""
    import inspect
    import threading
    from perceiverio_pytorch_amd import _lib as L
    from perceiverio_pytorch_amd import perceiver, transformer_primitives
    for mod in (perceiver, transformer_primitives):
        src = inspect.getsource(mod)
        assert "pio_ln_fold_enable(" not in src and "pio_set_cu_budget(" not in src, mod.__name__
    assert "pio_encoder_fwd_opts" in L.SIGNATURES and "pio_self_attention_fwd_opts" in L.SIGNATURES
    assert L.SIGNATURES["pio_encoder_fwd_opts"][1][-1]._type_ is L.CallOpts

    from perceiverio_pytorch_amd.transformer_primitives import SelfAttention
    results, errors = {}, []
    barrier = threading.Barrier(2)

    def worker(i):
        try:
            torch.manual_seed(i)
            m = SelfAttention(64, widening_factor=1, num_heads=4)
            barrier.wait(timeout=60)
            for _ in range(50):
                opts = L.CallOpts(1 + i, 64 * (i + 1))
                layers = (L.SelfAttention * 2)()
                layers[0].fold.range_flag = 1000 + i
                assert (opts.ln_fold, opts.cu_budget) == (1 + i, 64 * (i + 1))
                assert layers[0].fold.range_flag == 1000 + i
            results[i] = (opts.ln_fold, opts.cu_budget, m.layer_norm1.weight.shape[0])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errors, errors
    assert results == {0: (1, 64, 64), 1: (2, 128, 64)}
