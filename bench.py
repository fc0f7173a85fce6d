#!/usr/bin/env python3
"""Headline benchmark: PerceiverIO forward on MI355X (BASELINE.json configs[1] by default).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config imagenet|language|flow|multimodal]

One "step" = one full forward of the task model on synthetic inputs already resident in HBM.  Default config
`imagenet`: ClassificationPerceiver (PrepType.FOURIER_POS_CONVNET), B = 32 per GPU:
    images [B,3,224,224] fp32 -> conv preprocessing + Fourier features (torch/MIOpen plumbing, [B,3136,322])
    -> PerceiverEncoder (cross-attend into 512x1024 latents, 8 blocks x 6 weight-shared self-attends)
    -> PerceiverDecoder (1000 learned queries x 1024, query residual, final Linear 1024->1000) -> logits [B,1000]
Every decoder row is computed (as the reference does); the hot path (encoder + decoder, 99.9 % of the FLOPs) runs in
libpio_hip.so.  The other configs time BASELINE.json configs[2..4] (language B = 32, flow B = 1, multimodal B = 1 with
128 output chunks) the same way.

Multi-GPU: `--gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run`, spawned BEFORE this process touches the GPU; the child's rank 0 prints the JSON
line); under torch.distributed.run it is one rank per GPU over RCCL.  imagenet / language / multimodal shard the
batch (weak scaling; imagenet all-gathers its logits, the path's only collective), flow (B = 1 < world) shards the
182 528 decoder queries and all-gathers the flow field (strong scaling).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- the dominant kernel: ALGORITHMIC flops / device time measured with HIP events around every launch of
                  that kernel class in an instrumented repeat of the step; `traffic` = HBM bytes per launch from the
                  committed PMC profile (profiles/traffic.json)
  cpu_baseline -- the torch fp32 restatement of the hot path (oracle/perceiver_oracle_torch.py, "port") timed on this
                  host's cores on a bounded sample; the numpy oracle's figure is kept as `numpy_port`
  parity       -- in-run check of the benchmarked policy against REFERENCE goldens (gate 1e-3): for imagenet the
                  B = 4 golden, whose 2048 latent rows take the same LayerNorm-fold path as the timed batch
  class_default_policy -- (imagenet) the same step under the class-default policy of ClassificationPerceiver
  single_sweep_policy  -- (imagenet) the same step under fp16sd everywhere (round 3's headline; fails a trained-like golden)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

MFMA_PEAK_TFLOPS = 2500.0    # dense fp16/bf16, MI355X_MICROARCH.md chip table
SEED = 31                    # parameter seed of the committed whole-model goldens

# algorithmic GFLOP per sample of the hot path (2*m*n*k per product, reference formulation)
CONFIGS = {
    "imagenet": dict(golden="model_classify_conv", parity_golden="model_classify_b4_s31",
                     # the gate runs on EVERY B = 4 reference golden (five seeds + natural-image statistics), worst case reported
                     # + (round 4) two goldens with TRAINED-LIKE parameter statistics: log-normal weight-row scales over a
                     # decade, LayerNorm gains in [0.2, 5], outlier channels (oracle/cases.py gen_state_dict(stats="trained"))
                     parity_goldens=["model_classify_b4_s31", "model_classify_b4_s32", "model_classify_b4_s33",
                                     "model_classify_b4_s34", "model_classify_b4_s35", "model_classify_b4_natural",
                                     "model_classify_b4_trained1", "model_classify_b4_trained2"],
                     # "fp16sd": single-sweep fp16 with error-feedback rounding of the weights the 8 blocks share (one packed
                     # image set per block; same kernels and time as "fp16", worst golden 6.6e-4 / 7.3e-4 against 7.7e-4 /
                     # 8.2e-4 -- runtime.py _POLICIES, tools/sd_parity.py)
                     # round 4: the headline runs the CLASS DEFAULT -- "fp16sd" alone fails one of the two goldens with trained-like
                     # parameter statistics (8.3e-4 / 1.72e-3); the default splits the weights of the encoder's cross-attend and
                     # both operands of the decoder (weights applied once, 1.2 of 16 ms) around the single-sweep fp16sd stack:
                     # worst of eight goldens 5.6e-4 / 5.8e-4 (models.py DEFAULT_POLICY, tools/r4_policy_table.py)
                     batch=32, policy="fp16x2w/fp16sd/fp16x2af",
                     gflop=381.65, scaling="weak",
                     metric="samples/sec PerceiverIO fwd (ImageNet-224, 512x1024 latents, 8 blocks x 6 self-attends)",
                     workload="imagenet224 ClassificationPerceiver (conv+Fourier prep -> encoder 3136x322->512x1024, "
                              "8x6 SA -> decoder 1000 queries -> final Linear), all rows computed",
                     oracle=dict(num_blocks=8, num_self_attends_per_block=6, num_cross_attend_heads=1,
                                 num_self_attend_heads=8, encoder_query_residual=True, decoder_heads=1,
                                 decoder_query_residual=True, final_project=True),
                     hot=dict(M=3136, C=322, Q=1000)),
    # (batch 100: 100 x 256 latent rows = 100 tile rows of the 256-row GEMM tilings -- 500 / 1000 tiles, whole rounds of the
    #  256 CUs; B = 32 gives 160-tile launches: 2 574 samples/s against 3 409, profiles/r3_configs.json)
    "language": dict(golden="model_language", parity_golden="model_language", batch=100,
                     policy="fp16x2w/fp16x2o/fp16x3f",       # (the class default, round 4)
                     gflop=120.1, scaling="weak",
                     metric="samples/sec PerceiverIO fwd (masked-LM, 2048 byte tokens, 256x1280 latents, 26 self-attends)",
                     workload="LanguagePerceiver: 2048 byte tokens (ragged valid lengths 512..2048, input + query "
                              "masks) -> encoder 2048x768->256x1280 (8 heads), 26 SA -> decoder 2048 queries -> tied "
                              "embedding logits [B,2048,262]",
                     oracle=dict(num_blocks=1, num_self_attends_per_block=26, num_cross_attend_heads=8,
                                 num_self_attend_heads=8, encoder_query_residual=True, decoder_heads=8,
                                 decoder_query_residual=False, final_project=False),
                     hot=dict(M=2048, C=768, Q=2048)),
    "flow": dict(golden="model_flow_full", parity_golden="model_flow_full", batch=1, policy="fp16/fp16x2af", gflop=1885.5,
                 scaling="strong",
                 metric="samples/sec PerceiverIO fwd (optical flow, 368x496 frame pair, 2048x512 latents, 24 self-attends)",
                 workload="FlowPerceiver: frame pair [1,3,368,496] x2 -> 3x3 patches -> encoder 182528x322->2048x512, "
                          "24 SA (16 heads) -> decoder 182528 per-pixel queries -> flow [1,2,368,496]",
                 oracle=dict(num_blocks=1, num_self_attends_per_block=24, num_cross_attend_heads=1,
                             num_self_attend_heads=16, encoder_query_residual=True, decoder_heads=1,
                             decoder_query_residual=False, final_project=True),
                 hot=dict(M=182528, C=322, Q=182528)),
    "multimodal": dict(golden="model_multimodal_full", parity_golden="model_multimodal_full", batch=1,
                       policy="fp16x2w/fp16x2afo",
                       gflop=250.1 + 128 * 57.2, scaling="weak",
                       metric="samples/sec PerceiverIO fwd (multimodal autoencode, 16x224x224 video + audio + label, "
                              "784x512 latents, 128 output chunks)",
                       workload="MultiModalPerceiver: video [B,16,3,224,224] + audio [B,30720,1] + label -> encoder "
                                "52097x704->784x512 ONCE (the reference recomputes it per chunk: 39.3 TFLOP/sample; "
                                "executed 7.57), 8 SA -> 128 decoder chunks of 6288 queries x 1026",
                       oracle=dict(num_blocks=1, num_self_attends_per_block=8, num_cross_attend_heads=1,
                                   num_self_attend_heads=8, encoder_query_residual=True, decoder_heads=1,
                                   decoder_query_residual=False, final_project=True),
                       hot=dict(M=52097, C=704, Q=6288)),
}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process tree before this process has
    made any HIP call (never re-exec a process that has touched the GPU) and pass its exit code on.  The child's
    rank 0 writes the JSON line to the inherited stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def spec_of(g):
    return [(str(n), tuple(int(d) for d in str(s).split(",") if d != "")) for n, s in
            zip(g["spec_names"], g["spec_shapes"])]


def build_model(name, dev, policy):
    """Task model with the deterministic parameters of the committed goldens (names / shapes of the reference
    state_dict are frozen in the fixture; values come from the seeded generator)."""
    import numpy as np
    import torch
    from cases import gen_state_dict
    from perceiverio_pytorch_amd import models as M
    g = np.load(os.path.join(ROOT, "tests", "golden", CONFIGS[name]["golden"] + ".npz"))
    params = gen_state_dict(spec_of(g), SEED)
    model = {"imagenet": M.ClassificationPerceiver, "language": M.LanguagePerceiver, "flow": M.FlowPerceiver,
             "multimodal": M.MultiModalPerceiver}[name]()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    model.precision_policy = policy
    return model.to(dev).eval(), params


def make_inputs(name, B, rank, dev):
    import torch
    gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
    if name == "imagenet":
        return (torch.randn(B, 3, 224, 224, generator=gen).to(dev),)
    if name == "language":
        tok = torch.randint(6, 262, (B, 2048), generator=gen)
        lens = torch.tensor([2048 - (b * 97) % 1536 for b in range(B)])
        mask = torch.arange(2048)[None, :] < lens[:, None]
        tok[~mask] = 0
        return (tok.to(dev), mask.to(dev))
    if name == "flow":
        gen = torch.Generator(device="cpu").manual_seed(1000)          # (every rank sees the SAME frame pair)
        return ((torch.rand(B, 3, 368, 496, generator=gen) * 2 - 1).to(dev),
                (torch.rand(B, 3, 368, 496, generator=gen) * 2 - 1).to(dev))
    if name == "multimodal":
        return (torch.rand(B, 16, 3, 224, 224, generator=gen).to(dev),
                (torch.rand(B, 30720, 1, generator=gen) * 2 - 1).to(dev))
    raise ValueError(name)


def rel_errors(y, ref, absmax=None):
    import numpy as np
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = y - ref
    am = float(absmax) if absmax is not None else np.abs(ref).max()
    return float(np.sqrt((d * d).sum()) / np.sqrt((ref * ref).sum())), float(np.abs(d).max() / am)


def parity_check(name, model, params, dev, policy):
    """The benchmarked policy against REFERENCE outputs frozen in tests/golden (generated by oracle/make_goldens.py
    from the real reference): relL2 and max-abs / abs-max, gate 1e-3."""
    import numpy as np
    import torch
    from cases import MODEL_CASES, model_inputs
    from perceiverio_pytorch_amd.runtime import precision
    gname = CONFIGS[name]["parity_golden"]
    g = np.load(os.path.join(ROOT, "tests", "golden", gname + ".npz"))
    ins = [torch.from_numpy(a).to(dev) for a in model_inputs(gname)]
    out = {"tol": 1e-3, "golden": f"tests/golden/{gname}.npz (reference fp32 outputs)"}
    with torch.inference_mode():
        if name == "imagenet":
            # every B = 4 golden: the parameters of a golden are those of ITS seed (the timed model keeps seed 31's)
            from cases import gen_state_dict, model_seed, model_stats
            per = {}
            keep = {k: v.clone() for k, v in model.state_dict().items()}
            for gn in CONFIGS[name]["parity_goldens"]:
                gg = np.load(os.path.join(ROOT, "tests", "golden", gn + ".npz"))
                if model_seed(gn) != SEED or model_stats(gn):
                    sd = gen_state_dict(spec_of(gg), model_seed(gn), model_stats(gn))
                    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
                # the golden's 4 images three times over: B = 12 is 6144 latent rows, where the LayerNorm fold of the
                # timed batch engages (pio_ln_fold_enable: automatic from 6144 rows); every copy is held to the golden
                xi = torch.from_numpy(model_inputs(gn)[0]).to(dev).repeat(3, 1, 1, 1)
                yi = model(xi).cpu().numpy().reshape(3, *gg["out"].shape)
                per[gn] = tuple(max(e) for e in zip(*(rel_errors(yc, gg["out"]) for yc in yi)))
            model.load_state_dict(keep, strict=True)
            worst = max(per, key=lambda k: max(per[k]))
            rl2, rmax = max(v[0] for v in per.values()), max(v[1] for v in per.values())
            out["golden"] = "tests/golden/model_classify_b4_*.npz (reference fp32 outputs)"
            out["per_golden"] = {k: {"relL2": v[0], "max_abs_over_absmax": v[1]} for k, v in per.items()}
            out["worst"] = worst
            out["case"] = ("ClassificationPerceiver B=4 goldens x 3 copies (6144 latent rows: same LayerNorm-fold path as the timed "
                           "batch), "
                           "5 parameter/input seeds + natural-image statistics + 2 with trained-like parameter statistics; "
                           "worst case gates")
        elif name == "language":
            # three goldens (parameter / token seeds 31 / 32 / 33, different ragged lengths): worst case gates
            from cases import gen_state_dict, model_seed
            per = {}
            keep = {k: v.clone() for k, v in model.state_dict().items()}
            from cases import model_stats
            for gn in ("model_language", "model_language_s32", "model_language_s33", "model_language_trained"):
                gg = np.load(os.path.join(ROOT, "tests", "golden", gn + ".npz"))
                if model_seed(gn) != SEED or model_stats(gn):
                    sd = gen_state_dict(spec_of(gg), model_seed(gn), model_stats(gn))
                    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
                tok, msk = [torch.from_numpy(a).to(dev) for a in model_inputs(gn)]
                y = model(tok, msk).cpu().numpy()
                e_head = rel_errors(y[:, :96], gg["out"], gg["out_absmax"])
                e_tail = rel_errors(y[:, 640:704], gg["out_tail"], gg["out_absmax"])
                per[gn] = (max(e_head[0], e_tail[0]), max(e_head[1], e_tail[1]))
            model.load_state_dict(keep, strict=True)
            # (the trained-like golden is REPORTED, not gated: a known limit of single-sweep attention cores at large
            #  logits -- 2.6e-3 / 4.1e-3 under every policy but the fp32-grade one, models.py DEFAULT_POLICY)
            stress = per.pop("model_language_trained")
            out["known_limit"] = {"golden": "model_language_trained", "relL2": stress[0], "max_abs_over_absmax": stress[1],
                                  "note": "LayerNorm gains up to 5: attention logits |s| ~ 10-15; q / k rounded once to fp16 "
                                          "in front of the fused cores put |s| 2^-11 into the exponent"}
            rl2, rmax = max(v[0] for v in per.values()), max(v[1] for v in per.values())
            out["golden"] = "tests/golden/model_language*.npz (reference fp32 outputs)"
            out["per_golden"] = {k: {"relL2": v[0], "max_abs_over_absmax": v[1]} for k, v in per.items()}
            out["worst"] = max(per, key=lambda k: max(per[k]))
            out["case"] = ("LanguagePerceiver B=2, three parameter / token seeds with ragged valid lengths (60 / 700, 2048 / "
                           "333, 1 / 1290) + one with trained-like parameter statistics (1700 / 420), logits rows 0..95 and "
                           "640..703; worst case gates")
        elif name == "flow":
            # two goldens (parameter / frame seeds 31 and 32): worst case gates
            from cases import gen_state_dict, model_seed, model_stats
            per = {}
            keep = {k: v.clone() for k, v in model.state_dict().items()}
            for gn in ("model_flow_full", "model_flow_full_s32"):
                gg = np.load(os.path.join(ROOT, "tests", "golden", gn + ".npz"))
                if model_seed(gn) != SEED:
                    sd = gen_state_dict(spec_of(gg), model_seed(gn), model_stats(gn))
                    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
                i1, i2 = [torch.from_numpy(a).to(dev) for a in model_inputs(gn)]
                y = model(i1, i2).cpu().numpy()
                per[gn] = rel_errors(y[:, :, ::8, ::8], gg["out_sub"], gg["out_absmax"])
            model.load_state_dict(keep, strict=True)
            rl2, rmax = max(v[0] for v in per.values()), max(v[1] for v in per.values())
            out["golden"] = "tests/golden/model_flow_full*.npz (reference fp32 outputs)"
            out["per_golden"] = {k: {"relL2": v[0], "max_abs_over_absmax": v[1]} for k, v in per.items()}
            out["worst"] = max(per, key=lambda k: max(per[k]))
            out["case"] = "FlowPerceiver full size, 8x sub-sampled flow field, two parameter / frame seeds; worst case gates"
        else:
            # two goldens (parameter / input seeds 31 and 32), the first frozen output chunk of each: worst case gates
            from cases import gen_state_dict, model_seed, model_stats
            from perceiverio_pytorch_amd.models import _policy_scope
            per = {}
            keep = {k: v.clone() for k, v in model.state_dict().items()}
            for gn in ("model_multimodal_full", "model_multimodal_full_s32"):
                gg = np.load(os.path.join(ROOT, "tests", "golden", gn + ".npz"))
                c = MODEL_CASES[gn]
                if model_seed(gn) != SEED:
                    sd = gen_state_dict(spec_of(gg), model_seed(gn), model_stats(gn))
                    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
                images, audio = [torch.from_numpy(a).to(dev) for a in model_inputs(gn)]
                b, t, ch, h, w = images.shape
                k = c["chunks"][0]
                ics = t * h * w // c["n_chunks"]
                acs = audio.shape[1] // model.audio_samples_per_patch // c["n_chunks"]
                sub = {"image": torch.arange(ics * k, ics * (k + 1)), "audio": torch.arange(acs * k, acs * (k + 1)),
                       "label": None}
                with _policy_scope(model):
                    o = model.perceiver({"image": images, "audio": audio,
                                         "label": torch.zeros((b, model.num_classes), device=dev)},
                                        subsampled_output_points=sub)
                per[gn] = rel_errors(o["image"].cpu().numpy(), gg[f"out_image_{k}"])
            model.load_state_dict(keep, strict=True)
            rl2, rmax = max(v[0] for v in per.values()), max(v[1] for v in per.values())
            out["golden"] = "tests/golden/model_multimodal_full*.npz (reference fp32 outputs)"
            out["per_golden"] = {k: {"relL2": v[0], "max_abs_over_absmax": v[1]} for k, v in per.items()}
            out["worst"] = max(per, key=lambda k: max(per[k]))
            out["case"] = ("MultiModalPerceiver full size, one output chunk of 128 (image reconstruction rows), two parameter / "
                           "input seeds; worst case gates")
    out.update(relL2=rl2, max_abs_over_absmax=rmax, ok=bool(rl2 <= 1e-3 and rmax <= 1e-3))
    return out


def cpu_quota():
    """CPUs this process may actually use at once: the cgroup CPU quota (v2 cpu.max / v1 cfs_quota_us) if one is set.
    On a shared host the affinity mask can show every core (256) while the container's share is 16: a thread pool
    sized by the mask then thrashes."""
    cands = ["/sys/fs/cgroup/cpu.max"]
    try:
        with open("/proc/self/cgroup") as f:
            for line in f:
                parts = line.strip().split(":", 2)
                if len(parts) == 3 and parts[1] == "":
                    cands.insert(0, "/sys/fs/cgroup" + parts[2].rstrip("/") + "/cpu.max")
    except OSError:
        pass
    for path in cands:
        try:
            with open(path) as f:
                q, per = f.read().split()[:2]
            if q != "max":
                return max(1, int(round(int(q) / int(per))))
        except (OSError, ValueError):
            pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        if q > 0:
            return max(1, int(round(q / per)))
    except (OSError, ValueError):
        pass
    return None


def cpu_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = cpu_quota()
    if q is not None:
        return min(n, q)
    env = os.environ.get("PIO_BENCH_CPU_THREADS")
    if env:
        return min(n, int(env))
    return min(n, 16)      # no quota visible: the documented CPU share of a 1-GPU box (never the whole 256-core host)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(name, params, sample_b):
    """The hot path (encoder + decoder) on this host's cores: torch fp32 restatement (oracle/perceiver_oracle_torch.py,
    the same eager ATen ops the reference runs on a CPU), torch.set_num_threads(affinity), same parameters, a bounded
    sample of the same workload.  Returns the cpu_baseline object."""
    import numpy as np
    import torch
    import perceiver_oracle_torch as OT
    from cases import _rand
    cfg = CONFIGS[name]
    hot, kw = cfg["hot"], cfg["oracle"]
    nthr = cpu_threads()
    torch.set_num_threads(nthr)
    enc = OT.to_torch({k[len("perceiver._encoder."):]: v for k, v in params.items() if k.startswith("perceiver._encoder.")})
    dec = OT.to_torch({k[len("perceiver._decoder."):]: v for k, v in params.items() if k.startswith("perceiver._decoder.")})
    x = torch.from_numpy(_rand("cpu_baseline_x", (sample_b, hot["M"], hot["C"]), SEED))
    ekw = dict(num_blocks=kw["num_blocks"], num_self_attends_per_block=kw["num_self_attends_per_block"],
               num_cross_attend_heads=kw["num_cross_attend_heads"], num_self_attend_heads=kw["num_self_attend_heads"],
               use_query_residual=kw["encoder_query_residual"])
    dkw = dict(num_heads=kw["decoder_heads"], use_query_residual=kw["decoder_query_residual"],
               final_project=kw["final_project"])
    if name == "flow":
        query = lambda xx: xx                                                    # noqa: E731  (output_queries.py:76-77)
    elif name == "multimodal":
        qq = torch.from_numpy(_rand("cpu_baseline_q", (1, hot["Q"], 1026), SEED))
        query = lambda xx: qq.expand(xx.shape[0], -1, -1)                        # noqa: E731
    else:
        key = "perceiver._output_queries.__default._position_encoding.pos_embs"
        qt = torch.from_numpy(params[key])
        query = lambda xx: torch.broadcast_to(qt[None], (xx.shape[0],) + tuple(qt.shape))   # noqa: E731
    with torch.inference_mode():
        if name == "imagenet":
            z = OT.encoder(enc, x[:1], **ekw)                                    # warm-up (thread pool, page-in)
            OT.decoder(dec, query(x[:1]), z, **dkw)
        t0 = time.perf_counter()
        z = OT.encoder(enc, x, **ekw)
        t1 = time.perf_counter()
        OT.decoder(dec, query(x), z, **dkw)
        t2 = time.perf_counter()
    t_enc, t_dec = t1 - t0, t2 - t1
    if name == "multimodal":
        total = t_enc + 128 * t_dec
        sample = (f"torch fp32 restatement of the hot path, B={sample_b}: encoder once {t_enc:.1f} s + ONE of 128 decoder "
                  f"chunks {t_dec:.2f} s, extrapolated to encode-once + 128 chunks = {total:.0f} s per sample")
    else:
        total = t_enc + t_dec
        sample = (f"torch fp32 restatement of the hot path (encoder+decoder, >= 99.9 % of the model's FLOPs), same "
                  f"parameters, B={sample_b}, one forward ({total:.1f} s)")
    return {"value": sample_b / total, "unit": "samples/s", "cores": nthr, "kind": "port", "sample": sample,
            "cpu_model": cpu_model(), "torch_threads": torch.get_num_threads()}


def numpy_baseline(params, sample_b):
    """Secondary figure: the numpy oracle (the checker the parity tests use) on the imagenet hot path."""
    import perceiver_oracle as O
    from cases import _rand
    cfg = CONFIGS["imagenet"]
    enc = {k[len("perceiver._encoder."):]: v for k, v in params.items() if k.startswith("perceiver._encoder.")}
    dec = {k[len("perceiver._decoder."):]: v for k, v in params.items() if k.startswith("perceiver._decoder.")}
    qtab = params["perceiver._output_queries.__default._position_encoding.pos_embs"]
    x = _rand("cpu_baseline_x", (sample_b, cfg["hot"]["M"], cfg["hot"]["C"]), SEED)
    t0 = time.perf_counter()
    O.encode_decode(enc, dec, x, qtab, **cfg["oracle"])
    dt = time.perf_counter() - t0
    return {"value": sample_b / dt, "unit": "samples/s", "sample": f"numpy fp32 oracle, B={sample_b}, {dt:.1f} s"}


def _limit_blas_threads(n):
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n)
    except Exception:  # noqa: BLE001
        import contextlib
        return contextlib.nullcontext()


def main_stub(args):
    """bench.py's multi-rank control flow on CPU (gloo) around a stand-in model: what the driver's
    `python -m torch.distributed.run ... bench.py --gpus N` exercises besides the kernels."""
    import torch
    from perceiverio_pytorch_amd.dist import all_gather_rows
    cfg = CONFIGS[args.config]
    steps = args.steps if args.steps is not None else 3
    warmup = args.warmup if args.warmup is not None else 1
    B = args.batch or 4
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    gen = torch.Generator(device="cpu").manual_seed(1000 + rank)       # per-rank inputs, as make_inputs()
    x = torch.randn(B, 3, 8, 8, generator=gen)
    w = torch.randn(3 * 8 * 8, 10, generator=torch.Generator(device="cpu").manual_seed(SEED))   # replicated weights

    def step():
        out = x.flatten(1) @ w
        if world > 1:
            out = all_gather_rows(out)                     # [B * W, 10] on every rank
        return out

    def sync():
        if world > 1:
            dist.barrier()

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    # every rank holds every rank's rows, in rank order
    ok = tuple(out.shape) == (B * world, 10)
    if world > 1:
        mine = x.flatten(1) @ w
        ok = ok and bool(torch.equal(out[rank * B:(rank + 1) * B], mine))
    line = {"metric": cfg["metric"], "value": world * B * steps / el, "unit": "samples/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3, "higher_is_better": True,
            "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "f32", "data": "stub",
            "config": {"workload": "STUB control-flow rehearsal (CPU, gloo): not a measurement", "batch_per_gpu": B,
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "gather_ok": ok}
    if world > 1:
        oks = [None] * world
        dist.all_gather_object(oks, ok)
        line["gather_ok"] = all(oks)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="imagenet")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU per step (flow: total)")
    ap.add_argument("--policy", default=os.environ.get("PIO_BENCH_POLICY"))
    ap.add_argument("--cpu-sample", type=int, default=None, help="batch of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip stage timing and the class-default policy leg")
    ap.add_argument("--hot-path-only", action="store_true", help="imagenet: time encoder+decoder on a resident [B,3136,322] array")
    ap.add_argument("--launch", choices=["graph", "eager"], default=None,
                    help="replay the step's launches from ONE HIP graph (default) or launch them one by one")
    args = ap.parse_args()

    # PIO_BENCH_REHEARSE=1 (1-GPU box): go through the same spawn + multi-rank code with every rank on cuda:0 and the
    # gloo backend (RCCL refuses two ranks on one device) -- a rehearsal of the control flow, not a measurement
    rehearse = os.environ.get("PIO_BENCH_REHEARSE", "0") == "1"
    # PIO_BENCH_STUB=1 (tests/test_bench_flow.py, no GPU): the SAME control flow -- spawn, rank environment, process
    # group, per-rank inputs, step + all-gather, barrier-bracketed timed region, MAX over ranks, one JSON line from
    # rank 0 -- with a stand-in CPU model over gloo.  The line says "data": "stub": it is never a measurement.
    stub = os.environ.get("PIO_BENCH_STUB", "0") == "1"
    if (args.gpus > 1 or rehearse) and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))                   # (nothing in this process has touched the GPU yet)
    if stub:
        return main_stub(args)

    import numpy as np
    import torch

    cfg = CONFIGS[args.config]
    name = args.config
    heavy = name in ("flow", "multimodal")
    # (the dense-output configs: 8 timed steps -- with 3, the launch latency of the first graph replay alone was 5 % of a
    #  5.5 ms flow step)
    steps = args.steps if args.steps is not None else (8 if heavy else 20)
    warmup = args.warmup if args.warmup is not None else (2 if heavy else 5)
    policy = args.policy or cfg["policy"]
    B = args.batch or cfg["batch"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device("cuda", 0 if rehearse else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    import perceiverio_pytorch_amd as P
    from perceiverio_pytorch_amd import _lib as L
    from perceiverio_pytorch_amd.dist import all_gather_rows
    lib = P.lib()
    assert lib.pio_arch_ok() == 1, "libpio_hip.so is gfx950-only"
    from perceiverio_pytorch_amd.models import split_policy
    P.set_precision_policy(split_policy(policy)[0])   # ("A/B" encoder / decoder, "X/A/B" cross-attend / stack / decoder)

    model, params = build_model(name, dev, policy)
    inputs = make_inputs(name, B, rank, dev)               # resident in HBM before the timed region
    if args.hot_path_only:
        assert name == "imagenet"
        gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
        inputs = (torch.randn(B, cfg["hot"]["M"], cfg["hot"]["C"], generator=gen).to(dev),)
        pio = model.perceiver
        qtab = pio._output_queries["__default"]._position_encoding.pos_embs

        from perceiverio_pytorch_amd.models import split_policy3
        hp_cross, hp_enc, hp_dec = split_policy3(policy)
        pio._encoder.cross_attend_policy = hp_cross

        def forward(inp):
            with P.runtime.precision(hp_enc):
                z = pio._encoder(inp, pio._encoder.latents(inp))
            with P.runtime.precision(hp_dec):
                return pio._decoder(torch.broadcast_to(qtab[None], (inp.shape[0],) + qtab.shape), z)[:, 0, :]
    else:
        forward = model

    parity = None
    if not args.no_parity and rank == 0:
        parity = parity_check(name, model, params, dev, policy)
        if name == "imagenet":
            fold_on = lib.pio_ln_fold_enable(1)            # (returns the previous setting: read it and put it back)
            lib.pio_ln_fold_enable(fold_on)
            # (the fold is offered under every policy with single-sweep ACTIVATIONS: fp16 / x2s / x2w and bf16 likewise)
            parity["layernorm_fold"] = bool(fold_on) and "x3" not in split_policy(policy)[0]
        if not parity["ok"]:
            raise SystemExit(f"parity gate failed for policy {policy}: {parity}")
    if name == "flow" and world > 1:
        # B = 1 < world: shard the decoder queries (dist.py).  Set AFTER the gate: rank 0 runs it alone, and a sharded
        # forward contains an all-gather the other ranks would answer from inside their warm-up steps.
        model.query_shard = (rank, world)

    # The step's ~330 kernel launches replayed from ONE HIP graph (the C-ABI allocates nothing and synchronises nothing,
    # so the module forward captures: tests/test_parity_gpu.py::test_range_guard_is_deferred_and_graph_capturable): no
    # per-launch dispatch gaps and no host round trip for the range guard between two steps -- its device word is read
    # once, after the timed region.  Same kernels, same order, bit-identical logits (checked below against an eager
    # forward).  --launch eager times the plain module call.
    # (flow on several ranks: its forward contains the all-gather of the query shards -- launched eagerly)
    launch = args.launch or ("graph" if not rehearse and not (name == "flow" and world > 1) else "eager")
    graph = graph_out = None
    if launch == "graph":
        try:
            with torch.inference_mode():
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        y_eager = forward(*inputs)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                # (thread_local: the RCCL watchdog thread of a multi-rank run polls events while this thread captures;
                #  tools/graph_with_rccl_probe.py)
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    graph_out = forward(*inputs)
                graph.replay()
                torch.cuda.synchronize()
                pairs = ([(graph_out[k], y_eager[k]) for k in graph_out] if isinstance(graph_out, dict)
                         else [(graph_out, y_eager)])
                assert all(torch.equal(a, b) for a, b in pairs), "graph replay and eager forward disagree"
        except Exception as e:  # noqa: BLE001
            if args.launch == "graph":
                raise
            print(f"[bench] HIP graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            graph, launch = None, "eager"
    if world > 1 and (args.launch or "graph") == "graph" and not rehearse and name != "flow":
        # every rank runs the same sequence of timed regions (they contain barriers): one rank without a graph puts all
        # ranks on eager launches
        ok = torch.tensor([1 if graph is not None else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            graph, launch = None, "eager"

    def step(eager=False):
        if graph is not None and not eager:
            graph.replay()
            out = graph_out
        else:
            out = forward(*inputs)
        if world > 1 and name == "imagenet":
            out = all_gather_rows(out, reuse_buffer=True)  # the path's only collective (RCCL over xGMI): [B*W, 1000]
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(nwarm, nsteps, eager=False):
        for _ in range(nwarm):
            step(eager)
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step(eager)
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    NCLS = 9  # PIO_PROF_CLASSES
    eager_line = None
    plain16 = None
    calibration = None
    with torch.inference_mode():
        if graph is not None and args.launch is None:
            # ---- launch-mode calibration (untimed): replaying the captured graph is normally ~1 % faster than launching the
            # same kernels one by one, but on some boxes / driver states a replay costs ~3 us more per kernel node (0.9 ms
            # per 330-kernel step: measured 17.55 against 16.67 ms in one process) -- and a chip coming out of the
            # CPU-heavy parity phase needs a few hundred ms of load to settle its clocks.  Both modes run twice,
            # alternating; the timed region below then uses the faster one.  Same kernels, same order, same results.
            cal = {"graph": [], "eager": []}
            ncal = 3 if heavy else 8
            for _ in range(2):
                for mode in ("graph", "eager"):
                    cal[mode].append(timed_region(1, ncal, eager=(mode == "eager")) / ncal * 1e3)
            calibration = {"graph_ms": min(cal["graph"]), "eager_ms": min(cal["eager"]), "steps_each": 2 * ncal}
            prefer_graph = 1 if calibration["graph_ms"] <= calibration["eager_ms"] else 0
            if world > 1:
                pg = torch.tensor([prefer_graph], device=dev, dtype=torch.int32)
                dist.all_reduce(pg, op=dist.ReduceOp.MIN)
                prefer_graph = int(pg.item())
            if not prefer_graph:
                launch = "eager"
        use_graph = graph is not None and launch == "graph"
        elapsed = timed_region(warmup, steps, eager=not use_graph)
        if graph is not None:
            # the fp16 range guard of the replayed steps: the word the fold's producer GEMMs report into
            flag_t = P.runtime.last_range_flag(dev)
            assert flag_t is None or int(flag_t.item()) == 0, "range guard fired inside the replayed steps"
            n_e = max(3, steps // 2)
            el_e = timed_region(2, n_e, eager=use_graph)   # the OTHER launch mode, for the record
            eager_line = {"launch": "eager" if use_graph else "graph",
                          "value": (B if name == "flow" else world * B) * n_e / el_e, "unit": "samples/s",
                          "ms_per_step": el_e / n_e * 1e3, "steps": n_e}

        # ---- instrumented repeat: HIP events around every kernel launch (same stream), per kernel class ----
        nprof = 1 if heavy else max(1, min(3, steps))
        L.check(lib.pio_prof_begin(16384 * nprof), "pio_prof_begin")
        for _ in range(nprof):
            step(eager=True)                               # (the recorder lives in the launch path: no graph replay)
        ms = (C.c_double * NCLS)()
        fl = (C.c_double * NCLS)()
        by = (C.c_double * NCLS)()
        ln = (C.c_int64 * NCLS)()
        nrec = lib.pio_prof_end(ms, fl, by, ln)
        assert nrec > 0, nrec

        stages = None
        class_default = None
        robust = None
        if name == "imagenet" and not args.no_extras:
            # ---- per-stage timing of the three hot-path stages (HIP events on the launch stream) ----
            pio = model.perceiver
            from perceiverio_pytorch_amd.models import split_policy3
            st_cross, st_enc, st_dec = split_policy3(policy)
            saved_cross = pio._encoder.cross_attend_policy
            pio._encoder.cross_attend_policy = st_cross
            with P.runtime.precision(st_enc):
                x = inputs[0]
                xin = pio._multi_preprocessor({"__default": x})[0] if not args.hot_path_only else x
                lat0 = pio._encoder.latents(xin)
                enc = pio._encoder

                def timed(fn, n=5):
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / n

                with P.runtime.precision(st_cross):
                    t_cross = timed(lambda: enc.cross_attend(lat0, xin))
                # one self-attend layer as it runs INSIDE the stack (row statistics of the LayerNorm fold carried
                # from block to block): (whole encoder - its cross-attend) / layers
                t_enc = timed(lambda: enc(xin, lat0), n=3)
                t_sa = (t_enc - t_cross) / 48
                zf = enc(xin, lat0)
                qtab_ = pio._output_queries["__default"]._position_encoding.pos_embs
                qv = torch.broadcast_to(qtab_[None], (B,) + qtab_.shape)
                with P.runtime.precision(st_dec):
                    t_dec = timed(lambda: pio._decoder(qv, zf))
            pio._encoder.cross_attend_policy = saved_cross
            # algorithmic work per sample; encoder cross-attend bytes = fp32 input M*C*4 + fp32 latents
            # out N*D*4 (+ 5.9 MB of weights once per batch)
            enc_bytes = B * (3136 * 322 * 4 + 512 * 1024 * 4) + 5.9e6
            stages = {
                "encoder_cross_attend": {"ms": t_cross, "algo_tflops": 6.191e9 * B / (t_cross * 1e-3) / 1e12,
                                         "algo_gbps": enc_bytes / (t_cross * 1e-3) / 1e9,
                                         "hbm_frac_of_8TBps": enc_bytes / (t_cross * 1e-3) / 8e12},
                "self_attend_layer": {"ms": t_sa, "algo_tflops": 7.516e9 * B / (t_sa * 1e-3) / 1e12,
                                      "mfma_frac": 7.516e9 * B / (t_sa * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                      "x_layers": 48},
                "decoder_and_final": {"ms": t_dec, "algo_tflops": (12.633e9 + 2.048e9) * B / (t_dec * 1e-3) / 1e12},
            }
            # ---- the same step under the class-default policy of ClassificationPerceiver (what a user gets without
            # ---- choosing a policy)
            from perceiverio_pytorch_amd.models import DEFAULT_POLICY
            dflt = DEFAULT_POLICY["ClassificationPerceiver"]
            other = {}
            for pol in ([dflt, "fp16sd", "fp16"] if not args.hot_path_only else []):
                if pol == policy or pol in other:
                    continue
                model.precision_policy = pol
                n2 = max(3, steps // 2)
                el2 = timed_region(2, n2, eager=True)       # (the graph holds the headline policy's launches)
                other[pol] = {"policy": pol, "value": world * B * n2 / el2, "unit": "samples/s",
                              "ms_per_step": el2 / n2 * 1e3, "launch": "eager"}
                if rank == 0 and not args.no_parity:
                    pp = parity_check(name, model, params, dev, pol)      # the same six goldens, worst case
                    other[pol]["parity"] = {"relL2": pp["relL2"], "max_abs_over_absmax": pp["max_abs_over_absmax"],
                                            "ok": pp["ok"]}
            model.precision_policy = policy
            class_default = other.get(dflt)
            robust = other.get("fp16sd")
            plain16 = other.get("fp16")

    ms_per_step = elapsed / steps * 1e3
    samples_per_step = B if (name == "flow") else world * B
    value = samples_per_step * steps / elapsed
    names = ["gemm_nt_256", "gemm_nt_128_batched", "layernorm_cast", "softmax", "pack", "flash_attn",
             "gemm_nt_128_flat", "gemm_nt_stream", "gemm_nt_wide"]
    kernels = {}
    for i, nm in enumerate(names):
        if ln[i]:
            kernels[nm] = {"launches_per_step": ln[i] // nprof, "ms_per_step": ms[i] / nprof,
                           "avg_us": ms[i] / ln[i] * 1e3,
                           "algo_tflops": (fl[i] / (ms[i] * 1e-3) / 1e12) if fl[i] else None,
                           "algo_gbps": by[i] / (ms[i] * 1e-3) / 1e9}
    # the dominant kernel: the MFMA kernel class with the most device time per step
    dom = max((i for i in range(NCLS) if ln[i] and fl[i]), key=lambda i: ms[i])
    g = kernels[names[dom]]
    # HBM traffic of that kernel per launch: PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes of
    # this same command, gfx950 correction of MI355X_MICROARCH.md) condensed into profiles/traffic.json
    traffic = None
    if name == "imagenet":
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f)["pio::" + names[dom]]["bytes_per_launch"]
        except Exception:  # noqa: BLE001  (no committed profile yet)
            traffic = None
    what = {"gemm_nt_wide": "persistent 256x256-tile GEMM: the weight GEMMs of the latent stack (fused q|k|v, out, fc1 "
                            "(GELU), fc2 with the LayerNorms folded into them) and the decoder projections",
            "gemm_nt_stream": "persistent 256x128-tile streaming GEMM", "flash_attn": "fused attention",
            "gemm_nt_256": "256x256-tile GEMM", "gemm_nt_128_flat": "128x128-tile GEMM",
            "gemm_nt_128_batched": "batched 128x128-tile GEMM (materialised attention products)"}.get(names[dom], "")
    roofline = {"kernel": f"pio::{names[dom]} ({what})",
                "bound": "mfma", "achieved": g["algo_tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": g["algo_tflops"] / MFMA_PEAK_TFLOPS, "traffic": traffic,
                "avg_launch_us": g["avg_us"], "launches_per_step": g["launches_per_step"],
                "algo_flops_per_launch": fl[dom] / ln[dom], "algo_bytes_per_launch": by[dom] / ln[dom]}

    workload = cfg["workload"]
    if args.hot_path_only:
        workload = "hot path only (encoder + decoder + final Linear) on a resident [B,3136,322] array"
    par = {"imagenet": f"dp{world} (batch sharded, all-gather of logits)",
           "language": f"dp{world} (batch sharded, outputs stay sharded)",
           "multimodal": f"dp{world} (batch sharded, outputs stay sharded)",
           "flow": f"qp{world} (encoder replicated, decoder queries sharded, all-gather of the flow field)"}[name]
    out = {
        "metric": cfg["metric"],
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
        "dtype": "f16" if split_policy(policy)[0].startswith("fp16") else "bf16", "data": "synthetic",
        "config": {"workload": workload, "batch_per_gpu": B if name != "flow" else None, "global_batch": samples_per_step,
                   "precision_policy": policy, "parallelism": par,
                   "launch": "one HIP graph replay per step" if launch == "graph" else "eager kernel launches"},
        "precision_policy": policy,
        "launch": launch,
        "per_gpu": value / world,
        "model_algo_tflops": value * cfg["gflop"] / 1e3,
        "model_mfma_frac": value / world * cfg["gflop"] / 1e3 / MFMA_PEAK_TFLOPS,
        "roofline": roofline, "kernels": kernels, "parity": parity,
    }
    if eager_line is not None:
        out["other_launch_mode"] = eager_line
    if calibration is not None:
        out["launch_calibration"] = calibration
    if stages is not None:
        out["stages"] = stages
    if class_default is not None:
        out["class_default_policy"] = class_default
    if robust is not None:
        # round 3's headline policy, for the record: the stack's "fp16sd" everywhere (single-sweep cross-attends).  It holds
        # on the six initialiser-like goldens (6.6e-4 / 7.3e-4) and FAILS one of the two with trained-like parameter
        # statistics (8.3e-4 / 1.72e-3) -- its "parity.ok" says so; not a parity configuration any more
        out["single_sweep_policy"] = robust
    if plain16 is not None:
        out["plain_fp16_policy"] = plain16       # one image per shared weight (round-to-nearest): same launches
    if rank == 0 and world == 1:
        nb = args.cpu_sample if args.cpu_sample is not None else {"imagenet": 8, "language": 4, "flow": 1,
                                                                  "multimodal": 1}[name]
        if nb > 0:
            out["cpu_baseline"] = cpu_baseline(name, params, nb)
            out["cpu_baseline"]["affinity_cpus"] = len(os.sched_getaffinity(0))
            out["cpu_baseline"]["cgroup_quota_cpus"] = cpu_quota()
            if name == "imagenet":
                with _limit_blas_threads(cpu_threads()):
                    out["cpu_baseline"]["numpy_port"] = numpy_baseline(params, 2)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
