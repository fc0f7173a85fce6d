#!/usr/bin/env python3
"""Headline benchmark: PerceiverIO forward, ImageNet-224 classifier (BASELINE.json configs[1]).

One "step" = one full forward of ClassificationPerceiver (PrepType.FOURIER_POS_CONVNET) on a batch of synthetic
images that is already resident in HBM:
    images [B,3,224,224] fp32 -> conv preprocessing + Fourier features (torch/MIOpen plumbing, [B,3136,322])
    -> PerceiverEncoder (cross-attend into 512x1024 latents, 8 blocks x 6 weight-shared self-attends)
    -> PerceiverDecoder (1000 learned queries x 1024, query residual, final Linear 1024->1000)
    -> logits of query row 0 [B,1000]
with B = 32 per GPU.  Every decoder row is computed (as the reference does); the hot path (encoder + decoder,
99.9 % of the FLOPs) runs in libpio_hip.so.  `--hot-path-only` times just that on a resident [B,3136,322] array.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, RCCL)

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- the dominant kernel (gemm_nt_wide / gemm_nt_stream, the weight GEMMs of the latent stack): ALGORITHMIC flops / device
                  time measured with HIP events around every launch of that kernel in an instrumented repeat of the
                  step; `traffic` = HBM bytes per launch from the committed PMC profile (profiles/traffic.json)
  cpu_baseline -- the numpy oracle ("port") of the hot path timed on this host's cores on a bounded sample
  parity       -- in-run check of the same model at B=2 against the committed REFERENCE golden (gate 1e-3)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CFG = dict(M=3136, C=322, N=512, D=1024, L=6, blocks=8, xh=1, sh=8, Q=1000, Dq=1024, out=1000)
# algorithmic GFLOP per sample of the hot path (2*m*n*k per product, reference formulation);
# the conv preprocessing adds 0.24 GFLOP (0.06 %) and is not counted.
GFLOP_PER_SAMPLE = 381.65
MFMA_PEAK_TFLOPS = 2500.0    # dense fp16/bf16, MI355X_MICROARCH.md chip table
GOLDEN = "model_classify_conv"
SEED = 31                    # parameter seed of the committed whole-model goldens


def build_model(dev, policy):
    """ClassificationPerceiver with the deterministic parameters of the committed golden (names/shapes of the
    reference state_dict are frozen in the fixture; values come from the seeded generator)."""
    import perceiverio_pytorch_amd as P
    from cases import gen_state_dict
    from perceiverio_pytorch_amd.models import ClassificationPerceiver
    P.set_precision_policy(policy)
    g = np.load(os.path.join(ROOT, "tests", "golden", GOLDEN + ".npz"))
    spec = [(str(n), tuple(int(d) for d in str(s).split(",") if d != "")) for n, s in
            zip(g["spec_names"], g["spec_shapes"])]
    params = gen_state_dict(spec, SEED)
    model = ClassificationPerceiver()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    model.precision_policy = policy         # the benchmarked policy (the class default is fp16x2w)
    return model.to(dev).eval(), params, g


def parity_check(model, g, dev):
    """B=2 run on the golden's seeded images vs the reference's float32 logits frozen in tests/golden."""
    from cases import model_inputs
    x = torch.from_numpy(model_inputs(GOLDEN)[0]).to(dev)
    with torch.inference_mode():
        y = model(x).cpu().numpy().astype(np.float64)
    ref = g["out"].astype(np.float64)
    d = y - ref
    return float(np.sqrt((d * d).sum()) / np.sqrt((ref * ref).sum())), float(np.abs(d).max() / np.abs(ref).max())


def cpu_threads():
    """Threads the numpy oracle can actually use: the BLAS pool size (threadpoolctl) capped by this process's CPU
    affinity -- not os.cpu_count(), which reports the whole host even inside a 16-CPU share."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info
        pools = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        if pools:
            return min(avail, max(pools))
    except Exception:  # noqa: BLE001
        pass
    return avail


def cpu_baseline(params, sample_b):
    """numpy oracle of the hot path (encoder + decoder) on a [B,3136,322] sample, same parameters."""
    import perceiver_oracle as O
    from cases import _rand
    enc = {k[len("perceiver._encoder."):]: v for k, v in params.items() if k.startswith("perceiver._encoder.")}
    dec = {k[len("perceiver._decoder."):]: v for k, v in params.items() if k.startswith("perceiver._decoder.")}
    qtab = params["perceiver._output_queries.__default._position_encoding.pos_embs"]
    kw = dict(num_blocks=CFG["blocks"], num_self_attends_per_block=CFG["L"], num_cross_attend_heads=1,
              num_self_attend_heads=CFG["sh"], encoder_query_residual=True, decoder_heads=1,
              decoder_query_residual=True, final_project=True)
    x = _rand("cpu_baseline_x", (sample_b, CFG["M"], CFG["C"]), SEED)
    O.encode_decode(enc, dec, x[:1], qtab, **kw)            # warm-up (BLAS threads, page-in)
    t0 = time.perf_counter()
    O.encode_decode(enc, dec, x, qtab, **kw)
    dt = time.perf_counter() - t0
    return sample_b / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU per step")
    ap.add_argument("--policy", default=os.environ.get("PIO_BENCH_POLICY", "fp16"))
    ap.add_argument("--cpu-sample", type=int, default=4, help="batch of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--hot-path-only", action="store_true", help="time encoder+decoder on a resident [B,3136,322] array")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    import perceiverio_pytorch_amd as P
    from perceiverio_pytorch_amd import _lib as L
    lib = P.lib()
    assert lib.pio_arch_ok() == 1, "libpio_hip.so is gfx950-only"

    model, params, golden = build_model(dev, args.policy)
    B = args.batch
    gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
    if args.hot_path_only:
        x = torch.randn(B, CFG["M"], CFG["C"], generator=gen).to(dev)
        pio = model.perceiver
        qtab = pio._output_queries["__default"]._position_encoding.pos_embs

        def forward(inp):
            z = pio._encoder(inp, pio._encoder.latents(inp))
            return pio._decoder(torch.broadcast_to(qtab[None], (inp.shape[0],) + qtab.shape), z)[:, 0, :]
    else:
        x = torch.randn(B, 3, 224, 224, generator=gen).to(dev)     # resident in HBM before the timed region
        forward = model

    parity = None
    if not args.no_parity and rank == 0:
        rl2, rmax = parity_check(model, golden, dev)
        parity = {"relL2": rl2, "max_abs_over_absmax": rmax, "tol": 1e-3,
                  "case": "ClassificationPerceiver B=2 vs reference fp32 logits (tests/golden/model_classify_conv.npz)",
                  "ok": bool(rl2 <= 1e-3 and rmax <= 1e-3)}
        if not parity["ok"]:
            raise SystemExit(f"parity gate failed for policy {args.policy}: {parity}")
        # The golden case has 1024 latent rows, below the 2048 from which the SelfAttention blocks fold their
        # LayerNorms into the GEMMs (pio_ln_fold_t): check THIS run's batch against the same model under the
        # float32-grade policy fp16x3 (three MFMA sweeps, no fold; 2e-6 against the reference on the golden case)
        if not args.hot_path_only and args.policy != "fp16x3":
            with torch.inference_mode():
                y_run = model(x).double()
                model.precision_policy = "fp16x3"
                y_ref = model(x).double()
                model.precision_policy = args.policy
            frel = float(((y_run - y_ref).norm() / y_ref.norm()).item())
            fold_on = lib.pio_ln_fold_enable(1)     # (returns the previous setting: read it and put it back)
            lib.pio_ln_fold_enable(fold_on)
            parity["bench_batch_vs_fp16x3"] = {"relL2": frel, "tol": 1e-3, "batch": B, "layernorm_fold": bool(fold_on)}
            if frel > 1e-3:
                raise SystemExit(f"parity gate failed on the benchmarked batch: {parity}")

    gathered = [torch.empty(B, CFG["out"], device=dev) for _ in range(world)] if world > 1 else None

    def step():
        logits = forward(x)                                   # [B,1000]: query row 0 (ClassificationPostprocessor)
        if world > 1:
            dist.all_gather(gathered, logits.contiguous())    # the path's only collective (RCCL over xGMI);
            # same call as perceiverio_pytorch_amd.dist.all_gather_rows, with the receive list pre-allocated
        return logits

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.inference_mode():
        for _ in range(args.warmup):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # ---- instrumented repeat: HIP events around every kernel launch (same stream), per kernel class ----
        nprof = max(1, min(3, args.steps))
        L.check(lib.pio_prof_begin(4096 * nprof), "pio_prof_begin")
        for _ in range(nprof):
            step()
        NCLS = 9  # PIO_PROF_CLASSES
        ms = (C.c_double * NCLS)()
        fl = (C.c_double * NCLS)()
        by = (C.c_double * NCLS)()
        ln = (C.c_int64 * NCLS)()
        nrec = lib.pio_prof_end(ms, fl, by, ln)
        assert nrec > 0, nrec

        # ---- per-stage timing of the three hot-path stages (HIP events on the launch stream, 5 repeats each) ----
        pio = model.perceiver
        with P.runtime.precision(args.policy):
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

            t_cross = timed(lambda: enc.cross_attend(lat0, xin))
            # one self-attend layer as it runs INSIDE the stack (row statistics of the LayerNorm fold carried from
            # block to block): (whole encoder - its cross-attend) / layers
            t_enc = timed(lambda: enc(xin, lat0), n=3)
            t_sa = (t_enc - t_cross) / (CFG["L"] * CFG["blocks"])
            zf = enc(xin, lat0)
            qtab_ = pio._output_queries["__default"]._position_encoding.pos_embs
            qv = torch.broadcast_to(qtab_[None], (B,) + qtab_.shape)
            t_dec = timed(lambda: pio._decoder(qv, zf))

    # algorithmic work per sample: FLOPs = 2mnk per product; encoder cross-attend bytes =
    # fp32 input M*C*4 + fp32 latents out N*D*4 (+ 5.9 MB of weights once per batch)
    enc_bytes = B * (CFG["M"] * CFG["C"] * 4 + CFG["N"] * CFG["D"] * 4) + 5.9e6
    stages = {
        "encoder_cross_attend": {"ms": t_cross, "algo_tflops": 6.191e9 * B / (t_cross * 1e-3) / 1e12,
                                 "algo_gbps": enc_bytes / (t_cross * 1e-3) / 1e9,
                                 "hbm_frac_of_8TBps": enc_bytes / (t_cross * 1e-3) / 8e12},
        "self_attend_layer": {"ms": t_sa, "algo_tflops": 7.516e9 * B / (t_sa * 1e-3) / 1e12,
                              "mfma_frac": 7.516e9 * B / (t_sa * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                              "x_layers": CFG["L"] * CFG["blocks"]},
        "decoder_and_final": {"ms": t_dec, "algo_tflops": (12.633e9 + 2.048e9) * B / (t_dec * 1e-3) / 1e12},
    }

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    names = ["gemm_nt_256", "gemm_nt_128_batched", "layernorm_cast", "softmax", "pack", "flash_attn",
             "gemm_nt_128_flat", "gemm_nt_stream", "gemm_nt_wide"]
    kernels = {}
    for i, nm in enumerate(names):
        if ln[i]:
            kernels[nm] = {"launches_per_step": ln[i] // nprof, "ms_per_step": ms[i] / nprof,
                           "avg_us": ms[i] / ln[i] * 1e3,
                           "algo_tflops": (fl[i] / (ms[i] * 1e-3) / 1e12) if fl[i] else None,
                           "algo_gbps": by[i] / (ms[i] * 1e-3) / 1e9}
    # the dominant kernel: the MFMA kernel class with the most device time per step (the persistent 256x256 GEMM with
    # the fused q|k|v and fc1 projections of the latent stack, or the streaming GEMM with its out / fc2 projections)
    dom = max((i for i in range(NCLS) if ln[i] and fl[i]), key=lambda i: ms[i])
    g = kernels[names[dom]]
    # HBM traffic of that kernel per launch: PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes of
    # this same command, gfx950 correction of MI355X_MICROARCH.md) condensed into profiles/traffic.json
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            traffic = json.load(f)["pio::" + names[dom]]["bytes_per_launch"]
    except Exception:  # noqa: BLE001  (no committed profile yet)
        traffic = None
    what = {"gemm_nt_wide": "the weight GEMMs of the latent stack: fused q|k|v, out, fc1 (GELU), fc2 projections with the "
                            "LayerNorms folded into them, and the decoder projections",
            "gemm_nt_stream": "weight GEMMs of the latent stack"}.get(names[dom], "")
    roofline = {"kernel": f"pio::{names[dom]} ({what})",
                "bound": "mfma", "achieved": g["algo_tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": g["algo_tflops"] / MFMA_PEAK_TFLOPS, "traffic": traffic,
                "avg_launch_us": g["avg_us"], "launches_per_step": g["launches_per_step"],
                "algo_flops_per_launch": fl[dom] / ln[dom], "algo_bytes_per_launch": by[dom] / ln[dom]}

    workload = ("imagenet224 ClassificationPerceiver (conv+Fourier prep -> encoder 3136x322->512x1024, 8x6 SA -> decoder "
                "1000 queries -> final Linear), all rows computed")
    if args.hot_path_only:
        workload = "hot path only (encoder + decoder + final Linear) on a resident [B,3136,322] array"
    out = {
        "metric": "samples/sec PerceiverIO fwd (ImageNet-224, 512x1024 latents, 8 blocks x 6 self-attends)",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if args.policy.startswith("fp16") else "bf16", "data": "synthetic",
        "config": {"workload": workload, "batch_per_gpu": B, "global_batch": B * world,
                   "precision_policy": args.policy,
                   "parallelism": f"dp{world} (batch sharded, all-gather of logits)"},
        "per_gpu": value / world,
        "model_algo_tflops": value * GFLOP_PER_SAMPLE / 1e3,
        "model_mfma_frac": value / world * GFLOP_PER_SAMPLE / 1e3 / MFMA_PEAK_TFLOPS,
        "roofline": roofline, "stages": stages, "kernels": kernels, "parity": parity,
    }
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        v, dt = cpu_baseline(params, args.cpu_sample)
        out["cpu_baseline"] = {"value": v, "unit": "samples/s", "cores": cpu_threads(), "kind": "port",
                               "sample": f"numpy fp32 oracle of the hot path (encoder+decoder, 99.9% of the model's "
                                         f"FLOPs), same parameters, B={args.cpu_sample}, one forward ({dt:.1f} s) after "
                                         f"a B=1 warm-up"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
