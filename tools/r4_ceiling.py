"""Dev tool (GPU box): round-4 energy table of the latent self-attend layer (ImageNet config, B = 32: 16384 rows x 1024
channels, 8 heads) -- every kernel of the layer back to back on random data with the socket power sampled through amdsmi
at ~50 Hz, the power floor of an active chip without work, the whole 48-layer stack as the model runs it, and the main
loop of the fusion that was NOT shipped (fc1 -> GELU -> fc2 in one kernel, 64 rows per CU) as an energy skeleton.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC tools/microbench/energy_kernels.hip -o tools/_abl/libenergy.so
    python tools/r4_ceiling.py profiles/r4_ceiling.json

Per kernel: us per launch, socket power (mean / max over the loop, first 0.4 s skipped), the driver's sclk, energy per
launch = us x W, pJ per algorithmic flop, and the DYNAMIC energy = us x (W - W_floor), where W_floor is the power the
same chip draws with every CU holding sleeping waves (clocks running, nothing executing).  The static share of a kernel's
joules shrinks with its time; only the dynamic share is work.
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from perceiverio_pytorch_amd import _lib as L  # noqa: E402
from ceiling_probe import PowerSampler, loop  # noqa: E402  (the sampler / timing loop of round 3's probe)

lib = L.lib()
dev = torch.device("cuda:0")
M, K = 16384, 1024
CAP_W = 1400.0


def stream():
    return torch.cuda.current_stream().cuda_stream


def rec_of(us, iters, pw, flops):
    w = pw["power_w_mean"]
    r = {"us": round(us, 2), "iters": iters, "algo_tflops": round(flops / us / 1e6, 1) if flops else None, "power": pw}
    if w:
        r["energy_mj_per_launch"] = round(us * 1e-6 * w * 1e3, 2)
        if flops:
            r["energy_pj_per_algorithmic_flop"] = round(us * 1e-6 * w / flops * 1e12, 3)
    return r


def gemm_variants(sampler):
    x16 = torch.randn(M, K, device=dev).half()
    xlo = (torch.randn(M, K, device=dev) * 1e-3).half()
    part = torch.empty(M, 8, 2, device=dev)
    xf = x16.float().view(M, 8, 128)
    part[:, :, 0] = xf.sum(-1)
    part[:, :, 1] = (xf * xf).sum(-1)
    out = {}

    def base(N):
        W = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        b = torch.randn(N, device=dev)
        g = L.Gemm()
        g.A, g.B = x16.data_ptr(), W.data_ptr()
        g.M, g.N, g.K = M, N, K
        g.lda, g.ldb, g.ldc = K, K, N
        g.batch, g.nh = 1, 1
        g.bias, g.bias_mode, g.act, g.alpha = b.data_ptr(), 1, 0, 1.0
        g.n_store, g.dtype = N, L.PIO_DT_F16
        return g, (W, b)

    def consumer(N, act):
        g, keep = base(N)
        c = torch.randn(N, device=dev)
        Cc = torch.empty(M, N, device=dev, dtype=torch.float16)
        g.C, g.out_f32, g.act = Cc.data_ptr(), 0, act
        g.ln_part, g.ln_c, g.ln_eps = part.data_ptr(), c.data_ptr(), 1e-5
        return g, keep + (c, Cc)

    def producer():
        # (round 4: the residual stream is updated IN PLACE -- the pair read is the pair written)
        g, keep = base(1024)
        xh, xl = x16.clone(), xlo.clone()
        po = torch.empty(M, 8, 2, device=dev)
        g.C, g.out_f32 = None, 1
        g.X16, g.X16_lo, g.ld16 = xh.data_ptr(), xl.data_ptr(), 1024
        g.R16_hi, g.R16_lo = xh.data_ptr(), xl.data_ptr()
        g.row_part = po.data_ptr()
        # (back to back the in-place stream would random-walk out of range: the A operand is scaled down instead)
        A = (torch.randn(M, K, device=dev) * 1e-3).half()
        g.A = A.data_ptr()
        g.bias, g.bias_mode = None, 0
        return g, keep + (xh, xl, po, A)

    for name, (g, keep), N in [("qkv_consumer 16384x3072x1024", consumer(3072, 0), 3072),
                               ("out_or_fc2_producer 16384x1024x1024 (in place)", producer(), 1024),
                               ("fc1_consumer_gelu 16384x1024x1024", consumer(1024, 1), 1024)]:
        st = stream()
        fn = lambda: L.check(lib.pio_gemm_nt(C.byref(g), st), name)  # noqa: E731
        us, iters, pw = loop(fn, 2.5, sampler)
        out[name] = rec_of(us, iters, pw, 2.0 * M * N * K)
        print(name, json.dumps(out[name]), flush=True)
    return out


def flash(sampler):
    B, H, T, d = 32, 8, 512, 128
    qkv = torch.randn(B * T, 3 * H * d, device=dev).half()
    o = torch.empty(B * T, H * d, device=dev, dtype=torch.float16)
    ld3 = 3 * H * d
    st = stream()
    base = qkv.data_ptr()
    fn = lambda: L.check(lib.pio_flash_attention(L.PIO_DT_F16, d, d, d, base, base + H * d * 2, base + 2 * H * d * 2,  # noqa: E731
                                                 o.data_ptr(), B, H, T, T, ld3, ld3, ld3, H * d, T * ld3, T * ld3, T * ld3,
                                                 T * H * d, 1, st), "flash")
    us, iters, pw = loop(fn, 2.5, sampler)
    r = rec_of(us, iters, pw, 2.0 * B * H * T * T * 2 * d)
    print("flash_attn<128,128> row-major V", json.dumps(r), flush=True)
    return r


def energy_lib():
    path = os.path.join(ROOT, "tools", "_abl", "libenergy.so")
    if not os.path.exists(path):
        return None
    e = C.CDLL(path)
    e.ek_spin.argtypes = [C.c_void_p, C.c_ulonglong, C.c_int, C.c_void_p]
    e.ek_mlp64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return e


def floor_and_skeleton(sampler):
    e = energy_lib()
    if e is None:
        return {"error": "tools/_abl/libenergy.so not built"}
    out = {}
    cyc = torch.zeros(4, dtype=torch.int64, device=dev)
    st = stream()
    for busy, name in ((0, "active_floor_sleeping_waves"), (1, "active_floor_scalar_polling")):
        fn = lambda: e.ek_spin(st, 200000, busy, cyc.data_ptr())  # noqa: E731   (~0.1 ms per launch)
        us, iters, pw = loop(fn, 2.5, sampler)
        out[name] = {"us": round(us, 2), "power": pw}
        print(name, json.dumps(out[name]), flush=True)
    X = torch.randn(M, K, device=dev).half()
    W1 = (torch.randn(K, K, device=dev) / K ** 0.5).half()
    W2 = (torch.randn(K, K, device=dev) / K ** 0.5).half()
    o = torch.empty(256 * 256, device=dev)
    fn = lambda: e.ek_mlp64(st, X.data_ptr(), W1.data_ptr(), W2.data_ptr(), o.data_ptr())  # noqa: E731
    us, iters, pw = loop(fn, 2.5, sampler)
    out["fused_mlp_64row_main_loop_skeleton"] = rec_of(us, iters, pw, 2.0 * 2.0 * M * K * K)
    print("fused_mlp_64row_main_loop_skeleton", json.dumps(out["fused_mlp_64row_main_loop_skeleton"]), flush=True)
    # idle: nothing running at all
    with sampler:
        t0 = time.perf_counter()
        time.sleep(2.0)
        t1 = time.perf_counter()
    out["idle_no_kernel"] = sampler.summary(t0 + 0.5, t1)
    return out


def stack_mixture(sampler):
    import perceiverio_pytorch_amd as P
    from perceiverio_pytorch_amd.perceiver import PerceiverEncoder
    P.set_precision_policy("fp16")
    enc = PerceiverEncoder(322, 6, 8, 512, 1024, num_self_attend_heads=8).to(dev).eval()
    x = torch.randn(32, 3136, 322, device=dev)
    res = {}
    with torch.inference_mode():
        lat = enc.latents(x)
        for arm, env in (("in_place_stream (shipped)", "1"), ("ping_pong_pairs (round 3)", "0")):
            os.environ["PIO_FOLD_INPLACE"] = env
            fn = lambda: enc(x, lat)  # noqa: E731
            us, iters, pw = loop(fn, 4.0, sampler)
            t_cross = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            enc.cross_attend(lat, x)
            e0.record()
            for _ in range(5):
                enc.cross_attend(lat, x)
            e1.record()
            torch.cuda.synchronize()
            t_cross = e0.elapsed_time(e1) / 5
            layer_ms = (us / 1e3 - t_cross) / 48
            w = pw["power_w_mean"]
            res[arm] = {"encoder_ms": round(us / 1e3, 3), "cross_attend_ms": round(t_cross, 3), "layer_ms": round(layer_ms, 4),
                        "power": pw, "layer_energy_mj": round(w * layer_ms, 1) if w else None,
                        "layer_mfma_frac": round(7.516e9 * 32 / (layer_ms * 1e-3) / 2.5e15, 4)}
            print(arm, json.dumps(res[arm]), flush=True)
        os.environ.pop("PIO_FOLD_INPLACE", None)
    return res


if __name__ == "__main__":
    sampler = PowerSampler()
    res = {"device": torch.cuda.get_device_name(0), "socket_power_cap_w": CAP_W}
    res["floor_and_fusion_skeleton"] = floor_and_skeleton(sampler)
    res["gemm_kernels"] = gemm_variants(sampler)
    res["flash_attn"] = flash(sampler)
    res["stack"] = stack_mixture(sampler)
    # ---- the table: static / dynamic split
    fl = res["floor_and_fusion_skeleton"].get("active_floor_sleeping_waves", {}).get("power", {}).get("power_w_mean")
    if fl:
        rows = dict(res["gemm_kernels"])
        rows["flash_attn<128,128> 32x8x512x512"] = res["flash_attn"]
        rows["fused_mlp_64row_main_loop_skeleton"] = res["floor_and_fusion_skeleton"]["fused_mlp_64row_main_loop_skeleton"]
        table = {}
        for k, r in rows.items():
            w = r["power"]["power_w_mean"]
            if not w:
                continue
            table[k] = {"us": r["us"], "power_w": round(w, 1), "energy_mj": r.get("energy_mj_per_launch"),
                        "static_mj": round(r["us"] * 1e-6 * fl * 1e3, 2),
                        "dynamic_mj": round(r["us"] * 1e-6 * (w - fl) * 1e3, 2),
                        "dynamic_pj_per_flop": round(r["us"] * 1e-6 * (w - fl) / (r["algo_tflops"] * 1e12 * r["us"] * 1e-6) * 1e12, 3)
                        if r.get("algo_tflops") else None}
        mult = {"qkv": 1, "out_or": 2, "fc1": 1, "flash": 1}
        layer_us = sum(v["us"] * m for k, v in table.items() for p, m in mult.items() if k.startswith(p))
        layer_dyn = sum(v["dynamic_mj"] * m for k, v in table.items() for p, m in mult.items() if k.startswith(p))
        res["table"] = {"active_floor_w": round(fl, 1), "kernels": table,
                        "layer_back_to_back": {"us": round(layer_us, 1), "dynamic_mj": round(layer_dyn, 1),
                                               "static_mj_at_floor": round(layer_us * 1e-6 * fl * 1e3, 1)},
                        "fc1_plus_fc2_separate": {k: round(table["fc1_consumer_gelu 16384x1024x1024"][k] +
                                                           table["out_or_fc2_producer 16384x1024x1024 (in place)"][k], 2)
                                                  for k in ("us", "energy_mj", "dynamic_mj")},
                        "reading": "dynamic = (socket power - active floor) x time; a layer at the cap takes "
                                   "(dynamic_mj + floor x t) / cap"}
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r4_ceiling.json"
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", path)
