"""Dev tool (GPU box): what bounds the kernels of one latent self-attend layer -- held clock, socket power and the
main-loop-only time of each GEMM variant, measured back to back.  Writes profiles/r3_ceiling.json (run from the repo
root).

    python tools/wide_stamps.py --build 0            (here: tools/_abl/libpio_wide_0.so, -DPIO_GEMM_STAMPS)
    PIO_LIB_PATH=tools/_abl/libpio_wide_0.so python tools/ceiling_probe.py gpurun_out/r3_ceiling.json

Per GEMM variant of the layer (ImageNet config, B = 32: 16384 rows):
    us            device time per launch over a >= 2.5 s back-to-back loop on random data (HIP events)
    ghz           clock held INSIDE the kernel: d(s_memtime) / d(s_memrealtime) x 100 MHz around workgroup 0
    cycles        s_memtime cycles of workgroup 0: whole kernel, up to the start of its last tile's epilogue, epilogue
    power_w       socket power while the loop runs, sampled through amdsmi at >= 20 Hz (mean / max over the loop)
    sclk_mhz      the driver's current graphics clock at the same instants
    main_loop_us  (cycles up to the last epilogue) / ghz for one-tile-per-CU launches; for the three-tile q|k|v launch
                  whole - 3 x epilogue: what the launch would cost with free epilogues at the clock the chip holds
The same sampling runs over the whole 48-layer stack of the model (library as shipped) for the real mixture.
"""
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from perceiverio_pytorch_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")
have_stamps = hasattr(lib, "pio_debug_wide_stamps")
try:
    lib.pio_debug_wide_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.pio_debug_wide_stamps.restype = C.c_int
except AttributeError:
    have_stamps = False


class PowerSampler:
    """socket power [W], current sclk [MHz], hotspot temperature [C] through amdsmi, in a thread (the calls are ctypes:
    the GIL is released while the main thread enqueues)."""

    def __init__(self, hz=50.0):
        self.dt = 1.0 / hz
        self.samples = []
        self.err = None
        self._stop = threading.Event()
        try:
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            self.h = amdsmi.amdsmi_get_processor_handles()[0]
        except Exception as e:  # noqa: BLE001
            self.smi = None
            self.err = repr(e)

    def read(self):
        s = self.smi
        out = {"t": time.perf_counter()}
        try:
            p = s.amdsmi_get_power_info(self.h)
            for k in ("current_socket_power", "average_socket_power", "socket_power"):
                v = p.get(k)
                if isinstance(v, (int, float)) and v > 0:
                    out["w"] = float(v)
                    break
            out["cap"] = p.get("power_limit")
        except Exception as e:  # noqa: BLE001
            self.err = repr(e)
        try:
            c = s.amdsmi_get_clock_info(self.h, s.AmdSmiClkType.GFX)
            out["sclk"] = c.get("clk")
        except Exception as e:  # noqa: BLE001
            self.err = self.err or repr(e)
        try:
            out["temp"] = s.amdsmi_get_temp_metric(self.h, s.AmdSmiTemperatureType.HOTSPOT, s.AmdSmiTemperatureMetric.CURRENT)
        except Exception:  # noqa: BLE001
            pass
        return out

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(self.read())
            time.sleep(self.dt)

    def __enter__(self):
        self.samples = []
        self._stop.clear()
        if self.smi is not None:
            self.th = threading.Thread(target=self._run, daemon=True)
            self.th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.smi is not None:
            self.th.join()

    def summary(self, t0, t1):
        xs = [s for s in self.samples if t0 <= s["t"] <= t1]
        w = [s["w"] for s in xs if "w" in s]
        k = [s["sclk"] for s in xs if isinstance(s.get("sclk"), (int, float))]
        tp = [s["temp"] for s in xs if isinstance(s.get("temp"), (int, float))]
        cap = next((s["cap"] for s in xs if s.get("cap")), None)
        return {"samples": len(xs), "hz": len(xs) / max(1e-9, t1 - t0),
                "power_w_mean": sum(w) / len(w) if w else None, "power_w_max": max(w) if w else None,
                "power_w_min": min(w) if w else None, "power_cap_w": cap,
                "sclk_mhz_mean": sum(k) / len(k) if k else None, "sclk_mhz_min": min(k) if k else None,
                "hotspot_c_max": max(tp) if tp else None, "error": self.err}


def loop(fn, seconds, sampler):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    iters = max(50, int(seconds * 1e6 / us))
    with sampler:
        time.sleep(0.05)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    us = e0.elapsed_time(e1) / iters * 1e3
    # (skip the ramp: the first 0.4 s of the loop)
    return us, iters, sampler.summary(t0 + min(0.4, 0.2 * (t1 - t0)), t1)


def stamps():
    if not have_stamps:
        return None
    s = (C.c_ulonglong * 12)()
    if lib.pio_debug_wide_stamps(s) != 0:
        return None
    t = [int(v) for v in s[:8]]
    c = [int(v) for v in s[8:12]]
    ghz = (c[2] - c[0]) / max(1, c[3] - c[1]) * 0.1
    return {"ghz": ghz, "whole_cycles": c[2] - c[0], "to_last_epilogue_cycles": t[6] - c[0],
            "last_epilogue_cycles": t[7] - t[6]}


def gemm_variants():
    M, K = 16384, 1024
    st = torch.cuda.current_stream().cuda_stream
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

    # consumer: LayerNorm(x) W^T + b from the un-normalised x16 and the row sums (q|k|v: N = 3072; fc1: N = 1024 + GELU)
    def consumer(N, act):
        g, keep = base(N)
        c = torch.randn(N, device=dev)
        Cc = torch.empty(M, N, device=dev, dtype=torch.float16)
        g.C, g.out_f32, g.act = Cc.data_ptr(), 0, act
        g.ln_part, g.ln_c, g.ln_eps = part.data_ptr(), c.data_ptr(), 1e-5
        return g, keep + (c, Cc)

    # producer: x + A W^T + b with the residual stream as a 16-bit pair (out / fc2 projections)
    def producer():
        g, keep = base(1024)
        xh, xl = torch.empty(M, 1024, device=dev, dtype=torch.float16), torch.empty(M, 1024, device=dev, dtype=torch.float16)
        po = torch.empty(M, 8, 2, device=dev)
        g.C, g.out_f32 = None, 1
        g.X16, g.X16_lo, g.ld16 = xh.data_ptr(), xl.data_ptr(), 1024
        g.R16_hi, g.R16_lo = x16.data_ptr(), xlo.data_ptr()
        g.row_part = po.data_ptr()
        return g, keep + (xh, xl, po)

    variants = [("qkv_consumer 16384x3072x1024", consumer(3072, 0), 3072, 3),
                ("out_or_fc2_producer 16384x1024x1024", producer(), 1024, 1),
                ("fc1_consumer_gelu 16384x1024x1024", consumer(1024, 1), 1024, 1)]
    sampler = PowerSampler()
    for name, (g, keep), N, tiles_per_cu in variants:
        fn = lambda: L.check(lib.pio_gemm_nt(C.byref(g), st), name)  # noqa: E731
        us, iters, pw = loop(fn, 2.5, sampler)
        rec = {"us": us, "iters": iters, "algo_tflops": 2.0 * M * N * K / us / 1e6, "power": pw}
        s = stamps()
        if s:
            rec["stamps_workgroup0"] = s
            ghz = s["ghz"]
            if tiles_per_cu == 1:
                main_cycles = s["to_last_epilogue_cycles"]
            else:
                main_cycles = s["whole_cycles"] - tiles_per_cu * s["last_epilogue_cycles"]
            rec["main_loop_us_at_held_clock"] = main_cycles / (ghz * 1e3)
            rec["main_loop_tflops"] = 2.0 * M * N * K / rec["main_loop_us_at_held_clock"] / 1e6
            # matrix-pipe occupancy of the main loop at the held clock: MFMA cycles of the launch per SIMD / main cycles
            mfma_cycles = 2.0 * M * N * K / (256 * 4 * 1024.0)      # 16x16x32: 16384 flop per 16 cycles per SIMD
            rec["main_loop_pipe_occupancy"] = mfma_cycles / main_cycles
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
    return out


def stack_mixture():
    """the real mixture: the 48-layer latent stack of the classifier (B = 32), library as loaded"""
    import perceiverio_pytorch_amd as P
    from perceiverio_pytorch_amd.perceiver import PerceiverEncoder
    P.set_precision_policy("fp16")
    enc = PerceiverEncoder(322, 6, 8, 512, 1024, num_self_attend_heads=8).to(dev).eval()
    x = torch.randn(32, 3136, 322, device=dev)
    sampler = PowerSampler()
    with torch.inference_mode():
        lat = enc.latents(x)
        fn = lambda: enc(x, lat)  # noqa: E731
        us, iters, pw = loop(fn, 4.0, sampler)
        L.check(lib.pio_prof_begin(16384))
        fn()
        ms = (C.c_double * 9)(); fl = (C.c_double * 9)(); by = (C.c_double * 9)(); ln = (C.c_int64 * 9)()
        lib.pio_prof_end(ms, fl, by, ln)
    return {"encoder_ms": us / 1e3, "iters": iters, "power": pw,
            "per_class_ms": {i: ms[i] for i in range(9) if ln[i]}, "per_class_launches": {i: ln[i] for i in range(9) if ln[i]}}


if __name__ == "__main__":
    res = {"device": torch.cuda.get_device_name(0), "stamps_build": have_stamps,
           "gemm_variants": gemm_variants(), "stack": stack_mixture()}
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3_ceiling.json"
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", path)
