"""Dev tool (GPU): where the host time of a multimodal forward goes at a given decode_chunks_per_call -- wall time (with
a device synchronisation) of each decoder call and of the post-processing between two calls."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import bench as Bn  # noqa: E402

dev = torch.device("cuda:0")
cfg = Bn.CONFIGS["multimodal"]
model, params = Bn.build_model("multimodal", dev, cfg["policy"])
ins = Bn.make_inputs("multimodal", 1, 0, dev)
g = int(sys.argv[1])
model.decode_chunks_per_call = g
variant = sys.argv[2] if len(sys.argv) > 2 else ""
if variant == "nocache":
    model.cache_queries = False
if variant == "fp16dec":
    model.perceiver.decoder_policy = "fp16"
if variant == "norange":
    from perceiverio_pytorch_amd import runtime as R
    R.set_range_check(False) if hasattr(R, "set_range_check") else None
dec = model.perceiver._decoder
orig = dec.forward
log = []


def timed(*a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = orig(*a, **k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    log.append((t0, t1, time.perf_counter()))
    return y


with torch.inference_mode():
    for _ in range(2):
        model(*ins)
    dec.forward = timed
    st0 = torch.cuda.memory_stats()
    t_begin = time.perf_counter()
    model(*ins)
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    st1 = torch.cuda.memory_stats()
host = [(b - a) * 1e3 for a, b, c in log]
total = [(c - a) * 1e3 for a, b, c in log]
between = [(log[i + 1][0] - log[i][2]) * 1e3 for i in range(len(log) - 1)]
import glob
for f in glob.glob(f"/sys/class/kfd/kfd/proc/{os.getpid()}/stats_*/evicted_ms") + glob.glob(f"/sys/class/kfd/kfd/proc/{os.getpid()}/queues/*/size"):
    try:
        print("  ", f, open(f).read().strip())
    except OSError as e:
        print("  ", f, e)
print("   slow calls (index: call+sync ms):", [(i, round(t, 1)) for i, t in enumerate(total) if t > 10],
      "slow gaps:", [(i, round(t, 1)) for i, t in enumerate(between) if t > 10])
print(f"g={g} {variant}: forward {1e3 * (t_end - t_begin):.1f} ms; {len(log)} decoder calls: host part median {sorted(host)[len(host) // 2]:.2f} "
      f"max {max(host):.2f} ms, call+sync median {sorted(total)[len(total) // 2]:.2f} max {max(total):.2f} ms; between calls "
      f"median {sorted(between)[len(between) // 2]:.2f} max {max(between):.2f} ms")
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated", "segment.all.allocated"):
    print(f"   {k}: +{st1[k] - st0[k]}")
print("   reserved GiB", st1["reserved_bytes.all.current"] / 2**30, "allocated GiB", st1["allocated_bytes.all.current"] / 2**30)
