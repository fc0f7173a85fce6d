"""Dev tool (GPU): multimodal forward time against decode_chunks_per_call (how many of the 128 output chunks one decoder
call handles)."""
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
for g in [int(v) for v in (sys.argv[1:] or ["4", "5", "6", "8"])]:
    model.decode_chunks_per_call = g
    with torch.inference_mode():
        for _ in range(2):
            model(*ins)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model(*ins)
        torch.cuda.synchronize()
    elapsed = (time.perf_counter() - t0) / 3 * 1e3
    import ctypes as C
    from perceiverio_pytorch_amd import _lib as L
    lib = L.lib()
    with torch.inference_mode():
        L.check(lib.pio_prof_begin(20000))
        model(*ins)
        ms = (C.c_double * 9)(); fl = (C.c_double * 9)(); by = (C.c_double * 9)(); ln = (C.c_int64 * 9)()
        lib.pio_prof_end(ms, fl, by, ln)
    print("   kernel classes (ms/launches):", " ".join(f"{ms[i]:.2f}/{ln[i]}" for i in range(9)), f"sum {sum(ms):.2f}",
          "| hipMalloc count:", torch.cuda.memory_stats()["num_device_alloc"], flush=True)
    print(f"decode_chunks_per_call={g}: {elapsed:.2f} ms, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    model.__dict__.pop("_query_cache", None)
    torch.cuda.empty_cache()
