"""Dev tool (GPU): sample the main thread's Python stack every 2 ms during multimodal forwards and print the frames the
host sat in for more than 20 ms -- which call is the host blocked in during the ~90 ms queue stalls?"""
import os
import sys
import threading
import time
import traceback

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import bench as Bn  # noqa: E402

dev = torch.device("cuda:0")
cfg = Bn.CONFIGS["multimodal"]
model, params = Bn.build_model("multimodal", dev, cfg["policy"])
ins = Bn.make_inputs("multimodal", 1, 0, dev)
model.decode_chunks_per_call = int(sys.argv[1])
main_id = threading.main_thread().ident
samples, stop = [], False


def sampler():
    while not stop:
        fr = sys._current_frames().get(main_id)
        if fr is not None:
            st = traceback.extract_stack(fr)[-4:]
            samples.append((time.perf_counter(), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(st))))
        time.sleep(0.002)


with torch.inference_mode():
    for _ in range(2):
        model(*ins)
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    for _ in range(2):
        model(*ins)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    stop = True
    th.join()
print(f"2 forwards: launch phase {1e3 * (t1 - t0):.1f} ms, final sync {1e3 * (t2 - t1):.1f} ms, {len(samples)} samples")
runs, cur, start, last = [], None, 0.0, 0.0
for t, s in samples:
    if s != cur:
        if cur is not None and last - start > 0.02:
            runs.append((last - start, cur))
        cur, start = s, t
    last = t
if cur is not None and last - start > 0.02:
    runs.append((last - start, cur))
for d, s in runs:
    print(f"  {1e3 * d:7.1f} ms in {s}")
