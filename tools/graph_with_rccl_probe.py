"""Dev tool (GPU): does a HIP graph capture of the module forward survive an initialised RCCL process group (its
watchdog thread polls events)?  One rank, world_size 1."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench as Bn  # noqa: E402
from perceiverio_pytorch_amd.dist import all_gather_rows  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(4, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
model, _ = Bn.build_model("imagenet", dev, "fp16")
x = torch.randn(32, 3, 224, 224, device=dev)
for mode in sys.argv[1:] or ["global", "thread_local"]:
    try:
        with torch.inference_mode():
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    y = model(x)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=mode):
                yg = model(x)
            for _ in range(20):
                g.replay()
                out = all_gather_rows(yg)
            torch.cuda.synchronize()
            print(f"capture_error_mode={mode}: ok, identical {bool(torch.equal(yg, y))}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"capture_error_mode={mode}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
dist.destroy_process_group()
