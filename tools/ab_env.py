"""Dev tool (GPU): in-process A/B of library switches that are read from the environment at every call
(cdna_hip_programming.md rule 24: interleaved rounds in ONE process, median and min per arm).

    python tools/ab_env.py PIO_FOLD_INPLACE=0 PIO_FOLD_INPLACE=1 [--config imagenet] [--policy fp16sd] [--rounds 7]

Each arm is a comma-separated list of NAME=VALUE settings applied with os.environ before the arm's forwards.  Prints ms per
forward (eager launches, HIP events on the launch stream) and whether the arms' outputs are bit-identical.
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import bench as Bn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("arms", nargs="+")
ap.add_argument("--config", default="imagenet")
ap.add_argument("--policy", default=None)
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--graph", action="store_true", help="also time one HIP graph replay per forward, per arm")
args = ap.parse_args()

dev = torch.device("cuda:0")
cfg = Bn.CONFIGS[args.config]
policy = args.policy or cfg["policy"]
model, _ = Bn.build_model(args.config, dev, policy)
inputs = Bn.make_inputs(args.config, args.batch or cfg["batch"], 0, dev)


def apply(arm):
    for kv in arm.split(","):
        k, v = kv.split("=", 1)
        os.environ[k] = v


def flat(y):
    return torch.cat([v.float().flatten() for v in (y.values() if isinstance(y, dict) else [y])])


times = {a: [] for a in args.arms}
gtimes = {a: [] for a in args.arms}
graphs = {}
outs = {}
with torch.inference_mode():
    for a in args.arms:
        apply(a)
        for _ in range(2):
            y = model(*inputs)
        outs[a] = flat(y).clone()
        if args.graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    model(*inputs)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                yg = model(*inputs)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(flat(yg), outs[a])
            graphs[a] = (g, yg)
    for r in range(args.rounds):
        for a in (args.arms if r % 2 == 0 else args.arms[::-1]):
            apply(a)
            model(*inputs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                model(*inputs)
            e1.record()
            torch.cuda.synchronize()
            times[a].append(e0.elapsed_time(e1) / args.steps)
            if args.graph:
                g = graphs[a][0]
                g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                gtimes[a].append(e0.elapsed_time(e1) / args.steps)
ref = outs[args.arms[0]]
for a in args.arms:
    t = times[a]
    same = bool(torch.equal(outs[a], ref))
    print(f"{a:40s} median {statistics.median(t):8.3f} ms  min {min(t):8.3f} ms  max {max(t):8.3f}  "
          f"bit-identical to arm 0: {same}"
          + (f"  | graph replay median {statistics.median(gtimes[a]):8.3f} min {min(gtimes[a]):8.3f}" if args.graph else ""),
          flush=True)
