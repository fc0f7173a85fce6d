"""Dev tool (GPU): which kernel every pio_gemm_nt launch of one forward runs on (PIO_GEMM_LOG=1 must be set in the environment).
    PIO_GEMM_LOG=1 python tools/gemm_routing.py language 2> routing.txt"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import bench as Bn  # noqa: E402

name = sys.argv[1]
dev = torch.device("cuda:0")
m, _ = Bn.build_model(name, dev, Bn.CONFIGS[name]["policy"])
x = Bn.make_inputs(name, Bn.CONFIGS[name]["batch"], 0, dev)
with torch.inference_mode():
    m(*x)
    torch.cuda.synchronize()
    sys.stderr.write("=====LOGSTART\n")
    m(*x)
    torch.cuda.synchronize()
