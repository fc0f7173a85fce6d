"""Dev tool (GPU): the weight GEMMs of the optical-flow decoder (182 528 query rows x 322 channels, split activations)
with the channel pitch as it is (328) and rounded to 64 (384)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mm_dec_gemm_bench import run  # noqa: E402

M = 182528
for K1 in (328, 384):
    tag = f"pitch {K1}"
    run(f"q proj [{tag}]", M, 512, K1, False, False, 0, 512, 512, 0, iters=5)
    run(f"fc1 GELU [{tag}]", M, K1, K1, False, False, 1, K1, K1, 0, iters=5)
    run(f"fc2 + residual [{tag}]", M, 324, K1, True, True, 0, 324, 324, 0, iters=5)
    run(f"final N=2 [{tag}]", M, 2, K1, True, False, 0, 2, 2, 0, iters=5)
run("attention out, ldc 324", M, 324, 512, True, False, 0, 324, 324, 0, iters=5)
