"""Dev tool (GPU): where a 512-row projection's ~10 us go -- time against K (slope = the K loop, intercept = launch +
prologue + epilogue), against N, and with the tile choice forced (env PIO_GEMM_TILE / PIO_GEMM_T32 in separate runs)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import gemm_bench as G  # noqa: E402

print({k: v for k, v in os.environ.items() if k.startswith("PIO_")})
for M in (512, 1024, 2048):
    for K in (256, 512, 1024, 2048):
        G.run(M, 1024, K, False, False, 0, iters=200)
    G.run(M, 3072, 1024, False, False, 0, iters=200)
G.run(2048, 1536, 512, False, False, 0, iters=200)
G.run(2048, 512, 512, False, False, 0, iters=200)
