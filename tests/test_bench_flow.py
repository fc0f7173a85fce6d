"""bench.py's own multi-rank control flow on CPU: `python bench.py --gpus 2` with PIO_BENCH_STUB=1 spawns the ranks
through torch.distributed.run exactly as the driver's launch does (bench.spawn_ranks), every rank builds its own inputs,
steps, all-gathers through perceiverio_pytorch_amd.dist.all_gather_rows, the timed region is bracketed by barriers and
reduced with MAX, and rank 0 alone prints the JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_main_control_flow_stub(gpus):
    env = dict(os.environ, PIO_BENCH_STUB="1", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3",
                          "--warmup", "1", "--batch", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line (rank 0): {out.stdout[-500:]}"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == gpus and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["config"]["global_batch"] == 4 * gpus and rec["data"] == "stub"
    assert rec["gather_ok"] is True
    assert rec["value"] > 0 and rec["scaling"] == "weak" and rec["higher_is_better"] is True
