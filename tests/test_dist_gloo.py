"""CPU, world_size 2 over gloo: the batch sharding and the all-gather of logits used by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from perceiverio_pytorch_amd.dist import all_gather_rows, shard_batch
    full = torch.arange(total * 5, dtype=torch.float32).reshape(total, 5)
    mine = shard_batch(full)                      # every sample is independent: a rank computes only its rows
    logits = mine * 2.0 + 1.0                     # stand-in for the per-rank forward
    out = all_gather_rows(logits, total_rows=total)
    q.put((rank, mine.shape[0], out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [32, 7])
def test_shard_and_allgather_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = np.arange(total * 5, dtype=np.float32).reshape(total, 5) * 2.0 + 1.0
    assert sum(r[1] for r in res) == total
    for _, _, out in res:
        assert np.array_equal(out, full)          # rank order preserved, ragged shard trimmed


def test_shard_bounds_cover_exactly():
    from perceiverio_pytorch_amd.dist import shard_bounds
    for total in (1, 7, 32, 33):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


# ---- query-dimension sharding (B < world: optical flow) -- the same dist.decode_query_sharded that
# ---- PerceiverIO.forward(query_shard=...) and bench.py --config flow call, with a row-independent stand-in decoder
def _decode_stub(query, latents, query_mask=None):
    y = query * 3.0 + latents.sum(dim=(1, 2))[:, None, None]
    if query_mask is not None:
        y = torch.where(query_mask[:, :, None], y, torch.full_like(y, -7.0))
    return y


def _qworker(rank, world, port, total_q, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from perceiverio_pytorch_amd.dist import decode_query_sharded, shard_queries
    B, C = 1, 6
    query = torch.arange(B * total_q * C, dtype=torch.float32).reshape(B, total_q, C)
    latents = torch.ones(B, 4, 3)
    mask = (torch.arange(total_q) % 3 != 0)[None].expand(B, total_q)
    mine, _ = shard_queries(query)
    out = decode_query_sharded(_decode_stub, query, latents, mask)      # rank / world from the process group
    first = out.clone()
    # a second call of the same shape must not overwrite the first result (B = 1: the transposed gather result is already
    # contiguous, so nothing copies it out of a shared receive buffer -- the default is a fresh tensor per call)
    out2 = decode_query_sharded(_decode_stub, query + 100.0, latents, mask)
    assert torch.equal(out, first) and not torch.equal(out2, first)
    q.put((rank, mine.shape[1], out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total_q", [64, 13])
def test_query_shard_and_allgather_world2(total_q):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_qworker, args=(r, 2, port, total_q, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    query = torch.arange(total_q * 6, dtype=torch.float32).reshape(1, total_q, 6)
    mask = (torch.arange(total_q) % 3 != 0)[None]
    full = _decode_stub(query, torch.ones(1, 4, 3), mask).numpy()
    assert sum(r[1] for r in res) == total_q
    for _, _, out in res:
        assert out.shape == full.shape and np.array_equal(out, full)


def test_explicit_query_shard_without_process_group_raises():
    """decode_query_sharded(rank, world > 1) without an initialised process group must not return a [B, Q/W, C] block
    that the callers downstream would take for the full query set."""
    from perceiverio_pytorch_amd.dist import decode_query_sharded
    query = torch.zeros(1, 8, 6)
    with pytest.raises(RuntimeError, match="not initialised"):
        decode_query_sharded(_decode_stub, query, torch.ones(1, 4, 3), None, 0, 2)
    # world 1 (or unspecified without a group) is the whole query set: fine
    assert decode_query_sharded(_decode_stub, query, torch.ones(1, 4, 3), None, 0, 1).shape == (1, 8, 6)
