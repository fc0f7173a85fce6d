"""Data-parallel plumbing for the hot path (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Every sample is independent end to end, so the batch is sharded over ranks with replicated
weights and NO collective on the data path; the only exchange is an all-gather of the per-rank logits
[B/W, classes] (16 KiB per rank at B=32, W=8: latency-bound over xGMI).  `gloo` is used by the CPU tests."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the contiguous shard owned by `rank`; the first (total % world) ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """This rank's slice of a [B, ...] tensor (a view)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


_gather_buffers = {}


def _gather_buffer(shape, dtype, device) -> torch.Tensor:
    """One preallocated [rows * world, ...] receive buffer per (shape, dtype, device): the per-step all-gather of the
    logits writes into it directly (all_gather_into_tensor) instead of allocating `world` pieces and concatenating
    them.  The caller owns the result until its next call with the same shape."""
    key = (tuple(shape), dtype, str(device))
    buf = _gather_buffers.get(key)
    if buf is None:
        with torch.inference_mode(False):
            buf = torch.empty(shape, dtype=dtype, device=device)
        _gather_buffers[key] = buf
    return buf


def all_gather_rows(local: torch.Tensor, total_rows: Optional[int] = None, group=None,
                    reuse_buffer: bool = False) -> torch.Tensor:
    """Concatenate every rank's [b_r, ...] block along dim 0 in rank order.  Equal shards are ONE
    all_gather_into_tensor into a fresh [b * W, ...] tensor -- or, with reuse_buffer=True (a step loop that consumes
    the result before its next call: bench.py), into ONE preallocated buffer per shape that the next call of the same
    shape OVERWRITES; ragged shards (total_rows given, not divisible) are padded to the largest shard and trimmed."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    if total_rows is None or total_rows % world == 0:
        shape = (local.shape[0] * world,) + tuple(local.shape[1:])
        out = _gather_buffer(shape, local.dtype, local.device) if reuse_buffer else \
            torch.empty(shape, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    sizes = [shard_bounds(total_rows, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


# ---------------------------------------------------------------------------------------------------------------
# query-dimension sharding, for batches smaller than the world (optical flow: B = 1, Q = 182 528 decoder queries).
# Decoder query rows are mutually independent (each row's softmax runs over the latents only: reference
# perceiver_io/transformer_primitives.py:138-166, perceiver.py:166-180), so rank r decodes rows [lo_r, hi_r) of the
# query array against the (replicated, cheap) latents and the [B, Q/W, C] blocks are all-gathered along dim 1.
# ---------------------------------------------------------------------------------------------------------------
def shard_queries(query: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None,
                  query_mask: Optional[torch.Tensor] = None):
    """This rank's rows [lo, hi) of a [B, Q, C] query array (a view) and of its optional [B, Q] mask."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(query.shape[1], rank, world)
    return query[:, lo:hi], (query_mask[:, lo:hi] if query_mask is not None else None)


def all_gather_queries(local: torch.Tensor, total_queries: int, group=None) -> torch.Tensor:
    """Concatenate every rank's [B, q_r, C] block along dim 1 in rank order (ragged shards padded and trimmed)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    rows_first = local.transpose(0, 1).contiguous()                   # [q_r, B, C]: the sharded axis leads
    return all_gather_rows(rows_first, total_rows=total_queries, group=group).transpose(0, 1).contiguous()


def decode_query_sharded(decode, query: torch.Tensor, latents: torch.Tensor,
                         query_mask: Optional[torch.Tensor] = None, rank: Optional[int] = None,
                         world: Optional[int] = None, group=None) -> torch.Tensor:
    """`decode(query_rows, latents, query_mask=rows_of_mask)` on this rank's query rows, then the all-gather along
    the query axis: every rank returns the full [B, Q, C_out].  `decode` is PerceiverDecoder.forward in the product
    path (PerceiverIO.forward(query_shard=...), bench.py --config flow) and any row-independent function in tests."""
    if world is not None and world > 1:
        # an explicit shard of a world that does not exist would return a [B, Q/W, C] block that every caller downstream
        # (restructure, the postprocessors) takes for the full query set: refuse instead of returning wrong output
        if not dist.is_initialized():
            raise RuntimeError(f"decode_query_sharded(rank={rank}, world={world}): torch.distributed is not initialised; "
                               "a query shard can only be completed by the all-gather of an initialised process group")
        if dist.get_world_size(group) != world:
            raise RuntimeError(f"decode_query_sharded: world={world} but the process group has "
                               f"{dist.get_world_size(group)} ranks")
    q_local, m_local = shard_queries(query, rank, world, query_mask)
    y_local = decode(q_local, latents, query_mask=m_local)
    return all_gather_queries(y_local, query.shape[1], group=group)
