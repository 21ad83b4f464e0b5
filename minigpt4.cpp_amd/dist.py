"""Data-parallel request sharding + load-time weight broadcast (host-side plumbing over torch.distributed).

The path shards as independent requests: a request = (image, prompt) owns its embedding, KV cache and n_past, so ranks
never exchange anything per token.  The only collective is the load-time broadcast of the two weight arenas from rank 0
(`ncclBroadcast` over xGMI on the GPU box; the same code runs on `gloo` in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence


def shard_requests(n_requests: int, rank: int, world: int) -> List[int]:
    """Round-robin: rank r serves requests r, r+world, ... (BASELINE config 4: 32 images over 8 GPUs -> 4 per replica)."""
    return list(range(rank, n_requests, world))


def broadcast_arena(tensor, src: int = 0, chunk_bytes: int = 1 << 30):
    """Broadcast a flat uint8 tensor (a weight arena) from `src` in <= 1 GiB pieces (one large collective per piece: the ring is
    per-link bound on xGMI, so few large messages beat many small ones)."""
    import torch.distributed as dist
    n = tensor.numel()
    for off in range(0, n, chunk_bytes):
        dist.broadcast(tensor[off:min(n, off + chunk_bytes)], src=src)
    return tensor


def gather_objects(obj, world: int):
    """Results (token strings) stay on their rank; the host gathers them for reporting."""
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
