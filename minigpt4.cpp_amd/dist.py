"""Data-parallel replicas: request sharding + the load-time weight broadcast (host-side orchestration over torch.distributed).

The path shards as independent requests: a request = (image, prompt) owns its embedding, KV cache and n_past, so ranks
never exchange anything per token.  The only collective is the load-time broadcast of the two weight arenas from rank 0
(`ncclBroadcast` over xGMI on the GPU box; the same code runs on `gloo` in the CPU tests):

  rank 0        : minigpt4_model_load reads both files, repacks, fills its arenas                     (LOAD_FULL)
  ranks 1..N-1  : MINIGPT4_LOAD=recv -> headers only: the arenas are laid out and allocated, nothing is read / uploaded / repacked   (LOAD_RECV)
  all           : the layout plans (sizes + take()-sequence hashes) must agree; both arenas are broadcast in <= 1 GiB pieces;
                  receivers call minigpt4_amd_weights_received; the arena checksums must agree
so the 11.4 GB of the 13B pair cross the file system once per node instead of once per GPU.

Process set-up order: a process that uses this module on GPUs must initialise torch's device side (`torch.cuda.set_device(local_rank)`) BEFORE it loads
libminigpt4.so -- torch ships its own HIP runtime, and it only finds the GPUs when that runtime is the first one mapped (bench.py and tests/conftest.py do this).
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


def arena_tensor(lib, ctx, which: int, device):
    """A flat uint8 torch view of weight arena `which` (0: LLM, 1: vision) of a loaded context -- no copy (the engine owns the memory)."""
    import ctypes

    import torch
    ptr, nb = ctypes.c_void_p(), ctypes.c_size_t()
    assert lib.library.minigpt4_amd_weight_arena(ctx.ptr, which, ctypes.byref(ptr), ctypes.byref(nb)) == 0

    class _Arena:
        __cuda_array_interface__ = {"shape": (nb.value,), "typestr": "|u1", "data": (ptr.value, False), "version": 2}
    return torch.as_tensor(_Arena(), device=device)


def load_replica(lib, vision_path: str, llm_path: str, rank: int, world: int, device=None, **load_kw):
    """Load this rank's replica: rank 0 from the files, every other rank by receiving rank 0's arenas.  Returns (ctx, stats) with
    stats = {"mode", "load_s" (file load on rank 0 / header-only load elsewhere), "bcast_ms", "plan", "checksums"}; raises if the ranks disagree."""
    import os
    import time

    import torch
    import torch.distributed as dist
    recv = world > 1 and rank != 0
    t0 = time.time()
    if recv:
        os.environ["MINIGPT4_LOAD"] = "recv"
    try:
        ctx = lib.minigpt4_model_load(vision_path, llm_path, **load_kw)
    finally:
        os.environ.pop("MINIGPT4_LOAD", None)
    load_s = time.time() - t0
    stats = {"mode": "recv" if recv else "full", "load_s": load_s, "bcast_ms": None, "plan": lib.amd_arena_plan(ctx)}
    if world > 1:
        plans = gather_objects(stats["plan"], world)
        if any(p != plans[0] for p in plans):
            raise RuntimeError(f"rank {rank}: arena layouts differ between ranks: {plans}")
        if device is not None:
            torch.cuda.synchronize(device)
        dist.barrier()
        t0 = time.time()
        try:
            for which in (0, 1):
                broadcast_arena(arena_tensor(lib, ctx, which, device), src=0)
            if device is not None:
                torch.cuda.synchronize(device)
            stats["bcast_ms"] = (time.time() - t0) * 1e3       # only a broadcast that COMPLETED is a measurement
        except Exception as e:   # a launch-time refusal is raised by every rank at the same call: all of them take this branch and fall back to reading the files
            stats["bcast_error"] = f"{type(e).__name__}: {e}"[:300]
            stats["bcast_failed_after_ms"] = (time.time() - t0) * 1e3
            if recv:
                lib.minigpt4_free(ctx)
                t1 = time.time()
                ctx = lib.minigpt4_model_load(vision_path, llm_path, **load_kw)
                stats["mode"], stats["load_s"] = "full (broadcast refused)", time.time() - t1
                recv = False
        if recv:
            assert lib.library.minigpt4_amd_weights_received(ctx.ptr) == 0
        sums = gather_objects([lib.amd_arena_checksum(ctx, 0), lib.amd_arena_checksum(ctx, 1)], world)
        if any(c != sums[0] for c in sums):
            raise RuntimeError(f"rank {rank}: arena contents differ after the broadcast: {sums}")
        stats["checksums"] = sums[0]
        errs = gather_objects(stats.get("bcast_error"), world)
        if any(errs) and not stats.get("bcast_error"):          # some OTHER rank fell back to the files: this rank's timing is not a broadcast measurement either
            stats["bcast_error"], stats["bcast_ms"] = "another rank: " + next(e for e in errs if e), None
    return ctx, stats


def bcast_report(stats: dict):
    """The `weight_bcast` object of the bench line, from load_replica's stats: a measured broadcast, an explicit error record when the broadcast was refused
    (the replicas then came from the files -- never a GB/s figure), or None for a single-rank run."""
    if stats.get("bcast_error"):
        return {"error": stats["bcast_error"], "mode": stats.get("mode"), "failed_after_ms": stats.get("bcast_failed_after_ms"),
                "note": "the arena broadcast was refused; every rank loaded from the files instead -- no broadcast rate was measured"}
    if not stats.get("bcast_ms"):
        return None
    nbytes = stats["plan"]["llm_bytes"] + stats["plan"]["vision_bytes"]
    return {"bytes": nbytes, "ms": stats["bcast_ms"], "GBps": nbytes / stats["bcast_ms"] / 1e6, "xgmi_link_GBps": 153.0,
            "note": "both weight arenas, rank 0 -> all, ncclBroadcast in <= 1 GiB pieces; a ring broadcast is bound by one xGMI link"}
