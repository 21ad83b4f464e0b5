"""world_size-2 gloo test of the N>1 path: request sharding, the load-time arena broadcast, result gathering, max-over-ranks timing."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load_package()
    import torch
    import torch.distributed as dist
    from minigpt4_cpp_amd import dist as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        arena = torch.zeros(3 * 1024 * 1024 + 17, dtype=torch.uint8)
        if rank == 0:
            arena.copy_(torch.from_numpy(np.random.default_rng(9).integers(0, 256, arena.numel(), dtype=np.uint8)))
        D.broadcast_arena(arena, src=0, chunk_bytes=1 << 20)
        want = np.random.default_rng(9).integers(0, 256, arena.numel(), dtype=np.uint8)
        assert np.array_equal(arena.numpy(), want)
        mine = D.shard_requests(7, rank, world)
        results = {i: f"req{i}:" + "".join(chr(97 + (i * 7 + k) % 26) for k in range(4)) for i in mine}   # stand-in for decoded text
        allr = D.gather_objects(results, world)
        merged = {}
        for r in allr:
            merged.update(r)
        assert sorted(merged) == list(range(7))
        t = D.max_over_ranks(1.0 + rank)
        assert t == float(world)
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
