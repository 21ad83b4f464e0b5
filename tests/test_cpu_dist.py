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
        # the receive path's planner, metadata only (no GPU): every rank derives the arena layout from the file HEADERS; the layouts must agree before any byte moves,
        # and a rank that only planned can size its receive buffers from the plan (here: CPU tensors standing in for the two HBM arenas)
        from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
        lib = ML.load_library()
        vp, lp = os.path.join(out_dir, "vision.bin"), os.path.join(out_dir, "llm.bin")
        if rank == 0:
            G.write_vision_file(vp, G.tiny_vision(n_embd_llm=4096), seed=3, std=0.05)
            G.write_llm_file(lp, G.tiny_llm(wtype="q5_k", n_embd=256, n_layer=2, n_head=4, n_vocab=512, mix="q5_k_m"), seed=1, std=0.05)
        dist.barrier()
        plan = lib.amd_plan_arenas(vp, lp)
        plans = D.gather_objects(plan, world)
        assert plans[0] == plans[1] and plan["llm_bytes"] > 0 and plan["vision_bytes"] > 0
        assert plan["llm_bytes"] >= os.path.getsize(lp) * 0.9 and plan["vision_bytes"] >= os.path.getsize(vp) * 0.9   # planes have the files' byte volume
        for nbytes in (plan["llm_bytes"], plan["vision_bytes"]):
            buf = torch.zeros(nbytes, dtype=torch.uint8)
            if rank == 0:
                buf.copy_(torch.from_numpy(np.random.default_rng(nbytes % 1000).integers(0, 256, nbytes, dtype=np.uint8)))
            D.broadcast_arena(buf, src=0, chunk_bytes=1 << 20)
            sums = D.gather_objects(int(buf.to(torch.int64).sum()), world)
            assert sums[0] == sums[1]
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver runs `--gpus 1`): bench.py itself must fan out to 2 ranks under torch.distributed.run.  There is
    no GPU here, so every rank stops at its device check -- after announcing itself; the launcher's failure is propagated as a non-zero exit code."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""          # also on a GPU box: no device for the ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "tiny", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    err = r.stderr
    assert "launching 2 ranks" in err and "--nproc-per-node=2" in err, err[-2000:]
    assert "rank 0 of 2 started (launcher: self)" in err and "rank 1 of 2 started (launcher: self)" in err, err[-2000:]
    assert r.returncode != 0 and "no HIP device" in err, (r.returncode, err[-2000:])
    assert r.stdout.strip() == ""            # no JSON line from a run that measured nothing


def test_bench_world_size_must_match_gpus_flag():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "tiny"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
