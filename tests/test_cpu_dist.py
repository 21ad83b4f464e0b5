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


# ---- load_replica on two gloo ranks with a stand-in library (no GPU): the good path, and a REFUSED broadcast -- the JSON must say so, never a GB/s figure
class _FakeCtx:
    def __init__(self, mode, torch, np):
        self.mode, self.ptr, self.received = mode, id(self), False
        rng = np.random.default_rng(5)
        self.arenas = [torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8)) if mode == "full" else torch.zeros(n, dtype=torch.uint8) for n in (70001, 3001)]


class _FakeLib:
    """What dist.load_replica / serve.ReplicaServer use of MiniGPT4SharedLibrary."""
    def __init__(self, torch, np):
        self.torch, self.np, self.loads, self.freed, self.conversations = torch, np, [], 0, None
        outer = self

        class _Raw:
            @staticmethod
            def minigpt4_amd_weights_received(ptr):
                outer.received = True
                return 0
        self.library, self.received = _Raw, False

    def minigpt4_model_load(self, vp, lp, **kw):
        mode = "recv" if os.environ.get("MINIGPT4_LOAD") == "recv" else "full"
        self.loads.append(mode)
        return _FakeCtx(mode, self.torch, self.np)

    def amd_arena_plan(self, ctx):
        return {"llm_bytes": 70001, "vision_bytes": 3001, "llm_hash": 11, "vision_hash": 12}

    def amd_arena_checksum(self, ctx, which):
        return int(ctx.arenas[which].to(self.torch.int64).sum())

    def minigpt4_free(self, ctx):
        self.freed += 1

    def amd_set_conversations(self, ctx, n):
        self.conversations = n


def _replica_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load_package()
    import json
    import numpy as np
    import torch
    import torch.distributed as dist
    from minigpt4_cpp_amd import dist as D, serve as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D.arena_tensor = lambda lib, ctx, which, device: ctx.arenas[which]
        # 1) the broadcast works: rank 1 loaded headers only, received the arenas, finished the load; the report carries a rate
        lib = _FakeLib(torch, np)
        ctx, st = D.load_replica(lib, "v.bin", "l.bin", rank, world)
        assert lib.loads == (["full"] if rank == 0 else ["recv"]) and lib.received == (rank != 0) and lib.freed == 0
        assert st["bcast_ms"] is not None and "bcast_error" not in st and st["mode"] == ("full" if rank == 0 else "recv")
        rep = D.bcast_report(st)
        assert rep["bytes"] == 73002 and rep["GBps"] > 0 and "error" not in rep
        # 2) the broadcast is refused at its first call on every rank (what a launch-time RCCL refusal looks like)
        real_bcast = D.broadcast_arena

        def refuse(*a, **k):
            raise RuntimeError("ncclInternalError: injected")
        D.broadcast_arena = refuse
        lib = _FakeLib(torch, np)
        ctx, st = D.load_replica(lib, "v.bin", "l.bin", rank, world)
        assert lib.loads == (["full"] if rank == 0 else ["recv", "full"]) and lib.freed == (0 if rank == 0 else 1) and not lib.received
        assert st["bcast_ms"] is None and "injected" in st["bcast_error"] and (rank == 0 or st["mode"] == "full (broadcast refused)")
        rep = D.bcast_report(st)
        assert "injected" in rep["error"] and "GBps" not in rep and "ms" not in rep
        json.dumps(rep)
        # 3) the request-level entry point loads its replicas the same way (serve.ReplicaServer -> dist.load_replica)
        D.broadcast_arena = real_bcast
        lib = _FakeLib(torch, np)
        srv = S.ReplicaServer("v.bin", "l.bin", conversations=3, library=lib, rank=rank, world=world)
        assert lib.loads == (["full"] if rank == 0 else ["recv"]) and lib.conversations == 3 and srv.load_stats["bcast_ms"] is not None
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_load_replica_reports_a_refused_broadcast_as_an_error_not_as_a_rate(tmp_path):
    import torch.multiprocessing as mp
    port = 33500 + os.getpid() % 2000
    mp.spawn(_replica_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_bcast_report_single_rank_is_none():
    from minigpt4_cpp_amd import dist as D
    assert D.bcast_report({"mode": "full", "load_s": 1.0, "bcast_ms": None, "plan": {"llm_bytes": 1, "vision_bytes": 1}}) is None


# ---- the native (in-library) broadcast's environment contract (csrc/dist.cpp), host only
def _dist_env(lib, env):
    import ctypes
    keys = ("MINIGPT4_WORLD_SIZE", "MINIGPT4_RANK", "MINIGPT4_NCCL_ID_FILE", "MINIGPT4_DIST_TIMEOUT_S")
    old = {k: os.environ.pop(k, None) for k in keys}
    os.environ.update(env)
    try:
        f = lib.library.minigpt4_amd_dist_env
        f.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        w, r, idf, err = ctypes.c_int(), ctypes.c_int(), ctypes.create_string_buffer(512), ctypes.create_string_buffer(512)
        rc = f(ctypes.byref(w), ctypes.byref(r), idf, 512, err, 512)
        return rc, w.value, r.value, idf.value.decode(), err.value.decode()
    finally:
        for k in keys:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]


def test_native_broadcast_environment_contract(lib):
    assert _dist_env(lib, {}) == (0, 1, 0, "", "")                                            # nothing set: an ordinary single-GPU load
    assert _dist_env(lib, {"MINIGPT4_WORLD_SIZE": "8", "MINIGPT4_RANK": "3", "MINIGPT4_NCCL_ID_FILE": "/tmp/job17.id"})[:4] == (0, 8, 3, "/tmp/job17.id")
    assert _dist_env(lib, {"MINIGPT4_WORLD_SIZE": "1", "MINIGPT4_NCCL_ID_FILE": "/tmp/x.id"})[:4] == (0, 1, 0, "/tmp/x.id")   # a communicator of one rank (the single-GPU test)
    for env, needle in (({"MINIGPT4_WORLD_SIZE": "2", "MINIGPT4_RANK": "1"}, "MINIGPT4_NCCL_ID_FILE"),
                        ({"MINIGPT4_WORLD_SIZE": "2", "MINIGPT4_RANK": "2", "MINIGPT4_NCCL_ID_FILE": "/tmp/a"}, "MINIGPT4_RANK"),
                        ({"MINIGPT4_WORLD_SIZE": "0"}, "MINIGPT4_WORLD_SIZE"), ({"MINIGPT4_WORLD_SIZE": "eight"}, "not an integer"),
                        ({"MINIGPT4_RANK": "-1"}, "MINIGPT4_RANK"), ({"MINIGPT4_DIST_TIMEOUT_S": "0", "MINIGPT4_NCCL_ID_FILE": "/tmp/a"}, "TIMEOUT")):
        rc, _, _, _, err = _dist_env(lib, env)
        assert rc == 1 and needle in err, (env, err)
