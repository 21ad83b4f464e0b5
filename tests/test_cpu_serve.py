"""Host logic of the request-level server (minigpt4.cpp_amd/serve.py) without a GPU: wave planning, answer merging, and the world_size-2 gloo path of
`serve()` with the replica replaced by a stand-in (the real replica is exercised on the GPU by tests/test_gpu_serve.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_waves_and_merge():
    from minigpt4_cpp_amd import serve as S
    assert S.plan_waves(0, 4) == []
    assert S.plan_waves(3, 4) == [[0, 1, 2]]
    assert S.plan_waves(9, 4) == [[0, 1, 2, 3], [4, 5, 6, 7], [8]]
    assert S.merge_answers(4, [{0: "a", 2: "c"}, {1: "b", 3: "d"}]) == ["a", "b", "c", "d"]
    with pytest.raises(AssertionError):
        S.merge_answers(3, [{0: "a"}, {2: "c"}])


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load_package()
    import torch.distributed as dist
    from minigpt4_cpp_amd import serve as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class FakeReplica:                                   # stands in for ReplicaServer: answers identify the rank and the request
        def __init__(self, vp, lp, conversations=4, **kw):
            self.c = conversations

        def run(self, requests, **kw):
            return [f"r{rank}:{r.prompt}:{len(S.plan_waves(len(requests), self.c))}" for r in requests]

        def close(self):
            pass
    S.ReplicaServer = FakeReplica
    try:
        reqs = [S.Request(image=b"", prompt=f"p{i}") for i in range(7)]
        out = S.serve(reqs, "v.bin", "l.bin", conversations=2)
        if rank == 0:
            # rank 0 served 0, 2, 4, 6 (2 waves of 2), rank 1 served 1, 3, 5 (2 waves)
            assert out == [f"r{i % 2}:p{i}:2" for i in range(7)], out
        else:
            assert out is None
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_serve_shards_and_gathers_on_two_gloo_ranks(tmp_path):
    import torch.multiprocessing as mp
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
