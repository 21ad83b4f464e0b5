"""One rank of the two-rank serve() rehearsal (tests/test_gpu_serve.py): RANK / WORLD_SIZE / MASTER_* from the environment, gloo collectives, every rank on GPU 0
(RCCL refuses two ranks of one communicator on one device -- this exercises the request sharding, the receive-mode load with the arena broadcast out of device
memory and the gather of the answers on real hardware, not the transport)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402


def main():
    vp, lp, out_path = sys.argv[1:4]
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)                     # torch's HIP runtime first (conftest.py explains the order)
    dist.init_process_group(backend="gloo")
    _pkg.load_package()
    from minigpt4_cpp_amd import modelgen as G, serve as S
    reqs = [S.Request(G.synth_image(3 + i), p, n) for i, (p, n) in enumerate([("what is the text in the picture?", 6), ("describe it", 5), ("colour?", 7), ("how many?", 4), ("where?", 6)])]
    try:
        out = S.serve(reqs, vp, lp, conversations=2, n_ctx=512, n_batch=64, device=torch.device("cuda", 0), temp=0.0, ignore_eos=True)
        if int(os.environ["RANK"]) == 0:
            json.dump(out, open(out_path, "w"))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
