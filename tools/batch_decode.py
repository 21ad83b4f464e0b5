#!/usr/bin/env python3
"""Batched decode on the 13B headline file: B conversations, one weight pass per step.  usage: tools/batch_decode.py B [steps]
Prints one JSON line (tokens/s per GPU, ms per step).  Run under `rocprofv3 --kernel-trace --stats` with MINIGPT4_NO_GRAPH=1 for the per-kernel table."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    B = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    import _pkg
    _pkg.load_package()
    import bench
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    lib = ML.load_library()
    vp, lp, vcfg, lcfg = bench.make_models("13b", 0, 1, lambda: None)
    ctx = lib.minigpt4_model_load(vp, lp, verbosity=0, seed=1337, n_ctx=2048, n_batch=512)
    emb = lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(G.synth_image(42)))
    lib.amd_set_conversations(ctx, B)
    for sl in range(B):
        lib.amd_select_conversation(ctx, sl)
        lib.minigpt4_system_prompt(ctx)
        lib.minigpt4_begin_chat_image(ctx, emb, bench.PROMPT)
    lib.amd_select_conversation(ctx, 0)
    for _ in range(4):
        lib.amd_end_chat_batch(ctx, list(range(B)), temp=0.0)
    lib.library.minigpt4_amd_sync(ctx.ptr)
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.amd_end_chat_batch(ctx, list(range(B)), temp=0.0)
    lib.library.minigpt4_amd_sync(ctx.ptr)
    dt = time.perf_counter() - t0
    print(json.dumps({"B": B, "steps": steps, "tokens_per_s": B * steps / dt, "ms_per_step": dt * 1e3 / steps,
                      "env": {k: v for k, v in os.environ.items() if k.startswith("MINIGPT4_")}}))


if __name__ == "__main__":
    main()
