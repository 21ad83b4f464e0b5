#!/usr/bin/env python3
"""Decode rate of the bit-exact mode (MINIGPT4_PARITY / minigpt4_amd_set_parity: every fp32 accumulation in the CPU oracle's order) next to the fast mode, same context.
usage: tools/parity_speed.py [--config 13b] [--steps 64]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load_package()
import bench
from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G


def main():
    a = sys.argv[1:]
    config = a[a.index("--config") + 1] if "--config" in a else "13b"
    steps = int(a[a.index("--steps") + 1]) if "--steps" in a else 64
    lib = ML.load_library()
    vp, lp, vcfg, lcfg = bench.make_models(config, 0, 1, lambda: None)
    ctx = lib.minigpt4_model_load(vp, lp, verbosity=0, seed=1337, n_ctx=2048, n_batch=512)
    emb = lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(G.synth_image(42)))
    out = {"config": config}
    for name, par in (("fast", False), ("parity", True)):
        lib.amd_set_parity(ctx, par)
        lib.minigpt4_reset_chat(ctx)
        lib.minigpt4_system_prompt(ctx); lib.minigpt4_begin_chat_image(ctx, emb, bench.PROMPT)
        lib.minigpt4_end_chat_image(ctx, temp=0.0); lib.library.minigpt4_amd_sync(ctx.ptr)
        lib.minigpt4_reset_chat(ctx)                                   # second pass of the same prompt: the image turn's wall time without first-use effects
        lib.minigpt4_system_prompt(ctx); lib.library.minigpt4_amd_sync(ctx.ptr)
        tp = time.perf_counter()
        lib.minigpt4_begin_chat_image(ctx, emb, bench.PROMPT); lib.minigpt4_end_chat_image(ctx, temp=0.0); lib.library.minigpt4_amd_sync(ctx.ptr)
        prefill_ms = (time.perf_counter() - tp) * 1e3                   # the prompt pass + one decode step
        for _ in range(3): lib.minigpt4_end_chat_image(ctx, temp=0.0)
        lib.library.minigpt4_amd_sync(ctx.ptr)
        t0 = time.perf_counter()
        for _ in range(steps): lib.minigpt4_end_chat_image(ctx, temp=0.0)
        lib.library.minigpt4_amd_sync(ctx.ptr)
        dt = time.perf_counter() - t0
        out[name] = {"tok_s": steps / dt, "ms_tok": dt * 1e3 / steps, "image_turn_plus_one_step_ms": prefill_ms}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
