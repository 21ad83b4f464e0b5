#!/usr/bin/env python3
"""Persistent decode engine vs the launch-per-op step on one model, in ONE process (minigpt4_amd_set_engine): device-resident greedy loop ms / token, C-ABI tokens/s,
bit-identity of the logits, and the per-site table of both steps.  usage: tools/ab_engine.py [--config 13b] [--steps 64] [--sites]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load_package()
import numpy as np
import bench
from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G


def main():
    a = sys.argv[1:]
    config = a[a.index("--config") + 1] if "--config" in a else "13b"
    steps = int(a[a.index("--steps") + 1]) if "--steps" in a else 64
    lib = ML.load_library()
    vp, lp, vcfg, lcfg = bench.make_models(config, 0, 1, lambda: None)
    ctx = lib.minigpt4_model_load(vp, lp, verbosity=1, seed=1337, n_ctx=2048, n_batch=512)
    emb = lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(G.synth_image(42)))
    res = {"config": config, "engine_active": bool(lib.amd_engine_active(ctx))}
    logits = {}
    for name, on in (("engine", True), ("launches", False), ("engine_again", True)):
        lib.amd_set_engine(ctx, on)
        lib.minigpt4_reset_chat(ctx)
        lib.minigpt4_system_prompt(ctx); lib.minigpt4_begin_chat_image(ctx, emb, bench.PROMPT)
        for _ in range(8): lib.minigpt4_end_chat_image(ctx, temp=0.0)
        lib.library.minigpt4_amd_sync(ctx.ptr)
        t0 = time.perf_counter()
        toks = [lib.minigpt4_end_chat_image(ctx, temp=0.0) for _ in range(steps)]
        lib.library.minigpt4_amd_sync(ctx.ptr)
        dt = time.perf_counter() - t0
        logits[name] = lib.amd_logits(ctx).copy()
        ids, loop_ms = lib.amd_decode_loop(ctx, 33)
        r = {"tok_s": steps / dt, "ms_tok": dt * 1e3 / steps, "loop_ms_tok": loop_ms / 32.0, "ids_head": [int(i) for i in ids[:6]]}
        if "--sites" in a:
            prof = lib.amd_profile_sites(ctx, 4)
            agg = {}
            for s in prof["sites"]:
                k = agg.setdefault(s["site"], [0.0, 0.0]); k[0] += s["calls_per_step"]; k[1] += s["calls_per_step"] * s["avg_us"]
            r["sites"] = {k: {"calls": round(v[0], 2), "us_per_token": round(v[1], 1), "avg_us": round(v[1] / max(v[0], 1e-9), 2)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
        res[name] = r
    res["logits_bit_identical"] = bool(np.array_equal(logits["engine"], logits["launches"]) and np.array_equal(logits["engine"], logits["engine_again"]))
    res["max_abs_delta"] = float(np.abs(logits["engine"] - logits["launches"]).max())
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
