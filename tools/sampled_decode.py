#!/usr/bin/env python3
"""Decode rate through the C ABI with the reference's DEFAULT sampling parameters (temp 0.8, top_k 40, top_p 0.9: the sampler chain runs on the host on a copy of the logits)
next to the greedy rate the bench line reports.  tools/sampled_decode.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load_package()
import bench
from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = ML.load_library()
vp, lp, vcfg, lcfg = bench.make_models("13b", 0, 1, lambda: None)
ctx = lib.minigpt4_model_load(vp, lp, verbosity=0, seed=1337, n_ctx=2048, n_batch=512)
emb = lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(G.synth_image(42)))
for name, kw in (("greedy", dict(temp=0.0)), ("default sampling", dict()), ("mirostat 2", dict(mirostat=2)), ("greedy again", dict(temp=0.0)), ("default sampling again", dict())):
    lib.minigpt4_reset_chat(ctx)
    lib.minigpt4_system_prompt(ctx)
    lib.minigpt4_begin_chat_image(ctx, emb, "what is the text in the picture?")
    for _ in range(8):
        lib.minigpt4_end_chat_image(ctx, **kw)
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.minigpt4_end_chat_image(ctx, **kw)
    dt = time.perf_counter() - t0
    print(f"{name:24s} {steps / dt:7.1f} tok/s   {dt / steps * 1e3:.3f} ms/token", flush=True)
