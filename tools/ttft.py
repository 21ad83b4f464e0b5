#!/usr/bin/env python3
"""Time to first token of one image turn through the reference's call sequence (13B Q5_K_M): wall milliseconds of every C-ABI call for three turns of a fresh context (the first
one pays code-object loads and the hipGraph capture of the decode step).  tools/ttft.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load_package()
import bench
from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G

lib = ML.load_library()
vp, lp, vcfg, lcfg = bench.make_models("13b", 0, 1, lambda: None)
t0 = time.perf_counter()
ctx = lib.minigpt4_model_load(vp, lp, verbosity=0, seed=1337, n_ctx=2048, n_batch=512)
print(f"model_load {1e3 * (time.perf_counter() - t0):8.1f} ms")
img = ML.array_to_image_struct(G.synth_image(42))
# the native image path in front of it (reference: OpenCV imread + PillowResize): a 12-megapixel photograph-like JPEG from disk
try:
    import io, numpy as np
    from PIL import Image
    h, w = 3000, 4000
    yy, xx = np.mgrid[0:h, 0:w]
    ph = np.stack([xx * 255 // w, yy * 255 // h, (xx + yy) * 255 // (w + h)], -1).astype(np.float32) + np.random.default_rng(0).normal(0, 8, (h, w, 3))
    path = "/dev/shm/mg4_bench/photo_12mp.jpg"
    Image.fromarray(np.clip(ph, 0, 255).astype(np.uint8)).save(path, "JPEG", quality=90)
    for rep in range(3):
        t = time.perf_counter(); raw = lib.minigpt4_image_load_from_file(ctx, path); t_load = 1e3 * (time.perf_counter() - t)
        t = time.perf_counter(); pre = lib.minigpt4_preprocess_image(ctx, raw); t_pre = 1e3 * (time.perf_counter() - t)
        print(f"12 MP JPEG ({os.path.getsize(path) / 1e6:.1f} MB): minigpt4_image_load_from_file {t_load:.1f} ms, minigpt4_preprocess_image {t_pre:.1f} ms")
        lib.minigpt4_free_image(raw)
        if rep < 2:
            lib.minigpt4_free_image(pre)
    img = pre
except ImportError:
    pass
for turn in range(3):
    T = {}
    def timed(name, f, *a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[name] = 1e3 * (time.perf_counter() - t); return r     # (no sync between the calls: a sync would flush the queued prompt rows)
    timed("reset_chat", lib.minigpt4_reset_chat, ctx)
    lib.library.minigpt4_amd_sync(ctx.ptr)
    t_all = time.perf_counter()
    emb = timed("encode_image", lib.minigpt4_encode_image, ctx, img)
    timed("system_prompt", lib.minigpt4_system_prompt, ctx)
    timed("begin_chat_image", lib.minigpt4_begin_chat_image, ctx, emb, "what is the text in the picture?")
    timed("end_chat_image #1", lib.minigpt4_end_chat_image, ctx, temp=0.0)
    timed("end_chat_image #2", lib.minigpt4_end_chat_image, ctx, temp=0.0)
    timed("end_chat_image #3", lib.minigpt4_end_chat_image, ctx, temp=0.0)
    lib.minigpt4_free_embedding(emb)
    ttft = T["encode_image"] + T["system_prompt"] + T["begin_chat_image"] + T["end_chat_image #1"]     # the prompt rows are queued by the calls and evaluated in ONE pass when the first token is asked for
    print(f"turn {turn}: " + "  ".join(f"{k} {v:.2f}" for k, v in T.items()) + f"   | image -> first token {ttft:.2f} ms")
