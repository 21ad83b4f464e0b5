#!/usr/bin/env python3
"""In-kernel timeline of ONE persistent decode-engine launch (layer --layer of the 13B file) from a -DMG4_TIMELINE build:
  make -C minigpt4.cpp_amd/csrc variants && MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so python tools/timeline_engine.py [--config 13b] [--layer 20]
Per stage: min / median / max over the workgroups, microseconds after the earliest consumer entry.  Stamps: csrc/decode_engine.hip (EG_TL)."""
import ctypes, os, sys
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
    layer = int(a[a.index("--layer") + 1]) if "--layer" in a else 20
    lib = ML.load_library()
    L = lib.library
    L.minigpt4_amd_timeline_engine.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
    vp, lp, vcfg, lcfg = bench.make_models(config, 0, 1, lambda: None)
    ctx = lib.minigpt4_model_load(vp, lp, verbosity=0, seed=1337, n_ctx=2048, n_batch=512)
    emb = lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(G.synth_image(42)))
    lib.minigpt4_system_prompt(ctx); lib.minigpt4_begin_chat_image(ctx, emb, bench.PROMPT)
    assert L.minigpt4_amd_timeline_engine(None, 0, layer) == 0, "library built without MG4_TIMELINE?"
    for _ in range(6): lib.minigpt4_end_chat_image(ctx, temp=0.0)
    _, loop_ms = lib.amd_decode_loop(ctx, 17)
    L.minigpt4_amd_sync(ctx.ptr)
    buf = (ctypes.c_ulonglong * (520 * 64))()
    n = L.minigpt4_amd_timeline_engine(buf, 520, -1)
    print(f"{config}: device loop {loop_ms / 16:.3f} ms per token (timeline build); stamps of layer {layer}, {n} workgroup slots")
    if n <= 0:
        return
    fl = np.frombuffer(buf, np.uint64)[512 * 64:512 * 64 + 512].reshape(4, 128).astype(np.float64)
    t = np.frombuffer(buf, np.uint64)[:512 * 64].reshape(512, 64).astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    n_ops = 6
    names = ["wo", "w1|w3", "w2", "op3", "op4", "op5"]

    def show(label, col, rel=True):
        v = t[:, col]
        v = v[v > 0]
        if not len(v):
            return
        v = (v - t0) / 100.0 if rel else v / 100.0
        print(f"  {label:46s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f} us")
    show("consumer entry", 0)
    show("loader entry", 32)
    for o in range(n_ops):
        show(f"[{names[o]}] loader: first fill issued", 33 + o)
        show(f"[{names[o]}] loader: last fill issued", 44 + o)
        show(f"[{names[o]}] loader: time waiting for free slots", 56 + o, rel=False)
        show(f"[{names[o]}] loader: time in vmcnt waits (landing)", 26 + o, rel=False)
        show(f"[{names[o]}] loader: time issuing the DMA statements", 50 + o, rel=False)
        show(f"[{names[o]}] preparation begins (wave 0)", 1 + 4 * o)
        show(f"[{names[o]}] row gathered (wave 0)", 2 + 4 * o)
        show(f"[{names[o]}] image complete", 3 + 4 * o)
        show(f"[{names[o]}] last consumer wave done", 4 + 4 * o)
    show("loader: everything landed", 43)
    if "--fills" in a:
        print("  workgroup 7, per fill (us after its first issue): issued | published | consumer starts | consumer done")
        f0 = fl[0][fl[0] > 0].min() if (fl[0] > 0).any() else 0
        for f in range(128):
            if fl[0][f] > 0:
                print(f"   fill {f:3d}: " + "  ".join(f"{(fl[k][f] - f0) / 100.0:8.2f}" if fl[k][f] > 0 else "       -" for k in range(4)))


if __name__ == "__main__":
    main()
