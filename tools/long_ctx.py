#!/usr/bin/env python3
"""Decode rate of the 13B Q5_K_M file at long contexts (device-resident greedy loop, hipEvents), key-split attention vs the one-workgroup-per-head kernel.
usage: tools/long_ctx.py [contexts ...]   (default 256 512 1024 2040)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load_package()
import numpy as np
from minigpt4_cpp_amd import minigpt4_library as ML
import headline as H
ctxs = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2040]
vp, lp = H.headline_files(os.environ.get("MG4_CONFIG", "13b"))
lib = ML.load_library()
rng = np.random.default_rng(3)
for thr in os.environ.get("MG4_THRS", "512,0,1").split(","):
    os.environ["MINIGPT4_ATTN_SPLIT_T"] = thr
    ctx = lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=2048, n_batch=512)
    os.environ.pop("MINIGPT4_ATTN_SPLIT_T")
    have = 0
    out = []
    for C in ctxs:
        steps = 24
        need = C - steps - have
        toks = [1] + [int(t) for t in rng.integers(3, 30000, need - 1)] if have == 0 else [int(t) for t in rng.integers(3, 30000, need)]
        for i in range(0, len(toks), 512):
            lib.amd_eval_tokens(ctx, toks[i:i + 512])
        lib.amd_logits(ctx)
        ids, ms = lib.amd_decode_loop(ctx, steps + 1)
        have = C + 1
        out.append((C, ms / steps))
    print(f"MINIGPT4_ATTN_SPLIT_T={thr:>3s}: " + "  ".join(f"ctx {c}: {m:.3f} ms = {1e3 / m:.1f} tok/s" for c, m in out), flush=True)
    lib.minigpt4_free(ctx)
