#!/usr/bin/env python3
"""Micro-benchmark (+ in-kernel timeline in a -DMG4_TIMELINE build) of the prompt-row attention k_attn_prefill_h.
  [MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so] python tools/timeline_attn_prefill.py [n_head hd N n_past]...
Stamps (thread 0 of every workgroup): 0 entry, 1 scores in LDS, 2 softmax done, 3 P.V done, 4 stored."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

_pkg.load_package()
import numpy as np  # noqa: E402
from minigpt4_cpp_amd import minigpt4_library as ML  # noqa: E402

NAMES = ["entry", "scores in LDS", "softmax done", "P.V done", "stored"]


def main():
    lib = ML.load_library()
    L = lib.library
    L.minigpt4_amd_bench_attn_prefill.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
    L.minigpt4_amd_timeline_attn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    a = [int(x) for x in sys.argv[1:]]
    cases = [tuple(a[i:i + 4]) for i in range(0, len(a) - 3, 4)] or [(40, 128, 512, 0), (40, 128, 142, 0), (40, 128, 64, 448)]
    for n_head, hd, N, n_past in cases:
        us = ctypes.c_float()
        rc = L.minigpt4_amd_bench_attn_prefill(n_head, hd, N, n_past, 100, ctypes.byref(us))
        assert rc == 0, rc
        print(f"heads {n_head} hd {hd} rows {N} past {n_past}: {us.value:7.2f} us per launch", flush=True)
        buf = (ctypes.c_ulonglong * (2048 * 8))()
        n = L.minigpt4_amd_timeline_attn(buf, 2048)
        if n <= 0:
            continue
        t = np.frombuffer(buf, np.uint64).reshape(2048, 8).astype(np.float64)
        t = t[t[:, 0] > 0]
        t = t[t[:, 0] > t[:, 0].max() - 2e4]
        t = t[t[:, 4] >= t[:, 0]]
        t0 = t[:, 0].min()
        dur = (t[:, 4] - t[:, 0]) / 100.0
        print(f"  {len(t)} workgroups; per-workgroup duration min {dur.min():.2f} median {np.median(dur):.2f} max {dur.max():.2f} us; last store at {(t[:, 4].max() - t0) / 100:.2f} us")
        long = t[dur >= np.percentile(dur, 90)]                                   # the longest tiles (most keys)
        for i, name in enumerate(NAMES):
            v = (long[:, i] - long[:, 0]) / 100.0
            print(f"  longest 10 %: {name:16s} median +{np.median(v):6.2f} us after entry")
        starts = (t[:, 0] - t0) / 100.0
        print(f"  entry times: median {np.median(starts):.2f} max {starts.max():.2f} us (workgroups wait for a CU)")


if __name__ == "__main__":
    main()
