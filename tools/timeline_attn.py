#!/usr/bin/env python3
"""Micro-benchmark (+ in-kernel timeline in a diagnostic build, see tools/timeline_gemm.py) of the ViT / Q-Former attention kernel k_attn_vit.

  [MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so] python tools/timeline_attn.py [heads hd nq nk]...

Stamps (thread 0 of every workgroup): 0 entry, 1 Q / K requested, 2 score MFMAs issued, 3 loads + table DMA landed, 4 V requested, 5 row maxima exchanged,
6 exp + partial sums done, 7 sums exchanged, 8 P.V MFMAs issued, 9 partial outputs exchanged, 10 stores issued, 11 stores drained."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

_pkg.load_package()
import numpy as np  # noqa: E402
from minigpt4_cpp_amd import minigpt4_library as ML  # noqa: E402

NAMES = ["entry", "Q / K requested", "score MFMAs issued", "loads + table landed", "V requested", "maxima exchanged", "exp + partial sums", "sums exchanged",
         "P.V MFMAs issued", "partials exchanged", "stores issued", "stores drained"]


def main():
    lib = ML.load_library()
    L = lib.library
    L.minigpt4_amd_bench_attn_f32.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
    L.minigpt4_amd_timeline_vision.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    a = [int(x) for x in sys.argv[1:]]
    cases = [tuple(a[i:i + 4]) for i in range(0, len(a) - 3, 4)] or [(16, 88, 257, 257), (12, 64, 32, 257), (12, 64, 32, 32)]
    for heads, hd, nq, nk in cases:
        us = ctypes.c_float()
        rc = L.minigpt4_amd_bench_attn_f32(heads, hd, nq, nk, 200, ctypes.byref(us))
        assert rc == 0, rc
        print(f"heads {heads} hd {hd} nq {nq} nk {nk}: {us.value:7.2f} us per launch", flush=True)
        buf = (ctypes.c_ulonglong * (1024 * 32))()
        n = L.minigpt4_amd_timeline_vision(buf, 1024)
        if n <= 0:
            continue
        t = np.frombuffer(buf, np.uint64).reshape(1024, 32).astype(np.float64)
        t = t[t[:, 0] > 0]
        t = t[t[:, 0] > t[:, 0].max() - 1e4]
        t = t[t[:, 11] >= t[:, 0]]
        t0 = t[:, 0].min()
        prev = None
        for i, name in enumerate(NAMES):
            v = (t[:, i] - t0) / 100.0
            d = "" if prev is None else f"   (+{np.median(v) - prev:5.2f})"
            prev = np.median(v)
            print(f"  {name:22s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f} us{d}   ({len(v)} workgroups)")


if __name__ == "__main__":
    main()
