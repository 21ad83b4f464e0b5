#!/usr/bin/env python3
"""Micro-benchmark of k_attn_vit's query tiles per workgroup (round 6): us per launch at the ViT's shapes for B images per launch and qt = 1 (rounds 2-5), 2, 3, 5, 0 (the
launcher's choice).  python tools/attn_qt_bench.py [B ...]   GPU only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

_pkg.load_package()
from minigpt4_cpp_amd import minigpt4_library as ML  # noqa: E402

lib = ML.load_library()
L = lib.library
L.minigpt4_amd_bench_attn_f32_b.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
for B in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    row = []
    for qt in (1, 2, 3, 4, 5, 0):
        us = ctypes.c_float()
        rc = L.minigpt4_amd_bench_attn_f32_b(16, 88, 257, 257, B, qt, 200, ctypes.byref(us))
        assert rc == 0, rc
        row.append(f"qt {qt}: {us.value:6.2f}")
    print(f"ViT attention, {B} image(s) per launch (us per launch): " + "   ".join(row), flush=True)
