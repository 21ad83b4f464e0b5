#!/usr/bin/env python3
"""Micro-benchmark (+ in-kernel timeline in a diagnostic build) of the image path's fp16 GEMM.

  python tools/timeline_gemm.py [M N K flags variant]...                                        # us per launch (hipEvents, back to back, weights cycled through HBM)
  MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so python tools/timeline_gemm.py ...         # + per-workgroup clock stamps of the last launch
     (make -C minigpt4.cpp_amd/csrc EXTRA=-DMG4_TIMELINE OUT=../libminigpt4_tl.so OBJ=build_tl)

Stamps of k_gemm_f16 (thread 0 of every workgroup, 100 MHz clock): 0 entry, 1 three stages requested, 2 + k barrier of k tile k passed, 29 loop done, 30 epilogue
stores issued, 31 stores drained.  Printed as min / median / max over the workgroups, in microseconds after the EARLIEST workgroup's entry.  GPU only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

_pkg.load_package()
import numpy as np  # noqa: E402
from minigpt4_cpp_amd import minigpt4_library as ML  # noqa: E402

VIT = [(257, 4224, 1408, 0, 0), (257, 1408, 1408, 2, 0), (257, 6144, 1408, 1 | 4, 0), (257, 1408, 6144, 0, 2 | (4 << 8))]


def main():
    lib = ML.load_library()
    L = lib.library
    L.minigpt4_amd_bench_gemm_f16.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
    L.minigpt4_amd_timeline_vision.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    if os.environ.get("MG4_SPLITK_XCD") is not None:     # 0: split-K slices in grid.z (the rounds 3-5 mapping), A/B against the default
        L.minigpt4_amd_test_set_splitk_xcd.argtypes = [ctypes.c_int]
        L.minigpt4_amd_test_set_splitk_xcd.restype = None
        L.minigpt4_amd_test_set_splitk_xcd(int(os.environ["MG4_SPLITK_XCD"]))
    a = [int(x) for x in sys.argv[1:]]
    cases = [tuple(a[i:i + 5]) for i in range(0, len(a) - 4, 5)] or VIT
    for M, N, K, flags, variant in cases:
        us = ctypes.c_float()
        n_sets = max(2, int(600e6 // (N * K * 2)) + 1)
        rc = L.minigpt4_amd_bench_gemm_f16(M, N, K, flags, variant, 200, n_sets, ctypes.byref(us))
        assert rc == 0, (rc, lib.last_error() if hasattr(lib, "last_error") else "")
        fl = 2.0 * M * N * K
        print(f"M {M} N {N} K {K} flags {flags} variant {variant & 255} slices {(variant >> 8) & 255} arm {variant >> 16}: {us.value:7.2f} us per launch  {fl / us.value / 1e6:7.1f} TFLOP/s  weights {N * K * 2 / us.value / 1e6:6.2f} TB/s", flush=True)
        buf = (ctypes.c_ulonglong * (1024 * 32))()
        n = L.minigpt4_amd_timeline_vision(buf, 1024)
        if n <= 0:
            continue
        t = np.frombuffer(buf, np.uint64).reshape(1024, 32).astype(np.float64)
        t = t[t[:, 0] > 0]
        t = t[t[:, 0] > t[:, 0].max() - 1e4]                 # stale slots of an earlier, wider launch
        done = t[t[:, 31] >= t[:, 0]]                         # workgroups that own a tile (the others leave after stamp 0)
        t0 = t[:, 0].min()
        print(f"  {len(t)} workgroups entered over {(t[:, 0].max() - t0) / 100:.2f} us; {len(done)} own a tile")
        names = {0: "entry", 1: "stages requested", 29: "loop done", 30: "stores issued", 31: "stores drained"}
        prev = None
        for i in range(32):
            col = done[:, i]
            if not (col >= done[:, 0]).all() or col.max() < t0:
                continue
            v = (col - t0) / 100.0
            d = "" if prev is None else f"   (+{np.median(v) - prev:5.2f})"
            prev = np.median(v)
            print(f"  {names.get(i, 'k tile %d barrier' % (i - 2)):22s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f} us{d}")


if __name__ == "__main__":
    main()
