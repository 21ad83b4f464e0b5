#!/usr/bin/env python3
"""In-kernel timeline of ONE decode mat-vec launch (diagnostic build: `make -C minigpt4.cpp_amd/csrc EXTRA=-DMG4_TIMELINE OUT=../libminigpt4_tl.so OBJ=build_tl`).

  MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so python tools/timeline.py [type rows cols n_mat variant]...

variant 1 = activation row prepared by its own launch, 2 = rms-norm prologue inside the mat-vec (the decode's qkv / w1|w3 flavour), 3 = the batched decode's
multi-row launch (k_matvec_tn, 4 prepared rows), 4 = the same with 2 rows, 5 / 6 = 2 / 4 rows prepared inside the launch (rms-norm prologue).  Thread 0 of every workgroup
stamps the 100 MHz constant clock at entry / first weight request / row ready / first group done / last group done / results stored; printed per stage as the
min / median / max over workgroups, in microseconds after the EARLIEST workgroup's entry, next to the hipEvent time of the launch (which additionally holds the
dispatch + end-of-kernel cost).  GPU only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

_pkg.load_package()
import numpy as np  # noqa: E402
from minigpt4_cpp_amd import minigpt4_library as ML, quants as Q  # noqa: E402

STAGES = ["entry", "first weight request", "activation row ready", "first row group done", "last row group done", "results stored", "LAST wave of the workgroup done", "LAST wave of the workgroup entered"]


def main():
    lib = ML.load_library()
    L = lib.library
    L.minigpt4_amd_bench_matvec.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
    L.minigpt4_amd_timeline.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    args = sys.argv[1:]
    cases = [(args[i], int(args[i + 1]), int(args[i + 2]), int(args[i + 3]), int(args[i + 4])) for i in range(0, len(args) - 4, 5)] or [
        ("q5_k", 5120, 5120, 3, 2), ("q5_k", 5120, 5120, 1, 2), ("q5_k", 13824, 5120, 2, 2), ("q5_k", 5120, 13824, 1, 1), ("q6_k", 32000, 5120, 1, 2)]
    for t, rows, cols, n_mat, variant in cases:
        per = Q.nbytes(Q.NAME_TO_TYPE[t], rows * cols) * n_mat
        us, by = ctypes.c_float(), ctypes.c_double()
        n_sets = int(os.environ.get("TL_SETS", "0")) or max(2, int(700e6 // per) + 1)      # TL_SETS=1: the same weights every launch (Infinity-Cache resident when they fit)
        rc = L.minigpt4_amd_bench_matvec(Q.NAME_TO_TYPE[t], rows, cols, n_mat, variant, 50, n_sets, 0, ctypes.byref(us), ctypes.byref(by))
        assert rc == 0, rc
        buf = (ctypes.c_ulonglong * (1024 * 8))()
        n = L.minigpt4_amd_timeline(buf, 1024)
        print(f"{t} {rows}x{cols} x{n_mat} variant {variant}: {us.value:.1f} us per launch (hipEvents, back-to-back), {by.value / us.value / 1e3:.0f} GB/s")
        if n <= 0:
            print("  (library built without MG4_TIMELINE)" if n == 0 else "  timeline read failed")
            continue
        a = np.frombuffer(buf, np.uint64).reshape(1024, 8).astype(np.float64)
        a = a[a[:, 0] > 0]                                   # workgroups that ran in the last launch (all slots are rewritten by every launch of <= 1024 workgroups)
        a = a[a[:, 5] >= a[:, 0]]
        a = a[a[:, 0] > a[:, 0].max() - 1e4]                # drop stale slots of an earlier, wider launch (older than 100 us)
        t0 = a[:, 0].min()
        for i, name in enumerate(STAGES):
            v = (a[:, i] - t0) / 100.0                       # 100 MHz -> us
            print(f"  {name:22s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f} us   ({len(v)} workgroups)")


if __name__ == "__main__":
    main()
