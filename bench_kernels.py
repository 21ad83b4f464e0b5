#!/usr/bin/env python3
"""Kernel micro-benchmarks (decode mat-vec variants) on synthetic planes; prints GB/s per shape.  GPU only."""
import ctypes
import sys

import _pkg

_pkg.load_package()
from minigpt4_cpp_amd import minigpt4_library as ML, quants as Q  # noqa: E402

lib = ML.load_library()
L = lib.library
L.minigpt4_amd_bench_matvec.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]


def run(t, rows, cols, n_mat, variant, waves, iters=200):
    per = Q.nbytes(Q.NAME_TO_TYPE[t], rows * cols) * n_mat
    n_sets = max(2, int(700e6 // per) + 1)
    us, by = ctypes.c_float(), ctypes.c_double()
    rc = L.minigpt4_amd_bench_matvec(Q.NAME_TO_TYPE[t], rows, cols, n_mat, variant, iters, n_sets, waves, ctypes.byref(us), ctypes.byref(by))
    assert rc == 0, rc
    return us.value, by.value / us.value / 1e3   # us, GB/s


if __name__ == "__main__":
    shapes = [("q5_k", 5120, 5120, 1), ("q5_k", 5120, 5120, 3), ("q5_k", 13824, 5120, 2), ("q5_k", 5120, 13824, 1), ("q6_k", 5120, 13824, 1), ("q6_k", 32000, 5120, 1),
              ("q4_0", 4096, 4096, 3), ("q4_0", 11008, 4096, 2), ("q4_0", 4096, 11008, 1)]
    waves_list = [int(w) for w in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8]
    for t, r, c, n in shapes:
        us0, g0 = run(t, r, c, n, 0, 8)
        line = f"{t:5s} {r:6d}x{c:6d} x{n}  v1: {us0:7.1f} us {g0:7.0f} GB/s |"
        for w in waves_list:
            us1, g1 = run(t, r, c, n, 1, w)
            line += f" v2(w{w}): {us1:7.1f} us {g1:6.0f} GB/s |"
        print(line, flush=True)
