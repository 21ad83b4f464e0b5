#!/usr/bin/env python3
"""Batched decode, per matrix set of the 13B layer: microseconds per launch of the v_dot4 multi-row mat-vec (k_matvec_tn: minigpt4_amd_bench_matvec variants 12..14) and of the
row-interleaved MFMA mat-vec (k_matvec_ri: minigpt4_amd_bench_matvec_ri) at B = 2, 3, 4, rotating weight sets.  tools/batched_shapes_bench.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load_package()
from minigpt4_cpp_amd import minigpt4_library as ML, quants as Q
L = ML.load_library().library
F, D, I = ctypes.c_float, ctypes.c_double, ctypes.c_int
L.minigpt4_amd_bench_matvec.argtypes = [I] * 8 + [ctypes.POINTER(F), ctypes.POINTER(D)]
L.minigpt4_amd_bench_matvec_ri.argtypes = [I] * 7 + [ctypes.POINTER(F)]
SHAPES = [("qkv q5_k", "q5_k", 5120, 5120, 3), ("qk q5_k", "q5_k", 5120, 5120, 2), ("v q6_k", "q6_k", 5120, 5120, 1), ("wo q5_k", "q5_k", 5120, 5120, 1), ("w1w3 q5_k", "q5_k", 13824, 5120, 2),
          ("w2 q5_k", "q5_k", 5120, 13824, 1), ("w2 q6_k", "q6_k", 5120, 13824, 1), ("output q6_k", "q6_k", 32000, 5120, 1)]
us, by = F(), D()
for name, t, rows, cols, n_mat in SHAPES:
    tt = Q.NAME_TO_TYPE[t]
    mb = Q.nbytes(tt, rows * cols) * n_mat / 1e6
    sets = max(2, int(600 / mb))
    row = [f"{name:12s} {mb:6.1f} MB"]
    for B in (2, 3, 4):
        rc1 = L.minigpt4_amd_bench_matvec(tt, rows, cols, n_mat, 10 + B, 60, sets, 0, ctypes.byref(us), ctypes.byref(by)); a = us.value
        rc2 = L.minigpt4_amd_bench_matvec_ri(tt, rows, cols, n_mat, B, 60, sets, ctypes.byref(us)); b = us.value
        row.append(f"B={B}: dot4 {a:6.1f}  mfma {b:6.1f}" if rc1 == 0 and rc2 == 0 else f"B={B}: rc {rc1} {rc2}")
    print("   ".join(row), flush=True)
