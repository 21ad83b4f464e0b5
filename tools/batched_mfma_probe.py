#!/usr/bin/env python3
"""Batched decode on the matrix cores, measured (round 5): microseconds per launch of ONE 13B matrix set (w1|w3: 2 x 13824 rows, K = 5120, Q5_K) against B = 1..4 prepared rows
  * product kernel: k_matvec_tn / k_matvec_v2 (v_dot4_i32_i8, lane = weight unit)  -- minigpt4_amd_bench_matvec variants 1 / 12 / 13 / 14, rotating weight sets;
  * the int8-MFMA prompt kernel at one token tile (k_mmq2_q45k) and its fp16 form (k_mmqh_q45k) -- minigpt4_amd_bench_mmq generations 2 / 4;
  * v_mfma_i32_4x4x4_16B_i8 over row-interleaved synthetic planes (csrc/tn_mfma_probe.hip), validated against a scalar kernel at a small size first.
tools/batched_mfma_probe.py [rows cols]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load_package()
from minigpt4_cpp_amd import minigpt4_library as ML, quants as Q
L = ML.load_library().library
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (27648, 5120)
F, D, I = ctypes.c_float, ctypes.c_double, ctypes.c_int
L.minigpt4_amd_probe_tn_mfma.argtypes = [I] * 6 + [ctypes.POINTER(F)] * 2
L.minigpt4_amd_bench_matvec.argtypes = [I] * 8 + [ctypes.POINTER(F), ctypes.POINTER(D)]
L.minigpt4_amd_bench_mmq.argtypes = [I] * 8 + [ctypes.POINTER(F)]
us, rel, by = F(), F(), D()
if not L.minigpt4_amd_test_extras():
    raise SystemExit("libminigpt4_test.so has no closed-direction kernels: make -C minigpt4.cpp_amd/csrc test-extras")
for tn in (1, 2, 3, 4):                                   # correctness of the probe first: 512 rows x 2048, all token counts
    rc = L.minigpt4_amd_probe_tn_mfma(512, 2048, tn, 2, 1, 1, ctypes.byref(us), ctypes.byref(rel))
    print(f"check 512x2048 TN={tn}: rc {rc} relative difference to the scalar kernel {rel.value:.2e}", flush=True)
    assert rc == 0 and 0.0 <= rel.value < 1e-5, (rc, rel.value)
t = Q.NAME_TO_TYPE["q5_k"]
sets = 6                                                  # 6 x 97 MB: nothing survives in the 256 MB memory-side cache from one launch to the next
print(f"matrix set {rows} x {cols} Q5_K ({rows * cols / 256 * 176 / 1e6:.1f} MB), us per launch:")
for B in (1, 2, 3, 4):
    row = [f"B={B}"]
    variant = 1 if B == 1 else 10 + B
    rc = L.minigpt4_amd_bench_matvec(t, rows // 2, cols, 2, variant, 60, sets, 0, ctypes.byref(us), ctypes.byref(by))
    row.append(f"dot4 mat-vec (product) {us.value:7.1f}" if rc == 0 else f"dot4 mat-vec rc {rc}")
    for gen, name in ((2, "int8-MFMA prompt kernel, 1 tile"), (4, "fp16-MFMA prompt kernel, 1 tile")):
        rc = L.minigpt4_amd_bench_mmq(t, rows // 2, cols, 2, B, 30, 0, gen, ctypes.byref(us))
        row.append(f"{name} {us.value:7.1f}" if rc == 0 else f"{name} rc {rc}")
    rc = L.minigpt4_amd_probe_tn_mfma(rows, cols, B, 60, sets, 0, ctypes.byref(us), ctypes.byref(rel))
    row.append(f"mfma_4x4x4 row-interleaved {us.value:7.1f}" if rc == 0 else f"mfma_4x4x4 rc {rc}")
    print("   ".join(row), flush=True)
