#!/usr/bin/env python3
"""Issue cost of the vector instructions the decode mat-vec kernels are made of (gfx950): every wave issues independent instructions of one kind; ns per instruction
and wave at 1..3 waves per SIMD, and the ratio to v_and_b32.  usage: tools/probe_valu.py [iters]   (GPU only)"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg; _pkg.load_package()
from minigpt4_cpp_amd import minigpt4_library as ML
L = ML.load_library().library
L.minigpt4_amd_probe_valu.restype = ctypes.c_float
L.minigpt4_amd_probe_valu.argtypes = [ctypes.c_int] * 3
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
NAMES = ["v_and_b32", "v_dot4c_i32_i8", "v_mul_lo_u32", "v_mad_i32_i24", "v_fma_f32", "v_and_or_b32", "v_bfe_u32", "v_cvt_f32_i32", "v_dot4_i32_i8", "v_mad_u64_u32", "v_lshrrev_b32"]
base = {}
for w in (1, 2, 3):
    for op, name in enumerate(NAMES):
        ns = L.minigpt4_amd_probe_valu(op, w, iters)
        if op == 0: base[w] = ns
        print(f"waves/SIMD {w}  {name:16s} {ns:7.3f} ns per instruction and wave   SIMD-time per instruction {ns / w:6.3f} ns   x{ns / base[w]:5.2f} of v_and_b32", flush=True)
