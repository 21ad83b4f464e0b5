#!/usr/bin/env python3
"""LDS-DMA stream probe: what a loader-only kernel pulls into an LDS ring (chip-wide GB/s), by instruction form, cache policy, loader waves per CU, fill size, fills in
flight and deal -- the ceiling of the persistent decode engine's weight stream (csrc/decode_engine.hip), next to the same bytes through ordinary register loads."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load_package()
from minigpt4_cpp_amd import minigpt4_library as ML

lib = ML.load_library()
f = lib.library.minigpt4_amd_probe_dma
f.argtypes = [ctypes.c_int] * 6 + [ctypes.c_double]
f.restype = ctypes.c_float
FORMS = {0: "dma scalar-base+imm", 1: "dma per-lane addr", 2: "register loads", 3: "register loads + ds_write"}
print(f"{'form':26s} {'policy':8s} waves fill  depth deal       GB/s   GB/s per CU")
cases = []
for form in (0, 1):
    for waves, fill, depth in ((1, 16384, 2), (1, 16384, 3), (2, 16384, 2), (2, 16384, 3), (4, 8192, 3), (8, 4096, 3), (1, 8192, 4), (4, 16384, 2)):
        cases.append((form, 0, waves, fill, depth, 0))
cases += [(0, 1, 1, 16384, 3, 0), (0, 1, 2, 16384, 3, 0), (0, 0, 1, 16384, 3, 1), (0, 0, 2, 16384, 3, 1), (0, 0, 4, 8192, 3, 1)]
for waves, fill in ((8, 8192), (8, 16384), (4, 16384), (2, 16384), (1, 16384)):
    cases += [(2, 0, waves, fill, 1, 0), (2, 0, waves, fill, 1, 1), (3, 0, waves, fill, 2 if waves * 2 * fill <= 150 * 1024 else 1, 1)]
for c in cases:
    r = f(*c, 3.0)
    print(f"{FORMS[c[0]]:26s} {'nt' if c[1] == 0 else 'default':8s} {c[2]:5d} {c[3]:5d} {c[4]:5d} {'cyclic' if c[5] else 'blocked':8s} {r:8.0f} {r / 256:8.1f}", flush=True)
