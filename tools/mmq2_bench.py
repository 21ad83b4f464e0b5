#!/usr/bin/env python3
"""Prefill mat-mul micro-benchmark on the 13B layer shapes (GPU only):  tools/mmq2_bench.py [N ...]
Prints microseconds per launch of the layer's four prefill launches for generation 2 (mmq2_kernels.hip) and generation 1 (round-1 kernels, one launch per matrix),
with the launcher's own K split and with KS=1 (no split) -- each in its own process."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("qkv", "q5_k", 5120, 5120, 3), ("wo", "q5_k", 5120, 5120, 1), ("w1w3", "q5_k", 13824, 5120, 2), ("w2", "q5_k", 5120, 13824, 1), ("w2_q6k", "q6_k", 5120, 13824, 1)]


def child(ns):
    import _pkg
    _pkg.load_package()
    from minigpt4_cpp_amd import minigpt4_library as ML, quants as Q
    L = ML.load_library().library
    L.minigpt4_amd_bench_mmq.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
    for N in ns:
        row = []
        for name, t, rows, cols, n_mat in SHAPES:
            us = ctypes.c_float()
            gens = tuple(int(g) for g in os.environ.get("GENS", "4,2").split(","))
            vals = []
            for g in gens:
                rc = L.minigpt4_amd_bench_mmq(Q.NAME_TO_TYPE[t], rows, cols, n_mat, N, 20, int(os.environ.get("KS", "0")), g, ctypes.byref(us))
                vals.append(us.value if rc == 0 else float("nan"))
            row.append(f"{name} " + "/".join(f"{v:.1f}" for v in vals))
        print(f"N={N} dbg={os.environ.get('MINIGPT4_MMQ2_DBG', '0')} ks={os.environ.get('KS', 'auto')}: " + "  ".join(row) + "   (us per launch: generations " + os.environ.get("GENS", "4,2") + "; 4 = fp16-MFMA k_mmqh (round 5), 3 = digit planes mmq3, 2 = int8 mmq2, 1 = round 1)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child([int(x) for x in sys.argv[2:]])
    else:
        ns = sys.argv[1:] or ["142", "512"]
        for ks in ("0", "1"):
            env = dict(os.environ)
            if ks != "0":
                env["KS"] = ks
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + ns, env=env)
