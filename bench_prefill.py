#!/usr/bin/env python3
"""Prefill micro-benchmark: BASELINE.json configs[4] -- Vicuna-13B f16 (unquantised), one 512-token `llama_eval` on 1 MI355X, the MFMA-bound leg of the path.

  python bench_prefill.py [--tokens 512] [--config 13b-f16|13b|7b] [--reps 3]

Not part of the default `bench.py` run (the f16 file is 26 GB: it needs that much room in /dev/shm and ~1 min to write); prints ONE JSON line with the time
of a whole prompt evaluation through the C ABI hook `minigpt4_amd_eval_tokens`, the model FLOPs (2 x matrix parameters x tokens + causal attention) and the
fraction of the dense f16 MFMA peak.  For the quantised configs the same line is the int8-MFMA prefill (`k_mmq_*`) on the headline file.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

MFMA_F16_PEAK_TFLOPS = 2500.0     # dense f16 / bf16 (MI355X_MICROARCH.md); int8 dense is 2x that


def f16_files(bench, G, d):
    """The 13B f16 (unquantised) LLM file + a tiny vision file for the loader; None when the directory has no room for 26 GB."""
    lcfg = G.LLMConfig(n_vocab=32000, n_embd=5120, n_mult=256, n_head=40, n_layer=40, ftype=1, wtype="f16", mix="none")
    lp = os.path.join(d, "llm_13b_f16.bin")
    if not os.path.exists(lp + ".ok"):
        st = os.statvfs(d)
        if st.f_bavail * st.f_frsize < 27e9:
            return None, None, lcfg, f"{d} has {st.f_bavail * st.f_frsize / 1e9:.1f} GB free, the 13B f16 file needs 27 GB"
        t0 = time.time()
        G.write_llm_file(lp, lcfg, seed=1234, std=0.02, unique_layers=1, fast=True)
        open(lp + ".ok", "w").write("ok")
        print(f"[bench_prefill] wrote {lp} ({os.path.getsize(lp) / 1e9:.1f} GB) in {time.time() - t0:.0f}s", file=sys.stderr)
    vp = os.path.join(d, "vision_prefill_tiny.bin")
    if not os.path.exists(vp):
        G.write_vision_file(vp, G.tiny_vision(n_embd_llm=5120), seed=3, std=0.05)
    return vp, lp, lcfg, None


def prefill_leg(lib, vp, lp, lcfg, T=512, reps=3, dtype="f16", peak=MFMA_F16_PEAK_TFLOPS, workload=""):
    """One T-token prompt evaluation through the C ABI hook, best of `reps` after an untimed pass; returns the JSON object of the line."""
    import numpy as np
    ctx = lib.minigpt4_model_load(vp, lp, verbosity=1, seed=1, n_ctx=max(2048, T), n_batch=max(512, T))
    toks = [1] + [int(x) for x in np.random.default_rng(0).integers(259, lcfg.n_vocab, T - 1)]
    best = 1e9
    for i in range(reps + 1):                           # first pass untimed
        lib.minigpt4_reset_chat(ctx)
        lib.library.minigpt4_amd_sync(ctx.ptr)
        t0 = time.perf_counter()
        lib.amd_eval_tokens(ctx, toks)
        lib.library.minigpt4_amd_sync(ctx.ptr)
        dt = time.perf_counter() - t0
        if i:
            best = min(best, dt)
    E, L, V = lcfg.n_embd, lcfg.n_layer, lcfg.n_vocab
    F = ((2 * (4 * E) // 3 + lcfg.n_mult - 1) // lcfg.n_mult) * lcfg.n_mult
    mat_params = L * (4 * E * E + 3 * E * F)
    flops = 2.0 * mat_params * T + 2.0 * E * V + L * 2.0 * 2.0 * E * (T * (T + 1) / 2)    # layer matrices for T rows, output matrix for the last row, causal QK^T + PV
    out = {"metric": "prefill time", "value": best * 1e3, "unit": "ms", "higher_is_better": False, "n_gpus": 1, "tokens": T, "dtype": dtype, "data": "synthetic",
           "config": {"workload": workload}, "tokens_per_s": T / best,
           "roofline": {"bound": "mfma", "achieved": flops / best / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flops / best / 1e12 / peak, "traffic": None,
                        "flops": flops, "weight_bytes": lib.library.minigpt4_amd_weight_bytes_per_token(ctx.ptr)}}
    lib.minigpt4_free(ctx)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=512)
    ap.add_argument("--config", default="13b-f16", choices=["13b-f16", "13b", "7b"])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    _pkg.load_package()
    import bench
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    lib = ML.load_library()
    if lib.amd_device_count() <= 0:
        raise SystemExit("bench_prefill.py: no HIP device visible (the engine has no CPU fallback)")
    d = bench.model_dir()
    names = {"13b-f16": "Vicuna-13B f16 (unquantised), one 512-token llama_eval (BASELINE.json configs[4])",
             "13b": "Vicuna-13B Q5_K_M, one prompt evaluation on the int8 matrix cores", "7b": "Vicuna-7B Q4_0, one prompt evaluation"}
    if args.config == "13b-f16":
        vp, lp, lcfg, why = f16_files(bench, G, d)
        if why:
            raise SystemExit("bench_prefill.py: " + why)
        dtype, peak = "f16", MFMA_F16_PEAK_TFLOPS
    else:
        vp, lp, _, lcfg = bench.make_models(args.config, 0, 1, lambda: None)
        dtype, peak = "i8", 2 * MFMA_F16_PEAK_TFLOPS
    print(json.dumps(prefill_leg(lib, vp, lp, lcfg, args.tokens, args.reps, dtype, peak, names[args.config])), flush=True)


if __name__ == "__main__":
    main()
