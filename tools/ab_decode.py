#!/usr/bin/env python3
"""A/B runner for decode experiments on one GPU box: builds the synthetic model files once, then runs every variant in its own process
(the engine reads its knobs at model load).  usage: tools/ab_decode.py [--config 13b] [--steps 96] NAME[:ENV=VAL[,ENV=VAL...]] ...
The pseudo-variable LIB selects another build of the library (MINIGPT4_LIBRARY)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(config, steps):
    import _pkg
    _pkg.load_package()
    import bench
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    lib = ML.load_library()
    vp, lp, vcfg, lcfg = bench.make_models(config, 0, 1, lambda: None)
    ctx = lib.minigpt4_model_load(vp, lp, verbosity=0, seed=1337, n_ctx=2048, n_batch=512)
    img = ML.array_to_image_struct(G.synth_image(42))
    enc = []
    for i in range(3):
        emb = lib.minigpt4_encode_image(ctx, img); enc.append(lib.library.minigpt4_amd_last_encode_ms(ctx.ptr))
    t0 = time.perf_counter()
    lib.minigpt4_system_prompt(ctx); lib.minigpt4_begin_chat_image(ctx, emb, bench.PROMPT); lib.library.minigpt4_amd_sync(ctx.ptr)
    prefill = (time.perf_counter() - t0) * 1e3
    for _ in range(8): lib.minigpt4_end_chat_image(ctx, temp=0.0)
    lib.library.minigpt4_amd_sync(ctx.ptr)
    t0 = time.perf_counter()
    toks = [lib.minigpt4_end_chat_image(ctx, temp=0.0) for _ in range(steps)]
    lib.library.minigpt4_amd_sync(ctx.ptr)
    dt = time.perf_counter() - t0
    ids, loop_ms = lib.amd_decode_loop(ctx, 33)
    import zlib
    sig = zlib.crc32("".join(t if isinstance(t, str) else t.decode("latin1") for t in toks).encode("latin1", "replace"))
    print(json.dumps({"tok_s": steps / dt, "ms_tok": dt * 1e3 / steps, "loop_ms_tok": loop_ms / 32.0, "encode_ms": min(enc[1:]), "prefill_ms": prefill, "sig": sig}), flush=True)


def main():
    args = sys.argv[1:]
    config, steps = "13b", 96
    if args and args[0] == "--child":
        return child(args[1], int(args[2]))
    while args and args[0].startswith("--"):
        if args[0] == "--config": config = args[1]
        elif args[0] == "--steps": steps = int(args[1])
        args = args[2:]
    for spec in args:
        name, _, envs = spec.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            if k == "LIB": env["MINIGPT4_LIBRARY"] = os.path.join(ROOT, v)
            else: env[k] = v
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", config, str(steps)], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(f"{name:28s} {line[-1] if line else 'FAILED rc=%d %s' % (p.returncode, p.stderr[-400:])}", flush=True)


if __name__ == "__main__":
    main()
