#!/usr/bin/env python3
"""Worked example of localising a GPU-vs-oracle divergence with the two traces (how round 3 found hipcc folding `__float2half_rn(p * inv)` into v_fma_mixlo_f16):
the GPU runs in parity mode with MINIGPT4_PARITY_TRACE set, the oracle with orc_set_trace, tools/trace_diff.py names the first intermediate that differs.
Run on a GPU box:  python tools/trace_parity.py"""
import os, sys, tempfile
os.environ.setdefault("OMP_NUM_THREADS","16"); os.environ.setdefault("OMP_WAIT_POLICY","passive")
d = tempfile.mkdtemp()
os.environ["MINIGPT4_PARITY_TRACE"] = os.path.join(d, "gpu.trace")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")]
import _pkg; _pkg.load_package()
from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
import refcpu as R, trace_diff
lib = ML.load_library()
vp = os.path.join(d, "v.bin"); G.write_vision_file(vp, G.tiny_vision(n_embd_llm=4096), seed=3, std=0.05)
lp = os.path.join(d, "l.bin")
G.write_llm_file(lp, G.tiny_llm(wtype="q5_k", n_embd=256, n_layer=2, n_head=4, n_vocab=512, mix="none"), seed=1, std=0.05)
TOKS = [1, 5, 300, 44, 270, 99, 400, 17, 33, 260, 301, 302, 303, 304, 305, 306, 307, 308, 309, 310, 311]
ctx = lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=96, n_batch=16)
R.lib().orc_set_trace(os.path.join(d, "orc.trace").encode())
o = R.OracleLLM(G.read_llm_file(lp), n_ctx=96)
lib.amd_eval_tokens(ctx, TOKS); want = o.eval_tokens(TOKS); got = lib.amd_logits(ctx)
emb = (0.05 * np.random.default_rng(3).standard_normal((9, 256))).astype(np.float32)
lib.amd_eval_embd(ctx, emb); want = o.eval_embd(emb); got = lib.amd_logits(ctx)
tid = int(want.argmax())
lib.amd_eval_tokens(ctx, [tid]); want = o.eval_tokens([tid]); got = lib.amd_logits(ctx)
print("final", np.abs(got - want).max())
lib.minigpt4_free(ctx); R.lib().orc_set_trace(None)
print(trace_diff.diff(os.path.join(d, "gpu.trace"), os.path.join(d, "orc.trace")))
