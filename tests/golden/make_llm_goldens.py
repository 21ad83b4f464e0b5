#!/usr/bin/env python3
"""Generates tests/golden/llm_goldens.npz: outputs of the CPU oracle (oracle/refcpu.c) on seeded tiny models, committed so that
  * the oracle itself is pinned against drift (tests/test_cpu_goldens.py re-runs it and requires bit-equal logits), and
  * the GPU path is compared with fixed vectors, not only with an oracle rebuilt on the spot (tests/test_gpu_goldens.py).
These are goldens OF THE ORACLE -- a restatement of ggml's arithmetic; the reference itself cannot be built here (parity unpinned).

    python tests/golden/make_llm_goldens.py
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg  # noqa: E402

_pkg.load_package()

CASES = [("q4_0", "none"), ("q5_k", "q5_k_m"), ("q8_0", "none"), ("q4_1", "none"), ("q6_k", "none"), ("f16", "none")]
PROMPT = [1, 5, 300, 44, 270, 99, 400, 17, 33, 260, 301, 302, 303, 304, 305, 306, 307, 308, 309, 310, 311]
N_GREEDY = 16


def llm_case(wtype, mix, d):
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    p = os.path.join(d, f"llm_{wtype}.bin")
    G.write_llm_file(p, G.tiny_llm(wtype=wtype, n_embd=256, n_layer=2, n_head=4, n_vocab=512, mix=mix), seed=1, std=0.05, **G.TINY_CONDITIONED)   # = conftest tiny_files llm(..., conditioned=True)
    o = R.OracleLLM(G.read_llm_file(p), n_ctx=96)
    o.eval_tokens(PROMPT[:16])
    logits = o.eval_tokens(PROMPT[16:]).copy()
    ids, margins, lg = [], [], logits
    for _ in range(N_GREEDY):
        srt = np.sort(lg)
        margins.append(float((srt[-1] - srt[-2]) / (np.abs(lg).max() + 1e-30)))
        ids.append(int(lg.argmax()))
        lg = o.eval_tokens([ids[-1]])
    return hashlib.sha256(open(p, "rb").read()).hexdigest(), logits, np.array(ids, np.int32), np.array(margins, np.float32), lg.copy()


def vision_case(d):
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    p = os.path.join(d, "vision.bin")
    G.write_vision_file(p, G.tiny_vision(n_embd_llm=4096), seed=3, std=0.05)
    emb = R.OracleVision(G.read_vision_file(p)).encode(G.synth_image(42))
    return hashlib.sha256(open(p, "rb").read()).hexdigest(), emb


def main():
    out = {"prompt": np.array(PROMPT, np.int32)}
    with tempfile.TemporaryDirectory() as d:
        for wtype, mix in CASES:
            sha, logits, ids, margins, last = llm_case(wtype, mix, d)
            out[f"{wtype}/file_sha256"] = np.array(sha)
            out[f"{wtype}/prompt_logits"] = logits
            out[f"{wtype}/greedy_ids"] = ids
            out[f"{wtype}/greedy_margins"] = margins
            out[f"{wtype}/final_logits"] = last
        sha, emb = vision_case(d)
        out["vision/file_sha256"] = np.array(sha)
        out["vision/embedding_rows_0_3"] = emb[:4].copy()                  # 4 of the 32 query rows in full
        out["vision/embedding_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(emb).tobytes()).hexdigest())
        out["vision/embedding_absmax"] = np.array(np.abs(emb).max(), np.float32)
    np.savez_compressed(os.path.join(HERE, "llm_goldens.npz"), **out)
    print("wrote", os.path.join(HERE, "llm_goldens.npz"), os.path.getsize(os.path.join(HERE, "llm_goldens.npz")), "bytes")


if __name__ == "__main__":
    main()
