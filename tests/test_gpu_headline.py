"""GPU parity at the BASELINE.json headline shapes (round-1 verdict, "What's missing" #1): the 13B Q5_K_M graph (mixed Q5_K / Q6_K launches, the NU = 3 / 7
register tilings, split-K vision GEMMs, XCD tile order) and the 7B Q4_0 graph, compared with the CPU oracle through the reference's call sequence
`system_prompt -> begin_chat_image -> 32 x end_chat_image(temp 0)` (reference minigpt4.cpp:2671-2732), and one full-size ViT-g/14 (1408 x 39 blocks) +
Q-Former encode against OracleVision.

Observed errors are written to gpurun_out/parity_observed_<config>.json on every run; the committed copy (tests/golden/parity_observed.json) records what a
GPU box measured: a run must also stay within 2x the recorded maximum, so a numerics regression at the real shapes fails here even when the tiny-model tests
stay green.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 32
ABS_BAR_VISION = 3e-3       # fp16-weight tower: no int8 rounding in the path; observed 5.6e-4 at the full ViT-g/14 + Q-Former (tests/golden/parity_observed.json)


def _recorded():
    p = os.path.join(ROOT, "tests", "golden", "parity_observed.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def _dump(name, obj):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(obj, open(os.path.join(d, f"parity_observed_{name}.json"), "w"), indent=1, sort_keys=True)


@pytest.fixture(scope="module")
def omp_threads():
    # the oracle at these sizes is real work: give it the cores the box grants (tests/conftest.py pins 16 for the tiny problems)
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
    return max(1, min(n, 32))


@pytest.mark.parametrize("config", ["13b_l2", "13b", "7b"])
def test_headline_chat_flow_matches_oracle(gpu_lib, config, omp_threads):
    """What can and cannot be asserted end to end (measured, oracle/headline.py::oracle_self_noise): ggml rounds every activation row to int8 before every mat-mul, so the
    oracle ITSELF moves by 1.2-1.5 % of its logit range on the 2-layer full-width model and by ~5 % on the 40-layer one when its input is perturbed by 1e-6..1e-7
    relative -- the size of fp32 summation-order differences.  No implementation that adds the per-block fp32 terms in another order can be closer to it than that, so
    the end-to-end criterion is statistical: the GPU-vs-oracle difference must not exceed the oracle-vs-perturbed-oracle difference of the same run (x 1.5 on the mean,
    x 2 on the maximum), greedy ids must be identical wherever the oracle's top-2 margin exceeds twice the observed difference, and the teacher-forced argmax agreement
    must match the oracle's agreement with itself.  Bit-level claims live in the component tests (activation quantisation bit-exact, integer block dots exact,
    mat-mul 2e-5 at these row lengths: test_gpu_parity.py, test_gpu_mmq2.py)."""
    import headline as H
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    vp, lp = H.headline_files(config)
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=512, n_batch=512)
    try:
        img = G.synth_image(42)
        emb = gpu_lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(img))
        E = emb.n_embeddings // 32
        emb_np = np.ctypeslib.as_array(emb.data, shape=(emb.n_embeddings,)).copy().reshape(32, E)
        assert np.isfinite(emb_np).all()
        orc = H.oracle_run(lp, emb_np, STEPS, threads=omp_threads)
        noise = H.oracle_self_noise(lp, emb_np, orc, eps=1e-6, threads=omp_threads)
        pieces = H.gpu_free_run(gpu_lib, ctx, emb, STEPS)
        logits = H.gpu_teacher_forced(gpu_lib, ctx, emb, orc["ids"])
        res = H.compare(orc, pieces, logits)
        res["oracle_self_noise"] = noise
        res["oracle_prefill_s"], res["oracle_decode_s"] = orc["prefill_s"], orc["decode_s"]
        _dump(config, res)
        print(config, json.dumps(res))
        assert res["mean_logit_rel_range"] <= 1.5 * noise["mean_logit_rel_range"] + 1e-4, res
        assert res["max_logit_rel_range"] <= 2.0 * noise["max_logit_rel_range"] + 1e-4, res
        rec = _recorded().get(config, {})
        if "max_logit_rel_range" in rec:                       # and never more than twice what the committed record of this config shows
            assert res["max_logit_rel_range"] <= 2.0 * rec["max_logit_rel_range"], (res, rec)
        # bit-exact greedy ids wherever the oracle's own decision is outside the arithmetic's noise band
        assert res["decided_argmax_identical"] == res["decided"], res
        assert res["teacher_forced_argmax_identical"] >= noise["argmax_identical"] - 4, res
        # free-running text: identical up to the first undecided step at least
        ol = orc["logits"].astype(np.float64)
        srt = np.sort(ol, axis=1)
        undecided = [i for i in range(STEPS) if (srt[i, -1] - srt[i, -2]) <= 2.0 * np.abs(logits - ol).max()]
        assert res["free_running_first_divergence"] >= (undecided[0] if undecided else STEPS), res
        gpu_lib.minigpt4_free_embedding(emb)
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_full_size_vit_g_encode_matches_oracle(gpu_lib, omp_threads):
    """EVA ViT-g/14 at its real size (dim 1408, 16 heads x 88, MLP 6144, 39 blocks) + the 12-layer Q-Former + llama_proj (5120): every block has its own
    weights here (unique_blocks = None), so a wrong stride / tile order / split-K slab shows up."""
    import refcpu as R
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    import headline as H
    d = H.model_dir()
    vp = os.path.join(d, "vision_13b_unique.bin")
    if not os.path.exists(vp + ".ok"):
        G.write_vision_file(vp, G.vision_13b(), seed=777, std=0.02, unique_blocks=None, fast=True)
        open(vp + ".ok", "w").write("ok")
    lp = os.path.join(d, "llm_tiny_for_vision.bin")
    G.write_llm_file(lp, G.tiny_llm(wtype="q4_0", n_embd=256, n_layer=1, n_head=4, n_vocab=512), seed=2, std=0.05)   # the loader wants an LLM file; its width is irrelevant to the encode
    R.lib().orc_set_threads(int(omp_threads))
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=64, n_batch=32)
    try:
        img = G.synth_image(7)
        emb = gpu_lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(img))
        got = np.ctypeslib.as_array(emb.data, shape=(emb.n_embeddings,)).copy().reshape(32, -1)
        want = R.OracleVision(G.read_vision_file(vp)).encode(img)
        err = float(np.abs(got - want).max() / np.abs(want).max())
        _dump("vision_13b", {"max_rel": err, "shape": list(got.shape)})
        print("vision_13b", err)
        rec = _recorded().get("vision_13b", {})
        bar = min(ABS_BAR_VISION, 2.0 * rec["max_rel"]) if "max_rel" in rec else ABS_BAR_VISION
        assert got.shape == want.shape and err <= bar, err
        gpu_lib.minigpt4_free_embedding(emb)
    finally:
        gpu_lib.minigpt4_free(ctx)
