"""GPU parity at the BASELINE.json headline shapes: the 13B Q5_K_M graph (40 layers, all different; mixed Q5_K / Q6_K launches, the NU = 3 / 7 register tilings), the same
graph two layers deep, and the 7B Q4_0 graph -- against the CPU oracle through the reference's call sequence `system_prompt -> begin_chat_image -> K x end_chat_image(temp 0)`
(reference minigpt4.cpp:2671-2732, :2365-2382 add_tokens, :2425-2456 greedy sampling), plus one full-size ViT-g/14 (1408 x 39 blocks) + Q-Former encode against OracleVision.

north_star: "bit-exact token ids under greedy sampling with fp32 accumulation, logits within 1e-2 relative otherwise".  Both halves are asserted:
  * PARITY MODE (MINIGPT4_PARITY: the per-block fp32 terms added in the oracle's order): free-running, the logits behind EVERY sampled token equal the oracle's bit for bit and
    every greedy id is identical -- 256 steps on the 40-layer 13B file;
  * FAST MODE (the kernels bench.py measures): identical free-running greedy ids for all 32 steps, teacher-forced logits within 1e-2 of the largest |logit| at every step,
    at least 24 of 32 steps "decided" (oracle top-2 margin > 2 x the largest observed difference) with identical argmax on all of them.
The files (modelgen.headline_llm) are conditioned like a trained network -- scaled residual writers, every layer different, decisive output logits -- so that the criterion
is a statement about the kernels rather than about near-ties between 32000 i.i.d. logits.  The oracle's own sensitivity (1e-6 input perturbation: int8 activation
re-rounding, ~5e-3 of the logit range on the 13B file) is measured in the same run and recorded next to the GPU's numbers.

Observed numbers are written to gpurun_out/parity_observed_<config>.json on every run; tests/golden/parity_observed.json is the committed record of a GPU box.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 32
PARITY_STEPS = {"13b": 256, "13b_l2": 64, "7b": 64, "13b_v32001_l2": 48}   # free-running steps in parity mode (bit-identical logits at every one of them)
LOGIT_REL_TOL = 1e-2        # north_star: fast-mode logits within 1e-2 relative (max |delta| / max |logit| per step)
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


# 13b_v32001_l2 (round 5): the 13B width with Vicuna-v0's REAL vocabulary size, 32001 -> llama.cpp's k-quant fallback types (output.weight F16: a 327.7 MB mat-vec with an odd
# row count on the pipelined kernel; tok_embeddings Q4_0), two layers deep
@pytest.mark.parametrize("config", ["13b_l2", "13b", "7b", "13b_v32001_l2"])
def test_headline_chat_flow_matches_oracle(gpu_lib, config, omp_threads):
    import headline as H
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    vp, lp = H.headline_files(config)
    n_par = PARITY_STEPS[config]
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=512, n_batch=512)
    try:
        img = G.synth_image(42)
        emb = gpu_lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(img))
        E = emb.n_embeddings // 32
        emb_np = np.ctypeslib.as_array(emb.data, shape=(emb.n_embeddings,)).copy().reshape(32, E)
        assert np.isfinite(emb_np).all()
        orc = H.oracle_run(lp, emb_np, n_par, threads=omp_threads)
        # ---- parity mode: bit-identical logits, identical greedy ids, free-running, every step
        par = H.gpu_parity_mode_run(gpu_lib, ctx, emb, orc)
        # ---- fast mode on the first STEPS steps
        head = {"logits": orc["logits"][:STEPS], "ids": orc["ids"][:STEPS], "pieces": orc["pieces"][:STEPS], "n_prompt": orc["n_prompt"]}
        noise = H.oracle_self_noise(lp, emb_np, head, eps=1e-6, threads=omp_threads)
        pieces = H.gpu_free_run(gpu_lib, ctx, emb, STEPS)
        logits = H.gpu_teacher_forced(gpu_lib, ctx, emb, head["ids"])
        res = H.compare(head, pieces, logits)
        res["parity_mode"] = par
        res["oracle_self_noise"] = noise
        res["oracle_order_spread"] = H.oracle_order_spread(lp, emb_np, head, threads=omp_threads)   # the oracle's OTHER fp32 order (ggml-style lane partials) against the one used here
        res["oracle_prefill_s"], res["oracle_decode_s"] = orc["prefill_s"], orc["decode_s"]
        res["distinct_greedy_ids"] = len(set(orc["ids"]))
        _dump(config, res)
        print(config, json.dumps(res))
        assert par["logits_bit_identical"] == n_par and par["greedy_ids_identical"] == n_par and par["first_mismatch"] == -1, par
        assert res["distinct_greedy_ids"] >= n_par // 2, res            # the greedy walk is not a fixed point: identical ids are not a vacuous statement
        assert res["max_logit_rel"] <= LOGIT_REL_TOL, res
        assert res["free_running_identical"] == STEPS and res["free_running_first_divergence"] == STEPS, res
        assert res["teacher_forced_argmax_identical"] == STEPS, res
        assert res["decided"] >= 24 and res["decided_argmax_identical"] == res["decided"], res
        rec = _recorded().get(config, {})
        if "max_logit_rel" in rec:                              # and never more than twice what the committed record of this config shows
            assert res["max_logit_rel"] <= 2.0 * rec["max_logit_rel"], (res, rec)
        gpu_lib.minigpt4_free_embedding(emb)
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_f16_13b_width_512_token_prefill_matches_oracle(gpu_lib, omp_threads):
    """BASELINE.json configs[4] at its REAL shapes: Vicuna-13B f16 (unquantised) width -- n_embd 5120, 40 heads of 128, n_ff 13824, n_vocab 32000 -- and one 512-token
    `llama_eval`, i.e. the launches the 18 ms / 720 TFLOP/s figure is quoted on (`k_gemm_dma` 256x128 / 256x256 set launches with K split, the w1|w3 PAIR launch whose epilogue
    stores fp16(silu * mul), `k_attn_prefill_h8` storing fp16 rows for wo, split-K combines folded into the norms).  The file is the 40-layer graph cut to TWO layers (every launch
    shape of the full file occurs; the oracle's 512-row f16 pass over 40 layers would take minutes).  Last-token logits within 3e-3 of the CPU oracle (reference path:
    llama_eval behind minigpt4.cpp:2373, ggml f16 mat-mul with fp16-rounded activations), also after four decode steps on the cache the prompt launches filled."""
    import refcpu as R
    import headline as H
    from minigpt4_cpp_amd import modelgen as G
    d = H.model_dir()
    lp = os.path.join(d, "llm_13b_f16_l2.bin")
    if not os.path.exists(lp + ".ok"):
        lcfg = G.LLMConfig(n_vocab=32000, n_embd=5120, n_mult=256, n_head=40, n_layer=2, ftype=1, wtype="f16", mix="none")
        G.write_llm_file(lp, lcfg, seed=77, std=0.02, fast=True, **{k: v for k, v in G.TINY_CONDITIONED.items()})
        open(lp + ".ok", "w").write("ok")
    vp = os.path.join(d, "vision_tiny_5120.bin")
    if not os.path.exists(vp):
        G.write_vision_file(vp, G.tiny_vision(n_embd_llm=5120), seed=3, std=0.05)
    toks = [1] + [int(x) for x in np.random.default_rng(8).integers(259, 32000, 511)]
    R.lib().orc_set_threads(int(omp_threads))
    o = R.OracleLLM(G.read_llm_file(lp, in_memory=True), n_ctx=640)
    want = o.eval_tokens(toks)
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=640, n_batch=512)
    try:
        gpu_lib.amd_eval_tokens(ctx, toks)
        fast = gpu_lib.amd_logits(ctx).copy()
        ids = []
        for _ in range(4):                                      # and four decode steps that read the K / V rows the prompt launches wrote
            ids.append(int(gpu_lib.amd_logits(ctx).argmax()))
            gpu_lib.amd_eval_tokens(ctx, [ids[-1]])
        after = gpu_lib.amd_logits(ctx).copy()
    finally:
        gpu_lib.minigpt4_free(ctx)
    for t in ids:
        want2 = o.eval_tokens([t])
    rel = float(np.abs(fast - want).max() / np.abs(want).max())
    rel2 = float(np.abs(after - want2).max() / np.abs(want2).max())
    _dump("13b_f16_l2_prefill512", {"max_logit_rel": rel, "max_logit_rel_after_4_decode_steps": rel2})
    assert np.isfinite(fast).all() and rel <= 3e-3, rel
    assert rel2 <= 3e-3, rel2


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
        # parity mode at full size: every one of the 39 different blocks, the 12 Q-Former layers and the projection in the oracle's accumulation order -> bit-identical
        gpu_lib.amd_set_parity(ctx, True)
        emb_p = gpu_lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(img))
        par = np.ctypeslib.as_array(emb_p.data, shape=(emb_p.n_embeddings,)).copy().reshape(32, -1)
        gpu_lib.minigpt4_free_embedding(emb_p)
        gpu_lib.amd_set_parity(ctx, False)
        bit = bool(np.array_equal(par, want))
        _dump("vision_13b", {"max_rel": err, "shape": list(got.shape), "parity_mode_bit_identical": bit, "parity_mode_max_abs_delta": float(np.abs(par - want).max())})
        print("vision_13b", err, "parity mode bit-identical:", bit)
        assert bit, float(np.abs(par - want).max())
        rec = _recorded().get("vision_13b", {})
        bar = min(ABS_BAR_VISION, 2.0 * rec["max_rel"]) if "max_rel" in rec else ABS_BAR_VISION
        assert got.shape == want.shape and err <= bar, err
        gpu_lib.minigpt4_free_embedding(emb)
    finally:
        gpu_lib.minigpt4_free(ctx)
