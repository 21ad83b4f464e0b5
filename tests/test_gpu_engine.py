"""The persistent decode engine (csrc/decode_engine.hip: ONE launch per layer runs wo -> w1|w3 -> w2 -> the next layer's wq, wk, wv with in-launch granule hand-offs)
must reproduce the launch-per-op decode step BIT FOR BIT: same unit traits, same lane -> unit map, same per-lane fma order, same wave reduction, the fused prologue's
512-thread rms / quantisation order.  Every test runs the same conversation twice in one context -- engine on, engine off -- and compares every logits vector with
`np.array_equal` (reference path: llama_eval per token behind minigpt4.cpp:2373; greedy sampling :2425-2456)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOKS = [1, 5, 300, 44, 270, 99, 400, 17, 33, 260, 301, 302, 303, 304, 305, 306, 307, 308, 309, 310, 311]


def _walk(lib, ctx, toks, steps, engine):
    lib.amd_set_engine(ctx, engine)
    lib.minigpt4_reset_chat(ctx)
    lib.amd_eval_tokens(ctx, toks)
    out = [lib.amd_logits(ctx).copy()]
    for _ in range(steps):
        lib.amd_eval_tokens(ctx, [int(out[-1].argmax())])      # one row from a token id: the captured decode graph
        out.append(lib.amd_logits(ctx).copy())
    return out


@pytest.mark.parametrize("wtype,mix", [("q4_0", "none"), ("q5_k", "q5_k_m"), ("q4_k", "q5_k_m"), ("q6_k", "none"), ("q4_k", "none"), ("q5_k", "none")])
def test_engine_step_is_bit_identical_on_tiny_models(gpu_lib, tiny_files, wtype, mix):
    vp, llm = tiny_files
    ctx = gpu_lib.minigpt4_model_load(vp, llm(wtype, mix), verbosity=1, n_ctx=96, n_batch=16)
    try:
        assert gpu_lib.amd_engine_active(ctx), "the engine must serve this model (a silent fall-back would make this test vacuous)"
        on = _walk(gpu_lib, ctx, TOKS, 24, True)
        off = _walk(gpu_lib, ctx, TOKS, 24, False)
        assert not gpu_lib.amd_engine_active(ctx)
        for i, (a, b) in enumerate(zip(on, off)):
            assert np.isfinite(a).all() and np.array_equal(a, b), (wtype, mix, i, float(np.abs(a - b).max()))
        again = _walk(gpu_lib, ctx, TOKS, 24, True)              # and the engine against itself after the switch (graph re-captured, tags from a later epoch)
        for i, (a, b) in enumerate(zip(on, again)):
            assert np.array_equal(a, b), (wtype, mix, i)
    finally:
        gpu_lib.minigpt4_free(ctx)


@pytest.mark.parametrize("wtype", ["q8_0", "q4_1", "f16"])
def test_types_outside_the_engine_keep_the_launch_per_op_step(gpu_lib, tiny_files, wtype):
    vp, llm = tiny_files
    ctx = gpu_lib.minigpt4_model_load(vp, llm(wtype), verbosity=1, n_ctx=96, n_batch=16)
    try:
        assert not gpu_lib.amd_engine_active(ctx)
        out = _walk(gpu_lib, ctx, TOKS, 4, True)
        assert all(np.isfinite(o).all() for o in out)
    finally:
        gpu_lib.minigpt4_free(ctx)


@pytest.mark.parametrize("config", ["13b_l2", "7b"])
def test_engine_step_is_bit_identical_at_the_headline_shapes(gpu_lib, config):
    """13B Q5_K_M width (K = 5120 / 13824: 3 / 7 units per lane, Q6_K wv / w2 of a 'more bits' layer behind Q5_K wq / wk) and the 7B Q4_0 file (K = 4096 / 11008, 32 layers,
    Q6_K output): every workgroup owns several fills of every matrix, so the ring, the fill ownership and all three hand-offs of every layer are exercised."""
    import headline as H
    vp, lp = H.headline_files(config)
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=256, n_batch=64)
    try:
        assert gpu_lib.amd_engine_active(ctx)
        rng = np.random.default_rng(11)
        toks = [1] + [int(t) for t in rng.integers(3, 31000, 40)]
        on = _walk(gpu_lib, ctx, toks, 24, True)
        off = _walk(gpu_lib, ctx, toks, 24, False)
        for i, (a, b) in enumerate(zip(on, off)):
            assert np.isfinite(a).all() and np.array_equal(a, b), (config, i, float(np.abs(a - b).max()))
        # the device-resident greedy loop (graph replays back to back, nothing between the steps but the stream order)
        gpu_lib.amd_set_engine(ctx, True)
        gpu_lib.minigpt4_reset_chat(ctx); gpu_lib.amd_eval_tokens(ctx, toks); gpu_lib.amd_logits(ctx)
        ids_on, _ = gpu_lib.amd_decode_loop(ctx, 33)
        gpu_lib.amd_set_engine(ctx, False)
        gpu_lib.minigpt4_reset_chat(ctx); gpu_lib.amd_eval_tokens(ctx, toks); gpu_lib.amd_logits(ctx)
        ids_off, _ = gpu_lib.amd_decode_loop(ctx, 33)
        assert list(ids_on) == list(ids_off)
    finally:
        gpu_lib.minigpt4_free(ctx)
