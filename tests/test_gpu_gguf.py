"""A GGUF re-container of a GGJT model must run identically on the GPU: same logits bit for bit (the engine sees the same tensors; tests/test_cpu_gguf.py
proves the views equal), same greedy pieces through the chat API."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wtype,mix", [("q5_k", "q5_k_m"), ("q4_0", "none")])
def test_gguf_model_runs_identically(gpu_lib, tiny_files, tmp_path, wtype, mix):
    from minigpt4_cpp_amd import modelgen as G
    vp, llm = tiny_files
    src = llm(wtype, mix)
    dst = str(tmp_path / "m.gguf")
    G.ggjt_to_gguf(src, dst)
    outs = []
    for path in (src, dst):
        ctx = gpu_lib.minigpt4_model_load(vp, path, verbosity=0, n_ctx=256, n_batch=32)
        try:
            gpu_lib.minigpt4_system_prompt(ctx)
            gpu_lib.minigpt4_begin_chat(ctx, "hello there")
            lg = gpu_lib.amd_logits(ctx)
            pieces = [gpu_lib.minigpt4_end_chat(ctx, temp=0.0) for _ in range(6)]
            outs.append((lg, pieces))
        finally:
            gpu_lib.minigpt4_free(ctx)
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]


def test_committed_independent_gguf_fixture_runs_like_its_ggjt_source_and_like_the_oracle(gpu_lib, tiny_files, tmp_path):
    """The committed file of tests/gguf_independent.py (a writer that shares no code with the repo's converter): same logits bit for bit as the GGJT source on the fast
    path, and in parity mode bit-identical to the CPU oracle running the GGJT source."""
    import os
    import refcpu as R
    import gguf_independent as GI
    from minigpt4_cpp_amd import modelgen as G
    vp, _ = tiny_files
    src = str(tmp_path / "fixture_src.bin")
    GI.write_fixture_source(src)
    fx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_independent_v3.gguf")
    toks = [1, 5, 40, 44, 270, 99, 200, 17, 33, 260, 290]
    outs = []
    for path in (src, fx):
        ctx = gpu_lib.minigpt4_model_load(vp, path, verbosity=0, n_ctx=64, n_batch=32)
        try:
            gpu_lib.amd_eval_tokens(ctx, toks)
            fast = gpu_lib.amd_logits(ctx).copy()
            gpu_lib.minigpt4_reset_chat(ctx)
            gpu_lib.amd_set_parity(ctx, True)
            gpu_lib.amd_eval_tokens(ctx, toks)
            outs.append((fast, gpu_lib.amd_logits(ctx).copy()))
        finally:
            gpu_lib.minigpt4_free(ctx)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    want = R.OracleLLM(G.read_llm_file(src), n_ctx=64).eval_tokens(toks)
    assert np.array_equal(outs[1][1], want)
