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
