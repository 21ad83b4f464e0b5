"""GPU path against the COMMITTED golden vectors (tests/golden/llm_goldens.npz: oracle logits / greedy ids / image embedding of seeded tiny models).
Same bars as tests/test_gpu_parity.py: logits within LOGIT_TOL of the logit range (3e-3 for f16 weights), greedy ids identical wherever the golden
top-2 margin exceeds that noise (teacher-forced with the golden ids), image embedding within 3e-3."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_llm_goldens as M  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llm_goldens.npz"))
LOGIT_TOL = 5e-2


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("wtype,mix", M.CASES)
def test_llm_against_committed_goldens(gpu_lib, tiny_files, wtype, mix):
    vp, llm = tiny_files
    lp = llm(wtype, mix)                                   # same generator call as the golden script (seed 1, std 0.05)
    import hashlib
    if hashlib.sha256(open(lp, "rb").read()).hexdigest() != str(GOLD[f"{wtype}/file_sha256"]):
        pytest.skip("this host regenerated a different model file than the golden script's host (tests/test_cpu_goldens.py is the strict check)")
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=0, n_ctx=96, n_batch=16)
    try:
        gpu_lib.amd_eval_tokens(ctx, [int(t) for t in GOLD["prompt"]])
        got = gpu_lib.amd_logits(ctx)
        tol = 3e-3 if wtype == "f16" else LOGIT_TOL
        assert _rel(got, GOLD[f"{wtype}/prompt_logits"]) < tol
        ids, margins = GOLD[f"{wtype}/greedy_ids"], GOLD[f"{wtype}/greedy_margins"]
        decided = agree = 0
        for k in range(len(ids)):
            if margins[k] > LOGIT_TOL:
                decided += 1
                agree += int(got.argmax() == ids[k])
            gpu_lib.amd_eval_tokens(ctx, [int(ids[k])])    # teacher-forced: both sides consume the golden token
            got = gpu_lib.amd_logits(ctx)
        assert agree == decided and decided >= 1, (agree, decided)     # how many steps are decided depends on the model: q4_1's tiny model has 6 of 16
        assert _rel(got, GOLD[f"{wtype}/final_logits"]) < tol
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_image_embedding_against_committed_golden(gpu_lib, tiny_files):
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    vp, llm = tiny_files                                   # vision file: seed 3, std 0.05, as in the golden script
    import hashlib
    if hashlib.sha256(open(vp, "rb").read()).hexdigest() != str(GOLD["vision/file_sha256"]):
        pytest.skip("this host regenerated a different vision file than the golden script's host")
    ctx = gpu_lib.minigpt4_model_load(vp, llm("q4_0"), verbosity=0, n_ctx=64, n_batch=32)
    try:
        emb = gpu_lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(G.synth_image(42)))
        got = np.ctypeslib.as_array(emb.data, shape=(emb.n_embeddings,)).copy().reshape(32, -1)
        gpu_lib.minigpt4_free_embedding(emb)
        assert float(np.abs(got[:4] - GOLD["vision/embedding_rows_0_3"]).max() / float(GOLD["vision/embedding_absmax"])) < 3e-3
    finally:
        gpu_lib.minigpt4_free(ctx)
