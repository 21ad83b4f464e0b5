"""GPU tier, Q3_K (runs last on purpose: the file name sorts after the other GPU tests).  Q3_K tensors are re-encoded at load as value-identical Q6_K
super-blocks (tests/test_cpu_q3k.py proves the identity on the CPU); here the engine's results for Q3_K inputs are compared with the oracle's Q3_K arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("shape", [(1, 512, 96), (5, 768, 70), (1, 5120, 64), (33, 768, 70), (142, 2048, 256)])
def test_q3k_mul_mat_matches_oracle(gpu_lib, shape):
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    N, n_in, n_out = shape
    rng = np.random.default_rng(311 + sum(shape))
    raw = Q.quantize(Q.GGML_Q3_K, (0.05 * rng.standard_normal((n_out, n_in))).astype(np.float32))
    x = rng.standard_normal((N, n_in)).astype(np.float32)
    want = R.mul_mat(Q.GGML_Q3_K, raw, n_in, n_out, x)
    got = gpu_lib.amd_test_mul_mat(Q.GGML_Q3_K, raw, n_in, n_out, x)
    assert _rel(got, want) < 2e-5, (shape, _rel(got, want))


def test_q3k_model_logits_and_greedy_tokens(gpu_lib, tiny_files):
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    vp, llm = tiny_files
    lp = llm("q3_k", conditioned=True)
    tol = 1e-2                                           # north_star's whole-model tolerance on the conditioned tiny model (tests/test_gpu_parity.py)
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=96, n_batch=16)
    try:
        o = R.OracleLLM(G.read_llm_file(lp), n_ctx=96)
        toks = [1, 5, 300, 44, 270, 99, 400, 17, 33, 260, 301, 302, 303, 304, 305, 306, 307, 308, 309, 310, 311]
        gpu_lib.amd_eval_tokens(ctx, toks)
        o.eval_tokens(toks[:16])
        want = o.eval_tokens(toks[16:])
        got = gpu_lib.amd_logits(ctx)
        errs = [_rel(got, want)]
        decided = agree = 0
        for _ in range(16):
            srt = np.sort(want)
            if (srt[-1] - srt[-2]) / (np.abs(want).max() + 1e-30) > tol:
                decided += 1
                agree += int(got.argmax() == want.argmax())
            tid = int(want.argmax())
            gpu_lib.amd_eval_tokens(ctx, [tid])
            want = o.eval_tokens([tid])
            got = gpu_lib.amd_logits(ctx)
            errs.append(_rel(got, want))
        assert max(errs) < tol, errs
        assert agree == decided and decided >= 12, (agree, decided)
    finally:
        gpu_lib.minigpt4_free(ctx)
