"""MINIGPT4_PARITY mode: the language path with every fp32 accumulation in the CPU oracle's order must equal the oracle BIT FOR BIT.

north_star: "bit-exact token ids under greedy sampling with fp32 accumulation, logits within 1e-2 relative otherwise".  The fast kernels add the per-block fp32
terms of a dot product in a parallel order (lanes, waves, K splits, MFMA tiles); everything else -- activation quantisation, integer block dots, fp16 tables, RoPE,
softmax sum -- is exact and shared.  Parity mode (Engine::forward_ref: k_mul_mat_ref / k_attn_ref / sequential rms sum) uses the same unit traits and the same
HBM planes, but adds in oracle/refcpu.c's order, so:
  * parity-mode logits == oracle logits, `np.array_equal`, for every weight type, prompt chunks, embedding rows and decode steps;
  * any residual difference would localise a real bug (layout, integer arithmetic, table, stride) rather than "summation noise";
  * the fast path is then compared with parity mode ON THE GPU (same weights in HBM): that difference is summation order only.
Reference call sites: /root/reference/minigpt4.cpp:2365-2382 (add_tokens -> llama_eval), :2399-2422 (llama_eval_embd), :2425-2456 (greedy).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QTYPES = ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "q2_k", "q3_k", "q4_k", "q5_k", "q6_k", "f16", "f32"]


@pytest.mark.parametrize("wtype", QTYPES)
@pytest.mark.parametrize("shape", [(1, 512, 96), (5, 768, 70), (2, 5120, 64), (3, 13824, 40), (33, 1024, 130), (7, 352, 100), (4, 1408, 64)])
def test_mul_mat_ref_bit_identical(gpu_lib, wtype, shape):
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    N, n_in, n_out = shape
    t = Q.NAME_TO_TYPE[wtype]
    if n_in % Q.BLOCK[t][0]:
        pytest.skip("row length is not a whole number of blocks of this type")
    rng = np.random.default_rng(sum(map(ord, wtype)) * 999 + sum(shape))
    w = (0.05 * rng.standard_normal((n_out, n_in))).astype(np.float32)
    raw = Q.quantize(t, w)
    x = rng.standard_normal((N, n_in)).astype(np.float32)
    want = R.mul_mat(t, raw, n_in, n_out, x)
    got = gpu_lib.amd_test_mul_mat(t, raw, n_in, n_out, x, ref=True)
    assert np.array_equal(got, want), (wtype, shape, float(np.abs(got - want).max()))
    fast = gpu_lib.amd_test_mul_mat(t, raw, n_in, n_out, x)            # the fast dispatch differs by summation order only
    assert np.abs(fast - want).max() <= 2e-5 * np.abs(want).max()


TOKS = [1, 5, 300, 44, 270, 99, 400, 17, 33, 260, 301, 302, 303, 304, 305, 306, 307, 308, 309, 310, 311]   # 21 tokens: 2 chunks of n_batch = 16


@pytest.mark.parametrize("wtype,mix", [("q4_0", "none"), ("q5_k", "q5_k_m"), ("q4_k", "q5_k_m"), ("q4_1", "none"), ("q8_0", "none"), ("q6_k", "none"), ("q5_0", "none"),
                                       ("q5_1", "none"), ("q4_k", "none"), ("f16", "none"), ("f32", "none"), ("q2_k", "none"), ("q3_k", "none")])
def test_llm_parity_mode_is_bit_identical_to_the_oracle(gpu_lib, tiny_files, wtype, mix):
    """Prompt chunks (16 + 5 rows), 9 embedding rows (llama_eval_embd), then 24 FREE-RUNNING greedy steps on both sides: every logits vector equal bit for bit,
    every greedy id equal."""
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    vp, llm = tiny_files
    lp = llm(wtype, mix)
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=96, n_batch=16)
    try:
        gpu_lib.amd_set_parity(ctx, True)
        o = R.OracleLLM(G.read_llm_file(lp), n_ctx=96)
        gpu_lib.amd_eval_tokens(ctx, TOKS)
        o.eval_tokens(TOKS[:16])
        want = o.eval_tokens(TOKS[16:])
        got = gpu_lib.amd_logits(ctx)
        assert np.array_equal(got, want), float(np.abs(got - want).max())
        emb = (0.05 * np.random.default_rng(3).standard_normal((9, 256))).astype(np.float32)
        gpu_lib.amd_eval_embd(ctx, emb)
        want = o.eval_embd(emb)
        got = gpu_lib.amd_logits(ctx)
        assert np.array_equal(got, want), float(np.abs(got - want).max())
        for step in range(24):
            gid, oid = int(got.argmax()), int(want.argmax())
            assert gid == oid
            gpu_lib.amd_eval_tokens(ctx, [gid])
            want = o.eval_tokens([oid])
            got = gpu_lib.amd_logits(ctx)
            assert np.array_equal(got, want), (step, float(np.abs(got - want).max()))
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_fast_path_differs_from_parity_mode_by_summation_order_only(gpu_lib, tiny_files):
    """Same context, same HBM planes, same KV cache: switching the mode mid-conversation changes the logits by fp32 summation noise re-rounded through the int8
    activation steps of two layers -- and switching back to parity mode for a fresh conversation reproduces the oracle exactly again."""
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    vp, llm = tiny_files
    lp = llm("q5_k", "q5_k_m")
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=96, n_batch=16)
    try:
        o = R.OracleLLM(G.read_llm_file(lp), n_ctx=96)
        want = None
        o.eval_tokens(TOKS[:16]); want = o.eval_tokens(TOKS[16:])
        gpu_lib.amd_eval_tokens(ctx, TOKS)
        fast = gpu_lib.amd_logits(ctx)
        gpu_lib.minigpt4_reset_chat(ctx)
        gpu_lib.amd_set_parity(ctx, True)
        gpu_lib.amd_eval_tokens(ctx, TOKS)
        par = gpu_lib.amd_logits(ctx)
        assert np.array_equal(par, want)
        assert np.abs(fast - par).max() <= 5e-2 * (par.max() - par.min())
        gpu_lib.amd_set_parity(ctx, False)
        gpu_lib.minigpt4_reset_chat(ctx)
        gpu_lib.amd_eval_tokens(ctx, TOKS)
        assert np.array_equal(gpu_lib.amd_logits(ctx), fast)          # the fast path itself is deterministic
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_parity_mode_through_the_reference_chat_flow(gpu_lib, tmpdir_models):
    """system_prompt -> begin_chat_image(oracle's embedding) -> 16 x end_chat_image(temp 0) (minigpt4.cpp:2671-2732) in parity mode: pieces identical to OracleChat and
    the logits behind every sampled token bit-identical."""
    import os
    import refcpu as R
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    vp = os.path.join(tmpdir_models, "vision_pm.bin")
    lp = os.path.join(tmpdir_models, "llm_pm.bin")
    G.write_vision_file(vp, G.tiny_vision(n_embd_llm=4096), seed=11, std=0.05)
    G.write_llm_file(lp, G.tiny_llm(wtype="q5_k", n_embd=4096, n_layer=2, n_head=32, n_vocab=512, mix="q5_k_m"), seed=2, std=0.02)
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=256, n_batch=64)
    try:
        gpu_lib.amd_set_parity(ctx, True)
        emb_np = R.OracleVision(G.read_vision_file(vp)).encode(G.synth_image(42)).astype(np.float32)
        emb = ML.MiniGPT4Embedding()
        flat = np.ascontiguousarray(emb_np.reshape(-1))
        emb.data = flat.ctypes.data_as(ML.FLOAT_PTR)
        emb.n_embeddings = flat.size
        chat = R.OracleChat(R.OracleLLM(G.read_llm_file(lp), n_ctx=256), n_batch=64)
        chat.system_prompt()
        chat.begin_chat_image(emb_np, b"what is the text in the picture?")
        gpu_lib.minigpt4_system_prompt(ctx)
        gpu_lib.minigpt4_begin_chat_image(ctx, emb, "what is the text in the picture?")
        for _ in range(16):
            assert np.array_equal(gpu_lib.amd_logits(ctx), chat.llm.logits)
            want = chat.end_chat(temp=0.0)[1].decode("utf-8", errors="replace")
            got = gpu_lib.minigpt4_end_chat_image(ctx, temp=0.0)
            assert got == want
    finally:
        gpu_lib.minigpt4_free(ctx)


def _encode(lib, ctx, img):
    from minigpt4_cpp_amd import minigpt4_library as ML
    emb = lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(img))
    got = np.ctypeslib.as_array(emb.data, shape=(emb.n_embeddings,)).copy().reshape(32, -1)
    lib.minigpt4_free_embedding(emb)
    return got


def test_image_encode_parity_mode_is_bit_identical_to_the_oracle(gpu_lib, tiny_files):
    """MINIGPT4_PARITY covers the image path too (Engine::encode_images_ref): patch embedding, ViT blocks, ln_vision, Q-Former and llama_proj with every fp32 chain in the
    oracle's order -- the embedding of minigpt4_encode_image equals OracleVision's bit for bit; the fast path stays within fp16-GEMM summation noise of it."""
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    vp, llm = tiny_files
    ctx = gpu_lib.minigpt4_model_load(vp, llm("q4_0"), verbosity=1, n_ctx=64, n_batch=32)
    try:
        o = R.OracleVision(G.read_vision_file(vp))
        for seed in (42, 7):
            img = G.synth_image(seed)
            want = o.encode(img)
            fast = _encode(gpu_lib, ctx, img)
            gpu_lib.amd_set_parity(ctx, True)
            par = _encode(gpu_lib, ctx, img)
            gpu_lib.amd_set_parity(ctx, False)
            assert np.array_equal(par, want), float(np.abs(par - want).max())
            assert np.abs(fast - want).max() <= 2e-3 * np.abs(want).max()
    finally:
        gpu_lib.minigpt4_free(ctx)


@pytest.mark.parametrize("qtype", ["q4_0", "q5_k", "q8_0"])
def test_quantised_vision_file_parity_mode_is_bit_identical(gpu_lib, tiny_files, tmp_path, qtype):
    """A vision file re-quantised by minigpt4_quantize_model (the generic image path; dims that hold whole k-quant super-blocks): parity mode equals the oracle's
    evaluation of the same file bit for bit."""
    import refcpu as R
    import test_gpu_quantized_vision as TQ
    from minigpt4_cpp_amd import modelgen as G
    _, llm = tiny_files
    src = str(tmp_path / "vision_f16.bin")
    G.write_vision_file(src, G.tiny_vision(n_embd_llm=4096, embed_dim=352, mlp_dim=512, q_inter=256), seed=21, std=0.05)
    qp = str(tmp_path / f"vision_{qtype}.bin")
    assert gpu_lib.library.minigpt4_quantize_model(src.encode(), qp.encode(), TQ.MG4[qtype]) == 0
    ctx = gpu_lib.minigpt4_model_load(qp, llm("q4_0"), verbosity=1, n_ctx=64, n_batch=32)
    try:
        gpu_lib.amd_set_parity(ctx, True)
        img = G.synth_image(11)
        want = R.OracleVision(G.read_vision_file(qp)).encode(img)
        par = _encode(gpu_lib, ctx, img)
        assert np.array_equal(par, want), float(np.abs(par - want).max())
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_parity_mode_chat_from_the_raw_image_is_bit_identical(gpu_lib, tmpdir_models):
    """The whole reference flow in parity mode -- encode_image (GPU, parity) -> system_prompt -> begin_chat_image -> 12 x end_chat_image -- against OracleVision + OracleChat:
    nothing is handed from one side to the other; image embedding, every logits vector and every greedy piece are identical."""
    import os
    import refcpu as R
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    vp = os.path.join(tmpdir_models, "vision_pm2.bin")
    lp = os.path.join(tmpdir_models, "llm_pm2.bin")
    G.write_vision_file(vp, G.tiny_vision(n_embd_llm=4096), seed=12, std=0.05)
    G.write_llm_file(lp, G.tiny_llm(wtype="q5_k", n_embd=4096, n_layer=2, n_head=32, n_vocab=512, mix="q5_k_m"), seed=4, std=0.02, **G.TINY_CONDITIONED)
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=256, n_batch=64)
    try:
        gpu_lib.amd_set_parity(ctx, True)
        img = G.synth_image(5)
        emb = gpu_lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(img))
        got = np.ctypeslib.as_array(emb.data, shape=(emb.n_embeddings,)).copy().reshape(32, -1)
        want = R.OracleVision(G.read_vision_file(vp)).encode(img)
        assert np.array_equal(got, want)
        chat = R.OracleChat(R.OracleLLM(G.read_llm_file(lp), n_ctx=256), n_batch=64)
        chat.system_prompt()
        chat.begin_chat_image(want, b"what is the text in the picture?")
        gpu_lib.minigpt4_system_prompt(ctx)
        gpu_lib.minigpt4_begin_chat_image(ctx, emb, "what is the text in the picture?")
        for _ in range(12):
            assert np.array_equal(gpu_lib.amd_logits(ctx), chat.llm.logits)
            assert gpu_lib.minigpt4_end_chat_image(ctx, temp=0.0) == chat.end_chat(temp=0.0)[1].decode("utf-8", errors="replace")
        gpu_lib.minigpt4_free_embedding(emb)
    finally:
        gpu_lib.minigpt4_free(ctx)


def _rms_boundary_row(K, rng):
    """A row whose sum of squares sits on a float rounding boundary of the mean: any parallel sum T has (float)(T(1-d)/K) != (float)(T(1+d)/K), so the kernels must take
    the literal element-order loop (k_rms_quant's / the EPI_REF prologue's fallback, ~1e-4 of real rows) to land on the oracle's side of the boundary."""
    x = rng.standard_normal(K).astype(np.float32)
    sq = (x * x).astype(np.float32).astype(np.float64)
    rest = 0.0
    for v in sq[:-1]:
        rest += v
    m0 = np.float32(rest / K)
    up = np.nextafter(m0, np.float32(np.inf))
    mid = (np.float64(m0) + np.float64(up)) / 2.0
    if mid * K <= rest:                                                  # the midpoint above rest / K
        mid = (np.float64(up) + np.float64(np.nextafter(up, np.float32(np.inf)))) / 2.0
    need = mid * K - rest                                                # what the last element's fl(x^2) has to add
    c0 = np.float32(np.sqrt(need))
    best, best_err = c0, np.inf
    c = c0
    for _ in range(4000):
        c = np.nextafter(c, np.float32(0))
    for _ in range(8000):
        err = abs(rest + np.float64(np.float32(c * c)) - mid * K)
        if err < best_err:
            best, best_err = c, err
        c = np.nextafter(c, np.float32(np.inf))
    x[-1] = best
    T = rest + np.float64(np.float32(best * best))
    d = K * 4e-16
    assert np.float32(T * (1 - d) / K) != np.float32(T * (1 + d) / K), "the crafted row does not straddle a boundary"
    return x


@pytest.mark.parametrize("wtype,K", [("q4_k", 512), ("q5_k", 5120), ("q6_k", 5120), ("q5_k", 13824), ("q6_k", 13824), ("q4_k", 4096)])
@pytest.mark.parametrize("prep,fuse", [(1, True), (1, False), (2, True), (2, False)])
def test_oracle_order_matvec_is_bit_identical(gpu_lib, wtype, K, prep, fuse):
    """The decode mat-vec in its oracle-order form (MATVEC_EPI_REF: block chain on the DPP crossbar, the oracle's rms mean in the prologue) against oracle/refcpu.c's
    quantise + mul_mat on the identically prepared row: every bit, 1 and 3 matrices per launch, with and without residual -- including a row that forces the rms
    interval check's fallback."""
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    from test_gpu_parity import _prepared_row
    t = Q.NAME_TO_TYPE[wtype]
    rng = np.random.default_rng(K * 11 + prep * 5 + int(fuse) + sum(map(ord, wtype)))
    n_mat = 3 if prep == 1 else 1
    rows = 2300 if K <= 5120 else 1100
    raw = Q.quantize(t, (0.03 * rng.standard_normal((n_mat * rows, K))).astype(np.float32))
    x2 = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float32) if prep == 1 else None
    res = rng.standard_normal(n_mat * rows).astype(np.float32) if prep == 2 else None
    for x in ((rng.standard_normal(K)).astype(np.float32),) + ((_rms_boundary_row(K, rng),) if prep == 1 else ()):
        got = gpu_lib.amd_test_matvec(t, raw, n_mat, K, rows, x, x2, prep=prep, fuse=fuse, epi=2, residual=res).reshape(-1)
        want = R.mul_mat(t, raw, K, n_mat * rows, _prepared_row(prep, x, x2, None)[None, :])[0]
        if res is not None:
            want = want + res
        assert np.array_equal(got, want), (wtype, K, prep, fuse, float(np.abs(got - want).max()))


@pytest.mark.parametrize("t1,K", [("q5_k", 5120), ("q4_k", 4096)])
@pytest.mark.parametrize("fuse", [False, True])
def test_oracle_order_mixed_type_launch_is_bit_identical(gpu_lib, t1, K, fuse):
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    from test_gpu_parity import _prepared_row
    ta, tb = Q.NAME_TO_TYPE[t1], Q.NAME_TO_TYPE["q6_k"]
    rng = np.random.default_rng(K * 3 + int(fuse) + 17)
    rows = 1500
    ra = Q.quantize(ta, (0.03 * rng.standard_normal((2 * rows, K))).astype(np.float32))
    rb = Q.quantize(tb, (0.03 * rng.standard_normal((rows, K))).astype(np.float32))
    x = rng.standard_normal(K).astype(np.float32)
    nw = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float32)
    got = gpu_lib.amd_test_matvec(ta, ra, 2, K, rows, x, nw, prep=1, fuse=fuse, epi=2, type2=tb, raw2=rb).reshape(-1)
    row = _prepared_row(1, x, nw, None)[None, :]
    want = np.concatenate([R.mul_mat(ta, ra, K, 2 * rows, row)[0], R.mul_mat(tb, rb, K, rows, row)[0]])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("wtype", ["q4_k", "q5_k", "q6_k", "q4_0"])
@pytest.mark.parametrize("case", [(142, 5120, 384, 1, True), (70, 13824, 256, 1, False), (33, 2048, 130, 3, False), (512, 2560, 128, 1, True), (5, 768, 70, 2, False)],
                         ids=lambda c: "N%d_K%d_R%d_m%d_res%d" % c)
def test_prompt_matmul_without_k_split_is_the_oracle_bit_for_bit(gpu_lib, wtype, case):
    """The int8-MFMA prompt kernels (mmq2) add an output's per-block fp32 terms block after block; with the K split forced to 1 that IS oracle/refcpu.c's order, so the
    result must equal the oracle's in every bit (integer block sums are exact in any order) -- this is how parity mode runs its prompt rows."""
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    N, K, rows, nm, with_res = case
    t = Q.NAME_TO_TYPE[wtype]
    rng = np.random.default_rng(sum(map(ord, wtype)) * 31 + sum(case))
    raw = Q.quantize(t, (0.05 * rng.standard_normal((nm * rows, K))).astype(np.float32))
    x = rng.standard_normal((N, K)).astype(np.float32)
    x[N // 2, : min(256, K)] = 0.0
    res = rng.standard_normal((nm, N, rows)).astype(np.float32) if with_res else None
    got = gpu_lib.amd_test_mmq2(t, raw, nm, K, rows, x, residual=res, ks=1)
    want = R.mul_mat(t, raw, K, nm * rows, x).reshape(N, nm, rows).transpose(1, 0, 2)
    if with_res:
        want = want + res
    assert np.array_equal(got, want), (wtype, case, float(np.abs(got - want).max()))
