"""GPU parity of the several-conversations-per-context path (include/minigpt4_amd.h "batched decode"):
B conversations with their own KV caches share one pass over the weights per decode step.  Every conversation must behave exactly like the
reference's single conversation (minigpt4.cpp:2513-2521 holds one per context): greedy pieces identical to an independent oracle chat, logits equal to
the single-conversation path of the same engine up to the fp32 summation order of a different mat-mul kernel (2e-3 of the logit range, the bar the
other whole-model tests use).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


PROMPTS = [b"what is the text in the picture?", b"describe the colours", b"hello", b"and now something longer to shift the positions apart", b"a", b"b c d", b"zzz", b"tell me more"]


def oracle_chats(lp, prompts, n_ctx):
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    f = G.read_llm_file(lp)
    chats = []
    for p in prompts:
        c = R.OracleChat(R.OracleLLM(f, n_ctx=n_ctx), n_batch=32)
        c.system_prompt()
        c.begin_chat(p)
        chats.append(c)
    return chats


def start(lib, ctx, prompts):
    lib.amd_set_conversations(ctx, len(prompts))
    for s, p in enumerate(prompts):
        lib.amd_select_conversation(ctx, s)
        lib.minigpt4_system_prompt(ctx)
        lib.minigpt4_begin_chat(ctx, p.decode())
    lib.amd_select_conversation(ctx, 0)


@pytest.mark.parametrize("wtype,mix", [("q4_0", "none"), ("q5_k", "q5_k_m"), ("f16", "none")])
@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 8])        # <= 4: v_dot4 4-token tiles; >= 5: int8-MFMA tiles (k-quants / Q4_0)
def test_batched_greedy_equals_independent_oracle_chats(gpu_lib, tiny_files, wtype, mix, B):
    """Conditioned model (decisive greedy choices): EVERY conversation's 8 greedy pieces equal its own independent oracle chat -- no near-tie allowance."""
    vp, llm = tiny_files
    lp = llm(wtype, mix, conditioned=True)
    prompts = PROMPTS[:B]
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=0, n_ctx=256, n_batch=32)
    try:
        start(gpu_lib, ctx, prompts)
        assert gpu_lib.library.minigpt4_amd_n_conversations(ctx.ptr) == B
        got = [[] for _ in range(B)]
        for _ in range(8):
            for s, piece in enumerate(gpu_lib.amd_end_chat_batch(ctx, list(range(B)), temp=0.0)):
                got[s].append(piece)
        chats = oracle_chats(lp, prompts, 256)
        for s, c in enumerate(chats):
            want = [c.end_chat(temp=0.0)[1].decode("utf-8", errors="replace") for _ in range(8)]
            assert got[s] == want, (s, got[s], want)
        for s in range(B):                                 # positions advanced per conversation
            gpu_lib.amd_select_conversation(ctx, s)
            assert gpu_lib.library.minigpt4_amd_n_past(ctx.ptr) == chats[s].llm.n_past
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_batch_logits_equal_single_conversation_path(gpu_lib, tiny_files):
    vp, llm = tiny_files
    lp = llm("q5_k", "q5_k_m")
    prompts = PROMPTS[:3]
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=0, n_ctx=256, n_batch=32)
    ref = gpu_lib.minigpt4_model_load(vp, lp, verbosity=0, n_ctx=256, n_batch=32)
    try:
        start(gpu_lib, ctx, prompts)
        for _ in range(3):
            gpu_lib.amd_end_chat_batch(ctx, [0, 1, 2], temp=0.0)
        for s, p in enumerate(prompts):
            gpu_lib.minigpt4_reset_chat(ref)
            gpu_lib.minigpt4_system_prompt(ref)
            gpu_lib.minigpt4_begin_chat(ref, p.decode())
            for _ in range(3):
                gpu_lib.minigpt4_end_chat(ref, temp=0.0)
            want = gpu_lib.amd_logits(ref)
            gpu_lib.amd_select_conversation(ctx, s)
            got = gpu_lib.amd_logits(ctx)
            assert float(np.abs(got - want).max() / (want.max() - want.min())) < 2e-3
    finally:
        gpu_lib.minigpt4_free(ctx)
        gpu_lib.minigpt4_free(ref)


@pytest.mark.parametrize("B", [2, 4])
def test_mixed_type_layer_in_one_launch_full_width(gpu_lib, B, monkeypatch):
    """k_matvec_tn_mix (wq|wk Q5_K + wv Q6_K of a "more bits" layer in ONE launch) at the 13B width (K = 5120: the 3-units-per-lane register tiling; B = 2 prepares
    the rows inside the launch, B = 4 uses the standalone preparation): every row's dot products are summed in the same order as in the two separate launches
    (MINIGPT4_BATCH_MIX=0), so the logits must be bit-identical to that form -- and agree with the single-conversation path within this file's own re-rounding noise."""
    import headline as H
    vp, lp = H.headline_files("13b_l2")                        # layer 0 is a "more bits" layer, layer 1 a plain Q5_K one
    prompts = PROMPTS[:B]

    def run(mix, ri="0"):
        monkeypatch.setenv("MINIGPT4_BATCH_MIX", mix)
        monkeypatch.setenv("MINIGPT4_RI", ri)          # "0": the v_dot4 launches this identity is about; "1": from 3 rows the MFMA launch k_matvec_ri_mix takes the mixed layer
        ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=0, n_ctx=256, n_batch=64)
        try:
            start(gpu_lib, ctx, prompts)
            for _ in range(3):
                gpu_lib.amd_end_chat_batch(ctx, list(range(B)), temp=0.0)
            out = []
            for sl in range(B):
                gpu_lib.amd_select_conversation(ctx, sl)
                out.append(gpu_lib.amd_logits(ctx).copy())
            return np.stack(out)
        finally:
            gpu_lib.minigpt4_free(ctx)

    one, two = run("1"), run("0")
    assert np.isfinite(one).all() and np.array_equal(one, two)
    mfma = run("1", "1")                                # round 5: the product default (at B = 4 the mixed layer on the matrix cores; at B = 2 the same launches as `one`)
    assert np.isfinite(mfma).all() and (B > 2 or np.array_equal(mfma, one))
    monkeypatch.delenv("MINIGPT4_BATCH_MIX")
    monkeypatch.delenv("MINIGPT4_RI")
    ref = gpu_lib.minigpt4_model_load(vp, lp, verbosity=0, n_ctx=256, n_batch=64)
    try:
        for sl, p in enumerate(prompts):
            gpu_lib.minigpt4_reset_chat(ref)
            gpu_lib.minigpt4_system_prompt(ref)
            gpu_lib.minigpt4_begin_chat(ref, p.decode())
            for _ in range(3):
                gpu_lib.minigpt4_end_chat(ref, temp=0.0)
            want = gpu_lib.amd_logits(ref)
            err = float(np.abs(one[sl] - want).max() / (want.max() - want.min()))
            err_mfma = float(np.abs(mfma[sl] - want).max() / (want.max() - want.min()))
            print(f"B={B} conversation {sl}: batched vs single-conversation logits {err:.2e} of the range (MFMA launches: {err_mfma:.2e})")
            # two GPU summation orders of the same exact integer dots: each differs from the oracle by <= 1e-2 of the largest |logit| (test_configs3_operating_point_...,
            # test_gpu_headline.py: observed 5-6e-3), so from each other by at most the sum; of the RANGE (about 1.4 x the largest |logit| on this file) that is <= 1.5e-2
            assert err < 1.5e-2 and err_mfma < 1.5e-2
    finally:
        gpu_lib.minigpt4_free(ref)


@pytest.mark.parametrize("config,B,steps", [("13b_l2", 3, 16), ("13b_l2", 4, 16), ("13b", 4, 32), ("13b_l2", 2, 16), ("13b_l2", 5, 12), ("13b_l2", 8, 8)])
def test_configs3_operating_point_matches_independent_oracle_chats(gpu_lib, config, B, steps):
    """BASELINE.json configs[3]'s per-GPU operating point AS THE ENGINE RUNS IT (round-5 verdict, missing #1): B = 3 / 4 image conversations per replica at the 13B width with
    MINIGPT4_RI at its default, i.e. the batched step on `k_matvec_ri` / `k_matvec_ri_mix` (row-interleaved weight image, v_mfma_i32_4x4x4i8) + the K-split w2 launch at
    B = 4 -- asserted from the launch counters, so a silent fallback to the v_dot4 launches cannot pass for it.  Every conversation is compared with ITS OWN independent
    OracleChat (reference: one conversation per context, minigpt4.cpp:2513-2521, 2704-2718) by the criteria of the single-conversation headline test
    (oracle/headline.py::compare): free-running greedy pieces identical at every step, teacher-forced logits of every BATCHED step within 1e-2 of the largest |logit|,
    argmax identical on every step.  13b_l2 = the 40-layer graph cut to two layers (layer 0 a "more bits" layer -> the mixed-type launch, layer 1 plain Q5_K); 13b = the
    full 40-layer file, 32 steps."""
    import json
    import os
    import headline as H
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    assert os.environ.get("MINIGPT4_RI", "1") != "0" and os.environ.get("MINIGPT4_RI_W2", "1") != "0"
    vp, lp = H.headline_files(config)
    threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8, 32))
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=512, n_batch=512)
    try:
        embs = gpu_lib.amd_encode_images(ctx, [G.synth_image(200 + i) for i in range(B)])           # a different image per conversation
        prompts = H.BATCH_PROMPTS[:B]
        res = H.batched_vs_oracle(gpu_lib, ctx, lp, embs, prompts, steps, n_ctx=512, threads=threads)
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        json.dump(res, open(os.path.join(d, f"parity_observed_batched_{config}_b{B}.json"), "w"), indent=1, sort_keys=True)
        print(config, B, json.dumps({k: v for k, v in res.items() if k != "per_conversation"}))
        path = res["launches_of_the_batched_step"]
        n_layer = 2 if config.endswith("_l2") else 40
        if B in (3, 4):
            # the operating point itself: same-type sets and the output matrix on k_matvec_ri, every "more bits" layer's wq|wk + wv on k_matvec_ri_mix, w2 on the K-split form at B = 4
            assert path["rows"] == B and path["ri"] >= n_layer + 1 and path["ri_mix"] >= 1 and path["dot4_mix"] == 0 and path["mul_mat"] == 0 and path["sets"] == 0, path
            assert (path["ri_ksplit"] == n_layer) if B == 4 else (path["ri_ksplit"] == 0), path
        elif B == 2:
            # two conversations (round 6: the other batch sizes at the 13B width against the oracle too): the v_dot4 multi-row launches, rows prepared in their prologues
            assert path["rows"] == 2 and path["ri"] == 0 and path["ri_mix"] == 0 and path["dot4"] + path["dot4_mix"] >= 4 * n_layer and path["sets"] == 0, path
        else:
            # five and more: the prompt pass's int8-MFMA set launches, every layer
            assert path["rows"] == B and path["sets"] == n_layer and path["ri"] == 0 and path["ri_mix"] == 0, path
        assert res["launches_free_running_step"] == path
        assert len(set(res["prompt_tokens"])) > 1                                  # the conversations sit at different positions
        assert res["free_running_identical_min"] == steps, res
        assert res["teacher_forced_argmax_identical_min"] == steps, res
        assert res["max_logit_rel"] <= 1e-2, res
        assert res["decided_min"] >= steps * 3 // 4 and res["decided_argmax_mismatches"] == 0, res
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_interleaving_single_and_batched_steps_reset_and_subsets(gpu_lib, tiny_files):
    """Batched steps, single-conversation steps (the reference entry points on the selected conversation), a subset batch in a different order and a
    reset of one conversation interleave freely; each conversation still equals its own oracle chat."""
    vp, llm = tiny_files
    lp = llm("q4_0")
    prompts = PROMPTS[:4]
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=0, n_ctx=256, n_batch=32)
    try:
        start(gpu_lib, ctx, prompts)
        chats = oracle_chats(lp, prompts, 256)
        got = [[] for _ in range(4)]
        want = [[] for _ in range(4)]

        def step_oracle(slots):
            for s in slots:
                want[s].append(chats[s].end_chat(temp=0.0)[1].decode("utf-8", errors="replace"))

        def step_batch(slots):
            for s, piece in zip(slots, gpu_lib.amd_end_chat_batch(ctx, slots, temp=0.0)):
                got[s].append(piece)
            step_oracle(slots)

        step_batch([0, 1, 2, 3])
        step_batch([3, 1])                                  # subset, permuted
        gpu_lib.amd_select_conversation(ctx, 2)
        for _ in range(2):                                  # single-conversation decode through the reference ABI (hipGraph of conversation 2)
            got[2].append(gpu_lib.minigpt4_end_chat(ctx, temp=0.0))
        step_oracle([2])
        step_oracle([2])
        step_batch([0, 1, 2, 3])
        # conversation 1 starts over; the others keep their caches
        gpu_lib.amd_select_conversation(ctx, 1)
        gpu_lib.minigpt4_reset_chat(ctx)
        gpu_lib.minigpt4_system_prompt(ctx)
        gpu_lib.minigpt4_begin_chat(ctx, "fresh start")
        import refcpu as R
        from minigpt4_cpp_amd import modelgen as G
        chats[1] = R.OracleChat(R.OracleLLM(G.read_llm_file(lp), n_ctx=256), n_batch=32)
        chats[1].system_prompt()
        chats[1].begin_chat(b"fresh start")
        step_batch([0, 1, 2, 3])
        step_batch([2, 0, 1])
        assert got == want
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_full_context_is_sampled_but_not_advanced_and_bad_arguments(gpu_lib, tiny_files):
    vp, llm = tiny_files
    ctx = gpu_lib.minigpt4_model_load(vp, llm("q4_0"), verbosity=0, n_ctx=64, n_batch=32)
    try:
        gpu_lib.amd_set_conversations(ctx, 2)
        gpu_lib.amd_select_conversation(ctx, 0)
        gpu_lib.amd_eval_tokens(ctx, [1] + [5] * 62)        # one free position
        gpu_lib.amd_select_conversation(ctx, 1)
        gpu_lib.amd_eval_tokens(ctx, [1, 7, 9])
        for k in range(3):
            gpu_lib.amd_end_chat_batch(ctx, [0, 1], temp=0.0)
        gpu_lib.amd_select_conversation(ctx, 0)
        assert gpu_lib.library.minigpt4_amd_n_past(ctx.ptr) == 64           # filled by the first step, then left alone
        gpu_lib.amd_select_conversation(ctx, 1)
        assert gpu_lib.library.minigpt4_amd_n_past(ctx.ptr) == 6
        with pytest.raises(RuntimeError):
            gpu_lib.amd_end_chat_batch(ctx, [0, 0], temp=0.0)                # duplicates
        with pytest.raises(RuntimeError):
            gpu_lib.amd_end_chat_batch(ctx, [0, 2], temp=0.0)                # out of range
        with pytest.raises(RuntimeError):
            gpu_lib.amd_select_conversation(ctx, 2)
        with pytest.raises(RuntimeError):
            gpu_lib.amd_set_conversations(ctx, 65)
    finally:
        gpu_lib.minigpt4_free(ctx)


def test_seeded_sampling_in_a_batch_uses_the_host_sampler_per_conversation(gpu_lib, tiny_files):
    """temp > 0: every conversation's logits go through the same sampler chain as minigpt4_end_chat (one mt19937 stream per context, consumed in
    slot-list order) -- equal to sampling the same logits with the host-only hook."""
    vp, llm = tiny_files
    lp = llm("q4_0")
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=0, seed=77, n_ctx=256, n_batch=32)
    try:
        start(gpu_lib, ctx, PROMPTS[:2])
        gpu_lib.amd_select_conversation(ctx, 0)
        gpu_lib.library.minigpt4_amd_sync(ctx.ptr)
        lg = []
        for s in range(2):
            gpu_lib.amd_select_conversation(ctx, s)
            lg.append(gpu_lib.amd_logits(ctx))
        pieces = gpu_lib.amd_end_chat_batch(ctx, [0, 1], temp=0.7, top_k=20, top_p=0.95)
        assert len(pieces) == 2 and all(isinstance(p, str) for p in pieces)
        # the first draw of a fresh context with seed 77 on conversation 0's logits
        first = gpu_lib.amd_sample_logits(lg[0], 77, temp=0.7, top_k=20, top_p=0.95)
        voc = gpu_lib.library.minigpt4_amd_vocab_load(lp.encode())
        try:
            n = np.zeros(1, np.int32)
            import ctypes
            ptr = gpu_lib.library.minigpt4_amd_vocab_piece(voc, first, n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
            want = "</s>" if first == 2 else ctypes.string_at(ptr, int(n[0])).decode("utf-8", errors="replace")
        finally:
            gpu_lib.library.minigpt4_amd_vocab_free(voc)
        assert pieces[0] == want
    finally:
        gpu_lib.minigpt4_free(ctx)


# (type, K): K / 32 / 64 = units per lane 1, 2, 3, 6, 7 -> every register tiling of k_matvec_tn (7: the 4-wave variant); both activation formats (Q8_0 / Q8_K)
ROWS_CASES = [("q4_0", 512), ("q5_k", 512), ("q4_1", 4096), ("q4_k", 4096), ("q5_0", 5120), ("q5_k", 5120), ("q6_k", 5120), ("q5_1", 5120), ("q4_0", 11008), ("q5_k", 13824), ("q6_k", 13824)]


@pytest.mark.parametrize("wtype,K", ROWS_CASES)
@pytest.mark.parametrize("N,n_mat,with_res", [(1, 1, True), (2, 3, False), (3, 2, False), (4, 1, True), (4, 3, False)])
def test_multi_row_matvec_matches_oracle(gpu_lib, wtype, K, N, n_mat, with_res):
    """k_matvec_tn against the oracle's quantise + mul_mat, row by row: integer block dots exact, fp32 order differs -> 2e-5 of the row maximum (the bar of
    test_mul_mat_matches_oracle); rows > waves so that every wave pipelines several groups; ragged row counts."""
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    t = Q.NAME_TO_TYPE[wtype]
    rng = np.random.default_rng(K * 11 + N * 5 + n_mat + sum(map(ord, wtype)))
    rows = 2300 if K <= 5120 else 1100
    w = (0.03 * rng.standard_normal((n_mat * rows, K))).astype(np.float32)
    raw = Q.quantize(t, w)
    x = rng.standard_normal((N, K)).astype(np.float32)
    x[0, :min(K, 256)] = 0.0                                  # an all-zero activation block
    res = rng.standard_normal((n_mat, N, rows)).astype(np.float32) if with_res else None
    got = gpu_lib.amd_test_matvec_rows(t, raw, n_mat, K, rows, x, res)
    want = R.mul_mat(t, raw, K, n_mat * rows, x).reshape(N, n_mat, rows).transpose(1, 0, 2)
    if res is not None:
        want = want + res
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), (wtype, K, N, n_mat)


# rows: multiples of 64 (the row-interleaved groups); 2304 rows x 3 matrices = 108 groups (< CUs: the 8-wave form), 5120 x 2 = 160, 13824 x 2 = 432 (>= CUs: 4 waves, two
# workgroups per CU); K = 256 (one super-block: three of the four waves idle), 5120 / 13824 (the 13B widths: 20 / 54 super-blocks over 4 or 8 waves, ragged)
RI_CASES = [("q4_k", 256, 128, 1), ("q5_k", 512, 192, 2), ("q6_k", 768, 64, 3), ("q4_k", 4096, 2304, 3), ("q5_k", 5120, 2304, 3), ("q5_k", 5120, 5120, 2), ("q6_k", 5120, 2304, 1),
            ("q5_k", 13824, 1152, 1), ("q6_k", 13824, 1152, 1), ("q5_k", 5120, 13824, 2)]


@pytest.mark.parametrize("wtype,K,rows,n_mat", RI_CASES)
@pytest.mark.parametrize("N,with_res", [(1, False), (2, True), (3, False), (4, True)])
def test_row_interleaved_mfma_matvec_matches_oracle(gpu_lib, wtype, K, rows, n_mat, N, with_res):
    """The batched decode's mat-vec of round 5 (csrc/ri_kernels.hip: v_mfma_i32_4x4x4_16B_i8, lane = weight row, row-interleaved image built from the ordinary planes)
    against the oracle's quantise + mul_mat: exact integer sub-block dots, the fp32 super-block terms of a K range added in order and the ranges in wave order -> 2e-5 of
    the row maximum, the bar of every other mat-vec kernel; and agreement with the v_dot4 multi-row kernel (k_matvec_tn) on the same planes to that bar."""
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    t = Q.NAME_TO_TYPE[wtype]
    rng = np.random.default_rng(K * 13 + rows + N * 5 + n_mat + sum(map(ord, wtype)))
    w = (0.03 * rng.standard_normal((n_mat * rows, K))).astype(np.float32)
    w[3, :] = 0.11                                            # a constant row: scales 0, mins maximal
    raw = Q.quantize(t, w)
    x = rng.standard_normal((N, K)).astype(np.float32)
    x[0, :min(K, 256)] = 0.0                                  # an all-zero activation block
    if N > 1:
        x[1] *= 37.0                                          # rows of very different magnitude: per-row scales
    res = rng.standard_normal((n_mat, N, rows)).astype(np.float32) if with_res else None
    got = gpu_lib.amd_test_matvec_ri(t, raw, n_mat, K, rows, x, res)
    want = R.mul_mat(t, raw, K, n_mat * rows, x).reshape(N, n_mat, rows).transpose(1, 0, 2)
    if res is not None:
        want = want + res
    assert got.shape == want.shape and np.isfinite(got).all()
    scale = np.abs(want).max(axis=2, keepdims=True)
    assert (np.abs(got - want) <= 2e-5 * scale).all(), (wtype, K, rows, N, float((np.abs(got - want) / scale).max()))
    # the same launch preparing its rows itself: rms_norm(x_t) * w (ggml: fp32 squares summed in double, eps 1e-6) + Q8_K quantisation inside every workgroup
    if K >= 512:
        nw = (1.0 + 0.2 * rng.standard_normal(K)).astype(np.float32)
        rows_n = np.empty_like(x)
        for ti in range(N):
            ssq = np.float64(0)
            for v in x[ti]:
                ssq += np.float64(np.float32(v * v))
            rows_n[ti] = (x[ti] * (np.float32(1.0) / np.sqrt(np.float32(np.float32(ssq / K) + np.float32(1e-6)), dtype=np.float32))) * nw
        got2 = gpu_lib.amd_test_matvec_ri(t, raw, n_mat, K, rows, x, res, rms_w=nw)
        want2 = R.mul_mat(t, raw, K, n_mat * rows, rows_n).reshape(N, n_mat, rows).transpose(1, 0, 2)
        if res is not None:
            want2 = want2 + res
        scale2 = np.abs(want2).max(axis=2, keepdims=True)
        assert np.isfinite(got2).all() and (np.abs(got2 - want2) <= 2e-5 * scale2).all(), (wtype, K, rows, N, float((np.abs(got2 - want2) / scale2).max()))



@pytest.mark.parametrize("ta,K,rows", [("q5_k", 512, 128), ("q4_k", 4096, 4096), ("q5_k", 5120, 5120), ("q5_k", 5120, 2304)])
@pytest.mark.parametrize("N", [1, 3, 4])
def test_row_interleaved_mixed_type_launch_matches_the_single_type_launches(gpu_lib, ta, K, rows, N):
    """A "more bits" layer's wq | wk (Q4_K / Q5_K) + wv (Q6_K) in one MFMA launch (k_matvec_ri_mix): the same per-group stream as k_matvec_ri with the digit images of both
    types staged -> BIT-identical to the two single-type launches when the wave split of K is the same (both take the 8-wave or both the 4-wave form), the oracle's bar
    otherwise; 3 x 5120 rows = 240 groups (< CUs: 8 waves), 3 x 4096 = 192, 3 x 2304 = 108."""
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    tA, tB = Q.NAME_TO_TYPE[ta], Q.NAME_TO_TYPE["q6_k"]
    rng = np.random.default_rng(K * 7 + rows + N + sum(map(ord, ta)))
    wa = (0.03 * rng.standard_normal((2 * rows, K))).astype(np.float32)
    wb = (0.03 * rng.standard_normal((rows, K))).astype(np.float32)
    wb[5, :] = -0.07
    raw_a, raw_b = Q.quantize(tA, wa), Q.quantize(tB, wb)
    x = rng.standard_normal((N, K)).astype(np.float32)
    if N > 1:
        x[1] *= 19.0
    got = gpu_lib.amd_test_matvec_ri_mixed(tA, raw_a, 2, tB, raw_b, 1, K, rows, x)
    want = np.concatenate([R.mul_mat(tA, raw_a, K, 2 * rows, x).reshape(N, 2, rows).transpose(1, 0, 2), R.mul_mat(tB, raw_b, K, rows, x).reshape(N, 1, rows).transpose(1, 0, 2)])
    assert got.shape == want.shape and np.isfinite(got).all()
    scale = np.abs(want).max(axis=2, keepdims=True)
    assert (np.abs(got - want) <= 2e-5 * scale).all(), (ta, K, rows, N, float((np.abs(got - want) / scale).max()))
    # against the single-type launches of the same images
    one_a, one_b = gpu_lib.amd_test_matvec_ri(tA, raw_a, 2, K, rows, x), gpu_lib.amd_test_matvec_ri(tB, raw_b, 1, K, rows, x)
    single = np.concatenate([one_a, one_b])
    assert (np.abs(got - single) <= 2e-5 * scale).all()
