"""TEST INFRASTRUCTURE (never imported by the product): GPU-vs-oracle comparison at the BASELINE.json headline shapes.

Used by tests/test_gpu_headline.py and by bench.py's `cpu_baseline` leg (the `parity` object of the bench line).  The reference call sequence is
`minigpt4_system_prompt -> minigpt4_begin_chat_image -> K x minigpt4_end_chat_image(temp 0)` (reference minigpt4.cpp:2671-2718, 2720-2732;
examples/main.cpp:207-293); the oracle side is OracleChat over OracleLLM (oracle/refcpu.py, oracle/refcpu.c).

Three comparisons on the same file, the same image embedding and the same prompt:
  * parity mode (MINIGPT4_PARITY, oracle-order fp32 accumulation): free-running, logits BIT-IDENTICAL and greedy ids identical at every step (gpu_parity_mode_run);
  * free-running: both sides decode greedily on their own; the piece sequences are compared (north_star: "bit-exact token ids under greedy sampling");
  * teacher-forced: the GPU is fed the ORACLE's token at every step, so every step's logits are comparable even after a near-tie would have made the
    free-running sequences part ways; reported as max |delta| / (max - min of the oracle's logits) per step, plus whether the argmax agrees and whether the
    oracle's own top-2 margin exceeds the observed noise ("decided" steps).
"""
import os
import time
from typing import Dict, List, Optional

import numpy as np

PROMPT = "what is the text in the picture?"   # reference examples/main.cpp:61


def model_dir() -> str:
    for d in ("/dev/shm", "/tmp"):
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize > 14e9:
                p = os.path.join(d, "mg4_bench")
                os.makedirs(p, exist_ok=True)
                return p
        except OSError:
            pass
    p = os.path.join("/tmp", "mg4_bench")
    os.makedirs(p, exist_ok=True)
    return p


def headline_files(config: str):
    """The synthetic files bench.py measures (same generator arguments -- modelgen.headline_llm -- same cache directory): returns (vision_path, llm_path)."""
    from minigpt4_cpp_amd import modelgen as G
    d = model_dir()
    lcfg, kw = G.headline_llm(config)
    vcfg = G.vision_7b() if config.startswith("7b") else G.vision_13b()
    vname = "7b" if config.startswith("7b") else "13b"          # the vision file depends on the LLM width only
    vp, lp = os.path.join(d, f"vision_{vname}.bin"), os.path.join(d, f"llm_{config}_r3.bin")
    if not os.path.exists(vp + ".ok"):
        G.write_vision_file(vp, vcfg, seed=4321, std=0.02, unique_blocks=1, fast=True)
        open(vp + ".ok", "w").write("ok")
    if not os.path.exists(lp + ".ok"):
        G.write_llm_file(lp, lcfg, **kw)
        open(lp + ".ok", "w").write("ok")
    return vp, lp


def oracle_run(lp: str, embedding: np.ndarray, steps: int, prompt: str = PROMPT, n_ctx: int = 512, native: bool = False, threads: Optional[int] = None,
               teacher_ids: Optional[List[int]] = None) -> Dict:
    """Oracle side of the chat flow: per-step logits (before sampling), greedy ids and pieces.  teacher_ids: feed these ids instead of the run's own greedy ones
    (the self-noise run of oracle_self_noise)."""
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    f = G.read_llm_file(lp, in_memory=True)
    if threads:
        R.lib(native).orc_set_threads(int(threads))
    llm = R.OracleLLM(f, n_ctx=n_ctx, native=native)
    chat = R.OracleChat(llm, n_batch=512)
    t0 = time.time()
    chat.system_prompt()
    chat.begin_chat_image(embedding, prompt.encode())
    prefill_s = time.time() - t0
    n_prompt = llm.n_past
    logits, ids, pieces = [], [], []
    t0 = time.time()
    for i in range(steps):
        logits.append(llm.logits.copy())
        if teacher_ids is not None:
            tid, piece = int(teacher_ids[i]), b""
            llm.eval_tokens([tid])
        else:
            tid, piece = chat.end_chat(temp=0.0)
        ids.append(int(tid))
        pieces.append(piece.decode("utf-8", errors="replace"))
    return {"logits": np.stack(logits), "ids": ids, "pieces": pieces, "n_prompt": n_prompt, "prefill_s": prefill_s, "decode_s": time.time() - t0}


def oracle_self_noise(lp: str, embedding: np.ndarray, base: Dict, eps: float = 1e-6, **kw) -> Dict:
    """The reference arithmetic's own conditioning on this file: the oracle against ITSELF with the image embedding perturbed by `eps` relative (the size of fp32
    summation-order differences), teacher-forced on the unperturbed run's ids.  Every activation row is rounded to int8 before every mat-mul (ggml's Q8_K / Q8_0),
    so a 1e-6 perturbation flips a few roundings, the flips decorrelate the rounding errors of the two runs layer by layer, and a deep stack ends up differing by its
    whole accumulated quantisation noise.  Any implementation that does not add the fp32 terms in ggml's exact order differs from it by this much; a GPU-vs-oracle
    difference can only be judged against it."""
    rng = np.random.default_rng(12345)
    emb2 = (np.asarray(embedding, np.float32) * (1.0 + eps * rng.standard_normal(embedding.shape))).astype(np.float32)
    pert = oracle_run(lp, emb2, len(base["ids"]), teacher_ids=base["ids"], **kw)
    ol, pl = base["logits"].astype(np.float64), pert["logits"].astype(np.float64)
    rel = np.abs(pl - ol).max(axis=1) / (ol.max(axis=1) - ol.min(axis=1))
    return {"eps": eps, "max_logit_rel_range": float(rel.max()), "mean_logit_rel_range": float(rel.mean()), "argmax_identical": int((pl.argmax(axis=1) == ol.argmax(axis=1)).sum())}


def oracle_order_spread(lp: str, embedding: np.ndarray, base: Dict, native: bool = False, **kw) -> Dict:
    """The oracle against ITSELF in its other fp32 accumulation order (refcpu.c `orc_set_order(1)`: ggml's own x86 structure as best recalled -- eight lane partials per
    row, fmadd per block, hsum at the end -- where `base` was computed in order 0, one fma chain per output), same image embedding, teacher-forced on base's ids.  Neither
    order is pinned to a ggml binary; the spread between them is what "bit-exact against the reference" can mean at best on this file, and it is the yardstick for the
    GPU's fast mode (which differs from BOTH only by its own summation order).  The GPU's parity mode reproduces order 0."""
    import refcpu as R
    L = R.lib(native)
    L.orc_set_order(1)
    try:
        alt = oracle_run(lp, embedding, len(base["ids"]), teacher_ids=base["ids"], native=native, **kw)
    finally:
        L.orc_set_order(0)
    ol, al = base["logits"].astype(np.float64), alt["logits"].astype(np.float64)
    rel = np.abs(al - ol).max(axis=1) / np.abs(ol).max(axis=1)
    return {"steps": len(base["ids"]), "max_logit_rel": float(rel.max()), "mean_logit_rel": float(rel.mean()), "logits_bit_identical_steps": int((al == ol).all(axis=1).sum()),
            "argmax_identical": int((al.argmax(axis=1) == ol.argmax(axis=1)).sum()),
            "orders": "0: one fma chain per output (GPU parity mode = this, bit for bit); 1: ggml-style 8 lane partials + hsum (k-quant min term in its own 4-lane accumulator)"}


def gpu_free_run(lib, ctx, emb_struct, steps: int, prompt: str = PROMPT) -> List[str]:
    """The reference API flow on the GPU: pieces of `steps` greedy tokens (EOS ignored)."""
    lib.minigpt4_reset_chat(ctx)
    lib.minigpt4_system_prompt(ctx)
    lib.minigpt4_begin_chat_image(ctx, emb_struct, prompt)
    return [lib.minigpt4_end_chat_image(ctx, temp=0.0) for _ in range(steps)]


def gpu_teacher_forced(lib, ctx, emb_struct, ids: List[int], prompt: str = PROMPT) -> np.ndarray:
    """Logits of every decode step with the ORACLE's ids fed back (the same decode path as minigpt4_end_chat_image: one-row eval through the decode graph)."""
    lib.minigpt4_reset_chat(ctx)
    lib.minigpt4_system_prompt(ctx)
    lib.minigpt4_begin_chat_image(ctx, emb_struct, prompt)
    out = []
    for tid in ids:
        out.append(lib.amd_logits(ctx).copy())
        lib.amd_eval_tokens(ctx, [int(tid)])
    return np.stack(out)


def gpu_parity_mode_run(lib, ctx, emb_struct, oracle: Dict, prompt: str = PROMPT) -> Dict:
    """The same chat flow with the engine in parity mode (MINIGPT4_PARITY: every fp32 accumulation in the oracle's order), FREE-RUNNING: the logits behind every sampled
    token must equal the oracle's bit for bit and every greedy piece must be the oracle's.  Returns counts (and the first mismatching step, -1 when none)."""
    lib.amd_set_parity(ctx, True)
    try:
        lib.minigpt4_reset_chat(ctx)
        lib.minigpt4_system_prompt(ctx)
        lib.minigpt4_begin_chat_image(ctx, emb_struct, prompt)
        n = len(oracle["ids"])
        bit, ids, first, worst = 0, 0, -1, 0.0
        for i in range(n):
            lg = lib.amd_logits(ctx)
            same = bool(np.array_equal(lg, oracle["logits"][i]))
            bit += same
            if not same:
                worst = max(worst, float(np.abs(lg.astype(np.float64) - oracle["logits"][i]).max()))
            piece = lib.minigpt4_end_chat_image(ctx, temp=0.0)
            ok = piece == oracle["pieces"][i]
            ids += ok
            if first < 0 and not (same and ok):
                first = i
            if not ok:                       # the two sequences have parted: later steps are not comparable
                break
        return {"steps": n, "logits_bit_identical": bit, "greedy_ids_identical": ids, "first_mismatch": first, "max_abs_logit_delta": worst}
    finally:
        lib.amd_set_parity(ctx, False)
        lib.minigpt4_reset_chat(ctx)


def compare(oracle: Dict, gpu_pieces: List[str], gpu_logits: np.ndarray) -> Dict:
    ol = oracle["logits"].astype(np.float64)
    gl = gpu_logits.astype(np.float64)
    rng = ol.max(axis=1) - ol.min(axis=1)
    rel = np.abs(gl - ol).max(axis=1) / rng                                   # per step, relative to the oracle's logit range
    rel_max = np.abs(gl - ol).max(axis=1) / np.abs(ol).max(axis=1)            # per step, relative to the largest |logit| (north_star's "1e-2 relative")
    noise = float(np.abs(gl - ol).max())
    srt = np.sort(ol, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    decided = margin > 2.0 * noise                                            # the oracle's top-2 gap exceeds twice the largest observed logit difference
    arg_eq = gl.argmax(axis=1) == ol.argmax(axis=1)
    n = len(oracle["ids"])
    first_div = next((i for i in range(n) if gpu_pieces[i] != oracle["pieces"][i]), n)
    return {"tokens_compared": n, "free_running_identical": int(sum(a == b for a, b in zip(gpu_pieces, oracle["pieces"]))), "free_running_first_divergence": first_div,
            "teacher_forced_argmax_identical": int(arg_eq.sum()), "decided": int(decided.sum()), "decided_argmax_identical": int((arg_eq & decided).sum()),
            "max_logit_rel_range": float(rel.max()), "max_logit_rel": float(rel_max.max()), "mean_logit_rel_range": float(rel.mean()),
            "min_top2_margin_over_range": float((margin / rng).min()), "prompt_tokens": int(oracle["n_prompt"])}


# ---- BASELINE.json configs[3]'s per-GPU operating point: B conversations per replica decoded in ONE pass over the weights per step -----------------------------------------
# Reference behaviour to match: one INDEPENDENT conversation per context (minigpt4.cpp:2513-2521 holds one n_past / one KV cache; :2704-2718 end_chat_image), so every
# conversation of a batched replica must behave like its own OracleChat.

BATCH_PROMPTS = [PROMPT, "hello", "describe the colours of the image", "and now something longer to shift the positions of this conversation apart from the others",
                 "a", "what do you see?", "list every object in the scene, one per line", "zzz"]


def embedding_struct(emb_np: np.ndarray):
    """A MiniGPT4Embedding over a float32 [32, n_embd] array (the array is kept alive by the returned pair)."""
    import ctypes
    from minigpt4_cpp_amd import minigpt4_library as ML
    a = np.ascontiguousarray(emb_np, np.float32)
    st = ML.MiniGPT4Embedding(a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.size)
    return st, a


def gpu_batched_start(lib, ctx, embeddings: List[np.ndarray], prompts: List[str]) -> None:
    """system_prompt + begin_chat_image on conversation i with embeddings[i] / prompts[i] (the prompt rows are only queued: the first batched step runs each conversation's
    own prefill pass, then the shared decode pass)."""
    B = len(prompts)
    if lib.library.minigpt4_amd_n_conversations(ctx.ptr) != B:
        lib.amd_set_conversations(ctx, B)
    for sl in range(B):
        lib.amd_select_conversation(ctx, sl)
        lib.minigpt4_reset_chat(ctx)
        lib.minigpt4_system_prompt(ctx)
        st, keep = embedding_struct(embeddings[sl])
        lib.minigpt4_begin_chat_image(ctx, st, prompts[sl])
        del keep
    lib.amd_select_conversation(ctx, 0)


def gpu_batched_free_run(lib, ctx, embeddings, prompts, steps: int) -> List[List[str]]:
    """`steps` batched greedy steps (minigpt4_amd_end_chat_batch, EOS ignored): pieces[conversation][step]."""
    gpu_batched_start(lib, ctx, embeddings, prompts)
    B = len(prompts)
    out: List[List[str]] = [[] for _ in range(B)]
    for _ in range(steps):
        for sl, piece in enumerate(lib.amd_end_chat_batch(ctx, list(range(B)), temp=0.0)):
            out[sl].append(piece)
    return out


def gpu_batched_teacher_forced(lib, ctx, embeddings, prompts, ids: List[List[int]]) -> np.ndarray:
    """Logits [conversation][step][n_vocab] of the BATCHED step with every conversation fed ITS oracle's ids (minigpt4_amd_eval_batch): step 0's logits come from each
    conversation's own prefill pass, every later step's from the shared batched pass."""
    gpu_batched_start(lib, ctx, embeddings, prompts)
    B, steps = len(prompts), len(ids[0])
    out = None
    for i in range(steps):
        for sl in range(B):
            lib.amd_select_conversation(ctx, sl)
            lg = lib.amd_logits(ctx)
            if out is None:
                out = np.empty((B, steps, lg.shape[0]), np.float32)
            out[sl, i] = lg
        lib.amd_eval_batch(ctx, list(range(B)), [ids[sl][i] for sl in range(B)])
    lib.amd_select_conversation(ctx, 0)
    return out


def batched_vs_oracle(lib, ctx, lp: str, embeddings: List[np.ndarray], prompts: List[str], steps: int, n_ctx: int = 512, threads: Optional[int] = None,
                      oracles: Optional[List[Dict]] = None) -> Dict:
    """B conversations of one context against B independent oracle conversations on the same file: free-running pieces + teacher-forced logits per conversation (compare()),
    the worst of each figure over the conversations, and the launch kinds the batched step took (minigpt4_amd_batch_path).  `oracles`: reuse oracle_run results."""
    B = len(prompts)
    if oracles is None:
        oracles = [oracle_run(lp, embeddings[i], steps, prompt=prompts[i], n_ctx=n_ctx, threads=threads) for i in range(B)]
    pieces = gpu_batched_free_run(lib, ctx, embeddings, prompts, steps)
    path_free = lib.amd_batch_path(ctx)
    logits = gpu_batched_teacher_forced(lib, ctx, embeddings, prompts, [o["ids"][:steps] for o in oracles])
    path = lib.amd_batch_path(ctx)
    per = []
    for i in range(B):
        head = {"logits": oracles[i]["logits"][:steps], "ids": oracles[i]["ids"][:steps], "pieces": oracles[i]["pieces"][:steps], "n_prompt": oracles[i]["n_prompt"]}
        per.append(compare(head, pieces[i], logits[i]))
    agg = {"conversations": B, "tokens_compared_each": steps, "launches_of_the_batched_step": path, "launches_free_running_step": path_free,
           "free_running_identical_min": min(p["free_running_identical"] for p in per),
           "teacher_forced_argmax_identical_min": min(p["teacher_forced_argmax_identical"] for p in per),
           "max_logit_rel": max(p["max_logit_rel"] for p in per), "max_logit_rel_range": max(p["max_logit_rel_range"] for p in per),
           "decided_min": min(p["decided"] for p in per), "decided_argmax_mismatches": sum(p["decided"] - p["decided_argmax_identical"] for p in per),
           "prompt_tokens": [p["prompt_tokens"] for p in per], "oracle_s": sum(o["prefill_s"] + o["decode_s"] for o in oracles),
           "per_conversation": per}
    return agg
