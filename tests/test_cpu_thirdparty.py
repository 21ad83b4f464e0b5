"""CPU-only tier: the oracle (and the product's host logic) against REAL third-party implementations that exist in this image.

The reference's arithmetic lives in llama.cpp @ master-31cfbb1, which is not on this machine and ships no golden vectors, so the
oracle is a restatement.  These tests pin the parts of it that have an independent, widely used implementation here:

* the tokenizer against Google's `sentencepiece` library (llama.cpp's `llama_tokenizer` is a restatement of SentencePiece's BPE encoder; the
  ggml vocabulary stores "▁" as a plain space, `convert.py` of that revision);
* the LLaMA forward (RMSNorm eps, RoPE convention, causal attention, SwiGLU, KV-cache decode) against Hugging Face `LlamaForCausalLM` in fp32 --
  the weights go through the same interleaved -> rotate-half permutation `convert_llama_weights_to_hf.py` applies to Meta's checkpoints;
* the vision tower against Hugging Face BLIP-2 (`Blip2VisionModel` + `Blip2QFormerModel`), the very model MiniGPT-4 is built on
  (eva_vit_g with `[q_bias, 0, v_bias]`, `ln_vision` = `post_layernorm`, Q-Former with cross-attention every 2nd layer).  The reference
  evaluates GELU with ggml's tanh form (`ggml_gelu`), so the HF configs say `gelu_pytorch_tanh`.

They are tolerance pins (fp32 third-party math vs ggml's fp16 rounding points), not bit pins; the block formats stay pinned only by the
hand-computed vectors and the two independent implementations in tests/test_cpu_host.py.
"""
import ctypes
import os

import numpy as np
import pytest


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(b).max())


# ------------------------------------------------------------------------------------------------ tokenizer vs sentencepiece
def _sentencepiece_model(vocab):
    spm = pytest.importorskip("sentencepiece")
    from sentencepiece import sentencepiece_model_pb2 as pb
    m = pb.ModelProto()
    m.trainer_spec.model_type = pb.TrainerSpec.BPE
    m.trainer_spec.byte_fallback = True
    m.trainer_spec.vocab_size = len(vocab)
    m.normalizer_spec.name = "identity"
    m.normalizer_spec.add_dummy_prefix = False          # llama_tokenize adds no dummy prefix: callers prepend the space themselves
    m.normalizer_spec.remove_extra_whitespaces = False
    m.normalizer_spec.escape_whitespaces = True
    kind = pb.ModelProto.SentencePiece
    for i, (piece, score) in enumerate(vocab):
        sp = m.pieces.add()
        sp.score = score
        if i == 0:
            sp.piece, sp.type = "<unk>", kind.UNKNOWN
        elif i < 3:
            sp.piece, sp.type = piece.decode(), kind.CONTROL
        elif i < 259:
            sp.piece, sp.type = "<0x%02X>" % (i - 3), kind.BYTE
        else:
            sp.piece, sp.type = piece.decode().replace(" ", "▁"), kind.NORMAL
    return spm.SentencePieceProcessor(model_proto=m.SerializeToString())


def _random_texts(rng, n):
    alphabet = list("etaoinshrdlucmfwypvbgkjqxz") * 3 + [" "] * 14 + list("#<>/:?.,HAIGCP") + ["é", "世", "界", "\U0001F600", "ß"]
    out = []
    for _ in range(n):
        ln = int(rng.integers(1, 60))
        out.append("".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), ln)))
    return out


@pytest.mark.parametrize("n_vocab", [2000, 32000])
def test_tokenizer_equals_sentencepiece(lib, tmpdir_models, n_vocab):
    """Vocabularies that hold a piece for the space, as every real LLaMA vocabulary does ("▁"): without one SentencePiece byte-encodes its
    escaped "▁" (3 bytes) where llama.cpp byte-encodes the space itself -- an artefact of the escape, so the 512-piece test vocabulary is left out."""
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    texts = [G.SYSTEM_PROMPT, "Human: <Img>", "</Img> ", "### Assistant:", "Human: ", "what is the text in the picture?", " ", "###", "##", "#",
             "héllo 世界 \U0001F600", "the the  the   the", "a" * 33, "Give the following image", "   leading", "trailing   "]
    texts += _random_texts(np.random.default_rng(n_vocab), 300)
    # the product's tokenizer on the same vocabulary (a model file with this vocabulary; only its vocab section is read)
    lp = os.path.join(tmpdir_models, f"llm_vocab_{n_vocab}.bin")
    G.write_llm_file(lp, G.tiny_llm(wtype="q4_0", n_embd=64, n_layer=1, n_head=2, n_vocab=n_vocab, n_mult=32), seed=1, std=0.05)
    vocab = G.read_llm_file(lp).vocab                 # float32 scores, as the file stores them
    assert [p for p, _ in vocab] == [p for p, _ in G.synth_vocab(n_vocab)]
    proc = _sentencepiece_model(vocab)
    v = lib.library.minigpt4_amd_vocab_load(lp.encode())
    assert v
    for t in texts:
        want = proc.encode(t)
        b = t.encode()
        assert R.tokenize(vocab, b, False) == want, t
        out = (ctypes.c_int32 * (len(b) + 4))()
        n = lib.library.minigpt4_amd_vocab_tokenize(v, b, 0, out, len(b) + 4)
        assert list(out[:n]) == want, t
    lib.library.minigpt4_amd_vocab_free(v)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_tokenizer_equals_sentencepiece_on_random_vocabularies(lib, tmpdir_models, seed):
    """Random piece sets with random (partly tied) scores: merge order, leftmost tie-break and byte fallback must agree with SentencePiece's BPE encoder."""
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    rng = np.random.default_rng(1000 + seed)
    chars = list("abcdefgh ") + ["é", "世"]
    vocab = [(b"<unk>", 0.0), (b"<s>", 0.0), (b"</s>", 0.0)] + [(bytes([b]), 0.0) for b in range(256)]
    seen = set()
    while len(seen) < 220:
        seen.add("".join(chars[int(i)] for i in rng.integers(0, len(chars), int(rng.integers(2, 6)))))
    for pce in sorted(seen):
        vocab.append((pce.encode(), float(-int(rng.integers(1, 40)))))          # few distinct scores: many ties
    for c in chars[:-1]:                                                          # every character but one has a piece; "世" falls back to bytes
        vocab.append((c.encode(), -100.0 - float(len(vocab))))
    n_vocab = len(vocab)
    lp = os.path.join(tmpdir_models, f"llm_rndvocab_{seed}.bin")
    G.write_llm_file(lp, G.tiny_llm(wtype="q4_0", n_embd=64, n_layer=1, n_head=2, n_vocab=n_vocab, n_mult=32), seed=1, std=0.05, vocab=vocab)
    vocab = G.read_llm_file(lp).vocab
    proc = _sentencepiece_model(vocab)
    v = lib.library.minigpt4_amd_vocab_load(lp.encode())
    assert v
    for _ in range(400):
        t = "".join(chars[int(i)] for i in rng.integers(0, len(chars), int(rng.integers(1, 50))))
        want = proc.encode(t)
        b = t.encode()
        assert R.tokenize(vocab, b, False) == want, t
        out = (ctypes.c_int32 * (len(b) + 4))()
        n = lib.library.minigpt4_amd_vocab_tokenize(v, b, 0, out, len(b) + 4)
        assert list(out[:n]) == want, t
    lib.library.minigpt4_amd_vocab_free(v)


# ------------------------------------------------------------------------------------------------ LLaMA vs transformers
def _hf_llama(f):
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    hp = f.hparams
    E, H, L, V = hp["n_embd"], hp["n_head"], hp["n_layer"], hp["n_vocab"]
    n_ff = f.tensors["layers.0.feed_forward.w1.weight"].ne[1]
    cfg = tr.LlamaConfig(vocab_size=V, hidden_size=E, intermediate_size=n_ff, num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=H,
                         rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=256, tie_word_embeddings=False, attention_bias=False,
                         hidden_act="silu")
    cfg._attn_implementation = "eager"
    m = tr.LlamaForCausalLM(cfg).to(torch.float32).eval()

    def T(name):
        return torch.from_numpy(f.f64(name).astype(np.float32))

    def permute(w):   # Meta / ggml interleaved-pair RoPE rows -> HF rotate-half rows (convert_llama_weights_to_hf.py: permute())
        return w.view(H, E // H // 2, 2, E).transpose(1, 2).reshape(E, E)

    sd = {"model.embed_tokens.weight": T("tok_embeddings.weight"), "model.norm.weight": T("norm.weight"), "lm_head.weight": T("output.weight")}
    for i in range(L):
        p, q = f"layers.{i}.", f"model.layers.{i}."
        sd[q + "self_attn.q_proj.weight"] = permute(T(p + "attention.wq.weight"))
        sd[q + "self_attn.k_proj.weight"] = permute(T(p + "attention.wk.weight"))
        sd[q + "self_attn.v_proj.weight"] = T(p + "attention.wv.weight")
        sd[q + "self_attn.o_proj.weight"] = T(p + "attention.wo.weight")
        sd[q + "mlp.gate_proj.weight"] = T(p + "feed_forward.w1.weight")
        sd[q + "mlp.down_proj.weight"] = T(p + "feed_forward.w2.weight")
        sd[q + "mlp.up_proj.weight"] = T(p + "feed_forward.w3.weight")
        sd[q + "input_layernorm.weight"] = T(p + "attention_norm.weight")
        sd[q + "post_attention_layernorm.weight"] = T(p + "ffn_norm.weight")
    m.load_state_dict(sd, strict=True)
    return torch, m


@pytest.mark.parametrize("wtype,tol", [("f16", 3e-3), ("f32", 3e-3), ("q8_0", 4e-2), ("q5_k", 6e-2), ("q3_k", 6e-2)])
def test_oracle_llama_matches_transformers(tiny_files, wtype, tol):
    """Prefill of 24 tokens (every row's logits), then 6 teacher-forced decode steps against the KV cache.  f16 / f32 weights: ggml rounds the
    activations to fp16 for the mat-mul -> 3e-3; quantised weights: HF runs on the dequantised weights, ggml additionally rounds every activation
    row to int8 -> the activation-quantisation noise documented in DESIGN.md §3."""
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    _, llm = tiny_files
    f = G.read_llm_file(llm(wtype))
    torch, m = _hf_llama(f)
    rng = np.random.default_rng(5)
    toks = rng.integers(3, f.hparams["n_vocab"], 24)
    nxt = rng.integers(3, f.hparams["n_vocab"], 6)
    o = R.OracleLLM(f, n_ctx=64)
    with torch.no_grad():
        out = m(torch.from_numpy(toks)[None], use_cache=True)
        assert _rel(o.eval_tokens(toks, all_logits=True), out.logits[0].numpy()) < tol
        past = out.past_key_values
        for t in nxt:
            out = m(torch.tensor([[int(t)]]), past_key_values=past, use_cache=True)
            past = out.past_key_values
            assert _rel(o.eval_tokens([int(t)]), out.logits[0, 0].numpy()) < tol


# ------------------------------------------------------------------------------------------------ vision tower vs transformers BLIP-2
def _hf_blip2(vf):
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    ve, qf = vf.models["visual_encoder"], vf.models["Qformer"]
    D = ve["pos_embed"].ne[0]
    depth = sum(1 for k in ve if k.endswith(".norm1.weight"))
    mlp = ve["blocks.0.mlp.fc1.weight"].ne[1]
    ql = sum(1 for k in qf if k.endswith(".attention.self.query.weight"))
    q_inter = qf["bert.encoder.layer.0.intermediate_query.dense.weight"].ne[1]

    def T(model, name, shape=None):
        a = vf.f64(model, name).astype(np.float32)
        return torch.from_numpy(a.reshape(shape) if shape else a)

    vc = tr.Blip2VisionConfig(hidden_size=D, intermediate_size=mlp, num_hidden_layers=depth, num_attention_heads=D // 88, image_size=224, patch_size=14,
                              hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-5, qkv_bias=True)
    vc._attn_implementation = "eager"
    vm = tr.Blip2VisionModel(vc).to(torch.float32).eval()
    sd = {"embeddings.class_embedding": T("visual_encoder", "cls_token", (1, 1, D)),
          "embeddings.position_embedding": T("visual_encoder", "pos_embed", (1, 257, D)),
          "embeddings.patch_embedding.weight": T("visual_encoder", "patch_embed.proj.weight", (D, 3, 14, 14)),
          "embeddings.patch_embedding.bias": T("visual_encoder", "patch_embed.proj.bias"),
          "post_layernorm.weight": T("ln_vision", "weight"), "post_layernorm.bias": T("ln_vision", "bias")}
    for i in range(depth):
        p, q = f"blocks.{i}.", f"encoder.layers.{i}."
        sd[q + "self_attn.qkv.weight"] = T("visual_encoder", p + "attn.qkv.weight")
        sd[q + "self_attn.qkv.bias"] = torch.cat([T("visual_encoder", p + "attn.q_bias"), torch.zeros(D), T("visual_encoder", p + "attn.v_bias")])
        for a, b in (("self_attn.projection", "attn.proj"), ("layer_norm1", "norm1"), ("layer_norm2", "norm2"), ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2")):
            sd[q + a + ".weight"] = T("visual_encoder", p + b + ".weight")
            sd[q + a + ".bias"] = T("visual_encoder", p + b + ".bias")
    vm.load_state_dict(sd, strict=True)

    qc = tr.Blip2QFormerConfig(vocab_size=8, hidden_size=768, num_hidden_layers=ql, num_attention_heads=12, intermediate_size=q_inter,
                               hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-5, cross_attention_frequency=2, encoder_hidden_size=D)
    qc._attn_implementation = "eager"
    qm = tr.Blip2QFormerModel(qc).to(torch.float32).eval()
    sd = {"layernorm.weight": T("Qformer", "bert.embeddings.LayerNorm.weight"), "layernorm.bias": T("Qformer", "bert.embeddings.LayerNorm.bias")}
    for name in qf:
        if not name.startswith("bert.encoder."):
            continue
        sd[name[len("bert."):].replace("attention.self.", "attention.attention.")] = T("Qformer", name)
    qm.load_state_dict(sd, strict=True)
    return torch, vm, qm


def test_oracle_vision_matches_transformers_blip2(tiny_files):
    import refcpu as R
    from minigpt4_cpp_amd import modelgen as G
    vp, _ = tiny_files
    vf = G.read_vision_file(vp)
    torch, vm, qm = _hf_blip2(vf)
    ov = R.OracleVision(vf)
    for seed in (42, 7):
        img = G.synth_image(seed)
        with torch.no_grad():
            img_e = vm(pixel_values=torch.from_numpy(img)[None]).last_hidden_state          # ln_vision(ViT(x))  [1, 257, D]
            query = torch.from_numpy(vf.f64("query_tokens", "weight").astype(np.float32)).reshape(1, 32, 768)
            hs = qm(query_embeds=query, encoder_hidden_states=img_e, encoder_attention_mask=torch.ones(1, 257, dtype=torch.long)).last_hidden_state
            proj_w = torch.from_numpy(vf.f64("llama_proj", "weight").astype(np.float32))
            proj_b = torch.from_numpy(vf.f64("llama_proj", "bias").astype(np.float32))
            want = (hs[0] @ proj_w.T + proj_b).numpy()
        _, st2 = ov.encode(img, stage=2)
        assert _rel(st2, img_e[0].numpy()) < 3e-3
        _, st3 = ov.encode(img, stage=3)
        assert _rel(st3, hs[0].numpy()) < 3e-3
        assert _rel(ov.encode(img), want) < 3e-3
