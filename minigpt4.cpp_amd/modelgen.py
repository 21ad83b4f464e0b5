"""Synthetic MiniGPT-4 model files in the reference's two on-disk formats (host tooling).

No checkpoints are available offline, so tests and `bench.py` run on deterministic synthetic
weights written byte-for-byte in the formats the reference consumes:

* format A -- the MiniGPT-4 vision container ("ggml" magic, version 1), normative writer
  /root/reference/minigpt4/convert.py:74-180, reader /root/reference/minigpt4.cpp:1478-1596;
* format B -- the Vicuna LLM file (GGJT v3) read by llama.cpp@master-31cfbb1 through
  `llama_load_model_from_file` (/root/reference/minigpt4.cpp:1783); layout per SURVEY.md 2.5.

Also provides pure-numpy *readers* for both formats: the tests use them as an independent view
of a file (float64 dequantised tensors) and to hand raw tensor bytes to the CPU oracle.
"""
from __future__ import annotations

import json
import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import quants as Q

PAGE = 4096
GGJT_MAGIC = 0x67676A74
GGJT_VERSION = 3

SYSTEM_PROMPT = ("Give the following image: <Img>ImageContent</Img>. You will be able to see the image once I "
                 "provide it to you. Please answer my questions.###")


def _align(pos: int, a: int) -> int:
    return (pos + a - 1) // a * a


# =========================================================================== configs
@dataclass
class VisionConfig:
    embed_dim: int = 1408          # must be a multiple of 88 (head size is hard-coded, minigpt4.cpp:1271)
    depth: int = 39
    mlp_dim: int = 6144
    image_size: int = 224          # fixed by the reference (minigpt4.cpp:131)
    patch: int = 14
    q_hidden: int = 768            # fixed: 12 heads x 64 (minigpt4.cpp:127-129)
    q_layers: int = 12
    q_inter: int = 3072
    q_queries: int = 32
    cross_freq: int = 2
    n_embd_llm: int = 5120         # 4096 -> 7B, 5120 -> 13B (minigpt4.cpp:1614-1627)
    ftype: str = "f16"

    @property
    def n_pos(self) -> int:
        return (self.image_size // self.patch) ** 2 + 1


@dataclass
class LLMConfig:
    n_vocab: int = 32000
    n_embd: int = 5120
    n_mult: int = 256
    n_head: int = 40
    n_layer: int = 40
    ftype: int = 17
    wtype: str = "q5_k"            # base type of the 2-D weights
    mix: str = "q5_k_m"            # "none" | "q5_k_m" (wv/w2 -> q6_k in the 'more bits' layers, output q6_k)
    output_type: Optional[str] = None
    tok_type: Optional[str] = None
    more_bits_layers: Optional[Tuple[int, ...]] = None   # mix "q5_k_m": explicit 'more bits' layers instead of llama.cpp's use_more_bits(i, n_layer) rule

    @property
    def n_rot(self) -> int:
        return self.n_embd // self.n_head

    @property
    def n_ff(self) -> int:
        return ((2 * (4 * self.n_embd) // 3 + self.n_mult - 1) // self.n_mult) * self.n_mult


def vision_7b() -> VisionConfig:
    return VisionConfig(n_embd_llm=4096)


def vision_13b() -> VisionConfig:
    return VisionConfig(n_embd_llm=5120)


def llm_7b(wtype="q4_0") -> LLMConfig:
    return LLMConfig(n_vocab=32000, n_embd=4096, n_head=32, n_layer=32, wtype=wtype, mix="none", ftype=2,
                     output_type="q6_k")


def llm_13b(wtype="q5_k") -> LLMConfig:
    return LLMConfig(n_vocab=32000, n_embd=5120, n_head=40, n_layer=40, wtype=wtype, mix="q5_k_m", ftype=17)


def headline_llm(config: str):
    """(LLMConfig, write_llm_file keyword arguments) of the synthetic BASELINE.json files bench.py measures and tests/test_gpu_headline.py checks -- one definition for both.
    Round 3: the files are CONDITIONED like a trained network rather than an i.i.d. Gaussian stack -- wo / w2 scaled by 1 / sqrt(2 n_layer) (GPT-2 / Megatron scaled init),
    token embeddings at a scale that keeps the residual stream tied to the current token, an output matrix whose logits make decisive greedy choices (`output_tie`), and
    every layer different (`rotate_layers`; the 2-layer file has genuinely independent layers).  Same shapes, types and byte volume as before."""
    if config == "13b":
        cfg = llm_13b()
    elif config == "7b":
        cfg = llm_7b("q4_0")
    elif config in ("7b_q8_0", "7b_q4_1"):
        # the other two block types north_star names next to Q4_0 / Q5_K ("Q4_0/Q4_1/Q5_K/Q8_0 block dequant fused into the matvec"): every 2-D tensor of the 7B graph in that type
        cfg = LLMConfig(n_vocab=32000, n_embd=4096, n_head=32, n_layer=32, wtype=config[3:], mix="none", ftype=7 if config == "7b_q8_0" else 3)
    elif config == "13b_v32001":
        # the type mix of the file a user of the reference really loads: Vicuna-v0 has n_vocab 32001 -> llama.cpp's k-quant fallback (llm_tensor_types): output.weight F16
        # (327.7 MB streamed per token), tok_embeddings Q4_0; 9.310 GB per token
        cfg = LLMConfig(n_vocab=32001, n_embd=5120, n_head=40, n_layer=40, wtype="q5_k", mix="q5_k_m", ftype=17)
    elif config == "13b_v32001_l2":
        cfg = LLMConfig(n_vocab=32001, n_embd=5120, n_head=40, n_layer=2, wtype="q5_k", mix="q5_k_m", ftype=17, more_bits_layers=(0,))
    elif config == "13b_l2":
        # the 13B graph at full width, two layers deep: layer 0 a "more bits" layer (wv / w2 in Q6_K: the mixed-type qkv launch, the Q6_K NU = 7 tiling), layer 1 a plain
        # Q5_K layer, output Q6_K -- every kernel instantiation / tiling / launch geometry of the 40-layer headline model
        cfg = LLMConfig(n_vocab=32000, n_embd=5120, n_head=40, n_layer=2, wtype="q5_k", mix="q5_k_m", ftype=17, more_bits_layers=(0,))
    else:
        raise ValueError(config)
    kw = dict(seed=1234, std=0.02, unique_layers=(None if cfg.n_layer <= 2 else 1), fast=True, resid_scale=float(1.0 / np.sqrt(2.0 * cfg.n_layer)), rotate_layers=True,
              tok_std=6.0, output_tie=0.15)
    return cfg, kw


def use_more_bits(i: int, n: int) -> bool:
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def llm_tensor_types(cfg: LLMConfig) -> Dict[str, int]:
    """name -> ggml type for every tensor of the LLM file (per-tensor mix)."""
    base = Q.NAME_TO_TYPE[cfg.wtype]
    out: Dict[str, int] = {}
    otype = Q.NAME_TO_TYPE[cfg.output_type] if cfg.output_type else (Q.GGML_Q6_K if cfg.mix == "q5_k_m" else base)
    ttype = Q.NAME_TO_TYPE[cfg.tok_type] if cfg.tok_type else base
    # llama.cpp's quantiser with k-quants on (the reference forces GGML_USE_K_QUANTS, /root/reference/CMakeLists.txt:317): a tensor whose ne[0] or ne[1] is not a multiple
    # of QK_K = 256 cannot take a k-quant type and falls back -- output.weight to F16, tok_embeddings.weight to Q4_0.  Vicuna-v0 has n_vocab = 32001,
    # so the file the reference's README names (ggml-vicuna-13B-v0-q5_k.bin, /root/reference/README.md:134) carries exactly that pair.
    if cfg.n_vocab % 256 or cfg.n_embd % 256:
        if not cfg.output_type and Q.BLOCK[otype][0] == 256:
            otype = Q.GGML_F16
        if not cfg.tok_type and Q.BLOCK[ttype][0] == 256:
            ttype = Q.GGML_Q4_0
    for t in (otype, ttype, base):
        e, _ = Q.BLOCK[t]
        assert cfg.n_embd % e == 0 and cfg.n_ff % e == 0, "row length must be a multiple of the block size"
    out["tok_embeddings.weight"] = ttype
    out["norm.weight"] = Q.GGML_F32
    out["output.weight"] = otype
    for i in range(cfg.n_layer):
        p = f"layers.{i}."
        more = cfg.mix == "q5_k_m" and (i in cfg.more_bits_layers if cfg.more_bits_layers is not None else use_more_bits(i, cfg.n_layer))
        out[p + "attention_norm.weight"] = Q.GGML_F32
        out[p + "attention.wq.weight"] = base
        out[p + "attention.wk.weight"] = base
        out[p + "attention.wv.weight"] = Q.GGML_Q6_K if more else base
        out[p + "attention.wo.weight"] = base
        out[p + "ffn_norm.weight"] = Q.GGML_F32
        out[p + "feed_forward.w1.weight"] = base
        out[p + "feed_forward.w2.weight"] = Q.GGML_Q6_K if more else base
        out[p + "feed_forward.w3.weight"] = base
    return out


def llm_tensor_shapes(cfg: LLMConfig) -> Dict[str, Tuple[int, ...]]:
    """name -> ggml ne (innermost first). 2-D weights: ne = (n_in, n_out)."""
    E, F, V = cfg.n_embd, cfg.n_ff, cfg.n_vocab
    out: Dict[str, Tuple[int, ...]] = {
        "tok_embeddings.weight": (E, V), "norm.weight": (E,), "output.weight": (E, V)}
    for i in range(cfg.n_layer):
        p = f"layers.{i}."
        out[p + "attention_norm.weight"] = (E,)
        for w in ("wq", "wk", "wv", "wo"):
            out[p + f"attention.{w}.weight"] = (E, E)
        out[p + "ffn_norm.weight"] = (E,)
        out[p + "feed_forward.w1.weight"] = (E, F)
        out[p + "feed_forward.w2.weight"] = (F, E)
        out[p + "feed_forward.w3.weight"] = (E, F)
    return out


def llm_weight_bytes_per_token(cfg: LLMConfig) -> int:
    """Algorithmic HBM bytes one decode step must stream: every layer matrix + norms + output matrix
    (tok_embeddings contributes one row only)."""
    types, shapes = llm_tensor_types(cfg), llm_tensor_shapes(cfg)
    total = 0
    for name, ne in shapes.items():
        n = int(np.prod(ne))
        if name == "tok_embeddings.weight":
            n = ne[0]
        total += Q.nbytes(types[name], n)
    return total


# =========================================================================== vocab
def synth_vocab(n_vocab: int) -> List[Tuple[bytes, float]]:
    """Synthetic sentencepiece-like vocab: <unk>,<s>,</s>, 256 byte tokens, then ASCII pieces with
    descending scores.  Every printable ASCII char is present so prompts tokenise without byte
    fallback, and '#', '##', '###' exist so the reference's stop rules (minigpt4.cpp:2764-2782) fire."""
    vocab: List[Tuple[bytes, float]] = [(b"<unk>", 0.0), (b"<s>", 0.0), (b"</s>", 0.0)]
    for b in range(256):
        vocab.append((bytes([b]), 0.0))
    pieces: List[bytes] = []
    seen = set()

    def add(p: bytes):
        if p not in seen and len(vocab) + len(pieces) < n_vocab:
            seen.add(p)
            pieces.append(p)

    words = ["##", "###", " the", " image", "Human", "Assistant", ": ", "<Img>", "</Img>", " is", " of", " a",
             "Img", "age", " you", " to", " in", " and", "Content", " Please", " answer", " my", " questions",
             " picture", " text", " what", "what", " see", " be", " able", " will", " once", " provide", " it",
             "Give", " following", "ing", "er", "es", "ed", "th", "he", "in", "an", "re", "on", "at", "en",
             "nd", "ti", "or", "te", " t", " a", " s", " w", " i", " o", " b", " m", "ou", "it", "is", "ll"]
    for w in words:
        add(w.encode())
    rng = np.random.default_rng(7)
    letters = "etaoinshrdlucmfwypvbgkjqxz"
    for a in letters:
        for b in letters:
            add((a + b).encode())
    k = 0
    while len(vocab) + len(pieces) < n_vocab - 95 and k < 4 * n_vocab:
        ln = int(rng.integers(3, 7))
        s = "".join(letters[int(i)] for i in rng.integers(0, 14, ln))
        if rng.random() < 0.4:
            s = " " + s
        add(s.encode())
        k += 1
    for i, p in enumerate(pieces):
        vocab.append((p, -1.0 - 0.001 * i))
    # single printable ASCII characters (lowest scores: they are never the *result* of a merge)
    for c in range(32, 127):
        if len(vocab) < n_vocab:
            vocab.append((bytes([c]), -1000.0 - c))
    i = 0
    while len(vocab) < n_vocab:
        vocab.append((f"<pad{i}>".encode(), -5000.0 - i))
        i += 1
    return vocab[:n_vocab]


# =========================================================================== format B writer / reader
class _Pool:
    """Gaussian pool for bench-size files: tensors are cut from one 16M-sample pool at rotating offsets
    (same statistics, ~20x faster than drawing every weight)."""

    def __init__(self, rng, n=1 << 24):
        self.p = rng.standard_normal(n, dtype=np.float32)
        self.off = 0

    def take(self, n: int) -> np.ndarray:
        out = np.empty(n, np.float32)
        pos = 0
        while pos < n:
            self.off = (self.off * 7 + 104729) % (self.p.size - 1)
            k = min(n - pos, self.p.size - self.off)
            out[pos:pos + k] = self.p[self.off:self.off + k]
            pos += k
        return out


def write_llm_file(path: str, cfg: LLMConfig, seed: int = 1234, std: float = 0.02,
                   unique_layers: Optional[int] = None, vocab: Optional[List[Tuple[bytes, float]]] = None,
                   fast: bool = False, resid_scale: float = 1.0, rotate_layers: bool = False, tok_std: Optional[float] = None,
                   output_tie: float = 0.0) -> None:
    """Write a GGJT-v3 file with Gaussian weights.  `unique_layers` < n_layer re-uses the quantised bytes
    of layer (i % unique_layers) for layer i (bench-size files: same byte volume, generation in seconds).
    `resid_scale` multiplies the two matrices that write into the residual stream (attention.wo, feed_forward.w2) -- the
    GPT-2 / Megatron "scaled init" 1 / sqrt(2 n_layer): without it a deep random-weight stack amplifies a 1e-6 input
    perturbation to percents of the logit range (the int8 activation roundings of the two runs decorrelate), which drowns
    any GPU-vs-oracle comparison in the arithmetic's own noise.
    `rotate_layers`: a layer that re-uses another layer's quantised bytes gets them with the ROWS of every matrix rotated by a layer- and tensor-dependent amount
    (norm vectors: their elements) -- whole quantised rows move, so the blocks stay valid and all n_layer layers differ (a wrong layer stride / weight pointer shows).
    `tok_std`: standard deviation of tok_embeddings (default `std`); ~1 keeps the residual stream correlated with the current token through a deep stack.
    `output_tie` = beta in (0, 1]: output row u = std * (beta * e[p[u]] + sqrt(1 - beta^2) * r_u), e = the unit-variance token embeddings, p a fixed random permutation,
    r Gaussian -- the logit of the token u with p[u] == current token stands out of the Gaussian rest by a margin beta controls: DECISIVE greedy decisions (top-2 margin
    far above the comparison tolerance) on a walk through the permutation instead of near-ties between 32000 i.i.d. logits."""
    rng = np.random.default_rng(seed)
    types, shapes = llm_tensor_types(cfg), llm_tensor_shapes(cfg)
    vocab = vocab if vocab is not None else synth_vocab(cfg.n_vocab)
    assert len(vocab) == cfg.n_vocab
    uniq = unique_layers if unique_layers else cfg.n_layer
    cache: Dict[Tuple[str, int], np.ndarray] = {}
    pool = _Pool(rng) if fast else None

    tok_unit: Dict[str, np.ndarray] = {}

    def gen(name: str) -> np.ndarray:
        ne, t = shapes[name], types[name]
        n = int(np.prod(ne))
        key = None
        if name.startswith("layers."):
            _, idx, rest = name.split(".", 2)
            key = (rest, int(idx) % uniq, t)
            if key in cache:
                raw = cache[key]
                if rotate_layers and int(idx) >= uniq:
                    rows = ne[1] if len(ne) == 2 else ne[0]
                    shift = (int(idx) * 977 + sum(map(ord, rest)) * 131) % rows or 1
                    raw = np.roll(raw.reshape(rows, -1), shift, axis=0).reshape(-1)
                return raw
        if name.endswith("norm.weight"):
            x = (1.0 + 0.02 * rng.standard_normal(n)).astype(np.float32)
        elif pool is not None:
            x = pool.take(n)
        else:
            x = rng.standard_normal(n, dtype=np.float32)
        if not name.endswith("norm.weight"):
            if name == "tok_embeddings.weight" and output_tie:
                tok_unit["e"] = x.reshape(cfg.n_vocab, cfg.n_embd).copy()
            if name == "output.weight" and output_tie:
                perm = np.random.default_rng(seed + 77).permutation(cfg.n_vocab)
                x = (np.float32(output_tie) * tok_unit["e"][perm] + np.float32(np.sqrt(1.0 - output_tie * output_tie)) * x.reshape(cfg.n_vocab, cfg.n_embd)).reshape(-1)
            x = x * np.float32(tok_std if (name == "tok_embeddings.weight" and tok_std is not None) else std)
        if resid_scale != 1.0 and (name.endswith("attention.wo.weight") or name.endswith("feed_forward.w2.weight")):
            x = x * np.float32(resid_scale)
        raw = Q.quantize(t, x)
        if key is not None:
            cache[key] = raw
        return raw

    with open(path, "wb") as f:
        f.write(struct.pack("<II", GGJT_MAGIC, GGJT_VERSION))
        f.write(struct.pack("<7I", cfg.n_vocab, cfg.n_embd, cfg.n_mult, cfg.n_head, cfg.n_layer, cfg.n_rot, cfg.ftype))
        for piece, score in vocab:
            f.write(struct.pack("<I", len(piece)))
            f.write(piece)
            f.write(struct.pack("<f", score))
        for name in shapes:
            ne, t = shapes[name], types[name]
            nm = name.encode()
            f.write(struct.pack("<III", len(ne), len(nm), t))
            f.write(struct.pack(f"<{len(ne)}I", *ne))
            f.write(nm)
            pos = f.tell()
            f.write(b"\0" * (_align(pos, 32) - pos))
            raw = gen(name)
            assert raw.nbytes == Q.nbytes(t, int(np.prod(ne))), name
            raw.tofile(f)


def ggjt_to_gguf(src: str, dst: str, version: int = 3, alignment: int = 32, extra_tensor: bool = True) -> None:
    """Re-container a GGJT v3 LLM file as GGUF (the layout current llama.cpp converters write): same tensor bytes under the GGUF names, SentencePiece-form
    vocabulary (U+2581 for spaces, "<0xXX>" byte tokens with token_type 6), llama.* metadata.  Test tooling for the GGUF reader (csrc/formats.cpp)."""
    f = read_llm_file(src)
    hp = f.hparams
    n_ff = ((2 * (4 * hp["n_embd"]) // 3 + hp["n_mult"] - 1) // hp["n_mult"]) * hp["n_mult"]

    def gs(b: bytes) -> bytes:
        return struct.pack("<Q", len(b)) + b

    def kv(key: str, vt: int, payload: bytes) -> bytes:
        return gs(key.encode()) + struct.pack("<I", vt) + payload
    toks, types = [], []
    for i, (piece, _) in enumerate(f.vocab):
        if 3 <= i < 259 and len(piece) == 1:
            toks.append(b"<0x%02X>" % piece[0])
            types.append(6)
        else:
            toks.append(piece.replace(b" ", "\u2581".encode()))
            types.append(3 if i < 3 else 1)
    meta = [kv("general.architecture", 8, gs(b"llama")), kv("general.name", 8, gs(b"synthetic")), kv("general.alignment", 4, struct.pack("<I", alignment)),
            kv("llama.context_length", 4, struct.pack("<I", 2048)), kv("llama.embedding_length", 4, struct.pack("<I", hp["n_embd"])),
            kv("llama.block_count", 4, struct.pack("<I", hp["n_layer"])), kv("llama.feed_forward_length", 4, struct.pack("<I", n_ff)),
            kv("llama.rope.dimension_count", 4, struct.pack("<I", hp["n_embd"] // hp["n_head"])), kv("llama.attention.head_count", 4, struct.pack("<I", hp["n_head"])),
            kv("llama.attention.head_count_kv", 4, struct.pack("<I", hp["n_head"])), kv("llama.attention.layer_norm_rms_epsilon", 6, struct.pack("<f", 1e-6)),
            kv("llama.rope.freq_base", 6, struct.pack("<f", 10000.0)), kv("general.file_type", 4, struct.pack("<I", hp["ftype"])),
            kv("tokenizer.ggml.model", 8, gs(b"llama")),
            kv("tokenizer.ggml.tokens", 9, struct.pack("<IQ", 8, len(toks)) + b"".join(gs(t) for t in toks)),
            kv("tokenizer.ggml.scores", 9, struct.pack("<IQ", 6, len(toks)) + struct.pack(f"<{len(toks)}f", *[s for _, s in f.vocab])),
            kv("tokenizer.ggml.token_type", 9, struct.pack("<IQ", 5, len(toks)) + struct.pack(f"<{len(toks)}i", *types)),
            kv("tokenizer.ggml.bos_token_id", 4, struct.pack("<I", 1)), kv("tokenizer.ggml.eos_token_id", 4, struct.pack("<I", 2)),
            kv("some.unknown.array", 9, struct.pack("<IQ", 2, 3) + struct.pack("<3H", 1, 2, 3)), kv("some.bool", 7, b"\x01"), kv("some.u64", 10, struct.pack("<Q", 7))]
    ren = {"tok_embeddings.weight": "token_embd.weight", "norm.weight": "output_norm.weight", "output.weight": "output.weight"}
    sub = {"attention_norm.weight": "attn_norm.weight", "attention.wq.weight": "attn_q.weight", "attention.wk.weight": "attn_k.weight", "attention.wv.weight": "attn_v.weight",
           "attention.wo.weight": "attn_output.weight", "ffn_norm.weight": "ffn_norm.weight", "feed_forward.w1.weight": "ffn_gate.weight", "feed_forward.w2.weight": "ffn_down.weight",
           "feed_forward.w3.weight": "ffn_up.weight"}
    infos, blobs, off = [], [], 0
    items = list(f.tensors.items())
    if extra_tensor:
        items.append(("rope_freqs.weight", None))                  # a tensor the graph does not use: must be skipped by the reader
    for name, t in items:
        if t is None:
            raw, ne, gt, gname = np.zeros(16, np.float32).view(np.uint8), (16,), Q.GGML_F32, name
        else:
            raw, ne, gt = f.raw(name), t.ne, t.gtype
            if name.startswith("layers."):
                _, idx, rest = name.split(".", 2)
                gname = f"blk.{idx}.{sub[rest]}"
            else:
                gname = ren[name]
        infos.append(gs(gname.encode()) + struct.pack("<I", len(ne)) + struct.pack(f"<{len(ne)}Q", *ne) + struct.pack("<IQ", gt, off))
        blobs.append((off, raw))
        off = _align(off + raw.nbytes, alignment)
    head = struct.pack("<IIQQ", 0x46554747, version, len(infos), len(meta)) + b"".join(meta) + b"".join(infos)
    with open(dst, "wb") as out:
        out.write(head)
        out.write(b"\0" * (_align(len(head), alignment) - len(head)))
        base = out.tell()
        for o, raw in blobs:
            out.seek(base + o)
            out.write(bytes(raw))
        out.truncate(base + off)


@dataclass
class TensorInfo:
    name: str
    gtype: int
    ne: Tuple[int, ...]
    offset: int
    nbytes: int


@dataclass
class LLMFile:
    path: str
    hparams: Dict[str, int]
    vocab: List[Tuple[bytes, float]]
    tensors: Dict[str, TensorInfo]
    _mm: np.memmap = field(repr=False, default=None)

    def raw(self, name: str) -> np.ndarray:
        t = self.tensors[name]
        return np.asarray(self._mm[t.offset:t.offset + t.nbytes])

    def f64(self, name: str) -> np.ndarray:
        """Dequantised tensor as float64 in torch layout (reversed ne: [n_out, n_in])."""
        t = self.tensors[name]
        n = int(np.prod(t.ne))
        return Q.dequantize(t.gtype, self.raw(name), n).reshape(tuple(reversed(t.ne)))


def read_llm_file(path: str, in_memory: bool = False) -> LLMFile:
    """in_memory=True reads the whole file into anonymous memory (no page-fault storm when many threads stream it)."""
    mm = np.fromfile(path, dtype=np.uint8) if in_memory else np.memmap(path, dtype=np.uint8, mode="r")
    pos = 0

    def u32(k=1):
        nonlocal pos
        v = struct.unpack_from(f"<{k}I", mm, pos)
        pos += 4 * k
        return v if k > 1 else v[0]

    magic, version = u32(2)
    assert magic == GGJT_MAGIC and version == GGJT_VERSION, (hex(magic), version)
    keys = ["n_vocab", "n_embd", "n_mult", "n_head", "n_layer", "n_rot", "ftype"]
    hp = dict(zip(keys, u32(7)))
    vocab = []
    for _ in range(hp["n_vocab"]):
        ln = u32()
        piece = bytes(mm[pos:pos + ln])
        pos += ln
        (score,) = struct.unpack_from("<f", mm, pos)
        pos += 4
        vocab.append((piece, score))
    tensors: Dict[str, TensorInfo] = {}
    size = mm.shape[0]
    while pos < size:
        nd, nl, t = u32(3)
        ne = u32(nd) if nd > 1 else (u32(),)
        name = bytes(mm[pos:pos + nl]).decode()
        pos += nl
        pos = _align(pos, 32)
        nb = Q.nbytes(t, int(np.prod(ne)))
        tensors[name] = TensorInfo(name, t, tuple(int(x) for x in ne), pos, nb)
        pos += nb
    return LLMFile(path, hp, vocab, tensors, mm)


# =========================================================================== format A writer / reader
def vision_state(cfg: VisionConfig, seed: int = 4321, std: float = 0.02,
                 unique_blocks: Optional[int] = None, fast: bool = False) -> Dict[str, Dict[str, np.ndarray]]:
    """Synthetic state dicts (torch shapes, float32/int64) for the 5 sub-models, in convert.py's order."""
    rng = np.random.default_rng(seed)
    D, M = cfg.embed_dim, cfg.mlp_dim

    pool = _Pool(rng) if fast else None

    def w(*shape):
        if pool is not None:
            return (pool.take(int(np.prod(shape))) * np.float32(std)).reshape(shape)
        return (std * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)

    def g(n):
        return (1.0 + 0.02 * rng.standard_normal(n)).astype(np.float32)

    ve: Dict[str, np.ndarray] = {}
    ve["cls_token"] = w(1, 1, D)
    ve["pos_embed"] = w(1, cfg.n_pos, D)
    ve["patch_embed.proj.weight"] = w(D, 3, cfg.patch, cfg.patch)
    ve["patch_embed.proj.bias"] = w(D)
    uniq = unique_blocks if unique_blocks else cfg.depth
    blocks: List[Dict[str, np.ndarray]] = []
    for i in range(cfg.depth):
        if i < uniq:
            b = {
                "norm1.weight": g(D), "norm1.bias": w(D),
                "attn.q_bias": w(D), "attn.v_bias": w(D),
                "attn.qkv.weight": w(3 * D, D),
                "attn.proj.weight": w(D, D), "attn.proj.bias": w(D),
                "norm2.weight": g(D), "norm2.bias": w(D),
                "mlp.fc1.weight": w(M, D), "mlp.fc1.bias": w(M),
                "mlp.fc2.weight": w(D, M), "mlp.fc2.bias": w(D),
            }
            blocks.append(b)
        else:
            b = blocks[i % uniq]
        for k, v in b.items():
            ve[f"blocks.{i}.{k}"] = v
    ln = {"weight": g(D), "bias": w(D)}
    qt = {"weight": w(1, cfg.q_queries, cfg.q_hidden)}
    H, I = cfg.q_hidden, cfg.q_inter
    qf: Dict[str, np.ndarray] = {}
    qf["bert.embeddings.position_ids"] = np.arange(512, dtype=np.int64).reshape(1, 512)
    qf["bert.embeddings.LayerNorm.weight"] = g(H)
    qf["bert.embeddings.LayerNorm.bias"] = w(H)
    for i in range(cfg.q_layers):
        p = f"bert.encoder.layer.{i}."
        for att, kv_in in (("attention", H), ("crossattention", D)):
            if att == "crossattention" and i % cfg.cross_freq != 0:
                continue
            qf[p + f"{att}.self.query.weight"] = w(H, H)
            qf[p + f"{att}.self.query.bias"] = w(H)
            qf[p + f"{att}.self.key.weight"] = w(H, kv_in)
            qf[p + f"{att}.self.key.bias"] = w(H)
            qf[p + f"{att}.self.value.weight"] = w(H, kv_in)
            qf[p + f"{att}.self.value.bias"] = w(H)
            qf[p + f"{att}.output.dense.weight"] = w(H, H)
            qf[p + f"{att}.output.dense.bias"] = w(H)
            qf[p + f"{att}.output.LayerNorm.weight"] = g(H)
            qf[p + f"{att}.output.LayerNorm.bias"] = w(H)
        qf[p + "intermediate_query.dense.weight"] = w(I, H)
        qf[p + "intermediate_query.dense.bias"] = w(I)
        qf[p + "output_query.dense.weight"] = w(H, I)
        qf[p + "output_query.dense.bias"] = w(H)
        qf[p + "output_query.LayerNorm.weight"] = g(H)
        qf[p + "output_query.LayerNorm.bias"] = w(H)
    lp = {"weight": w(cfg.n_embd_llm, H), "bias": w(cfg.n_embd_llm)}
    return {"visual_encoder": ve, "ln_vision": ln, "query_tokens": qt, "Qformer": qf, "llama_proj": lp}


def qformer_config(cfg: VisionConfig) -> dict:
    """The subset of BertConfig.__dict__ the reference reads (minigpt4.cpp:2146,2227,2293) plus the usual keys."""
    return {
        "vocab_size": 30522, "hidden_size": cfg.q_hidden, "num_hidden_layers": cfg.q_layers,
        "num_attention_heads": 12, "hidden_act": "gelu", "intermediate_size": cfg.q_inter,
        "hidden_dropout_prob": 0.1, "attention_probs_dropout_prob": 0.1, "max_position_embeddings": 512,
        "type_vocab_size": 2, "initializer_range": 0.02, "layer_norm_eps": 1e-12,
        "position_embedding_type": "absolute", "use_cache": True, "classifier_dropout": None,
        "encoder_width": cfg.embed_dim, "add_cross_attention": True, "cross_attention_freq": cfg.cross_freq,
        "query_length": cfg.q_queries, "model_type": "bert",
    }


def write_vision_file(path: str, cfg: VisionConfig, seed: int = 4321, std: float = 0.02,
                      unique_blocks: Optional[int] = None,
                      state: Optional[Dict[str, Dict[str, np.ndarray]]] = None, fast: bool = False) -> None:
    """Byte layout of convert.py:146-180 (`write_file`) + :74-144 (`write_model`), incl. its dtype rule."""
    state = state if state is not None else vision_state(cfg, seed, std, unique_blocks, fast)
    ftype_id = 0 if cfg.ftype == "f16" else 1
    with open(path, "wb") as f:
        f.write(b"ggml")
        f.write(struct.pack("i", 1))
        f.write(struct.pack("i", ftype_id))
        cj = json.dumps({"ftype": cfg.ftype, "Qformer": qformer_config(cfg)}).encode()
        f.write(struct.pack("i", len(cj)))
        f.write(cj)
        for model_name in ("visual_encoder", "ln_vision", "query_tokens", "Qformer", "llama_proj"):
            model = state[model_name]
            nm = model_name.encode()
            f.write(struct.pack("i", len(nm)))
            f.write(nm)
            f.write(struct.pack("i", len(model)))
            arrays = {}
            conv_cache = {}
            for lname, arr in model.items():
                src_id = id(arr)
                arr = np.squeeze(np.asarray(arr))
                shape = list(reversed(arr.shape))
                dt = None
                if cfg.ftype == "f16":
                    if model_name not in ("query_tokens", "ln_vision") and lname.endswith("weight") and len(shape) >= 2:
                        if src_id not in conv_cache:
                            conv_cache[src_id] = arr.astype(np.float16)
                        arr, dt = conv_cache[src_id], Q.MG4_F16
                elif lname == "patch_embed.proj.weight":
                    arr, dt = arr.astype(np.float16), Q.MG4_F16
                if dt is None:
                    arr, dt = arr.astype(np.float32), Q.MG4_F32
                ln = lname.encode()
                f.write(struct.pack("i", len(ln)))
                f.write(ln)
                f.write(struct.pack("i", len(shape)))
                f.write(struct.pack(f"{len(shape)}i", *shape))
                f.write(struct.pack("i", dt))
                arrays[lname] = arr
            for lname in model:
                pos = f.tell()
                if pos & (PAGE - 1):
                    f.seek((pos + PAGE) & ~(PAGE - 1))
                np.ascontiguousarray(arrays[lname]).tofile(f)


@dataclass
class VisionFile:
    path: str
    version: int
    ftype: int
    config: dict
    models: Dict[str, Dict[str, TensorInfo]]
    _mm: np.memmap = field(repr=False, default=None)

    def raw(self, model: str, name: str) -> np.ndarray:
        t = self.models[model][name]
        return np.asarray(self._mm[t.offset:t.offset + t.nbytes])

    def f64(self, model: str, name: str) -> np.ndarray:
        """float64 array in torch layout (reversed ne)."""
        t = self.models[model][name]
        n = int(np.prod(t.ne))
        if t.gtype == Q.GGML_I64:
            return np.frombuffer(self.raw(model, name), np.int64, n).astype(np.float64).reshape(tuple(reversed(t.ne)))
        return Q.dequantize(t.gtype, self.raw(model, name), n).reshape(tuple(reversed(t.ne)))


def read_vision_file(path: str) -> VisionFile:
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    pos = 0

    def i32():
        nonlocal pos
        (v,) = struct.unpack_from("i", mm, pos)
        pos += 4
        return v

    def string():
        nonlocal pos
        ln = i32()
        s = bytes(mm[pos:pos + ln]).decode()
        pos += ln
        return s

    assert bytes(mm[0:4]) == b"ggml"
    pos = 4
    version = i32()
    ftype = i32()
    config = json.loads(string())
    models: Dict[str, Dict[str, TensorInfo]] = {}
    size = mm.shape[0]
    while pos < size:
        mname = string()
        n = i32()
        metas = []
        for _ in range(n):
            lname = string()
            nd = i32()
            ne = tuple(i32() for _ in range(nd))
            dt = i32()
            metas.append((lname, ne, Q.MG4_TO_GGML[dt]))
        tensors: Dict[str, TensorInfo] = {}
        for lname, ne, gt in metas:
            if pos & (PAGE - 1):
                pos = (pos + PAGE) & ~(PAGE - 1)
            nb = Q.nbytes(gt, int(np.prod(ne)) if len(ne) else 1)
            tensors[lname] = TensorInfo(lname, gt, ne, pos, nb)
            pos += nb
        models[mname] = tensors
    return VisionFile(path, version, ftype, config, models, mm)


# =========================================================================== tiny test models
def tiny_vision(n_embd_llm: int = 4096, depth: int = 2, q_layers: int = 2, embed_dim: int = 176,
                mlp_dim: int = 352, q_inter: int = 256) -> VisionConfig:
    """Small tower that still obeys the reference's hard-coded geometry: 224x224 input, 257 positions,
    head size 88 (so embed_dim = 88 * heads), Q-Former 12 x 64 = 768, 32 queries, and an LLM width the C API
    accepts (32*4096 or 32*5120 elements, minigpt4.cpp:2682)."""
    return VisionConfig(embed_dim=embed_dim, depth=depth, mlp_dim=mlp_dim, q_layers=q_layers, q_inter=q_inter,
                        n_embd_llm=n_embd_llm)


def tiny_llm(wtype: str = "q4_0", n_embd: int = 4096, n_layer: int = 2, n_head: int = 32, n_vocab: int = 512,
             n_mult: int = 256, mix: str = "none", output_type: Optional[str] = None,
             tok_type: Optional[str] = None) -> LLMConfig:
    return LLMConfig(n_vocab=n_vocab, n_embd=n_embd, n_mult=n_mult, n_head=n_head, n_layer=n_layer, ftype=2,
                     wtype=wtype, mix=mix, output_type=output_type, tok_type=tok_type)


# write_llm_file keyword arguments that condition a TINY test model like modelgen.headline_llm conditions the headline files (scaled residual writers, token-scale
# embeddings, decisive output logits): whole-model comparisons on such a file can be held to north_star's 1e-2 and its greedy decisions are far outside the noise.
TINY_CONDITIONED = dict(resid_scale=0.5, tok_std=4.0, output_tie=0.4)


def synth_image(seed: int = 42) -> np.ndarray:
    return np.random.default_rng(seed).standard_normal((3, 224, 224)).astype(np.float32)
