"""ctypes front-end of the CPU oracle (oracle/refcpu.c).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Also holds the pure-Python restatements of the host-side pieces of the path:
  * the llama.cpp (master-31cfbb1) SentencePiece-style tokenizer used by `llama_tokenize`
    (reference call site /root/reference/minigpt4.cpp:2389, BOS per fragment :2387);
  * the sampler chain of `MiniGPT4::sample_token` (/root/reference/minigpt4.cpp:2425-2483);
  * the chat templating of /root/reference/minigpt4.cpp:2671-2753.
PARITY UNPINNED (see refcpu.c header): no upstream source / golden vectors exist offline.
"""
from __future__ import annotations

import ctypes as C
import heapq
import math
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(native: bool = False) -> str:
    target = "native" if native else "all"
    subprocess.check_call(["make", "-s", "-C", _HERE, target])
    return os.path.join(_HERE, "_build", "librefcpu_native.so" if native else "librefcpu.so")


def lib(native: bool = False):
    global _LIB
    if _LIB is not None and not native:
        return _LIB
    path = os.path.join(_HERE, "_build", "librefcpu_native.so" if native else "librefcpu.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "refcpu.c")):
        try:
            path = build(native)
        except Exception:
            if not os.path.exists(path):
                raise
    L = C.CDLL(path)
    vp, i64, i32, f32p = C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_float)
    L.orc_table.restype = C.POINTER(C.c_uint16)
    L.orc_table.argtypes = [i32]
    L.orc_type_block.restype = i64
    L.orc_type_bytes.restype = i64
    L.orc_dequantize_row.argtypes = [i32, vp, vp, i64]
    L.orc_vec_dot_type.argtypes = [i32]
    L.orc_quantize_row.restype = i64
    L.orc_quantize_row.argtypes = [i32, vp, vp, i64]
    L.orc_vec_dot.restype = C.c_float
    L.orc_vec_dot.argtypes = [i32, i64, vp, vp]
    L.orc_mul_mat.argtypes = [i32, vp, i64, i64, vp, i64, vp]
    L.orc_set_simd.argtypes = [i32]
    L.orc_set_trace.argtypes = [C.c_char_p]
    L.orc_set_simd.restype = None
    L.orc_set_order.argtypes = [i32]
    L.orc_set_order.restype = None
    L.orc_llama_new.restype = vp
    L.orc_llama_new.argtypes = [i32] * 6
    L.orc_llama_free.argtypes = [vp]
    L.orc_llama_set_tensor.argtypes = [vp, C.c_char_p, i32, vp]
    L.orc_llama_eval.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    L.orc_rope_table.argtypes = [i32, i32, vp, vp]
    L.orc_vision_new.restype = vp
    L.orc_vision_new.argtypes = [i32] * 7
    L.orc_vision_free.argtypes = [vp]
    L.orc_vision_set_tensor.argtypes = [vp, C.c_char_p, C.c_char_p, i32, vp]
    L.orc_vision_encode.argtypes = [vp, vp, vp, i32, vp]
    L.orc_set_threads.argtypes = [i32]
    if not native:
        _LIB = L
    return L


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------------------ kernels
def dequantize_row(gtype: int, raw: np.ndarray, n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    raw = np.ascontiguousarray(raw)
    assert lib().orc_dequantize_row(gtype, _ptr(raw), _ptr(out), n) == 0
    return out


def quantize_row(qtype: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    buf = np.zeros(x.size * 4 + 64, np.uint8)
    nb = lib().orc_quantize_row(qtype, _ptr(x), _ptr(buf), x.size)
    assert nb >= 0
    return buf[:nb].copy()


def mul_mat(gtype: int, raw_w: np.ndarray, n_in: int, n_out: int, x: np.ndarray) -> np.ndarray:
    """x: [N, n_in] float32 -> [N, n_out] with ggml's quantised-activation arithmetic."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1, n_in)
    y = np.empty((x.shape[0], n_out), np.float32)
    raw_w = np.ascontiguousarray(raw_w)
    assert lib().orc_mul_mat(gtype, _ptr(raw_w), n_in, n_out, _ptr(x), x.shape[0], _ptr(y)) == 0
    return y


def table(which: int) -> np.ndarray:
    p = lib().orc_table(which)
    return np.ctypeslib.as_array(p, shape=(65536,)).copy()


# ------------------------------------------------------------------------------------------ LLaMA
class OracleLLM:
    """llama_eval / llama_eval_embd on a GGJT file read through minigpt4.cpp_amd.modelgen.read_llm_file."""

    def __init__(self, llm_file, n_ctx: int = 2048, native: bool = False):
        self.L = lib(native)
        self.f = llm_file
        hp = llm_file.hparams
        self.n_vocab, self.n_embd, self.n_head, self.n_layer = hp["n_vocab"], hp["n_embd"], hp["n_head"], hp["n_layer"]
        n_mult = hp["n_mult"]
        self.n_ff = ((2 * (4 * self.n_embd) // 3 + n_mult - 1) // n_mult) * n_mult
        self.n_ctx = n_ctx
        self.h = self.L.orc_llama_new(self.n_vocab, self.n_embd, self.n_head, self.n_layer, self.n_ff, n_ctx)
        self._keep = []
        for name, t in llm_file.tensors.items():
            raw = llm_file.raw(name)
            self._keep.append(raw)
            rc = self.L.orc_llama_set_tensor(self.h, name.encode(), t.gtype, _ptr(raw))
            assert rc == 0, name
        self.n_past = 0
        self.logits: Optional[np.ndarray] = None

    def __del__(self):
        try:
            self.L.orc_llama_free(self.h)
        except Exception:
            pass

    def eval_tokens(self, toks: Sequence[int], n_past: Optional[int] = None, all_logits: bool = False) -> np.ndarray:
        n_past = self.n_past if n_past is None else n_past
        t = np.ascontiguousarray(toks, np.int32)
        logits = np.empty(self.n_vocab, np.float32)
        al = np.empty((len(t), self.n_vocab), np.float32) if all_logits else None
        rc = self.L.orc_llama_eval(self.h, _ptr(t), None, len(t), n_past, _ptr(logits), _ptr(al) if al is not None else None)
        assert rc == 0
        self.n_past = n_past + len(t)
        self.logits = logits
        return al if all_logits else logits

    def eval_embd(self, embd: np.ndarray, n_past: Optional[int] = None) -> np.ndarray:
        n_past = self.n_past if n_past is None else n_past
        e = np.ascontiguousarray(embd, np.float32).reshape(-1, self.n_embd)
        logits = np.empty(self.n_vocab, np.float32)
        rc = self.L.orc_llama_eval(self.h, None, _ptr(e), e.shape[0], n_past, _ptr(logits), None)
        assert rc == 0
        self.n_past = n_past + e.shape[0]
        self.logits = logits
        return logits


# ------------------------------------------------------------------------------------------ vision
class OracleVision:
    def __init__(self, vision_file):
        self.L = lib()
        self.f = vision_file
        ve = vision_file.models["visual_encoder"]
        D = ve["pos_embed"].ne[0]
        depth = 0
        while f"blocks.{depth}.norm1.weight" in ve:
            depth += 1
        M = ve["blocks.0.mlp.fc1.weight"].ne[1]
        qf = vision_file.models["Qformer"]
        ql = 0
        while f"bert.encoder.layer.{ql}.attention.self.query.weight" in qf:
            ql += 1
        q_inter = qf["bert.encoder.layer.0.intermediate_query.dense.weight"].ne[1]
        n_q = vision_file.models["query_tokens"]["weight"].ne[1]
        n_out = vision_file.models["llama_proj"]["weight"].ne[1]
        self.D, self.n_q, self.n_out = D, n_q, n_out
        self.h = self.L.orc_vision_new(D, depth, M, ql, q_inter, n_q, n_out)
        self._keep = []
        for mname, tensors in vision_file.models.items():
            for name, t in tensors.items():
                raw = vision_file.raw(mname, name)
                self._keep.append(raw)
                rc = self.L.orc_vision_set_tensor(self.h, mname.encode(), name.encode(), t.gtype, _ptr(raw))
                assert rc >= 0, (mname, name)

    def __del__(self):
        try:
            self.L.orc_vision_free(self.h)
        except Exception:
            pass

    def encode(self, image_chw: np.ndarray, stage: int = 0):
        img = np.ascontiguousarray(image_chw, np.float32)
        assert img.size == 3 * 224 * 224
        out = np.empty((self.n_q, self.n_out), np.float32)
        st = None
        if stage in (1, 2):
            st = np.empty((257, self.D), np.float32)
        elif stage == 3:
            st = np.empty((self.n_q, 768), np.float32)
        rc = self.L.orc_vision_encode(self.h, _ptr(img), _ptr(out), stage, _ptr(st) if st is not None else None)
        assert rc == 0
        return (out, st) if stage else out


# ------------------------------------------------------------------------------------------ tokenizer
def tokenize(vocab: List[Tuple[bytes, float]], text: bytes, add_bos: bool = True) -> List[int]:
    """llama_tokenizer (llama.cpp master-31cfbb1): split into UTF-8 characters, repeatedly merge the
    adjacent pair with the highest vocab score (ties: leftmost), then map leftovers byte-wise (id = byte + 3)."""
    tok2id = {}
    for i, (p, _) in enumerate(vocab):
        tok2id[p] = i  # later duplicates win, as llama.cpp's token_to_id map is filled
    if not text:                 # llama_tokenize returns an empty vector for an empty text BEFORE it would push BOS
        return []
    out: List[int] = [1] if add_bos else []
    # symbols
    syms: List[List] = []  # [start, length, prev, next]
    i = 0
    while i < len(text):
        b = text[i]
        ln = 1 if b < 0x80 else 2 if (b & 0xE0) == 0xC0 else 3 if (b & 0xF0) == 0xE0 else 4 if (b & 0xF8) == 0xF0 else 1
        ln = min(ln, len(text) - i)
        syms.append([i, ln, len(syms) - 1, len(syms) + 1])
        i += ln
    syms[-1][3] = -1
    heap: List[Tuple[float, int, int, int]] = []

    def try_add(l: int, r: int):
        if l == -1 or r == -1:
            return
        piece = text[syms[l][0]:syms[l][0] + syms[l][1] + syms[r][1]]
        tid = tok2id.get(piece)
        if tid is None:
            return
        heapq.heappush(heap, (-vocab[tid][1], l, r, len(piece)))

    for k in range(1, len(syms)):
        try_add(k - 1, k)
    while heap:
        _, l, r, size = heapq.heappop(heap)
        if syms[l][1] == 0 or syms[r][1] == 0 or syms[l][1] + syms[r][1] != size:
            continue
        syms[l][1] += syms[r][1]
        syms[r][1] = 0
        syms[l][3] = syms[r][3]
        if syms[r][3] >= 0:
            syms[syms[r][3]][2] = l
        try_add(syms[l][2], l)
        try_add(l, syms[l][3])
    k = 0
    while k != -1:
        s = syms[k]
        piece = text[s[0]:s[0] + s[1]]
        tid = tok2id.get(piece)
        if tid is None:
            out.extend(int(b) + 3 for b in piece)
        else:
            out.append(tid)
        k = s[3]
    return out


# ------------------------------------------------------------------------------------------ sampler
class MT19937:
    """std::mt19937 (same stream as numpy's legacy RandomState seeded with an int)."""

    def __init__(self, seed: int):
        self.rs = np.random.RandomState(seed & 0xFFFFFFFF)

    def u32(self) -> int:
        return int(self.rs.randint(0, 2 ** 32, dtype=np.uint64))

    def canonical(self) -> float:
        # libstdc++ generate_canonical<double, 53>(mt19937): two 32-bit draws
        a, b = self.u32(), self.u32()
        r = (a + b * 4294967296.0) / 18446744073709551616.0
        return r if r < 1.0 else math.nextafter(1.0, 0.0)


def _softmax_sorted(logits: np.ndarray) -> np.ndarray:
    mx = logits[0]  # llama_sample_softmax uses candidates[0].logit as the maximum once `sorted` is set
    p = np.exp((logits - mx).astype(np.float32)).astype(np.float32)
    s = np.float32(0)
    for v in p:  # sequential fp32 sum, as llama_sample_softmax does
        s = np.float32(s + v)
    return (p / s).astype(np.float32)


def sample(logits: np.ndarray, rng: Optional[MT19937], temp: float, top_k: int, top_p: float, tfs_z: float,
           typical_p: float) -> int:
    """Greedy when temp <= 0 (first maximum wins); otherwise top_k -> tail_free -> typical -> top_p ->
    temperature -> multinomial, mirroring minigpt4.cpp:2449-2478 (mirostat not restated here)."""
    n_vocab = logits.shape[0]
    if temp <= 0:
        return int(np.argmax(logits))
    top_k = n_vocab if top_k <= 0 else top_k
    ids = np.arange(n_vocab)
    lg = logits.astype(np.float32)
    # top_k: sort descending (std::sort / partial_sort by logit; ties are implementation-defined)
    k = max(min(top_k, n_vocab), 1)
    order = np.argsort(-lg, kind="stable")[:k]
    ids, lg = ids[order], lg[order]
    # tail free
    if 0 < tfs_z < 1.0 and len(ids) > 2:
        p = _softmax_sorted(lg)
        d1 = p[:-1] - p[1:]
        d2 = np.abs(d1[:-1] - d1[1:]).astype(np.float32)
        s = np.float32(0)
        for v in d2:
            s = np.float32(s + v)
        d2 = (d2 / s).astype(np.float32)
        cum, last = np.float32(0), len(ids)
        for i, v in enumerate(d2):
            cum = np.float32(cum + v)
            if cum > tfs_z and i >= 1:
                last = i
                break
        ids, lg = ids[:last], lg[:last]
    # typical
    if 0 < typical_p < 1.0:
        p = _softmax_sorted(lg)
        ent = np.float32(0)
        for v in p:
            ent = np.float32(ent + np.float32(-v * np.log(v)))
        shifted = np.abs(-np.log(p) - ent).astype(np.float32)
        idx = np.argsort(shifted, kind="stable")
        cum, last = np.float32(0), len(idx)
        for i, j in enumerate(idx):
            cum = np.float32(cum + p[j])
            if cum > typical_p and i >= 0:
                last = i + 1
                break
        keep = idx[:last]
        ids, lg = ids[keep], lg[keep]  # order is NOT restored and `sorted` stays set at this llama.cpp revision
    # top_p
    if top_p < 1.0:
        p = _softmax_sorted(lg)
        cum, last = np.float32(0), len(ids)
        for i, v in enumerate(p):
            cum = np.float32(cum + v)
            if cum >= top_p and i + 1 >= 1:
                last = i + 1
                break
        ids, lg = ids[:last], lg[:last]
    lg = (lg / np.float32(temp)).astype(np.float32)
    p = _softmax_sorted(lg)
    # std::discrete_distribution: normalise in double, cumulative, lower_bound of a canonical draw
    pd = p.astype(np.float64)
    pd = pd / pd.sum()
    cp = np.cumsum(pd)
    cp[-1] = 1.0
    r = rng.canonical()
    return int(ids[int(np.searchsorted(cp, r, side="left"))])


# ------------------------------------------------------------------------------------------ chat templating
SYSTEM_PROMPT = (b"Give the following image: <Img>ImageContent</Img>. You will be able to see the image once I "
                 b"provide it to you. Please answer my questions.###")


class OracleChat:
    """MiniGPT4 engine call sequence (minigpt4.cpp:2365-2422, 2671-2753) over OracleLLM."""

    def __init__(self, llm: OracleLLM, n_batch: int = 512):
        self.llm, self.vocab, self.n_batch = llm, llm.f.vocab, n_batch
        self.id2piece = [p for p, _ in self.vocab]
        self.tokens_fed: List[int] = []

    def reset(self):
        self.llm.n_past = 0

    def add_string(self, s: bytes):
        toks = tokenize(self.vocab, s, add_bos=True)
        for i in range(0, len(toks), self.n_batch):
            self.llm.eval_tokens(toks[i:i + self.n_batch])
        self.tokens_fed.extend(toks)

    def system_prompt(self):
        self.add_string(SYSTEM_PROMPT)

    def begin_chat_image(self, embedding: np.ndarray, s: bytes):
        self.add_string(b"Human: <Img>")
        self.llm.eval_embd(np.asarray(embedding, np.float32).reshape(32, -1))
        self.add_string(b"</Img> ")
        self.add_string(s)
        self.add_string(b"### Assistant:")

    def begin_chat(self, s: bytes):
        self.add_string(b"Human: ")
        self.add_string(s)
        self.add_string(b"### Assistant:")

    def end_chat(self, temp=0.0, top_k=40, top_p=0.9, tfs_z=1.0, typical_p=1.0, rng=None) -> Tuple[int, bytes]:
        tid = sample(self.llm.logits, rng, temp, top_k, top_p, tfs_z, typical_p)
        piece = b"</s>" if tid == 2 else self.id2piece[tid]
        self.llm.eval_tokens([tid])
        return tid, piece
