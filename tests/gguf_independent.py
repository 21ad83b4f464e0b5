#!/usr/bin/env python3
"""An INDEPENDENT GGUF writer for the reader tests (round-2 verdict, row f4: the reader had only ever seen files written by modelgen.ggjt_to_gguf).

Written from the public GGUF specification (ggml/docs/gguf.md), sharing no code with minigpt4.cpp_amd/modelgen.py:

    header      : u32 magic 'GGUF' | u32 version | u64 tensor_count | u64 metadata_kv_count
    metadata kv : gguf_string key | u32 value_type | value
                  value types 0 u8, 1 i8, 2 u16, 3 i16, 4 u32, 5 i32, 6 f32, 7 bool, 8 string, 9 array {u32 elem_type, u64 count, elems (arrays may nest)},
                  10 u64, 11 i64, 12 f64;  gguf_string = u64 length + bytes
    tensor info : gguf_string name | u32 n_dims | u64 dims[n_dims] | u32 ggml_type | u64 offset (relative to the data section, a multiple of general.alignment)
    padding to general.alignment, then the tensor data, each tensor padded to the alignment

Deliberate differences from the repo's own converter, all legal per the specification: 128-byte alignment carried as a u64 value; tokenizer keys FIRST and the llama.*
keys in reverse order; integer hyper-parameters in mixed widths (u16 / u64 / i32 / i64); f64 rope base; extra keys of every scalar type, a string array, an empty
array and a NESTED array; tensors in reverse order with an unused 1-D tensor in the middle; 0xAB filler in every padding gap; trailing bytes after the last tensor.
No third-party GGUF file or `gguf` package exists in this image -- this writer is the closest available stand-in for "a file somebody else wrote".

    python tests/gguf_independent.py      # rewrites tests/golden/tiny_independent_v3.gguf from the seeded GGJT model
"""
import os
import struct
import sys
from typing import Dict, List, Sequence, Tuple

U8, I8, U16, I16, U32, I32, F32, BOOL, STRING, ARRAY, U64, I64, F64 = range(13)
_FMT = {U8: "<B", I8: "<b", U16: "<H", I16: "<h", U32: "<I", I32: "<i", F32: "<f", BOOL: "<?", U64: "<Q", I64: "<q", F64: "<d"}

GGUF_NAMES = {"tok_embeddings.weight": "token_embd.weight", "norm.weight": "output_norm.weight", "output.weight": "output.weight"}
LAYER_NAMES = {"attention_norm.weight": "attn_norm.weight", "attention.wq.weight": "attn_q.weight", "attention.wk.weight": "attn_k.weight",
               "attention.wv.weight": "attn_v.weight", "attention.wo.weight": "attn_output.weight", "ffn_norm.weight": "ffn_norm.weight",
               "feed_forward.w1.weight": "ffn_gate.weight", "feed_forward.w2.weight": "ffn_down.weight", "feed_forward.w3.weight": "ffn_up.weight"}


class Value:
    """One typed metadata value; arrays hold a homogeneous list of Values' payloads."""

    def __init__(self, vtype: int, payload, elem_type: int = -1):
        self.vtype, self.payload, self.elem_type = vtype, payload, elem_type

    def body(self) -> bytes:
        if self.vtype == STRING:
            b = self.payload if isinstance(self.payload, bytes) else self.payload.encode()
            return struct.pack("<Q", len(b)) + b
        if self.vtype == ARRAY:
            out = struct.pack("<IQ", self.elem_type, len(self.payload))
            for e in self.payload:
                out += (e if isinstance(e, Value) else Value(self.elem_type, e)).body()
            return out
        return struct.pack(_FMT[self.vtype], self.payload)


def gguf_string(b: bytes) -> bytes:
    return struct.pack("<Q", len(b)) + b


class IndependentGGUF:
    def __init__(self, version: int = 3, alignment: int = 128, filler: int = 0xAB):
        self.version, self.alignment, self.filler = version, alignment, filler
        self.kv: List[Tuple[str, Value]] = []
        self.tensors: List[Tuple[str, Sequence[int], int, bytes]] = []

    def put(self, key: str, vtype: int, payload, elem_type: int = -1):
        self.kv.append((key, Value(vtype, payload, elem_type)))

    def tensor(self, name: str, dims: Sequence[int], ggml_type: int, data: bytes):
        self.tensors.append((name, tuple(int(d) for d in dims), int(ggml_type), bytes(data)))

    def _pad(self, n: int) -> bytes:
        return bytes([self.filler]) * ((-n) % self.alignment)

    def serialise(self, trailing: bytes = b"") -> bytes:
        head = struct.pack("<4sIQQ", b"GGUF", self.version, len(self.tensors), len(self.kv))
        for key, val in self.kv:
            head += gguf_string(key.encode()) + struct.pack("<I", val.vtype) + val.body()
        offsets, off = [], 0
        for _, _, _, data in self.tensors:
            offsets.append(off)
            off += len(data) + ((-len(data)) % self.alignment)
        for (name, dims, gt, _), o in zip(self.tensors, offsets):
            head += gguf_string(name.encode()) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", gt, o)
        out = bytearray(head + self._pad(len(head)))
        for _, _, _, data in self.tensors:
            out += data + self._pad(len(data))
        return bytes(out) + trailing


def sentencepiece_form(vocab: Sequence[Tuple[bytes, float]]):
    """GGJT-era pieces (raw bytes, plain spaces) -> what a current converter stores: U+2581 for spaces, "<0xXX>" + token_type 6 for the 256 byte tokens at ids 3..258."""
    toks, kinds = [], []
    for i, (piece, _) in enumerate(vocab):
        if 3 <= i <= 258 and len(piece) == 1:
            toks.append(("<0x%02X>" % piece[0]).encode())
            kinds.append(6)                       # BYTE
        else:
            toks.append(b"\xe2\x96\x81".join(piece.split(b" ")))
            kinds.append(2 if i == 0 else 3 if i in (1, 2) else 1)   # UNKNOWN / CONTROL / NORMAL
    return toks, kinds


def convert(ggjt_file, version: int = 3, alignment: int = 128) -> bytes:
    """ggjt_file: an object with .hparams (dict), .vocab [(bytes, float)], .tensors {name: info with .ne / .gtype} and .raw(name) -> uint8 array
    (modelgen.read_llm_file's view of a GGJT v3 file: the INPUT side; nothing of its writer is used)."""
    hp = ggjt_file.hparams
    n_embd, n_head, n_layer, n_mult = hp["n_embd"], hp["n_head"], hp["n_layer"], hp["n_mult"]
    n_ff = -(-(2 * (4 * n_embd) // 3) // n_mult) * n_mult
    toks, kinds = sentencepiece_form(ggjt_file.vocab)
    g = IndependentGGUF(version=version, alignment=alignment)
    # tokenizer first, then odds and ends, then the llama.* block back to front
    g.put("tokenizer.ggml.token_type", ARRAY, kinds, I32)
    g.put("tokenizer.ggml.scores", ARRAY, [float(s) for _, s in ggjt_file.vocab], F32)
    g.put("tokenizer.ggml.tokens", ARRAY, toks, STRING)
    g.put("tokenizer.ggml.model", STRING, "llama")
    g.put("tokenizer.ggml.unknown_token_id", U8, 0)
    g.put("tokenizer.ggml.eos_token_id", I64, 2)
    g.put("tokenizer.ggml.bos_token_id", U16, 1)
    g.put("tokenizer.ggml.add_bos_token", BOOL, True)
    g.put("writer.notes", ARRAY, ["written by tests/gguf_independent.py", "", "▁ é"], STRING)
    g.put("writer.empty", ARRAY, [], F64)
    g.put("writer.nested", ARRAY, [Value(ARRAY, [1, -2, 3], I16), Value(ARRAY, [], I16), Value(ARRAY, [7], I16)], ARRAY)
    g.put("writer.i8", I8, -5)
    g.put("writer.f64", F64, 2.5)
    g.put("general.quantization_version", U32, 2)
    g.put("general.file_type", I32, hp["ftype"])
    g.put("general.alignment", U64, alignment)
    g.put("llama.rope.freq_base", F64, 10000.0)
    g.put("llama.attention.layer_norm_rms_epsilon", F32, 1e-6)
    g.put("llama.attention.head_count_kv", U64, n_head)
    g.put("llama.attention.head_count", U16, n_head)
    g.put("llama.rope.dimension_count", I32, n_embd // n_head)
    g.put("llama.feed_forward_length", U64, n_ff)
    g.put("llama.block_count", U8, n_layer)
    g.put("llama.embedding_length", I64, n_embd)
    g.put("llama.context_length", U32, 2048)
    g.put("general.name", STRING, "independent")
    g.put("general.architecture", STRING, "llama")
    names = list(ggjt_file.tensors)[::-1]
    for k, name in enumerate(names):
        if k == len(names) // 2:
            g.tensor("rope_freqs.weight", (n_embd // n_head // 2,), 0, struct.pack(f"<{n_embd // n_head // 2}f", *[1.0] * (n_embd // n_head // 2)))   # unused by the graph
        info = ggjt_file.tensors[name]
        if name.startswith("layers."):
            _, idx, rest = name.split(".", 2)
            gname = f"blk.{idx}.{LAYER_NAMES[rest]}"
        else:
            gname = GGUF_NAMES[name]
        g.tensor(gname, info.ne, info.gtype, ggjt_file.raw(name).tobytes())
    return g.serialise(trailing=b"trailing bytes a reader must ignore")


FIXTURE_MODEL = dict(wtype="q4_0", n_embd=64, n_layer=2, n_head=2, n_vocab=300, n_mult=32)   # ~90 KB: small enough to commit; other types are converted on the fly by the tests


def write_fixture_source(path: str):
    """The seeded GGJT v3 model the committed fixtures were converted from."""
    from minigpt4_cpp_amd import modelgen as G
    G.write_llm_file(path, G.tiny_llm(**FIXTURE_MODEL), seed=21, std=0.05, **G.TINY_CONDITIONED)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import _pkg
    _pkg.load_package()
    import tempfile
    from minigpt4_cpp_amd import modelgen as G
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "src.bin")
        write_fixture_source(src)
        f = G.read_llm_file(src)
        out = os.path.join(here, "golden", "tiny_independent_v3.gguf")
        open(out, "wb").write(convert(f, version=3, alignment=128))
        print("wrote", out, os.path.getsize(out), "bytes")
