"""GGUF reader (csrc/formats.cpp LLMFile::load_gguf): a GGUF v2 / v3 re-container of a GGJT v3 model must give the engine exactly the same
view -- hyper-parameters, vocabulary in the pinned tokenizer's form, every tensor's type / shape / bytes -- checked through a digest of that view and through the
tokenizer; unsupported variants are refused with LoadLanguageModel.  Host logic only."""
import ctypes
import struct

import pytest


def digest(lib, path, with_data=1):
    d = ctypes.c_uint64()
    lib.library.minigpt4_amd_llm_file_digest.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    rc = lib.library.minigpt4_amd_llm_file_digest(path.encode(), ctypes.byref(d), with_data)
    return rc, d.value


@pytest.mark.parametrize("wtype,mix", [("q4_0", "none"), ("q5_k", "q5_k_m"), ("f16", "none"), ("q3_k", "none")])
@pytest.mark.parametrize("version,alignment", [(3, 32), (2, 64)])
def test_gguf_gives_the_same_view_as_ggjt(lib, tiny_files, tmp_path, wtype, mix, version, alignment):
    from minigpt4_cpp_amd import modelgen as G
    _, llm = tiny_files
    src = llm(wtype, mix)
    dst = str(tmp_path / "m.gguf")
    G.ggjt_to_gguf(src, dst, version=version, alignment=alignment)
    a, b = digest(lib, src), digest(lib, dst)
    assert a[0] == 0 and b[0] == 0, (a, b, lib.library.minigpt4_amd_last_error())
    assert a[1] == b[1]
    # the vocabulary arrives in the GGJT-era form (spaces, raw bytes), so the pinned tokenizer gives the same ids
    L = lib.library
    va, vb = L.minigpt4_amd_vocab_load(src.encode()), L.minigpt4_amd_vocab_load(dst.encode())
    try:
        assert va and vb and L.minigpt4_amd_vocab_size(va) == L.minigpt4_amd_vocab_size(vb)
        for text in (b"Human: <Img>", b"what is the text in the picture?### Assistant:", "héllo ▁ wörld \xff".encode("utf-8", "surrogatepass"), b"\x00\x01 a  b"):
            oa, ob = (ctypes.c_int32 * 256)(), (ctypes.c_int32 * 256)()
            na, nb = L.minigpt4_amd_vocab_tokenize(va, text, 1, oa, 256), L.minigpt4_amd_vocab_tokenize(vb, text, 1, ob, 256)
            assert na == nb and list(oa[:na]) == list(ob[:nb])
    finally:
        L.minigpt4_amd_vocab_free(va)
        L.minigpt4_amd_vocab_free(vb)
    nl = ctypes.c_int()
    wb1, wb2 = ctypes.c_int64(), ctypes.c_int64()
    assert L.minigpt4_amd_inspect_files(None, src.encode(), None, ctypes.byref(nl), ctypes.byref(wb1)) == 0
    assert L.minigpt4_amd_inspect_files(None, dst.encode(), None, ctypes.byref(nl), ctypes.byref(wb2)) == 0 and wb1.value == wb2.value


def _patch_kv_f32(data: bytearray, key: bytes, value: float):
    i = data.find(key)
    assert i > 0
    struct.pack_into("<f", data, i + len(key) + 4, value)          # key bytes, u32 type (6 = f32), then the value


def test_gguf_variants_outside_the_reference_graph_are_refused(lib, tiny_files, tmp_path):
    from minigpt4_cpp_amd import modelgen as G
    _, llm = tiny_files
    dst = str(tmp_path / "m.gguf")
    G.ggjt_to_gguf(llm("q4_0"), dst)
    good = bytearray(open(dst, "rb").read())
    bad = str(tmp_path / "bad.gguf")

    def rc_of(b):
        open(bad, "wb").write(bytes(b))
        return digest(lib, bad)[0]
    b = bytearray(good); _patch_kv_f32(b, b"llama.attention.layer_norm_rms_epsilon", 1e-5)
    assert rc_of(b) == 4 and b"epsilon" in lib.library.minigpt4_amd_last_error()           # LLaMA-2 style eps: the kernels implement the pinned 1e-6
    b = bytearray(good); _patch_kv_f32(b, b"llama.rope.freq_base", 1e6)
    assert rc_of(b) == 4
    b = bytearray(good); i = b.find(b"llama.attention.head_count_kv"); struct.pack_into("<I", b, i + len(b"llama.attention.head_count_kv") + 4, 1)
    assert rc_of(b) == 4 and b"grouped-query" in lib.library.minigpt4_amd_last_error()
    b = bytearray(good); struct.pack_into("<I", b, 4, 1)                                      # GGUF v1
    assert rc_of(b) == 4
    b = bytearray(good); i = b.find(b"llama", 24); b[i:i + 5] = b"gpt2x"                     # general.architecture value
    assert rc_of(b) == 4
    for cut in (10, 100, 3000, len(good) // 2, len(good) - 1):                              # truncations
        assert rc_of(good[:cut]) == 4
    b = bytearray(good); struct.pack_into("<Q", b, 8, 1 << 60)                                # absurd tensor count
    assert rc_of(b) == 4
    b = bytearray(good); struct.pack_into("<Q", b, 16, 1 << 60)                               # absurd kv count
    assert rc_of(b) == 4


# ---- files from an INDEPENDENT writer (tests/gguf_independent.py: written from the GGUF specification, no code shared with modelgen.ggjt_to_gguf) -------------------------
def _fixture_paths(tmp_path):
    import os
    import gguf_independent as GI
    src = str(tmp_path / "fixture_src.bin")
    GI.write_fixture_source(src)
    return src, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_independent_v3.gguf")


def test_committed_independent_gguf_fixture_matches_its_ggjt_source(lib, tmp_path):
    """tests/golden/tiny_independent_v3.gguf is a COMMITTED file (128-byte alignment as a u64, tokenizer keys first, mixed integer widths, f64 rope base, a nested array,
    tensors in reverse order, 0xAB padding, trailing bytes): the loader's view of it -- hyper-parameters, vocabulary, every tensor's type / shape / bytes -- must equal
    its view of the GGJT v3 model regenerated from the seed."""
    src, fx = _fixture_paths(tmp_path)
    a, b = digest(lib, src), digest(lib, fx)
    assert a[0] == 0 and b[0] == 0, (a, b, lib.library.minigpt4_amd_last_error())
    assert a[1] == b[1]
    L = lib.library
    va, vb = L.minigpt4_amd_vocab_load(src.encode()), L.minigpt4_amd_vocab_load(fx.encode())
    try:
        assert va and vb and L.minigpt4_amd_vocab_size(va) == L.minigpt4_amd_vocab_size(vb) == 300
        for text in (b"Human: <Img>", b"### Assistant: what  is this?", "▁ é \xff".encode("utf-8", "surrogatepass")):
            oa, ob = (ctypes.c_int32 * 128)(), (ctypes.c_int32 * 128)()
            na, nb = L.minigpt4_amd_vocab_tokenize(va, text, 1, oa, 128), L.minigpt4_amd_vocab_tokenize(vb, text, 1, ob, 128)
            assert na == nb and list(oa[:na]) == list(ob[:nb])
    finally:
        L.minigpt4_amd_vocab_free(va)
        L.minigpt4_amd_vocab_free(vb)


@pytest.mark.parametrize("wtype,mix", [("q5_k", "q5_k_m"), ("f16", "none"), ("q3_k", "none"), ("q8_0", "none")])
@pytest.mark.parametrize("version,alignment", [(3, 64), (2, 256)])
def test_independent_writer_other_types_and_versions(lib, tiny_files, tmp_path, wtype, mix, version, alignment):
    import gguf_independent as GI
    from minigpt4_cpp_amd import modelgen as G
    _, llm = tiny_files
    src = llm(wtype, mix)
    dst = str(tmp_path / "ind.gguf")
    open(dst, "wb").write(GI.convert(G.read_llm_file(src), version=version, alignment=alignment))
    a, b = digest(lib, src), digest(lib, dst)
    assert a[0] == 0 and b[0] == 0, (a, b, lib.library.minigpt4_amd_last_error())
    assert a[1] == b[1]
    # and the repo's own converter agrees with the independent one
    own = str(tmp_path / "own.gguf")
    G.ggjt_to_gguf(src, own, version=version, alignment=32)
    assert digest(lib, own)[1] == b[1]


def test_independent_writer_unknown_value_type_and_deep_nesting_are_refused(lib, tmp_path):
    import gguf_independent as GI
    from minigpt4_cpp_amd import modelgen as G
    src, _ = _fixture_paths(tmp_path)
    f = G.read_llm_file(src)
    good = GI.convert(f)
    bad = str(tmp_path / "bad.gguf")
    # a value type the specification does not define (13): its size is unknowable, the file cannot be parsed past it
    i = good.find(b"writer.i8")
    b = bytearray(good); struct.pack_into("<I", b, i + len(b"writer.i8"), 13)
    open(bad, "wb").write(bytes(b))
    assert digest(lib, bad)[0] == 4
    # arrays nested deeper than the loader's bound
    g = GI.IndependentGGUF()
    deep = GI.Value(GI.ARRAY, [1], GI.I16)
    for _ in range(6):
        deep = GI.Value(GI.ARRAY, [deep], GI.ARRAY)
    g.kv.append(("writer.deep", deep))
    g.put("general.architecture", GI.STRING, "llama")
    open(bad, "wb").write(g.serialise())
    assert digest(lib, bad)[0] == 4
