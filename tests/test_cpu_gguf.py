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
