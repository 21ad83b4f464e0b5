"""CPU tier of minigpt4_quantize_model (reference minigpt4.cpp:2817-2982): the C++ block quantisers (csrc/quantize.cpp) against the independent numpy
restatement of ggml's reference quantisers (oracle/refquant.py) byte for byte, hand-derived Q4_0 / Q8_0 blocks, reconstruction-error bounds through the
layout-normative dequantisers (minigpt4.cpp_amd/quants.py), and the file rewrite itself: which tensors change type, byte-identical copies of the rest,
4096-byte alignment, error codes.  No GPU involved -- quantisation is host work in the reference too."""
import ctypes
import os

import numpy as np
import pytest

TYPES = {"q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8, "q2_k": 10, "q3_k": 11, "q4_k": 12, "q5_k": 13, "q6_k": 14}
MG4 = {"f16": 0, "f32": 1, "q4_0": 4, "q4_1": 5, "q5_0": 6, "q5_1": 7, "q8_0": 8, "q2_k": 10, "q3_k": 11, "q4_k": 12, "q5_k": 13, "q6_k": 14}


def cxx_quantize(lib, t, x):
    from minigpt4_cpp_amd import quants as Q
    L = lib.library
    L.minigpt4_amd_quantize_chunk.restype = ctypes.c_int64
    L.minigpt4_amd_quantize_chunk.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    out = np.zeros(Q.nbytes(t, x.size), np.uint8)
    n = L.minigpt4_amd_quantize_chunk(t, x.ctypes.data, out.ctypes.data, x.size)
    assert n == out.size
    return out


def sample_blocks(blk, seed):
    rng = np.random.default_rng(seed)
    n = 6
    x = rng.standard_normal((n + 7, blk)).astype(np.float32)
    x[1] *= 1e-3
    x[2] = np.abs(x[2])                      # all positive: min clamps to 0 in the k-quant search
    x[3] = -np.abs(x[3])
    x[n] = 0.0                               # all-zero block
    x[n + 1] = 0.37                          # constant block (max == min)
    x[n + 2] = 0.0
    x[n + 2, 5] = -2.5                       # single spike
    x[n + 3] = np.linspace(-1, 1, blk)       # ramp
    x[n + 4] = rng.standard_normal(blk) * 50
    x[n + 5] = np.round(rng.standard_normal(blk) * 3) / 3      # many exact ties
    x[n + 6, : blk // 2] = 0.0
    return x


@pytest.mark.parametrize("name", list(TYPES))
def test_cxx_quantisers_equal_numpy_restatement(lib, name):
    import refquant as RQ
    t = TYPES[name]
    blk = RQ.BLOCK_FN[t][0]
    x = sample_blocks(blk, 100 + t)
    got = cxx_quantize(lib, t, x)
    want = RQ.quantize_chunk(t, x)
    per = got.size // x.shape[0]
    bad = [i for i in range(x.shape[0]) if not np.array_equal(got[i * per:(i + 1) * per], want[i * per:(i + 1) * per])]
    assert not bad, (name, bad)


def test_hand_derived_blocks(lib):
    # Q4_0 of the ramp -16..15: the first (and only) |max| is -16 -> d = -16 / -8 = 2, id = 0.5; q = trunc(x / 2 + 8.5) capped at 15
    x = np.arange(-16, 16, dtype=np.float32)
    got = cxx_quantize(lib, TYPES["q4_0"], x)
    q = np.minimum(15, np.trunc(x / 2 + 8.5)).astype(np.uint8)
    want = np.concatenate([np.frombuffer(np.float16(2.0).tobytes(), np.uint8), q[:16] | (q[16:] << 4)])
    assert np.array_equal(got, want)
    # Q8_0 of multiples of 1/127: d = 1/127 rounded to fp16, q = round(x * 127)
    k = np.arange(-16, 16)
    x = (k / 127.0).astype(np.float32)
    x[0] = -1.0                                               # amax = 1 -> d = 1/127
    got = cxx_quantize(lib, TYPES["q8_0"], x)
    assert np.array_equal(got[:2], np.frombuffer(np.float16(np.float32(1.0) / np.float32(127.0)).tobytes(), np.uint8))
    want_q = np.rint(x * (np.float32(1.0) / (np.float32(1.0) / np.float32(127.0)))).astype(np.int8)
    assert np.array_equal(got[2:].view(np.int8), want_q)
    # Q4_1 of 0..31 scaled: min 0, d = 31/15
    x = np.arange(32, dtype=np.float32)
    got = cxx_quantize(lib, TYPES["q4_1"], x)
    d = np.float32(31.0) / np.float32(15.0)
    assert np.array_equal(got[:4], np.frombuffer(np.float16(d).tobytes() + np.float16(0).tobytes(), np.uint8))
    q = np.minimum(15, np.trunc(x * (np.float32(1.0) / d) + np.float32(0.5))).astype(np.uint8)
    assert np.array_equal(got[4:], q[:16] | (q[16:] << 4))


@pytest.mark.parametrize("name,bound", [("q4_0", 0.12), ("q4_1", 0.10), ("q5_0", 0.06), ("q5_1", 0.05), ("q8_0", 0.01), ("q4_k", 0.09), ("q5_k", 0.045), ("q6_k", 0.025), ("q2_k", 0.45), ("q3_k", 0.22)])
def test_reconstruction_error(lib, name, bound):
    from minigpt4_cpp_amd import quants as Q
    t = TYPES[name]
    x = (0.02 * np.random.default_rng(7).standard_normal(256 * 64)).astype(np.float32)
    y = Q.dequantize(t, cxx_quantize(lib, t, x), x.size)
    rel = float(np.sqrt(np.mean((y - x) ** 2)) / np.sqrt(np.mean(x ** 2)))
    assert rel < bound, (name, rel)
    if name.endswith("_k"):                                   # the searched scales must not be worse than the naive min/max encoder of quants.py
        y2 = Q.dequantize(t, Q.quantize(t, x), x.size)
        assert rel <= float(np.sqrt(np.mean((y2 - x) ** 2)) / np.sqrt(np.mean(x ** 2))) * 1.02
    assert cxx_quantize(lib, t, x[:0] if False else x[:256]).size == Q.nbytes(t, 256)
    L = lib.library
    assert L.minigpt4_amd_quantize_chunk(t, x.ctypes.data, x.ctypes.data, 100) == 0      # ragged
    assert L.minigpt4_amd_quantize_chunk(1, x.ctypes.data, x.ctypes.data, 256) == 0        # F16 is not a quantised type


def _eligible(model, t):
    n = t.name
    return (t.gtype in (0, 1) and n.endswith("weight") and len(t.ne) >= 2 and "norm" not in n and "Norm" not in n and model not in ("ln_vision", "query_tokens", "llama_proj")
            and n != "patch_embed.proj.weight")


@pytest.mark.parametrize("ftype", ["f16", "f32"])
@pytest.mark.parametrize("target", ["q4_0", "q5_1", "q8_0", "q4_k", "q6_k", "q2_k", "q3_k"])
def test_quantize_model_rewrites_the_vision_file(lib, tmp_path, ftype, target):
    import refquant as RQ
    from minigpt4_cpp_amd import modelgen as G, quants as Q
    cfg = G.tiny_vision(n_embd_llm=4096, embed_dim=352, mlp_dim=512, q_inter=256)
    cfg.ftype = ftype
    src, dst = str(tmp_path / "v.bin"), str(tmp_path / f"v_{target}.bin")
    G.write_vision_file(src, cfg, seed=5, std=0.05)
    assert lib.library.minigpt4_quantize_model(src.encode(), dst.encode(), MG4[target]) == 0
    a, b = G.read_vision_file(src), G.read_vision_file(dst)
    gt = TYPES[target]
    blk = Q.BLOCK[gt][0]
    assert b.ftype == MG4[target] and b.config == a.config and list(b.models) == list(a.models)
    n_q = 0
    checked = False
    for mname, model in a.models.items():
        assert list(b.models[mname]) == list(model)                                        # same tensors, same order
        for tname, t in model.items():
            u = b.models[mname][tname]
            assert u.ne == t.ne and u.offset % 4096 == 0
            if _eligible(mname, t) and t.ne[0] % blk == 0:
                assert u.gtype == gt, (mname, tname)
                n_q += 1
                if not checked and int(np.prod(t.ne)) <= 352 * 512:                        # one tensor byte for byte against the numpy restatement (first 48 blocks)
                    raw = a.raw(mname, tname)
                    vals = raw.view(np.float16).astype(np.float32) if t.gtype == 1 else raw.view(np.float32)
                    nb = 48
                    want = RQ.quantize_chunk(gt, vals[:nb * blk])
                    assert np.array_equal(b.raw(mname, tname)[:want.size], want)
                    checked = True
            else:
                assert u.gtype == t.gtype and np.array_equal(a.raw(mname, tname), b.raw(mname, tname)), (mname, tname)
    assert n_q > 10 and checked
    # k-quants cannot hold the 352-wide rows: those Linears keep their type, the file stays loadable
    if target.endswith("_k"):
        assert b.models["visual_encoder"]["blocks.0.attn.qkv.weight"].gtype == a.models["visual_encoder"]["blocks.0.attn.qkv.weight"].gtype
        assert b.models["visual_encoder"]["blocks.0.mlp.fc2.weight"].gtype == gt
    assert b.models["visual_encoder"]["patch_embed.proj.weight"].gtype == 1 and b.models["llama_proj"]["weight"].gtype == a.models["llama_proj"]["weight"].gtype
    nv = ctypes.c_int()
    assert lib.library.minigpt4_amd_inspect_files(dst.encode(), None, ctypes.byref(nv), None, None) == 0 and nv.value > 20


def test_quantize_model_error_codes(lib, tmp_path):
    from minigpt4_cpp_amd import modelgen as G
    src = str(tmp_path / "v.bin")
    G.write_vision_file(src, G.tiny_vision(n_embd_llm=4096), seed=5, std=0.05)
    L = lib.library
    assert L.minigpt4_quantize_model(b"/nonexistent/in.bin", b"/tmp/x", MG4["q4_0"]) == 17       # PathDoesNotExist (minigpt4.cpp:2823-2826)
    assert L.minigpt4_quantize_model(src.encode(), b"/nonexistent_dir/out.bin", MG4["q4_0"]) == 18   # DumpModelFileOpen (:1636-1640)
    for bad in (MG4["f16"], MG4["f32"], 2, 3, 9, 15, 99, -1):
        assert L.minigpt4_quantize_model(src.encode(), str(tmp_path / "o.bin").encode(), bad) == 3  # LoadModelMiniGPT4DataType
    # output == input (same path, or a hard link to it): refused before the mmap'd input could be truncated, and the input survives intact
    import hashlib, os
    before = hashlib.sha256(open(src, "rb").read()).hexdigest()
    link = str(tmp_path / "v_link.bin")
    os.link(src, link)
    assert L.minigpt4_quantize_model(src.encode(), src.encode(), MG4["q4_0"]) == 18
    assert L.minigpt4_quantize_model(src.encode(), link.encode(), MG4["q4_0"]) == 18
    assert hashlib.sha256(open(src, "rb").read()).hexdigest() == before
    garbage = tmp_path / "g.bin"
    garbage.write_bytes(b"not a model")
    assert L.minigpt4_quantize_model(str(garbage).encode(), str(tmp_path / "o.bin").encode(), MG4["q4_0"]) == 1   # LoadModelFileHeader
