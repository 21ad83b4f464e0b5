"""CPU-only tier: Q3_K (ggml type 11).  The gfx950 kernels do not stream Q3_K natively; the loader re-encodes every Q3_K super-block as a Q6_K
super-block that has the same dequantised values AND the same integer block dot products (both formats are d * scale_16 * q over sixteen 16-wide
sub-blocks), so ggml's Q3_K arithmetic is reproduced exactly by the Q6_K path.  Checked here: the block layout (hand-built vector, numpy vs C oracle),
the converter (C++ product vs numpy twin, value identity, bit-identical oracle dot products) and the oracle's whole-model Q3_K forward."""
import ctypes
import os

import numpy as np
import pytest


def _random_q3k_bytes(rng, nb):
    """Arbitrary super-blocks: every bit pattern of hmask / qs / scales is valid; d is a sane fp16."""
    b = rng.integers(0, 256, (nb, 110), dtype=np.uint8)
    b[:, 108:110] = (rng.standard_normal(nb) * 0.01).astype(np.float16).view(np.uint8).reshape(nb, 2)
    return b.reshape(-1)


def test_hand_built_q3k_block():
    """One super-block written from the documented layout (k_quants.h block_q3_K): hmask[32] | qs[64] | scales[12] | d."""
    from minigpt4_cpp_amd import quants as Q
    import refcpu as R
    b = np.zeros(110, np.uint8)
    # element e = 128 n + 32 j + l: low 2 bits = (qs[32 n + l] >> 2 j) & 3, high bit = hmask[l] bit (4 n + j); value = low - (high ? 0 : 4)
    want_q = np.zeros(256, np.int32)
    for e in range(256):
        n, j, l = e // 128, (e % 128) // 32, e % 32
        v = (e * 7 + 3) % 8 - 4                        # -4..3
        want_q[e] = v
        low, high = (v + 4) & 3, (v + 4) >> 2
        b[32 + 32 * n + l] |= low << (2 * j)
        b[l] |= high << (4 * n + j)
    # scale of sub-block s (16 elements): 6-bit value L; low nibble in byte s (s < 8) / high nibble of byte s - 8, top 2 bits in byte 8 + s % 4 at 2 * (s / 4)
    L = [(5 * s + 11) % 64 for s in range(16)]
    for s, v in enumerate(L):
        if s < 8:
            b[96 + s] |= v & 15
        else:
            b[96 + s - 8] |= (v & 15) << 4
        b[96 + 8 + s % 4] |= (v >> 4) << (2 * (s // 4))
    b[108:110] = np.array([0.5], np.float16).view(np.uint8)
    want = np.array([0.5 * (L[e // 16] - 32) * want_q[e] for e in range(256)], np.float64)
    assert np.array_equal(Q.dequantize(Q.GGML_Q3_K, b, 256), want)
    assert np.array_equal(R.dequantize_row(Q.GGML_Q3_K, b, 256).astype(np.float64), want)


def test_q3k_numpy_and_c_oracle_agree_on_arbitrary_blocks():
    from minigpt4_cpp_amd import quants as Q
    import refcpu as R
    raw = _random_q3k_bytes(np.random.default_rng(1), 64)
    a = Q.dequantize(Q.GGML_Q3_K, raw, 64 * 256)
    b = R.dequantize_row(Q.GGML_Q3_K, raw, 64 * 256)
    assert np.array_equal(a.astype(np.float32), b)


def test_q3k_quantiser_round_trip_quality():
    from minigpt4_cpp_amd import quants as Q
    x = (0.05 * np.random.default_rng(2).standard_normal(256 * 32)).astype(np.float32)
    raw = Q.quantize(Q.GGML_Q3_K, x)
    assert raw.size == Q.nbytes(Q.GGML_Q3_K, x.size) == 32 * 110
    y = Q.dequantize(Q.GGML_Q3_K, raw, x.size)
    assert np.sqrt(((y - x) ** 2).mean()) / np.sqrt((x ** 2).mean()) < 0.25       # 3-bit weights
    assert not Q.dequantize(Q.GGML_Q3_K, Q.quantize(Q.GGML_Q3_K, np.zeros(256, np.float32)), 256).any()


@pytest.mark.parametrize("source", ["quantised", "arbitrary"])
def test_q3k_to_q6k_is_lossless(lib, source):
    """The product's load-time converter == its numpy twin; the Q6_K image dequantises to the same values and gives bit-identical dot products."""
    from minigpt4_cpp_amd import quants as Q
    import refcpu as R
    rng = np.random.default_rng(3)
    n_in, n_out = 1024, 24
    if source == "quantised":
        raw = Q.quantize(Q.GGML_Q3_K, (0.05 * rng.standard_normal((n_out, n_in))).astype(np.float32))
    else:
        raw = _random_q3k_bytes(rng, n_in // 256 * n_out)
    nb = n_in // 256 * n_out
    twin = Q.q3_k_to_q6_k(raw, n_in * n_out)
    got = np.zeros(nb * 210, np.uint8)
    lib.library.minigpt4_amd_convert_q3k_q6k.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    assert lib.library.minigpt4_amd_convert_q3k_q6k(np.ascontiguousarray(raw).ctypes.data, got.ctypes.data, nb) == 0
    assert np.array_equal(got, twin)
    assert np.array_equal(Q.dequantize(Q.GGML_Q6_K, got, n_in * n_out), Q.dequantize(Q.GGML_Q3_K, raw, n_in * n_out))
    x = rng.standard_normal((5, n_in)).astype(np.float32)
    y3 = R.mul_mat(Q.GGML_Q3_K, raw, n_in, n_out, x)
    y6 = R.mul_mat(Q.GGML_Q6_K, got, n_in, n_out, x)
    assert np.array_equal(y3, y6)                       # same Q8_K activations, same integer sums, same fp32 scale products
    f64 = x.astype(np.float64) @ Q.dequantize(Q.GGML_Q3_K, raw, n_in * n_out).reshape(n_out, n_in).T
    assert np.abs(y3 - f64).max() / np.abs(f64).max() < 3e-2


def test_oracle_q3k_model_matches_float64(tiny_files):
    import refcpu as R
    import f64ref as F
    from minigpt4_cpp_amd import modelgen as G
    _, llm = tiny_files
    f = G.read_llm_file(llm("q3_k"))
    assert {t.gtype for t in f.tensors.values()} == {0, 11}
    toks = np.random.default_rng(4).integers(3, 512, 20)
    got = R.OracleLLM(f, n_ctx=64).eval_tokens(toks, all_logits=True)
    want = F.LlamaF64(f).eval(tokens=toks)
    assert np.abs(got - want).max() / np.abs(want).max() < 6e-2


def test_q3k_file_is_accepted_by_the_parsers(lib, tiny_files):
    _, llm = tiny_files
    n = ctypes.c_int(0)
    bpt = ctypes.c_int64(0)
    assert lib.library.minigpt4_amd_inspect_files(None, llm("q3_k").encode(), None, ctypes.byref(n), ctypes.byref(bpt)) == 0
    assert n.value == 3 + 2 * 9 and bpt.value > 0
