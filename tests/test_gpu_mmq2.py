"""GPU parity of the second-generation prefill kernels (csrc/mmq2_kernels.hip) against the oracle's ggml mul_mat (Q8_K activations, exact integer sub-block dots):
every token-tile count (1..3 tiles per chunk, several chunks), ragged row / token counts, 1..3 matrices per launch, residual add, forced K splits (fixed-order
combination).  Integer dots are exact; only the order in which the per-super-block fp32 terms are added differs -> 2e-5 of the row maximum, like test_mul_mat_matches_oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # N, n_in, n_out, n_mat, ks, residual
    (5, 768, 70, 1, 0, False),
    (31, 256, 32, 1, 0, True),          # one super-block, one partial tile
    (33, 1024, 130, 2, 0, True),        # 2 tiles, ragged rows across the 4 waves of a workgroup
    (64, 512, 128, 3, 0, False),
    (97, 2048, 256, 1, 0, False),       # 4 tiles (last one 1 token)
    (142, 2048, 256, 3, 0, False),      # the image-turn prompt length: 5 tiles -> 2 chunks of 3 + 2
    (142, 5120, 384, 1, 4, True),       # forced K split with residual (combined in fixed order)
    (300, 1024, 200, 2, 2, False),      # 10 tiles -> 3 chunks, K split
    (512, 2560, 128, 1, 0, True),       # n_batch rows
    (129, 13824, 160, 1, 3, False),     # the w2 row length (54 super-blocks), ragged K split
]


@pytest.mark.parametrize("wtype", ["q4_k", "q5_k", "q6_k", "q4_0"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d_K%d_R%d_m%d_ks%d_res%d" % c)
def test_mmq2_matches_oracle(gpu_lib, wtype, case):
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    N, n_in, n_out, n_mat, ks, with_res = case
    t = Q.NAME_TO_TYPE[wtype]
    rng = np.random.default_rng(sum(map(ord, wtype)) * 977 + sum(case))
    w = (0.05 * rng.standard_normal((n_mat * n_out, n_in))).astype(np.float32)
    raw = Q.quantize(t, w)
    x = rng.standard_normal((N, n_in)).astype(np.float32)
    x[N // 2, : min(256, n_in)] = 0.0                                   # an all-zero Q8_K block (d = 0)
    res = rng.standard_normal((n_mat, N, n_out)).astype(np.float32) if with_res else None
    got = gpu_lib.amd_test_mmq2(t, raw, n_mat, n_in, n_out, x, residual=res, ks=ks)
    want = R.mul_mat(t, raw, n_in, n_mat * n_out, x).reshape(N, n_mat, n_out).transpose(1, 0, 2)
    scale = np.abs(want).max()
    if with_res:
        want = want + res
    assert got.shape == want.shape and np.isfinite(got).all()
    err = float(np.abs(got - want).max() / scale)
    assert err < 2e-5, (wtype, case, err)


@pytest.mark.parametrize("wtype", ["q4_k", "q5_k"])
@pytest.mark.parametrize("case", [CASES[1], CASES[2], CASES[5], CASES[6], CASES[8], CASES[9]], ids=lambda c: "N%d_K%d_R%d_m%d_ks%d_res%d" % c)
def test_fp16_mfma_form_with_scaled_operands_matches_oracle(gpu_lib, wtype, case):
    """k_mmqh_q45k (round 5; measured 10-17 % slower than the int8 kernels and NOT the default -- profiles/r05_prefill_fp16_scaled_operands.md; kept as the record of the
    direction): sub-block scale x quant as exact fp16 integers (<= 1953) in the B operand, the Q8_K activation values as fp16 in the A operand, one fp32 fma per output
    element and half super-block.  Same ggml arithmetic up to the fp32 rounding of the in-block sums -> the same 2e-5 bar as the int8 kernels, and within 2e-6 of them."""
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    N, n_in, n_out, n_mat, ks, with_res = case
    t = Q.NAME_TO_TYPE[wtype]
    rng = np.random.default_rng(sum(map(ord, wtype)) * 977 + sum(case) + 4)
    raw = Q.quantize(t, (0.05 * rng.standard_normal((n_mat * n_out, n_in))).astype(np.float32))
    x = rng.standard_normal((N, n_in)).astype(np.float32)
    x[N // 2, : min(256, n_in)] = 0.0
    res = rng.standard_normal((n_mat, N, n_out)).astype(np.float32) if with_res else None
    got = gpu_lib.amd_test_mmq2(t, raw, n_mat, n_in, n_out, x, residual=res, ks=ks, generation=4)
    int8 = gpu_lib.amd_test_mmq2(t, raw, n_mat, n_in, n_out, x, residual=res, ks=ks, generation=2)
    want = R.mul_mat(t, raw, n_in, n_mat * n_out, x).reshape(N, n_mat, n_out).transpose(1, 0, 2)
    scale = np.abs(want).max()
    if with_res:
        want = want + res
    assert np.isfinite(got).all()
    assert float(np.abs(got - want).max() / scale) < 2e-5
    assert float(np.abs(got - int8).max() / scale) < 2e-6


def test_mmq2_is_deterministic_with_k_split(gpu_lib):
    """The K-split partial sums are combined in a fixed order: two runs give identical bits."""
    from minigpt4_cpp_amd import quants as Q
    rng = np.random.default_rng(11)
    t = Q.NAME_TO_TYPE["q5_k"]
    raw = Q.quantize(t, (0.05 * rng.standard_normal((256, 5120))).astype(np.float32))
    x = rng.standard_normal((142, 5120)).astype(np.float32)
    a = gpu_lib.amd_test_mmq2(t, raw, 1, 5120, 256, x, ks=5)
    b = gpu_lib.amd_test_mmq2(t, raw, 1, 5120, 256, x, ks=5)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("wtype", ["q4_k", "q5_k"])
def test_mmq2_extreme_values_stay_exact(gpu_lib, wtype):
    """Weights that drive every quant and every 6-bit scale to its maximum against activation rows of +-127 everywhere (block sums at their extremes: the min term's
    digit split 128 hi + lo sees its largest values) must still equal the oracle's integers."""
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    t = Q.NAME_TO_TYPE[wtype]
    rng = np.random.default_rng(5)
    n_in, n_out, N = 512, 64, 40
    w = np.where(rng.random((n_out, n_in)) < 0.5, 1.0, -1.0).astype(np.float32) * rng.choice([0.01, 1.0, 7.5], size=(n_out, n_in // 32)).repeat(32, axis=1).astype(np.float32)
    w[0, :] = 3.0                                                        # constant rows: scales 0 / mins maximal
    w[1, :] = np.tile(np.linspace(-4, 4, 32), n_in // 32)                # full quant range in every sub-block
    raw = Q.quantize(t, w)
    x = np.where(rng.random((N, n_in)) < 0.5, 1.0, -1.0).astype(np.float32)
    x[0, :] = 1.0; x[1, :] = -1.0                                        # every int8 at +127 / -127... (Q8_K maps the max to -128 -> iscale sign)
    got = gpu_lib.amd_test_mmq2(t, raw, 1, n_in, n_out, x)
    want = R.mul_mat(t, raw, n_in, n_out, x).reshape(N, 1, n_out).transpose(1, 0, 2)
    assert float(np.abs(got - want).max() / np.abs(want).max()) < 2e-5


F16_CASES = [
    # N, n_in, n_out, n_mat, ks, residual -- the F16 language-model set path needs >= 512 rows, n_out % 128 == 0, n_in % 64 == 0
    (512, 512, 256, 3, 0, False),        # wq|wk|wv in one launch
    (512, 1024, 384, 2, 0, False),       # w1|w3
    (640, 2048, 128, 1, 2, True),        # split K with residual (wo / w2), ragged last row tile
    (512, 4096, 256, 1, 3, True),
    (515, 512, 128, 1, 0, True),         # no split, residual in the epilogue, ragged rows
]


@pytest.mark.parametrize("case", F16_CASES, ids=lambda c: "N%d_K%d_R%d_m%d_ks%d_res%d" % c)
def test_f16_set_gemm_matches_fp32_reference(gpu_lib, case):
    """Unquantised weights at prompt sizes (BASELINE configs[4]): ggml rounds the activation rows to fp16 and accumulates exact fp16 x fp16 products in fp32 -- which is
    what v_mfma_f32_32x32x16_f16 does; only the accumulation order differs -> 2e-5 of the row maximum against a float64 product of the same fp16-rounded operands."""
    from minigpt4_cpp_amd import quants as Q
    N, n_in, n_out, n_mat, ks, with_res = case
    rng = np.random.default_rng(sum(case))
    w = (0.05 * rng.standard_normal((n_mat * n_out, n_in))).astype(np.float16)
    x = rng.standard_normal((N, n_in)).astype(np.float32)
    res = rng.standard_normal((n_mat, N, n_out)).astype(np.float32) if with_res else None
    got = gpu_lib.amd_test_mmq2(Q.NAME_TO_TYPE["f16"], w.view(np.uint8).reshape(-1), n_mat, n_in, n_out, x, residual=res, ks=ks)
    want = (x.astype(np.float16).astype(np.float64) @ w.astype(np.float64).T).reshape(N, n_mat, n_out).transpose(1, 0, 2)
    scale = np.abs(want).max()
    if with_res:
        want = want + res
    assert got.shape == want.shape and np.isfinite(got).all()
    assert float(np.abs(got - want).max() / scale) < 2e-5


@pytest.mark.parametrize("case", [(512, 512, 256), (640, 1024, 768), (515, 256, 128), (512, 2048, 1408), (512, 256, 12416)], ids=lambda c: "N%d_K%d_F%d" % c)
def test_f16_silu_pair_launch_is_bit_identical_to_the_two_products(gpu_lib, case):
    """F16 weights at prompt sizes: w1 | w3 in ONE launch whose epilogue stores fp16(silu_table(w1 x) * (w3 x)) (k_gemm_dma PAIR, round 3).  Each product is accumulated in
    the order of every other tile shape, so the stored row must equal -- bit for bit -- what the two-product set launch followed by k_silu_mul_quant's fp16 leg produces:
    silu through ggml's fp16 table on the fp16-rounded w1 product, times the fp32 w3 product, rounded to fp16.  (The last case has enough column tiles for the 256x256
    form; the reference launch needs n_out % 128 == 0.)"""
    import ctypes
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    N, n_in, n_out = case
    rng = np.random.default_rng(sum(case))
    w = (0.05 * rng.standard_normal((2 * n_out, n_in))).astype(np.float16)
    x = rng.standard_normal((N, n_in)).astype(np.float32)
    h = gpu_lib.amd_test_mmq2(Q.NAME_TO_TYPE["f16"], w.view(np.uint8).reshape(-1), 2, n_in, n_out, x)          # [2][N][n_out] fp32, the set launch
    tab = R.table(1).view(np.float16)                                                                           # ggml's table_silu_f16
    want = (tab[h[0].astype(np.float16).view(np.uint16)].astype(np.float32) * h[1]).astype(np.float16)
    L = gpu_lib.library
    L.minigpt4_amd_test_f16_silu_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    got = np.zeros((N, n_out), np.uint16)
    xs = np.ascontiguousarray(x); ws = np.ascontiguousarray(w)
    rc = L.minigpt4_amd_test_f16_silu_pair(xs.ctypes.data, ws.ctypes.data, N, n_in, n_out, got.ctypes.data, None)
    assert rc == 0, rc
    assert np.array_equal(got, want.view(np.uint16)), int((got != want.view(np.uint16)).sum())
