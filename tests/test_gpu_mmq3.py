"""GPU parity of the digit-plane prompt mat-mul (csrc/mmq3_kernels.hip: Q4_K / Q5_K, sub-block scale x quant stored at load time as 128 hi + lo in MFMA-fragment order,
LDS-DMA ring) against the oracle's ggml mul_mat AND against k_mmq2_q45k: the integers are the same and the fp32 updates run in the same order, so for the same K split
the two kernels must agree bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_test_extras(gpu_lib):
    # the digit-plane kernels are a closed direction (round 4): they are in the test library only after `make -C minigpt4.cpp_amd/csrc test-extras`
    if not gpu_lib.library.minigpt4_amd_test_extras():
        pytest.skip("libminigpt4_test.so was built without the closed-direction kernels (`make test-extras`)")

CASES = [
    # N, n_in, n_out, n_mat, ks, residual
    (5, 768, 70, 1, 1, False),
    (31, 256, 32, 1, 1, True),          # one super-block, one partial tile
    (33, 1024, 130, 2, 1, True),        # ragged rows across the waves of a workgroup, 3 token tiles in a 4-tile kernel
    (64, 512, 128, 3, 1, False),
    (97, 2048, 256, 1, 1, False),       # 7 token tiles (last one 1 token) in the 8-tile kernel
    (142, 2048, 256, 3, 1, False),      # the image-turn prompt length: 9 tiles, one pass
    (142, 5120, 384, 1, 4, True),       # K split with residual (combined in fixed order)
    (300, 1024, 200, 2, 2, False),      # 19 tiles -> 3 chunks of 7
    (512, 2560, 128, 1, 1, True),       # n_batch rows: 4 chunks of 8 tiles
    (129, 13824, 160, 1, 3, False),     # the w2 row length (54 super-blocks), ragged K split
]


@pytest.mark.parametrize("wtype", ["q4_k", "q5_k"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d_K%d_R%d_m%d_ks%d_res%d" % c)
def test_mmq3_matches_oracle_and_mmq2(gpu_lib, wtype, case):
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
    got = gpu_lib.amd_test_mmq2(t, raw, n_mat, n_in, n_out, x, residual=res, ks=ks, generation=3)
    same = gpu_lib.amd_test_mmq2(t, raw, n_mat, n_in, n_out, x, residual=res, ks=ks, generation=2)
    want = R.mul_mat(t, raw, n_in, n_mat * n_out, x).reshape(N, n_mat, n_out).transpose(1, 0, 2)
    scale = np.abs(want).max()
    if with_res:
        want = want + res
    assert got.shape == want.shape and np.isfinite(got).all()
    err = float(np.abs(got - want).max() / scale)
    assert err < 2e-5, (wtype, case, err)
    assert np.array_equal(got, same), (wtype, case, float(np.abs(got - same).max()))


@pytest.mark.parametrize("wtype", ["q4_k", "q5_k"])
def test_mmq3_extreme_values_stay_exact(gpu_lib, wtype):
    """Every quant and every 6-bit scale at its maximum (scale x quant = 63 x 31: the 4-bit high digit at 15) against +-127 activations."""
    import refcpu as R
    from minigpt4_cpp_amd import quants as Q
    t = Q.NAME_TO_TYPE[wtype]
    rng = np.random.default_rng(5)
    n_in, n_out, N = 512, 64, 40
    w = np.where(rng.random((n_out, n_in)) < 0.5, 1.0, -1.0).astype(np.float32) * rng.choice([0.01, 1.0, 7.5], size=(n_out, n_in // 32)).repeat(32, axis=1).astype(np.float32)
    w[0, :] = 3.0
    w[1, :] = np.tile(np.linspace(-4, 4, 32), n_in // 32)
    raw = Q.quantize(t, w)
    x = np.where(rng.random((N, n_in)) < 0.5, 1.0, -1.0).astype(np.float32)
    x[0, :] = 1.0; x[1, :] = -1.0
    got = gpu_lib.amd_test_mmq2(t, raw, 1, n_in, n_out, x, ks=1, generation=3)
    want = R.mul_mat(t, raw, n_in, n_out, x).reshape(N, 1, n_out).transpose(1, 0, 2)
    assert float(np.abs(got - want).max() / np.abs(want).max()) < 2e-5
    assert np.array_equal(got, gpu_lib.amd_test_mmq2(t, raw, 1, n_in, n_out, x, ks=1, generation=2))
