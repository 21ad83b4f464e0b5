"""GPU parity of the image path for vision files whose Linear weights are not F16 (csrc/engine_vision_generic.cpp): files written by minigpt4_quantize_model
(reference minigpt4.cpp:2817-2982) and `--ftype f32` conversions (convert.py:113-121).  The reference runs the same graph on them; ggml's mul_mat then
quantises every activation row to the weight type's vec_dot_type and takes exact integer block dots -- the oracle (refcpu.c `linear` -> `orc_mul_mat`) and
the engine (LLM mat-mul kernels) both do exactly that, so they differ only by fp32 summation order, amplified by the int8 rounding of activations
(DESIGN.md "Whole-model tolerance"; observed 1.3e-2 .. 1.9e-2 for every block type): 4e-2 of the embedding range, inside the 5e-2 bar of the LLM
whole-model tests; F32 weights have no rounding step and must agree to 2e-3 (observed 2e-4)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MG4 = {"q4_0": 4, "q4_1": 5, "q5_0": 6, "q5_1": 7, "q8_0": 8, "q2_k": 10, "q3_k": 11, "q4_k": 12, "q5_k": 13, "q6_k": 14}


@pytest.fixture(scope="module")
def base(tmp_path_factory):
    from minigpt4_cpp_amd import modelgen as G
    d = tmp_path_factory.mktemp("qvision")
    cfg = G.tiny_vision(n_embd_llm=4096, embed_dim=352, mlp_dim=512, q_inter=256)      # 352 = 4 heads x 88 = 11 blocks of 32; fc2 / Q-Former rows hold k-quant super-blocks
    src = str(d / "vision_f16.bin")
    G.write_vision_file(src, cfg, seed=21, std=0.05)
    cfg32 = G.tiny_vision(n_embd_llm=4096, embed_dim=352, mlp_dim=512, q_inter=256)
    cfg32.ftype = "f32"
    src32 = str(d / "vision_f32.bin")
    G.write_vision_file(src32, cfg32, seed=21, std=0.05)
    return str(d), src, src32


def encode_both(gpu_lib, vp, lp, seed=42):
    import refcpu as R
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    img = G.synth_image(seed)
    ctx = gpu_lib.minigpt4_model_load(vp, lp, verbosity=0, n_ctx=64, n_batch=32)
    try:
        emb = gpu_lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(img))
        got = np.ctypeslib.as_array(emb.data, shape=(emb.n_embeddings,)).copy().reshape(32, -1)
        gpu_lib.minigpt4_free_embedding(emb)
    finally:
        gpu_lib.minigpt4_free(ctx)
    want = R.OracleVision(G.read_vision_file(vp)).encode(img)
    return got, want


@pytest.mark.parametrize("target", list(MG4))
def test_quantised_vision_file_matches_oracle(gpu_lib, base, tiny_files, target):
    d, src, _ = base
    _, llm = tiny_files
    dst = os.path.join(d, f"vision_{target}.bin")
    assert gpu_lib.library.minigpt4_quantize_model(src.encode(), dst.encode(), MG4[target]) == 0
    got, want = encode_both(gpu_lib, dst, llm("q4_0"))
    err = float(np.abs(got - want).max() / np.abs(want).max())
    print(f"quantised vision {target}: rel err {err:.2e}")
    assert np.isfinite(got).all() and err < 4e-2, (target, err)
    # and the quantised tower is a different function than the f16 one (the quantised weights were really used)
    got16, _ = encode_both(gpu_lib, src, llm("q4_0"))
    assert float(np.abs(got - got16).max() / np.abs(want).max()) > 1e-4


def test_f32_vision_file_matches_oracle(gpu_lib, base, tiny_files):
    _, _, src32 = base
    _, llm = tiny_files
    got, want = encode_both(gpu_lib, src32, llm("q4_0"))
    err = float(np.abs(got - want).max() / np.abs(want).max())
    print(f"f32 vision: rel err {err:.2e}")
    assert err < 2e-3, err


def test_generic_path_batches_images_bit_identically(gpu_lib, base, tiny_files):
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    d, src, _ = base
    _, llm = tiny_files
    dst = os.path.join(d, "vision_q5_0_batch.bin")
    assert gpu_lib.library.minigpt4_quantize_model(src.encode(), dst.encode(), MG4["q5_0"]) == 0
    ctx = gpu_lib.minigpt4_model_load(dst, llm("q4_0"), verbosity=0, n_ctx=64, n_batch=32)
    try:
        imgs = [G.synth_image(s) for s in (5, 6, 7)]
        structs = (ML.MiniGPT4Image * 3)(*[ML.array_to_image_struct(i) for i in imgs])
        batch, out = ML.MiniGPT4Images(structs, 3), ML.MiniGPT4Embeddings()
        assert gpu_lib.library.minigpt4_amd_encode_images(ctx.ptr, ctypes.byref(batch), ctypes.byref(out), 0) == 0
        for i, img in enumerate(imgs):
            single = gpu_lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(img))
            a = np.ctypeslib.as_array(single.data, shape=(single.n_embeddings,)).copy()
            b = np.ctypeslib.as_array(out.embeddings[i].data, shape=(out.embeddings[i].n_embeddings,)).copy()
            assert np.array_equal(a, b)
            gpu_lib.minigpt4_free_embedding(single)
        gpu_lib.library.minigpt4_amd_free_embeddings(ctypes.byref(out))
    finally:
        gpu_lib.minigpt4_free(ctx)
