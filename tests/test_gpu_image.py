"""GPU parity of minigpt4_preprocess_image (HIP kernels in csrc/image_kernels.hip) against the oracle (oracle/refimage.py, pinned to Pillow by
tests/test_cpu_image.py) and against the committed Pillow goldens, through the C ABI.

Bar: the resized bytes are integer work -> bit-exact (checked by inverting the float tail and by sha256 against Pillow's own output);
the float tail is three correctly rounded fp32 operations -> at most 1 ulp of the result (5e-7 absolute; equality is expected).
"""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from test_cpu_image import GOLD, IMG_DIR, resize_case

pytestmark = pytest.mark.gpu


def preprocess(lib, ctx_ptr, rgb):
    from minigpt4_cpp_amd import minigpt4_library as ML
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    img = ML.MiniGPT4Image(rgb.ctypes.data_as(ctypes.c_void_p), rgb.shape[1], rgb.shape[0], 3, 2)
    out = ML.MiniGPT4Image()
    rc = lib.library.minigpt4_preprocess_image(ctx_ptr, ctypes.byref(img), ctypes.byref(out), 0)
    assert rc == 0, (rc, lib.library.minigpt4_amd_last_error())
    assert (out.width, out.height, out.channels, out.format) == (1, 3 * 224 * 224, 1, 1)     # the reference's reported geometry (minigpt4.cpp:2639-2643)
    x = np.ctypeslib.as_array(ctypes.cast(out.data, ctypes.POINTER(ctypes.c_float)), shape=(3, 224, 224)).copy()
    assert lib.library.minigpt4_free_image(ctypes.byref(out)) == 0
    return x


def to_u8(x):
    import refimage as R
    mean, std = np.asarray(R.CLIP_MEAN, np.float32), np.asarray(R.CLIP_STD, np.float32)
    return np.rint((x.transpose(1, 2, 0).astype(np.float64) * std + mean) * 255.0).astype(np.int64)


@pytest.mark.parametrize("i", range(len(GOLD["resize_names"])))
def test_preprocess_equals_oracle_and_pillow(gpu_lib, i):
    import refimage as R
    a = resize_case(i)
    got = preprocess(gpu_lib, None, a)
    want = R.preprocess(a)
    u8 = to_u8(got)
    assert hashlib.sha256(u8.astype(np.uint8).tobytes()).hexdigest() == str(GOLD["resize_sha256"][i]), "resized bytes differ from Pillow's"
    assert np.array_equal(u8, R.pillow_resize_bicubic(a))
    assert float(np.abs(got - want).max()) <= 5e-7, float(np.abs(got - want).max())


@pytest.mark.parametrize("shape", [(1, 1), (2, 1000), (1000, 2), (224, 223), (223, 224), (4000, 3000), (31, 4096)])
def test_preprocess_edge_shapes(gpu_lib, shape):
    import refimage as R
    h, w = shape
    a = np.random.default_rng(h * 7919 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = preprocess(gpu_lib, None, a)
    assert np.array_equal(to_u8(got), R.pillow_resize_bicubic(a))
    assert float(np.abs(got - R.normalize_chw(R.pillow_resize_bicubic(a))).max()) <= 5e-7


def test_file_to_embedding_pipeline(gpu_lib, tiny_files):
    """minigpt4_image_load_from_file -> minigpt4_preprocess_image -> minigpt4_encode_image, the call sequence of the reference's
    test_native_image_implementation path (minigpt4_library.py:722-724) and of examples/main.cpp:214-224, against the oracle on the same pixels."""
    import refcpu as RC
    import refimage as R
    from minigpt4_cpp_amd import modelgen as G
    vp, llm = tiny_files
    ctx = gpu_lib.minigpt4_model_load(vp, llm("q4_0"), verbosity=0, n_ctx=128, n_batch=32)
    try:
        image = gpu_lib.minigpt4_image_load_from_file(ctx, os.path.join(IMG_DIR, "png_llama_small.png"))
        assert (image.width, image.height, image.channels) == (187, 140, 3)
        pre = gpu_lib.minigpt4_preprocess_image(ctx, image)
        x = np.ctypeslib.as_array(ctypes.cast(pre.data, ctypes.POINTER(ctypes.c_float)), shape=(3, 224, 224)).copy()
        assert np.array_equal(to_u8(x), GOLD["llama_224"])                       # Pillow's own resize of the same file
        emb = gpu_lib.minigpt4_encode_image(ctx, pre)
        got = np.ctypeslib.as_array(emb.data, shape=(emb.n_embeddings,)).copy().reshape(32, -1)
        want = RC.OracleVision(G.read_vision_file(vp)).encode(R.preprocess(GOLD["decoded/png_llama_small.png"]))
        assert float(np.abs(got - want).max() / np.abs(want).max()) < 3e-3
        gpu_lib.minigpt4_free_embedding(emb)
        gpu_lib.minigpt4_free_image(pre)
        gpu_lib.minigpt4_free_image(image)
    finally:
        gpu_lib.minigpt4_free(ctx)
