"""Host-side mirror of the reference's Python binding for the MI355X build of libminigpt4.so.

Same class and method names, argument meaning and error behaviour as the reference's
`minigpt4/minigpt4_library.py` (MiniGPT4SharedLibrary :74-523, load_library :525-566, MiniGPT4ChatBot :568-689),
so code and tests written against the reference binding read the same here.  The reference's *unmodified* binding
also works against this library; this mirror exists so the repo is self-contained, adds correct
ctypes prototypes (`size_t` / `bool` instead of int32) and exposes the additive `minigpt4_amd_*` entry points.

There is no CPU fallback: loading a model without a usable gfx950 device raises.
"""
from __future__ import annotations

import ctypes
import enum
import os
from typing import Iterator, List, Optional, Sequence

import numpy as np


class DataType(enum.IntEnum):
    F16 = 0; F32 = 1; I32 = 2; L64 = 3; Q4_0 = 4; Q4_1 = 5; Q5_0 = 6; Q5_1 = 7; Q8_0 = 8; Q8_1 = 9   # noqa: E702
    Q2_K = 10; Q3_K = 11; Q4_K = 12; Q5_K = 13; Q6_K = 14; Q8_K = 15                                   # noqa: E702

    def __str__(self):
        return str(self.name)


class Verbosity(enum.IntEnum):
    SILENT = 0; ERR = 1; INFO = 2; DEBUG = 3   # noqa: E702


class ImageFormat(enum.IntEnum):
    UNKNOWN = 0; F32 = 1; U8 = 2   # noqa: E702


I32, F32, SIZE_T, VOID_PTR = ctypes.c_int32, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p
CHAR_PTR = ctypes.c_char_p
FLOAT_PTR = ctypes.POINTER(ctypes.c_float)
INT_PTR = ctypes.POINTER(ctypes.c_int32)


class MiniGPT4Context:
    def __init__(self, ptr):
        self.ptr = ptr


class MiniGPT4Image(ctypes.Structure):
    _fields_ = [("data", VOID_PTR), ("width", I32), ("height", I32), ("channels", I32), ("format", I32)]


class MiniGPT4Embedding(ctypes.Structure):
    _fields_ = [("data", FLOAT_PTR), ("n_embeddings", SIZE_T)]   # 2nd field is `elements` in the C header


class MiniGPT4Images(ctypes.Structure):
    _fields_ = [("images", ctypes.POINTER(MiniGPT4Image)), ("n_images", SIZE_T)]


class MiniGPT4Embeddings(ctypes.Structure):
    _fields_ = [("embeddings", ctypes.POINTER(MiniGPT4Embedding)), ("n_embeddings", SIZE_T)]


_CHAT_ARGS = [SIZE_T, F32, I32, F32, F32, F32, I32, F32, F32, F32, I32, F32, F32, I32]

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _test_hook_names() -> frozenset:
    """The names include/minigpt4_amd_test.h declares: the ONLY symbols that may be served by libminigpt4_test.so."""
    import re
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "minigpt4_amd_test.h")
    try:
        with open(hdr) as f:
            return frozenset(re.findall(r"\b(minigpt4_amd_[a-z0-9_]+)\s*\(", f.read()))
    except OSError:
        return frozenset()


class _Symbols:
    """Attribute access to the product library's symbols.  A name that include/minigpt4_amd_test.h declares (kernel-level test hooks, micro-benchmarks, probes, host-only test
    helpers) is served by libminigpt4_test.so, which is loaded -- and has its prototypes declared -- on first use; only tests/ and tools/ ever get there.  Any other
    name the product does not export is an AttributeError (a typo must not load a second copy of the engine).  The two libraries are separate copies of the engine's
    state (tuning knobs, last error): contexts are never passed from one to the other except by `minigpt4_amd_copy_arenas`, and both must report the same build."""

    def __init__(self, product, test_path: str, declare_test):
        self.__dict__.update(_product=product, _test_path=test_path, _declare_test=declare_test, _test=None, _hook_names=_test_hook_names())

    def _hooks(self):
        if self._test is None:
            if hasattr(self._product, "minigpt4_amd_test_mul_mat"):      # MINIGPT4_LIBRARY names a test-hook build itself: ONE copy of the engine serves everything
                self.__dict__["_test"] = self._product                     # (what the in-kernel timeline tools need: the stamps live in the library that ran the kernel)
                self._declare_test(self._product)
                return self._test
            if not os.path.exists(self._test_path):
                raise AttributeError(f"{self._test_path} not found (the test-hook build of the library: `make -C minigpt4.cpp_amd/csrc`)")
            test = ctypes.cdll.LoadLibrary(self._test_path)
            for lib_ in (test, self._product):
                lib_.minigpt4_amd_build_info.restype = CHAR_PTR
            a, b = self._product.minigpt4_amd_build_info(), test.minigpt4_amd_build_info()
            if a != b:
                raise RuntimeError(f"libminigpt4.so and {os.path.basename(self._test_path)} come from different builds ({a!r} vs {b!r}): rebuild both (`make -C minigpt4.cpp_amd/csrc`)")
            self.__dict__["_test"] = test
            self._declare_test(test)
        return self._test

    @property
    def test_hooks(self):
        """libminigpt4_test.so itself (explicit handle)."""
        return self._hooks()

    def _last_error(self):
        """thread-local error text of the product library and, once it is loaded, of the test-hook library (each has its own copy of the engine's state)."""
        a = self._product.minigpt4_amd_last_error() or b""
        b = (self._test.minigpt4_amd_last_error() or b"") if self._test is not None else b""
        return a + (b" | " if a and b else b"") + b

    def __getattr__(self, name):
        if name == "minigpt4_amd_last_error":
            return self._last_error
        if name in self._hook_names:
            return getattr(self._hooks(), name)
        return getattr(self._product, name)


class MiniGPT4SharedLibrary:
    """ctypes wrapper around libminigpt4.so (reference class of the same name)."""

    def __init__(self, shared_library_path: str, test_library_path: Optional[str] = None):
        product = ctypes.cdll.LoadLibrary(shared_library_path)
        if test_library_path is None:
            test_library_path = os.environ.get("MINIGPT4_TEST_LIBRARY", shared_library_path[:-3] + "_test.so" if shared_library_path.endswith(".so") else shared_library_path + "_test")
        self.library = _Symbols(product, test_library_path, self._declare_test_hooks)
        L = product
        P = ctypes.POINTER
        L.minigpt4_model_load.argtypes = [CHAR_PTR, CHAR_PTR, I32, I32, I32, I32, ctypes.c_bool]
        L.minigpt4_model_load.restype = VOID_PTR
        L.minigpt4_image_load_from_file.argtypes = [VOID_PTR, CHAR_PTR, P(MiniGPT4Image), I32]
        L.minigpt4_preprocess_image.argtypes = [VOID_PTR, P(MiniGPT4Image), P(MiniGPT4Image), I32]
        L.minigpt4_encode_image.argtypes = [VOID_PTR, P(MiniGPT4Image), P(MiniGPT4Embedding), SIZE_T]
        L.minigpt4_begin_chat_image.argtypes = [VOID_PTR, P(MiniGPT4Embedding), CHAR_PTR, SIZE_T]
        L.minigpt4_end_chat_image.argtypes = [VOID_PTR, P(ctypes.c_char_p)] + _CHAT_ARGS
        L.minigpt4_system_prompt.argtypes = [VOID_PTR, SIZE_T]
        L.minigpt4_begin_chat.argtypes = [VOID_PTR, CHAR_PTR, SIZE_T]
        L.minigpt4_end_chat.argtypes = [VOID_PTR, P(ctypes.c_char_p)] + _CHAT_ARGS
        L.minigpt4_reset_chat.argtypes = [VOID_PTR]
        L.minigpt4_contains_eos_token.argtypes = [CHAR_PTR]
        L.minigpt4_is_eos.argtypes = [CHAR_PTR]
        L.minigpt4_free.argtypes = [VOID_PTR]
        L.minigpt4_free_image.argtypes = [P(MiniGPT4Image)]
        L.minigpt4_free_embedding.argtypes = [P(MiniGPT4Embedding)]
        L.minigpt4_error_code_to_string.argtypes = [I32]
        L.minigpt4_error_code_to_string.restype = CHAR_PTR
        L.minigpt4_quantize_model.argtypes = [CHAR_PTR, CHAR_PTR, I32]
        L.minigpt4_set_verbosity.argtypes = [I32]
        L.minigpt4_set_verbosity.restype = None
        for name in ("image_load_from_file", "preprocess_image", "encode_image", "begin_chat_image", "end_chat_image", "system_prompt",
                     "begin_chat", "end_chat", "reset_chat", "contains_eos_token", "is_eos", "free", "free_image", "free_embedding", "quantize_model"):
            getattr(L, "minigpt4_" + name).restype = I32
        # additive API (include/minigpt4_amd.h)
        L.minigpt4_amd_device_count.restype = I32
        L.minigpt4_amd_last_error.restype = CHAR_PTR
        L.minigpt4_amd_build_info.restype = CHAR_PTR
        L.minigpt4_amd_decode_image.argtypes = [CHAR_PTR, SIZE_T, P(MiniGPT4Image)]
        L.minigpt4_amd_decode_image.restype = I32
        for name in ("n_vocab", "n_embd", "n_past", "sync"):
            getattr(L, "minigpt4_amd_" + name).argtypes = [VOID_PTR]
            getattr(L, "minigpt4_amd_" + name).restype = I32
        L.minigpt4_amd_eval_tokens.argtypes = [VOID_PTR, INT_PTR, I32]
        L.minigpt4_amd_eval_embd.argtypes = [VOID_PTR, FLOAT_PTR, I32]
        L.minigpt4_amd_get_logits.argtypes = [VOID_PTR, FLOAT_PTR, SIZE_T]
        L.minigpt4_amd_tokenize.argtypes = [VOID_PTR, CHAR_PTR, I32, INT_PTR, I32]
        L.minigpt4_amd_sample.argtypes = [VOID_PTR, INT_PTR, F32, I32, F32, F32, F32, I32, F32, F32]
        L.minigpt4_amd_decode_loop.argtypes = [VOID_PTR, I32, INT_PTR, FLOAT_PTR]
        L.minigpt4_amd_profile_sites.argtypes = [VOID_PTR, I32, ctypes.c_char_p, SIZE_T]
        L.minigpt4_amd_weight_bytes_per_token.argtypes = [VOID_PTR]
        L.minigpt4_amd_weight_bytes_per_token.restype = ctypes.c_double
        L.minigpt4_amd_last_encode_ms.argtypes = [VOID_PTR]
        L.minigpt4_amd_last_encode_ms.restype = F32
        for name in ("minigpt4_amd_encode_images", "minigpt4_encode_images"):        # the second = deprecated alias (include/minigpt4_amd.h)
            getattr(L, name).argtypes = [VOID_PTR, P(MiniGPT4Images), P(MiniGPT4Embeddings), SIZE_T]
        for name in ("minigpt4_amd_free_embeddings", "minigpt4_free_embeddings"):
            getattr(L, name).argtypes = [P(MiniGPT4Embeddings)]
        L.minigpt4_amd_weight_arena.argtypes = [VOID_PTR, I32, P(VOID_PTR), P(SIZE_T)]
        U64P, SZP = P(ctypes.c_uint64), P(ctypes.c_size_t)
        L.minigpt4_amd_plan_arenas.argtypes = [CHAR_PTR, CHAR_PTR, SZP, SZP, U64P, U64P]
        L.minigpt4_amd_arena_plan.argtypes = [VOID_PTR, SZP, SZP, U64P, U64P]
        L.minigpt4_amd_load_mode.argtypes = [VOID_PTR]
        L.minigpt4_amd_weights_received.argtypes = [VOID_PTR]
        L.minigpt4_amd_arena_checksum.argtypes = [VOID_PTR, I32, U64P]
        L.minigpt4_amd_set_parity.argtypes = [VOID_PTR, I32]
        L.minigpt4_amd_dist_info.argtypes = [VOID_PTR, P(I32), P(I32), P(F32)]
        L.minigpt4_amd_dist_info.restype = I32
        L.minigpt4_amd_parity.argtypes = [VOID_PTR]
        L.minigpt4_amd_set_conversations.argtypes = [VOID_PTR, I32]
        L.minigpt4_amd_select_conversation.argtypes = [VOID_PTR, I32]
        L.minigpt4_amd_n_conversations.argtypes = [VOID_PTR]
        L.minigpt4_amd_end_chat_batch.argtypes = [VOID_PTR, INT_PTR, I32, P(ctypes.c_char_p), F32, I32, F32, F32, F32, I32, F32, F32]
        L.minigpt4_amd_eval_batch.argtypes = [VOID_PTR, INT_PTR, I32, INT_PTR, INT_PTR]
        L.minigpt4_amd_batch_path.argtypes = [VOID_PTR, INT_PTR]

    @staticmethod
    def _declare_test_hooks(L):
        """Prototypes of libminigpt4_test.so's extra symbols (include/minigpt4_amd_test.h)."""
        P = ctypes.POINTER
        U64P = P(ctypes.c_uint64)
        L.minigpt4_amd_convert_q3k_q6k.argtypes = [VOID_PTR, VOID_PTR, ctypes.c_int64]
        L.minigpt4_amd_test_mul_mat.argtypes = [I32, VOID_PTR, ctypes.c_int64, ctypes.c_int64, FLOAT_PTR, ctypes.c_int64, FLOAT_PTR]
        L.minigpt4_amd_test_mul_mat_ref.argtypes = L.minigpt4_amd_test_mul_mat.argtypes
        L.minigpt4_amd_test_mmq2.argtypes = [I32, VOID_PTR, I32, ctypes.c_int64, ctypes.c_int64, FLOAT_PTR, ctypes.c_int64, FLOAT_PTR, I32, I32, FLOAT_PTR]
        L.minigpt4_amd_test_matvec.argtypes = [I32, VOID_PTR, I32, I32, VOID_PTR, I32, ctypes.c_int64, ctypes.c_int64, FLOAT_PTR, FLOAT_PTR, I32, I32, I32, FLOAT_PTR, FLOAT_PTR]
        L.minigpt4_amd_test_matvec_rows.argtypes = [I32, VOID_PTR, I32, ctypes.c_int64, ctypes.c_int64, FLOAT_PTR, I32, FLOAT_PTR, FLOAT_PTR]
        L.minigpt4_amd_test_quantize.argtypes = [FLOAT_PTR, FLOAT_PTR, ctypes.c_int64, ctypes.c_int64, VOID_PTR, VOID_PTR, VOID_PTR, VOID_PTR, VOID_PTR]
        L.minigpt4_amd_test_gemm_f16.argtypes = [FLOAT_PTR, FLOAT_PTR, FLOAT_PTR, I32, I32, I32, I32, FLOAT_PTR]
        L.minigpt4_amd_test_gemm_f16_skinny.argtypes = L.minigpt4_amd_test_gemm_f16.argtypes
        L.minigpt4_amd_last_error.restype = CHAR_PTR
        L.minigpt4_amd_vocab_load.argtypes = [CHAR_PTR]
        L.minigpt4_amd_vocab_load.restype = VOID_PTR
        L.minigpt4_amd_vocab_free.argtypes = [VOID_PTR]
        L.minigpt4_amd_vocab_free.restype = None
        L.minigpt4_amd_vocab_size.argtypes = [VOID_PTR]
        L.minigpt4_amd_vocab_piece.argtypes = [VOID_PTR, I32, INT_PTR]
        L.minigpt4_amd_vocab_piece.restype = VOID_PTR
        L.minigpt4_amd_vocab_tokenize.argtypes = [VOID_PTR, CHAR_PTR, I32, INT_PTR, I32]
        L.minigpt4_amd_inspect_files.argtypes = [CHAR_PTR, CHAR_PTR, INT_PTR, INT_PTR, P(ctypes.c_int64)]
        L.minigpt4_amd_sample_logits.argtypes = [FLOAT_PTR, I32, I32, F32, I32, F32, F32, F32, I32, F32, F32]
        L.minigpt4_amd_resample_coeffs.argtypes = [I32, I32, INT_PTR, INT_PTR, INT_PTR, INT_PTR, SIZE_T]
        L.minigpt4_amd_copy_arenas.argtypes = [VOID_PTR, VOID_PTR]
        L.minigpt4_amd_llm_file_digest.argtypes = [CHAR_PTR, U64P, I32]
        L.minigpt4_amd_quantize_chunk.argtypes = [I32, FLOAT_PTR, VOID_PTR, ctypes.c_int64]
        L.minigpt4_amd_quantize_chunk.restype = ctypes.c_int64
        L.minigpt4_amd_probe_valu.restype = F32
        L.minigpt4_amd_probe_grid_barrier.restype = F32

    # ---------------------------------------------------------------- reference surface
    def panic_if_error(self, error_code: int) -> None:
        if error_code != 0:
            raise RuntimeError(self.library.minigpt4_error_code_to_string(I32(error_code)))

    def minigpt4_model_load(self, model_path: str, llm_model_path: str, verbosity: int = 1, seed: int = 1337, n_ctx: int = 2048,
                            n_batch: int = 512, numa: int = 0) -> MiniGPT4Context:
        ptr = self.library.minigpt4_model_load(model_path.encode(), llm_model_path.encode(), int(verbosity), seed, n_ctx, n_batch, bool(numa))
        if not ptr:
            raise RuntimeError("minigpt4_model_load failed: " + (self.library.minigpt4_amd_last_error() or b"").decode(errors="replace"))
        return MiniGPT4Context(ptr)

    def minigpt4_image_load_from_file(self, ctx: MiniGPT4Context, path: str, flags: int = 0) -> MiniGPT4Image:
        image = MiniGPT4Image()
        self.panic_if_error(self.library.minigpt4_image_load_from_file(ctx.ptr, path.encode(), ctypes.pointer(image), flags))
        return image

    def minigpt4_preprocess_image(self, ctx: MiniGPT4Context, image: MiniGPT4Image, flags: int = 0) -> MiniGPT4Image:
        out = MiniGPT4Image()
        self.panic_if_error(self.library.minigpt4_preprocess_image(ctx.ptr, ctypes.pointer(image), ctypes.pointer(out), flags))
        return out

    def minigpt4_encode_image(self, ctx: MiniGPT4Context, image: MiniGPT4Image, n_threads: int = 0) -> MiniGPT4Embedding:
        embedding = MiniGPT4Embedding()
        self.panic_if_error(self.library.minigpt4_encode_image(ctx.ptr, ctypes.pointer(image), ctypes.pointer(embedding), n_threads))
        return embedding

    def minigpt4_begin_chat_image(self, ctx: MiniGPT4Context, image_embedding: MiniGPT4Embedding, s: str, n_threads: int = 0):
        self.panic_if_error(self.library.minigpt4_begin_chat_image(ctx.ptr, ctypes.pointer(image_embedding), s.encode(), n_threads))

    def _end(self, fn, ctx, n_threads, temp, top_k, top_p, tfs_z, typical_p, repeat_last_n, repeat_penalty, alpha_presence, alpha_frequency,
             mirostat, mirostat_tau, mirostat_eta, penalize_nl) -> str:
        token = ctypes.c_char_p()
        self.panic_if_error(fn(ctx.ptr, ctypes.byref(token), n_threads, temp, top_k, top_p, tfs_z, typical_p, repeat_last_n, repeat_penalty,
                               alpha_presence, alpha_frequency, mirostat, mirostat_tau, mirostat_eta, penalize_nl))
        return (token.value or b"").decode("utf-8", errors="replace")

    def minigpt4_end_chat_image(self, ctx, n_threads=0, temp=0.8, top_k=40, top_p=0.9, tfs_z=1.0, typical_p=1.0, repeat_last_n=64,
                                repeat_penalty=1.1, alpha_presence=1.0, alpha_frequency=1.0, mirostat=0, mirostat_tau=5.0, mirostat_eta=1.0,
                                penalize_nl=1) -> str:
        return self._end(self.library.minigpt4_end_chat_image, ctx, n_threads, temp, top_k, top_p, tfs_z, typical_p, repeat_last_n, repeat_penalty,
                         alpha_presence, alpha_frequency, mirostat, mirostat_tau, mirostat_eta, penalize_nl)

    def minigpt4_system_prompt(self, ctx: MiniGPT4Context, n_threads: int = 0):
        self.panic_if_error(self.library.minigpt4_system_prompt(ctx.ptr, n_threads))

    def minigpt4_begin_chat(self, ctx: MiniGPT4Context, s: str, n_threads: int = 0):
        self.panic_if_error(self.library.minigpt4_begin_chat(ctx.ptr, s.encode(), n_threads))

    def minigpt4_end_chat(self, ctx, n_threads=0, temp=0.8, top_k=40, top_p=0.9, tfs_z=1.0, typical_p=1.0, repeat_last_n=64, repeat_penalty=1.1,
                          alpha_presence=1.0, alpha_frequency=1.0, mirostat=0, mirostat_tau=5.0, mirostat_eta=1.0, penalize_nl=1) -> str:
        return self._end(self.library.minigpt4_end_chat, ctx, n_threads, temp, top_k, top_p, tfs_z, typical_p, repeat_last_n, repeat_penalty,
                         alpha_presence, alpha_frequency, mirostat, mirostat_tau, mirostat_eta, penalize_nl)

    def minigpt4_reset_chat(self, ctx: MiniGPT4Context):
        self.panic_if_error(self.library.minigpt4_reset_chat(ctx.ptr))

    def minigpt4_contains_eos_token(self, s: str) -> bool:
        return bool(self.library.minigpt4_contains_eos_token(s.encode()))

    def minigpt4_is_eos(self, s: str) -> bool:
        return bool(self.library.minigpt4_is_eos(s.encode()))

    def minigpt4_free(self, ctx: MiniGPT4Context) -> None:
        self.panic_if_error(self.library.minigpt4_free(ctx.ptr))
        ctx.ptr = None

    def minigpt4_free_image(self, image: MiniGPT4Image) -> None:
        self.panic_if_error(self.library.minigpt4_free_image(ctypes.pointer(image)))

    def minigpt4_free_embedding(self, embedding: MiniGPT4Embedding) -> None:
        self.panic_if_error(self.library.minigpt4_free_embedding(ctypes.pointer(embedding)))

    def minigpt4_error_code_to_string(self, error_code: int) -> str:
        return self.library.minigpt4_error_code_to_string(error_code).decode()

    def minigpt4_quantize_model(self, in_path: str, out_path: str, data_type: DataType):
        self.panic_if_error(self.library.minigpt4_quantize_model(in_path.encode(), out_path.encode(), int(data_type)))

    def minigpt4_set_verbosity(self, verbosity: Verbosity):
        self.library.minigpt4_set_verbosity(int(verbosity))

    # ---------------------------------------------------------------- additive surface (numpy in / out)
    def amd_device_count(self) -> int:
        return int(self.library.minigpt4_amd_device_count())

    # multi-GPU load (include/minigpt4_amd.h)
    def amd_plan_arenas(self, vision_path: str, llm_path: str) -> dict:
        """Arena layout the two files produce, computed on the host (no GPU): {"llm_bytes", "vision_bytes", "llm_hash", "vision_hash"}."""
        lb, vb, lh, vh = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_uint64(), ctypes.c_uint64()
        rc = self.library.minigpt4_amd_plan_arenas(vision_path.encode(), llm_path.encode(), ctypes.byref(lb), ctypes.byref(vb), ctypes.byref(lh), ctypes.byref(vh))
        if rc:
            raise RuntimeError(f"plan_arenas failed ({rc}): " + self.library.minigpt4_amd_last_error().decode())
        return {"llm_bytes": lb.value, "vision_bytes": vb.value, "llm_hash": lh.value, "vision_hash": vh.value}

    def amd_arena_plan(self, ctx) -> dict:
        lb, vb, lh, vh = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_uint64(), ctypes.c_uint64()
        assert self.library.minigpt4_amd_arena_plan(ctx.ptr, ctypes.byref(lb), ctypes.byref(vb), ctypes.byref(lh), ctypes.byref(vh)) == 0
        return {"llm_bytes": lb.value, "vision_bytes": vb.value, "llm_hash": lh.value, "vision_hash": vh.value}

    def amd_arena_checksum(self, ctx, which: int) -> int:
        v = ctypes.c_uint64()
        assert self.library.minigpt4_amd_arena_checksum(ctx.ptr, which, ctypes.byref(v)) == 0
        return int(v.value)

    # several conversations per context (include/minigpt4_amd.h): the reference calls act on the selected one
    def amd_dist_info(self, ctx) -> dict:
        """what the native (in-library RCCL) weight broadcast of this context's load did: world size, rank, milliseconds (0.0 for an ordinary load)"""
        w, r, ms = I32(), I32(), F32()
        assert self.library.minigpt4_amd_dist_info(ctx.ptr, ctypes.byref(w), ctypes.byref(r), ctypes.byref(ms)) == 0
        return {"world": w.value, "rank": r.value, "bcast_ms": ms.value}

    def amd_set_parity(self, ctx, on: bool):
        """Parity mode (MINIGPT4_PARITY): the language path adds its fp32 terms in the CPU oracle's order -- bit-identical logits, slow."""
        if self.library.minigpt4_amd_set_parity(ctx.ptr, 1 if on else 0):
            raise RuntimeError("minigpt4_amd_set_parity failed")

    def amd_set_conversations(self, ctx, n: int):
        if self.library.minigpt4_amd_set_conversations(ctx.ptr, n):
            raise RuntimeError("set_conversations failed: " + self.library.minigpt4_amd_last_error().decode())

    def amd_select_conversation(self, ctx, slot: int):
        if self.library.minigpt4_amd_select_conversation(ctx.ptr, slot):
            raise RuntimeError("select_conversation: index out of range")

    def amd_end_chat_batch(self, ctx, slots: Sequence[int], temp=0.8, top_k=40, top_p=0.9, tfs_z=1.0, typical_p=1.0, mirostat=0, mirostat_tau=5.0,
                           mirostat_eta=1.0) -> List[str]:
        """One minigpt4_end_chat step for several conversations in one pass over the weights; returns one piece per conversation."""
        sl = np.ascontiguousarray(slots, np.int32)
        toks = (ctypes.c_char_p * len(sl))()
        rc = self.library.minigpt4_amd_end_chat_batch(ctx.ptr, sl.ctypes.data_as(INT_PTR), len(sl), toks, temp, top_k, top_p, tfs_z, typical_p, mirostat, mirostat_tau, mirostat_eta)
        if rc:
            raise RuntimeError("end_chat_batch failed: " + self.library.minigpt4_amd_last_error().decode())
        return [(t or b"").decode("utf-8", errors="replace") for t in toks]

    def amd_eval_batch(self, ctx, slots: Sequence[int], tokens: Sequence[int]) -> List[int]:
        """One batched decode step with GIVEN next tokens (teacher forcing); returns every conversation's own greedy choice."""
        n = len(slots)
        assert len(tokens) == n
        sl, tk, out = (ctypes.c_int32 * n)(*slots), (ctypes.c_int32 * n)(*[int(t) for t in tokens]), (ctypes.c_int32 * n)()
        if self.library.minigpt4_amd_eval_batch(ctx.ptr, sl, n, tk, out):
            raise RuntimeError("minigpt4_amd_eval_batch failed: " + self.library.minigpt4_amd_last_error().decode("utf-8", errors="replace"))
        return [int(x) for x in out]

    def amd_batch_path(self, ctx) -> dict:
        """Launch kinds of the batched step as last built (include/minigpt4_amd.h: minigpt4_amd_batch_path)."""
        out = (ctypes.c_int32 * 8)()
        if self.library.minigpt4_amd_batch_path(ctx.ptr, out):
            raise RuntimeError("minigpt4_amd_batch_path failed")
        return dict(zip(("rows", "ri", "ri_mix", "ri_ksplit", "dot4", "dot4_mix", "mul_mat", "sets"), [int(x) for x in out]))

    def amd_encode_images(self, ctx, images: Sequence[np.ndarray]) -> List[np.ndarray]:
        """minigpt4_amd_encode_images: the images (f32 CHW [3,224,224] arrays) in passes of up to 8 over the vision weights; one [32, n_embd] array per image."""
        arrs = [np.ascontiguousarray(i, dtype=np.float32) for i in images]
        structs = (MiniGPT4Image * len(arrs))(*[array_to_image_struct(a) for a in arrs])
        batch, out = MiniGPT4Images(structs, len(arrs)), MiniGPT4Embeddings()
        self.panic_if_error(self.library.minigpt4_amd_encode_images(ctx.ptr, ctypes.byref(batch), ctypes.byref(out), 0))
        try:
            return [np.ctypeslib.as_array(out.embeddings[i].data, shape=(out.embeddings[i].n_embeddings,)).copy().reshape(32, -1) for i in range(out.n_embeddings)]
        finally:
            self.library.minigpt4_amd_free_embeddings(ctypes.byref(out))

    def amd_decode_image(self, data: bytes) -> MiniGPT4Image:
        image = MiniGPT4Image()
        self.panic_if_error(self.library.minigpt4_amd_decode_image(data, len(data), ctypes.pointer(image)))
        return image

    def amd_eval_tokens(self, ctx, tokens: Sequence[int]):
        t = np.ascontiguousarray(tokens, np.int32)
        self.panic_if_error(self.library.minigpt4_amd_eval_tokens(ctx.ptr, t.ctypes.data_as(INT_PTR), len(t)))

    def amd_eval_embd(self, ctx, embd: np.ndarray):
        e = np.ascontiguousarray(embd, np.float32)
        n_embd = self.library.minigpt4_amd_n_embd(ctx.ptr)
        self.panic_if_error(self.library.minigpt4_amd_eval_embd(ctx.ptr, e.ctypes.data_as(FLOAT_PTR), e.size // n_embd))

    def amd_logits(self, ctx) -> np.ndarray:
        out = np.empty(self.library.minigpt4_amd_n_vocab(ctx.ptr), np.float32)
        assert self.library.minigpt4_amd_get_logits(ctx.ptr, out.ctypes.data_as(FLOAT_PTR), out.size) == 0
        return out

    def amd_tokenize(self, ctx, text: bytes, add_bos: bool = True) -> List[int]:
        cap = len(text) + 8
        out = (ctypes.c_int32 * cap)()
        n = self.library.minigpt4_amd_tokenize(ctx.ptr, text, int(add_bos), out, cap)
        return list(out[:n])

    def amd_decode_loop(self, ctx, steps: int):
        toks = np.zeros(steps, np.int32)
        ms = ctypes.c_float()
        rc = self.library.minigpt4_amd_decode_loop(ctx.ptr, steps, toks.ctypes.data_as(INT_PTR), ctypes.byref(ms))
        if rc:
            raise RuntimeError("decode_loop failed: " + self.library.minigpt4_amd_last_error().decode())
        return toks, float(ms.value)

    def amd_profile_sites(self, ctx, steps: int) -> dict:
        """Per-launch-site table of `steps` eager decode steps (the launch set of the captured graph): {"steps", "eager_ms_per_step", "sites": [{site, kernel, calls_per_step, avg_us, bytes_per_call}]}"""
        import json as _json
        buf = ctypes.create_string_buffer(1 << 16)
        rc = self.library.minigpt4_amd_profile_sites(ctx.ptr, steps, buf, len(buf))
        if rc:
            raise RuntimeError("profile_sites failed")
        return _json.loads(buf.value.decode())

    def amd_test_mul_mat(self, ggml_type: int, raw_w: np.ndarray, n_in: int, n_out: int, x: np.ndarray, ref: bool = False) -> np.ndarray:
        """ref: the parity-mode kernel (oracle accumulation order) instead of the engine's fast dispatch."""
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n_in)
        raw_w = np.ascontiguousarray(raw_w)
        y = np.empty((x.shape[0], n_out), np.float32)
        fn = self.library.minigpt4_amd_test_mul_mat_ref if ref else self.library.minigpt4_amd_test_mul_mat
        rc = fn(ggml_type, raw_w.ctypes.data_as(VOID_PTR), n_in, n_out, x.ctypes.data_as(FLOAT_PTR), x.shape[0],
                                                    y.ctypes.data_as(FLOAT_PTR))
        if rc:
            raise RuntimeError(f"test_mul_mat rc={rc}: " + self.library.minigpt4_amd_last_error().decode())
        return y

    def amd_test_mmq2(self, ggml_type: int, raw_w: np.ndarray, n_mat: int, n_in: int, n_out: int, x: np.ndarray, residual: Optional[np.ndarray] = None, ks: int = 0,
                      generation: int = 2) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n_in)
        raw_w = np.ascontiguousarray(raw_w)
        N = x.shape[0]
        y = np.empty((n_mat, N, n_out), np.float32)
        r = np.ascontiguousarray(residual, np.float32) if residual is not None else None
        rc = self.library.minigpt4_amd_test_mmq2(ggml_type, raw_w.ctypes.data_as(VOID_PTR), n_mat, n_in, n_out, x.ctypes.data_as(FLOAT_PTR), N,
                                                 r.ctypes.data_as(FLOAT_PTR) if r is not None else None, ks, generation, y.ctypes.data_as(FLOAT_PTR))
        if rc:
            raise RuntimeError(f"amd_test_mmq2 failed ({rc}): " + self.library.minigpt4_amd_last_error().decode())
        return y

    def amd_test_matvec(self, type1: int, raw1: np.ndarray, n1: int, n_in: int, n_out: int, x: np.ndarray, x2: Optional[np.ndarray] = None, prep: int = 2,
                        fuse: bool = False, epi: int = 0, residual: Optional[np.ndarray] = None, type2: int = 0, raw2: Optional[np.ndarray] = None) -> np.ndarray:
        """Decode mat-vec launches exactly as the engine issues them (see include/minigpt4_amd.h)."""
        x = np.ascontiguousarray(x, np.float32).reshape(n_in)
        x2c = None if x2 is None else np.ascontiguousarray(x2, np.float32).reshape(n_in)
        raw1 = np.ascontiguousarray(raw1)
        n2 = 0 if raw2 is None else 1
        raw2c = None if raw2 is None else np.ascontiguousarray(raw2)
        res = None if residual is None else np.ascontiguousarray(residual, np.float32).reshape(-1)
        y = np.empty(((1 if epi == 1 else n1 + n2), n_out), np.float32)      # epi 1: the SiLU pair epilogue writes one row; 2: the CPU oracle's fp32 order (k-quants)
        rc = self.library.minigpt4_amd_test_matvec(type1, raw1.ctypes.data_as(VOID_PTR), n1, type2, None if raw2c is None else raw2c.ctypes.data_as(VOID_PTR), n2,
                                                   n_in, n_out, x.ctypes.data_as(FLOAT_PTR), None if x2c is None else x2c.ctypes.data_as(FLOAT_PTR), prep, int(fuse), epi,
                                                   None if res is None else res.ctypes.data_as(FLOAT_PTR), y.ctypes.data_as(FLOAT_PTR))
        if rc:
            raise RuntimeError(f"test_matvec rc={rc}: " + self.library.minigpt4_amd_last_error().decode())
        return y

    def amd_test_matvec_rows(self, ggml_type: int, raw_w: np.ndarray, n_mat: int, n_in: int, n_out: int, x: np.ndarray, residual: Optional[np.ndarray] = None) -> np.ndarray:
        """The batched-decode mat-vec: x [N][n_in] (N <= 4) against n_mat matrices in one weight pass -> [n_mat][N][n_out]."""
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n_in)
        raw_w = np.ascontiguousarray(raw_w)
        res = None if residual is None else np.ascontiguousarray(residual, np.float32)
        y = np.empty((n_mat, x.shape[0], n_out), np.float32)
        rc = self.library.minigpt4_amd_test_matvec_rows(ggml_type, raw_w.ctypes.data_as(VOID_PTR), n_mat, n_in, n_out, x.ctypes.data_as(FLOAT_PTR), x.shape[0],
                                                        None if res is None else res.ctypes.data_as(FLOAT_PTR), y.ctypes.data_as(FLOAT_PTR))
        if rc:
            raise RuntimeError(f"test_matvec_rows rc={rc}: " + self.library.minigpt4_amd_last_error().decode())
        return y

    def amd_test_matvec_ri(self, ggml_type: int, raw_w: np.ndarray, n_mat: int, n_in: int, n_out: int, x: np.ndarray, residual: Optional[np.ndarray] = None,
                           rms_w: Optional[np.ndarray] = None) -> np.ndarray:
        """The batched decode's MFMA launch over the row-interleaved image (csrc/ri_kernels.hip): x [N][n_in] (N <= 4) -> [n_mat][N][n_out]."""
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n_in)
        raw_w = np.ascontiguousarray(raw_w)
        res = None if residual is None else np.ascontiguousarray(residual, np.float32)
        y = np.empty((n_mat, x.shape[0], n_out), np.float32)
        f = self.library.minigpt4_amd_test_matvec_ri
        f.argtypes = [I32, VOID_PTR, I32, ctypes.c_int64, ctypes.c_int64, FLOAT_PTR, I32, FLOAT_PTR, FLOAT_PTR, FLOAT_PTR]
        w = None if rms_w is None else np.ascontiguousarray(rms_w, np.float32)
        rc = f(ggml_type, raw_w.ctypes.data_as(VOID_PTR), n_mat, n_in, n_out, x.ctypes.data_as(FLOAT_PTR), x.shape[0], None if res is None else res.ctypes.data_as(FLOAT_PTR),
               None if w is None else w.ctypes.data_as(FLOAT_PTR), y.ctypes.data_as(FLOAT_PTR))
        if rc:
            raise RuntimeError(f"test_matvec_ri rc={rc}: " + self.library.minigpt4_amd_last_error().decode())
        return y

    def amd_test_matvec_ri_mixed(self, type_a: int, raw_a: np.ndarray, n_a: int, type_b: int, raw_b: np.ndarray, n_b: int, n_in: int, n_out: int, x: np.ndarray) -> np.ndarray:
        """The mixed-type MFMA launch of a "more bits" layer (wq | wk of Q4_K / Q5_K + a Q6_K wv): x [N][n_in] (N <= 4) -> [n_a + n_b][N][n_out]."""
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n_in)
        raw_a, raw_b = np.ascontiguousarray(raw_a), np.ascontiguousarray(raw_b)
        y = np.empty((n_a + n_b, x.shape[0], n_out), np.float32)
        f = self.library.minigpt4_amd_test_matvec_ri_mixed
        f.argtypes = [I32, VOID_PTR, I32, I32, VOID_PTR, I32, ctypes.c_int64, ctypes.c_int64, FLOAT_PTR, I32, FLOAT_PTR]
        rc = f(type_a, raw_a.ctypes.data_as(VOID_PTR), n_a, type_b, raw_b.ctypes.data_as(VOID_PTR), n_b, n_in, n_out, x.ctypes.data_as(FLOAT_PTR), x.shape[0], y.ctypes.data_as(FLOAT_PTR))
        if rc:
            raise RuntimeError(f"test_matvec_ri_mixed rc={rc}: " + self.library.minigpt4_amd_last_error().decode())
        return y

    def amd_test_quantize(self, x: np.ndarray, rms_w: Optional[np.ndarray] = None):
        x = np.ascontiguousarray(x, np.float32)
        N, K = x.shape
        q8k, dk, bs = np.empty((N, K), np.int8), np.empty((N, K // 256), np.float32), np.empty((N, K // 16), np.int16)
        q80, d0 = np.empty((N, K), np.int8), np.empty((N, K // 32), np.float32)
        w = None if rms_w is None else np.ascontiguousarray(rms_w, np.float32)
        rc = self.library.minigpt4_amd_test_quantize(x.ctypes.data_as(FLOAT_PTR), None if w is None else w.ctypes.data_as(FLOAT_PTR), N, K,
                                                     q8k.ctypes.data_as(VOID_PTR), dk.ctypes.data_as(VOID_PTR), bs.ctypes.data_as(VOID_PTR),
                                                     q80.ctypes.data_as(VOID_PTR), d0.ctypes.data_as(VOID_PTR))
        if rc:
            raise RuntimeError(f"test_quantize rc={rc}")
        return q8k, dk, bs, q80, d0

    def amd_test_gemm_f16(self, A: np.ndarray, W: np.ndarray, bias: Optional[np.ndarray] = None, gelu: bool = False, skinny: bool = False) -> np.ndarray:
        A = np.ascontiguousarray(A, np.float32)
        W = np.ascontiguousarray(W, np.float32)
        M, K = A.shape
        N = W.shape[0]
        C = np.empty((M, N), np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        fn = self.library.minigpt4_amd_test_gemm_f16_skinny if skinny else self.library.minigpt4_amd_test_gemm_f16
        rc = fn(A.ctypes.data_as(FLOAT_PTR), W.ctypes.data_as(FLOAT_PTR), None if b is None else b.ctypes.data_as(FLOAT_PTR), M, N, K, int(gelu), C.ctypes.data_as(FLOAT_PTR))
        if rc:
            raise RuntimeError(f"test_gemm_f16 rc={rc}")
        return C

    def amd_sample_logits(self, logits: np.ndarray, seed: int, temp=0.8, top_k=40, top_p=0.9, tfs_z=1.0, typical_p=1.0, mirostat=0, mirostat_tau=5.0,
                          mirostat_eta=1.0) -> int:
        lg = np.ascontiguousarray(logits, np.float32)
        return int(self.library.minigpt4_amd_sample_logits(lg.ctypes.data_as(FLOAT_PTR), lg.size, seed, temp, top_k, top_p, tfs_z, typical_p, mirostat,
                                                           mirostat_tau, mirostat_eta))


def default_library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libminigpt4.so")


def load_library() -> MiniGPT4SharedLibrary:
    """Reference `load_library` (:525-566) searches a few relative paths for libminigpt4.so; here the in-tree build is used."""
    path = os.environ.get("MINIGPT4_LIBRARY", default_library_path())
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    return MiniGPT4SharedLibrary(path)


def image_to_array(image, size: int = 224) -> np.ndarray:
    """Preprocessing of the reference's ChatBot (:589-600, 682-687) without torchvision: RGB, bicubic resize to 224x224, /255,
    CLIP mean/std, HWC -> CHW float32."""
    from PIL import Image
    if not isinstance(image, Image.Image):
        image = Image.open(image)
    image = image.convert("RGB").resize((size, size), Image.BICUBIC)
    a = np.asarray(image, np.float32) / 255.0
    a = (a - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
    return np.ascontiguousarray(a.transpose(2, 0, 1))


def array_to_image_struct(chw: np.ndarray) -> MiniGPT4Image:
    chw = np.ascontiguousarray(chw, np.float32)
    assert chw.shape == (3, 224, 224)
    img = MiniGPT4Image(chw.ctypes.data_as(VOID_PTR), 224, 224, 3, int(ImageFormat.F32))
    img._keepalive = chw
    return img


class MiniGPT4ChatBot:
    """Reference class of the same name (:568-689): upload_image / generate / reset_chat."""

    def __init__(self, model_path: str, llm_model_path: str, verbosity: Verbosity = Verbosity.SILENT, n_threads: int = 0, library: Optional[MiniGPT4SharedLibrary] = None,
                 n_ctx: int = 2048, n_batch: int = 512, seed: int = 1337):
        self.library = library or load_library()
        self.ctx = self.library.minigpt4_model_load(model_path, llm_model_path, int(verbosity), seed=seed, n_ctx=n_ctx, n_batch=n_batch)
        self.n_threads = n_threads
        self.embedding: Optional[MiniGPT4Embedding] = None
        self.is_image_uploaded = False

    def free(self):
        if self.ctx is not None and self.ctx.ptr:
            self.library.minigpt4_free(self.ctx)

    def generate(self, message: str, limit: int = 1024, temp: float = 0.8, top_k: int = 40, top_p: float = 0.9, tfs_z: float = 1.0, typical_p: float = 1.0,
                 repeat_last_n: int = 64, repeat_penalty: float = 1.1, alpha_presence: float = 1.0, alpha_frequency: float = 1.0, mirostat: int = 0,
                 mirostat_tau: float = 5.0, mirostat_eta: float = 1.0, penalize_nl: int = 1, ignore_eos: bool = False) -> Iterator[str]:
        if self.is_image_uploaded:
            self.library.minigpt4_begin_chat_image(self.ctx, self.embedding, message, self.n_threads)
            self.is_image_uploaded = False
        else:
            self.library.minigpt4_begin_chat(self.ctx, message, self.n_threads)
        chat = ""
        for _ in range(limit):
            token = self.library.minigpt4_end_chat_image(self.ctx, self.n_threads, temp, top_k, top_p, tfs_z, typical_p, repeat_last_n, repeat_penalty,
                                                          alpha_presence, alpha_frequency, mirostat, mirostat_tau, mirostat_eta, penalize_nl)
            chat += token
            if not ignore_eos:
                if self.library.minigpt4_contains_eos_token(token):
                    continue
                if self.library.minigpt4_is_eos(chat):
                    break
            yield token

    def reset_chat(self):
        self.is_image_uploaded = False
        if self.embedding is not None:
            self.library.minigpt4_free_embedding(self.embedding)
            self.embedding = None
        self.library.minigpt4_reset_chat(self.ctx)
        self.library.minigpt4_system_prompt(self.ctx, self.n_threads)

    def upload_image(self, image):
        """image: a preprocessed f32 [3][224][224] array, a file path / encoded bytes (decoded and preprocessed by the library itself:
        minigpt4_image_load_from_file + minigpt4_preprocess_image, the reference's `test_native_image_implementation` path, minigpt4_library.py:722-724),
        or a PIL image (the reference's torchvision-style path, via Pillow)."""
        self.reset_chat()
        if isinstance(image, (str, bytes, bytearray)):
            raw = self.library.amd_decode_image(bytes(image)) if not isinstance(image, str) else self.library.minigpt4_image_load_from_file(self.ctx, image)
            pre = None
            try:
                pre = self.library.minigpt4_preprocess_image(self.ctx, raw)
                self.embedding = self.library.minigpt4_encode_image(self.ctx, pre, self.n_threads)
            finally:
                self.library.minigpt4_free_image(raw)
                if pre is not None:
                    self.library.minigpt4_free_image(pre)
        else:
            chw = image if isinstance(image, np.ndarray) else image_to_array(image)
            self.embedding = self.library.minigpt4_encode_image(self.ctx, array_to_image_struct(chw), self.n_threads)
        self.is_image_uploaded = True
