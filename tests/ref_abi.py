"""The EFFECTIVE ABI of the reference's Python binding, restated (TEST INFRASTRUCTURE): the ctypes argtypes / restypes `/root/reference/minigpt4/minigpt4_library.py`
declares at :94-227 -- int32 where the C prototype says size_t / bool, POINTER(c_char) strings, void* contexts.  The reference is not mounted on the GPU box, so the GPU
boundary test binds the library through THIS table; tests/test_cpu_refwrapper.py checks, where the reference is mounted, that the table equals what the unmodified
reference module really declares (so it cannot drift)."""
import ctypes

I32, F32, SIZE_T, VOID_PTR = ctypes.c_int32, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p
CHAR_PTR = ctypes.POINTER(ctypes.c_char)
FLOAT_PTR = ctypes.POINTER(ctypes.c_float)
CHAR_PTR_PTR = ctypes.POINTER(ctypes.POINTER(ctypes.c_char))


class MiniGPT4Image(ctypes.Structure):                       # reference :56-63
    _fields_ = [("data", VOID_PTR), ("width", I32), ("height", I32), ("channels", I32), ("format", I32)]


class MiniGPT4Embedding(ctypes.Structure):                   # reference :65-69
    _fields_ = [("data", FLOAT_PTR), ("n_embeddings", SIZE_T)]


ImageP, EmbeddingP = ctypes.POINTER(MiniGPT4Image), ctypes.POINTER(MiniGPT4Embedding)
_END = [VOID_PTR, CHAR_PTR_PTR, I32, F32, I32, F32, F32, F32, I32, F32, F32, F32, I32, F32, F32, I32]

# name -> (argtypes, restype); minigpt4_preprocess_image has NO declaration in the reference (:303 calls it undeclared): ctypes defaults apply
ABI = {
    "minigpt4_model_load": ([CHAR_PTR, CHAR_PTR, I32, I32, I32, I32, I32], VOID_PTR),                       # :94-103 (numa, a C bool, travels as int32)
    "minigpt4_image_load_from_file": ([VOID_PTR, CHAR_PTR, ImageP, I32], I32),                             # :105-111
    "minigpt4_encode_image": ([VOID_PTR, ImageP, EmbeddingP, I32], I32),                                   # :113-119 (size_t n_threads as int32)
    "minigpt4_begin_chat_image": ([VOID_PTR, EmbeddingP, CHAR_PTR, I32], I32),                             # :121-127
    "minigpt4_end_chat_image": (_END, I32),                                                                 # :129-147
    "minigpt4_system_prompt": ([VOID_PTR, I32], I32),                                                       # :149-153
    "minigpt4_begin_chat": ([VOID_PTR, CHAR_PTR, I32], I32),                                                # :155-160
    "minigpt4_end_chat": (_END, I32),                                                                       # :162-180
    "minigpt4_reset_chat": ([VOID_PTR], I32),                                                               # :182-185
    "minigpt4_contains_eos_token": ([CHAR_PTR], I32),                                                       # :187-190
    "minigpt4_is_eos": ([CHAR_PTR], I32),                                                                   # :192-195
    "minigpt4_free": ([VOID_PTR], I32),                                                                     # :197-200
    "minigpt4_free_image": ([ImageP], I32),                                                                 # :202-205
    "minigpt4_free_embedding": ([EmbeddingP], I32),                                                         # :207-210
    "minigpt4_error_code_to_string": ([I32], CHAR_PTR),                                                     # :212-215
    "minigpt4_quantize_model": ([CHAR_PTR, CHAR_PTR, I32], I32),                                            # :217-222
    "minigpt4_set_verbosity": ([I32], None),                                                                # :224-227
}


def bind(path: str):
    lib = ctypes.cdll.LoadLibrary(path)
    for name, (args, res) in ABI.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    return lib


def cstr(s: str):
    """A `POINTER(c_char)` argument the way the reference passes strings: `s.encode('utf-8')` (bytes convert to char* through ctypes)."""
    return s.encode("utf-8")
