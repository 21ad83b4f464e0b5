"""Boundary, CPU side (round-1 verdict item 8): the reference's UNMODIFIED ctypes module and its MiniGPT4ChatBot against libminigpt4.so, as far as a box without a GPU
gets (every declaration binds, the constructor marshals `minigpt4_model_load` and fails where the reference would fail on a missing model), and the honesty check of the
restated ABI table the GPU-side test binds through."""
import ctypes
import importlib.util
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/minigpt4/minigpt4_library.py"
SO = os.path.join(ROOT, "minigpt4.cpp_amd", "libminigpt4.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not mounted (never on the GPU box)")


def _ref_module():
    spec = importlib.util.spec_from_file_location("ref_minigpt4_library_flow", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _StubTorchvision:
    """torchvision is absent from this image; MiniGPT4ChatBot imports it lazily in its constructor for the PIL transform (reference :584-602).  Installed only around
    the constructor call and removed again (a spec-less module left in sys.modules breaks `importlib.util.find_spec('torchvision')` in other packages)."""

    def __enter__(self):
        self.added = []
        if "torchvision" not in sys.modules:
            tv, tr, fn = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")

            class _T:
                def __init__(self, *a, **k):
                    pass

                def __call__(self, x):
                    return x
            tr.Compose = tr.RandomResizedCrop = tr.ToTensor = tr.Normalize = _T
            fn.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
            tv.transforms, tr.functional = tr, fn
            for k, v in {"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": fn}.items():
                sys.modules[k] = v
                self.added.append(k)
        return self

    def __exit__(self, *exc):
        for k in self.added:
            sys.modules.pop(k, None)


def test_restated_abi_table_equals_the_reference_declarations(lib):
    import ref_abi
    mod = _ref_module()
    ref = mod.MiniGPT4SharedLibrary(SO).library
    for name, (args, res) in ref_abi.ABI.items():
        fn = getattr(ref, name)
        got = [a.__name__ if hasattr(a, "__name__") else repr(a) for a in (fn.argtypes or [])]
        want = [a.__name__ if hasattr(a, "__name__") else repr(a) for a in args]
        assert got == want, name
        assert (fn.restype.__name__ if fn.restype is not None else None) == (res.__name__ if res is not None else None), name
    assert ctypes.sizeof(mod.MiniGPT4Image) == ctypes.sizeof(ref_abi.MiniGPT4Image) == 24
    assert ctypes.sizeof(mod.MiniGPT4Embedding) == ctypes.sizeof(ref_abi.MiniGPT4Embedding) == 16
    assert [f[0] for f in mod.MiniGPT4Embedding._fields_] == [f[0] for f in ref_abi.MiniGPT4Embedding._fields_]


def test_reference_chatbot_constructor_marshals_model_load(lib, tiny_files):
    """`MiniGPT4ChatBot(model, llm)` of the unmodified reference: load_library() -> minigpt4_model_load through ITS argtypes.  Without a GPU the engine refuses to load
    (no CPU fallback) and returns NULL exactly like the reference on a load failure, which the reference turns into its own AssertionError (:266)."""
    if lib.amd_device_count() > 0:
        pytest.skip("a GPU is visible: the GPU suite runs the whole flow")
    mod = _ref_module()
    mod.load_library = lambda: mod.MiniGPT4SharedLibrary(SO)
    vp, llm = tiny_files
    with _StubTorchvision(), pytest.raises(AssertionError, match="minigpt4_model_load failed"):
        mod.MiniGPT4ChatBot(vp, llm("q4_0"), verbosity=mod.Verbosity.SILENT)
    # the calls that need no context, through the reference's own wrappers
    w = mod.MiniGPT4SharedLibrary(SO)
    assert w.minigpt4_contains_eos_token("##") and not w.minigpt4_contains_eos_token("#")
    assert w.minigpt4_is_eos("xyz###") and not w.minigpt4_is_eos("xyz##")
    w.minigpt4_set_verbosity(mod.Verbosity.ERR)
    with pytest.raises(AssertionError):
        w.minigpt4_model_load("/nonexistent/a.bin", "/nonexistent/b.bin")
