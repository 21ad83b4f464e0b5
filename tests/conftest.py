import os
import sys

# The oracle is OpenMP code; the GPU boxes have 256 hardware threads and the test problems are tiny -- a parallel region per op with 256
# spinning threads costs far more than the op.  (bench.py's cpu_baseline leg sets its own thread count.)
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import _pkg  # noqa: E402

_pkg.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tmpdir_models(tmp_path_factory):
    return str(tmp_path_factory.mktemp("models"))


@pytest.fixture(scope="session")
def lib():
    # A process that uses torch's GPU side (tests of dist.arena_tensor / RCCL) next to libminigpt4.so must let torch bring up ITS bundled HIP runtime first: with
    # the library's /opt/rocm runtime mapped first, torch.cuda later fails with "No HIP GPUs are available" (seen on the GPU box, round 2).  bench.py has that order.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from minigpt4_cpp_amd import minigpt4_library as ML
    so = ML.default_library_path()
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    return ML.load_library()


@pytest.fixture(scope="session")
def gpu_lib(lib):
    if lib.amd_device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (the engine has no CPU fallback)")
    return lib


@pytest.fixture(scope="session")
def tiny_files(tmpdir_models):
    """(vision_path, {wtype: llm_path}) -- small models in the reference's two file formats."""
    from minigpt4_cpp_amd import modelgen as G
    vp = os.path.join(tmpdir_models, "vision_tiny.bin")
    G.write_vision_file(vp, G.tiny_vision(n_embd_llm=4096), seed=3, std=0.05)
    made = {}

    def llm(wtype, mix="none", n_vocab=512, output_type=None, tok_type=None, conditioned=False):
        """conditioned: modelgen.TINY_CONDITIONED (scaled residual writers, decisive logits) -- the whole-model tolerance tests; False: plain i.i.d. Gaussian weights."""
        key = (wtype, mix, n_vocab, output_type, tok_type, conditioned)
        if key not in made:
            p = os.path.join(tmpdir_models, f"llm_{wtype}_{mix}_{n_vocab}_{output_type}_{tok_type}_{int(conditioned)}.bin")
            G.write_llm_file(p, G.tiny_llm(wtype=wtype, n_embd=256, n_layer=2, n_head=4, n_vocab=n_vocab, mix=mix, output_type=output_type,
                                           tok_type=tok_type), seed=1, std=0.05, **(G.TINY_CONDITIONED if conditioned else {}))
            made[key] = p
        return made[key]
    return vp, llm


# ---- observed whole-model errors: every run writes what it measured to gpurun_out/parity_observed_tiny.json; tests/golden/parity_observed_tiny.json is the committed record
# of a GPU box, and a bar is min(north_star's 1e-2, 2 x the recorded value) -- so a numerics regression fails even when it stays inside the absolute tolerance.
_OBS = {}


def record_observed(name: str, value: float) -> None:
    import json
    _OBS[name] = max(float(value), _OBS.get(name, 0.0))
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, "parity_observed_tiny.json")
    cur = {}
    if os.path.exists(p):
        try:
            cur = json.load(open(p))
        except ValueError:
            cur = {}
    cur.update(_OBS)
    json.dump(cur, open(p, "w"), indent=1, sort_keys=True)


def observed_bar(name: str, absolute: float = 1e-2) -> float:
    import json
    p = os.path.join(ROOT, "tests", "golden", "parity_observed_tiny.json")
    rec = json.load(open(p)) if os.path.exists(p) else {}
    # 5e-3 floor: a run in which no int8 rounding happened to flip records ~1e-7; one flipped rounding (a benign change of summation order) costs ~2e-3 on these models
    return min(absolute, max(2.0 * rec[name], 5e-3)) if name in rec and rec[name] > 0 else absolute
