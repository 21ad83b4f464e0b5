"""Stand-alone vision-file converter (minigpt4.cpp_amd/convert.py): checkpoints in the layouts the reference's convert.py consumes (EVA ViT without prefix and with the
unused 40th block / head, BLIP-2 stage checkpoint with the Q-Former's text side, MiniGPT-4 projection checkpoint) must give byte for byte the file the reference's
`write_file` layout produces from the five sub-modules' state dicts (modelgen.write_vision_file restates it), and the product's parser must accept it."""
import ctypes
import os

import numpy as np
import pytest


@pytest.mark.parametrize("ftype", ["f16", "f32"])
def test_converter_reproduces_the_reference_file_layout(lib, tmp_path, ftype):
    torch = pytest.importorskip("torch")
    from minigpt4_cpp_amd import convert as C, modelgen as G
    cfg = G.tiny_vision(n_embd_llm=4096)
    cfg.ftype = ftype
    state = G.vision_state(cfg, seed=9, std=0.05)
    want = str(tmp_path / "want.bin")
    G.write_vision_file(want, cfg, state=state)

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    rng = np.random.default_rng(0)
    junk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    # EVA ViT checkpoint: no prefix, one block more than the tower uses, classifier leftovers
    eva = {k: T(v) for k, v in state["visual_encoder"].items()}
    for k, v in state["visual_encoder"].items():
        if k.startswith("blocks.0."):
            eva[k.replace("blocks.0.", f"blocks.{cfg.depth}.")] = junk(*v.shape)
    eva["norm.weight"], eva["norm.bias"], eva["head.weight"] = junk(cfg.embed_dim), junk(cfg.embed_dim), junk(10, cfg.embed_dim)
    torch.save(eva, str(tmp_path / "eva.pth"))
    # BLIP-2 stage checkpoint: {"model": ...} with the Q-Former's text side and an unrelated head
    b2 = {"ln_vision." + k: T(v) for k, v in state["ln_vision"].items()}
    b2["query_tokens"] = T(state["query_tokens"]["weight"])
    for k, v in state["Qformer"].items():
        b2["Qformer." + k] = T(v)
    b2["Qformer.cls.predictions.bias"] = junk(30522)
    b2["Qformer.bert.embeddings.word_embeddings.weight"] = junk(64, 768)
    b2["Qformer.bert.embeddings.position_embeddings.weight"] = junk(512, 768)
    for i in range(cfg.q_layers):
        b2[f"Qformer.bert.encoder.layer.{i}.intermediate.dense.weight"] = junk(cfg.q_inter, 768)
        b2[f"Qformer.bert.encoder.layer.{i}.output.dense.weight"] = junk(768, cfg.q_inter)
        b2[f"Qformer.bert.encoder.layer.{i}.output.LayerNorm.weight"] = junk(768)
    b2["t5_proj.weight"] = junk(8, 768)
    torch.save({"model": b2}, str(tmp_path / "blip2.pth"))
    torch.save({"model": {"llama_proj." + k: T(v) for k, v in state["llama_proj"].items()}}, str(tmp_path / "mg4.pth"))

    got = str(tmp_path / "got.bin")
    assert C.main(["--eva-vit", str(tmp_path / "eva.pth"), "--blip2", str(tmp_path / "blip2.pth"), "--minigpt4", str(tmp_path / "mg4.pth"), "--ftype", ftype, "--out", got]) == 0
    assert open(got, "rb").read() == open(want, "rb").read()
    n = ctypes.c_int(0)
    assert lib.library.minigpt4_amd_inspect_files(got.encode(), None, ctypes.byref(n), None, None) == 0 and n.value > 0


def test_converter_refuses_incomplete_or_foreign_checkpoints(tmp_path):
    torch = pytest.importorskip("torch")
    from minigpt4_cpp_amd import convert as C, modelgen as G
    cfg = G.tiny_vision(n_embd_llm=4096)
    state = G.vision_state(cfg, seed=9, std=0.05)
    torch.save({"model": {"llama_proj." + k: torch.from_numpy(v) for k, v in state["llama_proj"].items()}}, str(tmp_path / "mg4.pth"))
    with pytest.raises(ValueError, match="no tensors for sub-model"):
        C.convert(str(tmp_path / "x.bin"), "f16", minigpt4=str(tmp_path / "mg4.pth"))
    with pytest.raises(ValueError, match="ftype"):
        C.convert(str(tmp_path / "x.bin"), "q4_0", minigpt4=str(tmp_path / "mg4.pth"))
