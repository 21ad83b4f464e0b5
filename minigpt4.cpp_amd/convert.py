"""Stand-alone writer of the MiniGPT-4 vision file (the `"ggml"` v0 container minigpt4_model_load reads) from plain PyTorch state dicts.

The reference's `minigpt4/convert.py` builds the whole MiniGPT-4 model from a checkout of Vision-CAIR/MiniGPT-4 (its `Blip2Base`, the EVA ViT and BERT
sources, downloaded checkpoints; /root/reference/minigpt4/convert.py:13-18, 181-260) only to call `state_dict()` on five sub-modules and hand the tensors to
`write_file` (:146-180).  This converter needs no model code: it merges the checkpoints' own state dicts, keeps exactly the tensors those five sub-modules
would have had, and writes them with the byte layout of `write_file` / `write_model` (:74-144) -- same sub-model order, dtype rule, squeezed + reversed
shapes, page-aligned tensor data.

    python -m minigpt4_cpp_amd.convert --eva-vit eva_vit_g.pth --blip2 blip2_pretrained_flant5xxl.pth --minigpt4 pretrained_minigpt4.pth \\
        --ftype f16 --out minigpt4-13B-f16.bin            (through `_pkg.load_package()`; or run this file directly)

Inputs (each a `torch.save`d dict, optionally wrapped as {"model": ...} / {"state_dict": ...}; `.npz` works too):
  * `--eva-vit`   EVA-CLIP ViT-g weights without prefix (`cls_token`, `pos_embed`, `patch_embed.proj.*`, `blocks.N.*`): they become `visual_encoder.*`; MiniGPT-4
                  builds the tower with depth 39, so block 39 and the classifier head / final norm are dropped (`eva_vit.py: create_eva_vit_g`);
  * `--blip2`     BLIP-2 stage checkpoint (`ln_vision.*`, `query_tokens`, `Qformer.*`, possibly `visual_encoder.*`): MiniGPT-4 deletes the Q-Former's text side
                  (`cls`, word / position embeddings, every layer's text `intermediate` / `output`; mini_gpt4.py) -- dropped here as well;
  * `--minigpt4`  the MiniGPT-4 checkpoint (`llama_proj.*`);
  * `--state`     any further state dict whose keys already carry the five prefixes (later files override earlier ones).
Host-side tooling: nothing here runs on the product's compute path.
"""
from __future__ import annotations

import argparse
import re
import sys
from typing import Dict, Iterable, Optional

import numpy as np

SUBMODELS = ("visual_encoder", "ln_vision", "query_tokens", "Qformer", "llama_proj")
_QFORMER_TEXT_SIDE = re.compile(r"^Qformer\.(cls\.|bert\.embeddings\.(word_embeddings|position_embeddings)\.|bert\.encoder\.layer\.\d+\.(intermediate|output)\.)")


def _to_numpy(v) -> np.ndarray:
    if isinstance(v, np.ndarray):
        return v
    if hasattr(v, "detach"):                               # torch.Tensor
        v = v.detach().cpu()
        if str(v.dtype) in ("torch.bfloat16", "torch.float16"):
            v = v.float()
        return v.numpy()
    return np.asarray(v)


def load_state(path: str) -> Dict[str, np.ndarray]:
    """A checkpoint's flat {key: array}; unwraps the usual {"model": ...} / {"state_dict": ...} envelopes."""
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("model", "state_dict"):
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            obj = obj[key]
    return {k: _to_numpy(v) for k, v in obj.items() if hasattr(v, "shape")}


def _vit_depth_kept(keys: Iterable[str], drop_last_block: bool) -> int:
    n = 1 + max((int(m.group(1)) for k in keys for m in [re.match(r"^(?:visual_encoder\.)?blocks\.(\d+)\.", k)] if m), default=-1)
    return n - 1 if drop_last_block else n


def merge_states(eva_vit: Optional[Dict[str, np.ndarray]] = None, blip2: Optional[Dict[str, np.ndarray]] = None, minigpt4: Optional[Dict[str, np.ndarray]] = None,
                 extra: Iterable[Dict[str, np.ndarray]] = (), eva_has_extra_block: bool = True) -> Dict[str, Dict[str, np.ndarray]]:
    """{sub-model: {layer name: array}} exactly as the five `state_dict()` calls of convert.py:160-164 would have returned them (insertion order = checkpoint order)."""
    flat: Dict[str, np.ndarray] = {}
    if eva_vit:
        keep = _vit_depth_kept(eva_vit.keys(), eva_has_extra_block)
        for k, v in eva_vit.items():
            k = k[len("visual_encoder."):] if k.startswith("visual_encoder.") else k
            m = re.match(r"^blocks\.(\d+)\.", k)
            if (m and int(m.group(1)) >= keep) or k.startswith(("head.", "norm.", "fc_norm.")):
                continue                                    # create_eva_vit_g(depth = 39): the last block, the final norm and the head are not part of the tower
            flat["visual_encoder." + k] = v
    for sd in ([blip2] if blip2 else []) + ([minigpt4] if minigpt4 else []) + list(extra):
        for k, v in sd.items():
            if k.startswith(SUBMODELS):
                flat[k] = v
    out: Dict[str, Dict[str, np.ndarray]] = {m: {} for m in SUBMODELS}
    for k, v in flat.items():
        if _QFORMER_TEXT_SIDE.match(k):
            continue
        if k == "query_tokens":
            out["query_tokens"]["weight"] = v               # convert.py:162: {'weight': minigpt4.query_tokens}
            continue
        model, _, layer = k.partition(".")
        if model in out and layer:
            out[model][layer] = v
    missing = [m for m in SUBMODELS if not out[m]]
    if missing:
        raise ValueError("no tensors for sub-model(s): " + ", ".join(missing))
    return out


def config_from_state(state: Dict[str, Dict[str, np.ndarray]], ftype: str):
    """VisionConfig (shapes only) of a merged state; refuses geometries the reference hard-codes otherwise (minigpt4.cpp:127-131, 1271, 1614-1627)."""
    from . import modelgen as G
    ve, qf = state["visual_encoder"], state["Qformer"]
    D = int(np.squeeze(ve["pos_embed"]).shape[-1])
    depth = _vit_depth_kept(ve.keys(), False)
    mlp = int(ve["blocks.0.mlp.fc1.weight"].shape[0])
    ql = 1 + max(int(m.group(1)) for k in qf for m in [re.match(r"^bert\.encoder\.layer\.(\d+)\.", k)] if m)
    q_inter = int(qf["bert.encoder.layer.0.intermediate_query.dense.weight"].shape[0])
    nq = int(np.squeeze(state["query_tokens"]["weight"]).shape[0])
    n_out = int(state["llama_proj"]["weight"].shape[0])
    cross = [int(m.group(1)) for k in qf for m in [re.match(r"^bert\.encoder\.layer\.(\d+)\.crossattention\.self\.query\.weight$", k)] if m]
    freq = (sorted(cross)[1] - sorted(cross)[0]) if len(cross) > 1 else 2
    if D % 88 or int(np.squeeze(ve["pos_embed"]).shape[0]) != 257 or nq != 32 or n_out not in (4096, 5120):
        raise ValueError(f"geometry outside what the reference supports: embed_dim {D}, positions {np.squeeze(ve['pos_embed']).shape[0]}, queries {nq}, llama width {n_out}")
    return G.VisionConfig(embed_dim=D, depth=depth, mlp_dim=mlp, q_layers=ql, q_inter=q_inter, q_queries=nq, cross_freq=freq, n_embd_llm=n_out, ftype=ftype)


def convert(out_path: str, ftype: str = "f16", eva_vit: Optional[str] = None, blip2: Optional[str] = None, minigpt4: Optional[str] = None, states: Iterable[str] = (),
            eva_has_extra_block: bool = True) -> None:
    from . import modelgen as G
    if ftype not in ("f16", "f32"):
        raise ValueError("ftype must be f16 or f32")
    state = merge_states(load_state(eva_vit) if eva_vit else None, load_state(blip2) if blip2 else None, load_state(minigpt4) if minigpt4 else None,
                         [load_state(p) for p in states], eva_has_extra_block)
    G.write_vision_file(out_path, config_from_state(state, ftype), state=state)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--eva-vit")
    ap.add_argument("--blip2")
    ap.add_argument("--minigpt4")
    ap.add_argument("--state", action="append", default=[])
    ap.add_argument("--keep-all-vit-blocks", action="store_true", help="the --eva-vit file already holds exactly the tower's blocks (no extra last block)")
    ap.add_argument("--ftype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)
    convert(a.out, a.ftype, a.eva_vit, a.blip2, a.minigpt4, a.state, not a.keep_all_vit_blocks)
    print("wrote", a.out)
    return 0


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import _pkg
    _pkg.load_package()
    from minigpt4_cpp_amd import convert as _self
    sys.exit(_self.main())
