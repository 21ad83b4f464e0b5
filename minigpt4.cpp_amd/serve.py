"""Batched image+prompt serving on one replica, and the data-parallel wrapper around it (BASELINE.json configs[3]: 32 requests over 8 GPUs).

A request = (image, prompt) owns its embedding, KV cache and position, so a replica can run several of them side by side:

  * the images of a wave are encoded in ONE pass over the vision weights (`minigpt4_amd_encode_images`, up to 8 per pass);
  * every request gets its own conversation of the context (`minigpt4_amd_set_conversations` / `_select_conversation`) and is prompted through the
    reference entry points exactly as `MiniGPT4ChatBot.generate` does (`minigpt4_library.py:627-644` of the reference): system prompt, image turn;
  * decode steps run for all unfinished conversations at once (`minigpt4_amd_end_chat_batch`: one pass over the LLM weights per 4 conversations);
  * per conversation the reference's stop rule applies: a piece equal to "##" is swallowed, the answer ends when the accumulated text ends with "###"
    (`minigpt4_contains_eos_token` / `minigpt4_is_eos`, reference `generate`), or after `max_tokens`.

Across GPUs requests shard round-robin (`dist.shard_requests`); ranks exchange nothing per token and the answers are gathered at the end.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np

from . import dist as D
from . import minigpt4_library as ML


@dataclass
class Request:
    image: Union[str, bytes, np.ndarray]      # file path, encoded image bytes (PNG / JPEG / ...), or a preprocessed f32 [3][224][224] array
    prompt: str
    max_tokens: int = 64


def plan_waves(n_requests: int, conversations: int) -> List[List[int]]:
    """Requests are served in waves of `conversations`; inside a wave they decode together until each one stops."""
    return [list(range(i, min(n_requests, i + conversations))) for i in range(0, n_requests, conversations)]


class ReplicaServer:
    def __init__(self, vision_path: str, llm_path: str, conversations: int = 4, n_ctx: int = 2048, n_batch: int = 512, seed: int = 1337,
                 library: Optional[ML.MiniGPT4SharedLibrary] = None, verbosity: int = 0, rank: int = 0, world: int = 1, device=None):
        """world > 1 (inside an initialised `torch.distributed` job): the replica is loaded through `dist.load_replica` -- rank 0 reads the files, every other
        rank loads headers only and receives both weight arenas by broadcast, so the weights cross the file system once per node."""
        self.lib = library or ML.load_library()
        self.load_stats = None
        if world > 1:
            self.ctx, self.load_stats = D.load_replica(self.lib, vision_path, llm_path, rank, world, device=device, verbosity=verbosity, seed=seed, n_ctx=n_ctx, n_batch=n_batch)
        else:
            self.ctx = self.lib.minigpt4_model_load(vision_path, llm_path, verbosity=verbosity, seed=seed, n_ctx=n_ctx, n_batch=n_batch)
        self.conversations = conversations
        self.lib.amd_set_conversations(self.ctx, conversations)

    def close(self):
        if self.ctx is not None:
            self.lib.minigpt4_free(self.ctx)
            self.ctx = None

    # ---- image -> MiniGPT4Image (F32 CHW), through the library's own loader / preprocess kernels for files and bytes
    def _to_struct(self, image):
        if isinstance(image, np.ndarray):
            return ML.array_to_image_struct(image), []
        raw = self.lib.amd_decode_image(image) if isinstance(image, (bytes, bytearray)) else self.lib.minigpt4_image_load_from_file(self.ctx, image)
        pre = self.lib.minigpt4_preprocess_image(self.ctx, raw)
        return pre, [raw, pre]

    def run(self, requests: Sequence[Request], temp: float = 0.0, top_k: int = 40, top_p: float = 0.9, ignore_eos: bool = False) -> List[str]:
        import ctypes
        lib, ctx = self.lib, self.ctx
        answers: List[str] = [""] * len(requests)
        for wave in plan_waves(len(requests), self.conversations):
            structs, owned = [], []
            for i in wave:
                st, own = self._to_struct(requests[i].image)
                structs.append(st)
                owned += own
            arr = (ML.MiniGPT4Image * len(wave))(*structs)
            batch, embs = ML.MiniGPT4Images(arr, len(wave)), ML.MiniGPT4Embeddings()
            lib.panic_if_error(lib.library.minigpt4_amd_encode_images(ctx.ptr, ctypes.byref(batch), ctypes.byref(embs), 0))
            try:
                for slot, i in enumerate(wave):
                    lib.amd_select_conversation(ctx, slot)
                    lib.minigpt4_reset_chat(ctx)
                    lib.minigpt4_system_prompt(ctx)
                    lib.minigpt4_begin_chat_image(ctx, embs.embeddings[slot], requests[i].prompt)
                text = {slot: "" for slot in range(len(wave))}
                shown = {slot: "" for slot in range(len(wave))}
                left = {slot: requests[i].max_tokens for slot, i in enumerate(wave)}
                active = [slot for slot in range(len(wave)) if left[slot] > 0]
                while active:
                    pieces = lib.amd_end_chat_batch(ctx, active, temp=temp, top_k=top_k, top_p=top_p)
                    nxt = []
                    for slot, piece in zip(active, pieces):
                        left[slot] -= 1
                        text[slot] += piece
                        done = left[slot] <= 0
                        if not ignore_eos:
                            if lib.minigpt4_contains_eos_token(piece):
                                pass                                  # swallowed, like the reference's `continue`
                            elif lib.minigpt4_is_eos(text[slot]):
                                done = True
                            else:
                                shown[slot] += piece
                        else:
                            shown[slot] += piece
                        if not done:
                            nxt.append(slot)
                    active = nxt
                for slot, i in enumerate(wave):
                    answers[i] = shown[slot]
            finally:
                lib.library.minigpt4_amd_free_embeddings(ctypes.byref(embs))
                for im in owned:
                    lib.minigpt4_free_image(im)
        lib.amd_select_conversation(ctx, 0)
        return answers


def serve(requests: Sequence[Request], vision_path: str, llm_path: str, conversations: int = 4, **kw) -> Optional[List[str]]:
    """Data-parallel entry point: call it on every rank of a `torch.distributed` job (or alone).  Rank r serves requests r, r + world, ...;
    rank 0 returns all answers in request order, the other ranks return None.  With world > 1 the replicas are loaded through `dist.load_replica`
    (file load on rank 0, arena broadcast to the others); pass `device=torch.device("cuda", local_rank)` on GPUs."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    mine = D.shard_requests(len(requests), rank, world)
    server = ReplicaServer(vision_path, llm_path, conversations=conversations, rank=rank, world=world, device=kw.get("device"),
                           **{k: v for k, v in kw.items() if k in ("n_ctx", "n_batch", "seed", "library", "verbosity")})
    try:
        out = server.run([requests[i] for i in mine], **{k: v for k, v in kw.items() if k in ("temp", "top_k", "top_p", "ignore_eos")})
    finally:
        server.close()
    mapping = dict(zip(mine, out))
    if world == 1:
        return merge_answers(len(requests), [mapping])
    per_rank = D.gather_objects(mapping, world)            # a collective: every rank calls it exactly once
    return merge_answers(len(requests), per_rank) if rank == 0 else None


def merge_answers(n_requests: int, per_rank: Sequence[dict]) -> List[str]:
    merged = {}
    for d in per_rank:
        merged.update(d)
    assert sorted(merged) == list(range(n_requests)), "every request must be answered by exactly one rank"
    return [merged[i] for i in range(n_requests)]
