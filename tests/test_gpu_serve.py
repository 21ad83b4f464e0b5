"""GPU test of the request-level server (minigpt4.cpp_amd/serve.py): a wave of requests -- image as a preprocessed array, as a PNG path and as JPEG bytes
(native decode + preprocess kernels) -- is encoded in one pass, prompted into separate conversations and decoded together; every answer must equal the
answer of the same request served alone through the reference's single-conversation call sequence."""
import os

import numpy as np
import pytest

from test_cpu_image import IMG_DIR

pytestmark = pytest.mark.gpu


def test_batched_requests_equal_requests_served_alone(gpu_lib, tmpdir_models):
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G, serve as S
    vp = os.path.join(tmpdir_models, "vision_serve.bin")
    lp = os.path.join(tmpdir_models, "llm_serve.bin")
    G.write_vision_file(vp, G.tiny_vision(n_embd_llm=4096), seed=31, std=0.05)
    G.write_llm_file(lp, G.tiny_llm(wtype="q4_0", n_embd=4096, n_layer=1, n_head=32, n_vocab=512, output_type="q6_k"), seed=4, std=0.02)
    reqs = [S.Request(G.synth_image(3), "what is the text in the picture?", 8),
            S.Request(os.path.join(IMG_DIR, "png_llama_small.png"), "describe it", 6),
            S.Request(open(os.path.join(IMG_DIR, "jpg_420_64x64_base.jpg"), "rb").read(), "colour?", 7)]
    srv = S.ReplicaServer(vp, lp, conversations=2, n_ctx=512, n_batch=64, library=gpu_lib)     # 3 requests, 2 conversations: waves of 2 + 1
    try:
        got = srv.run(reqs, temp=0.0, ignore_eos=True)
        alone = []
        for r in reqs:
            one = srv.run([r], temp=0.0, ignore_eos=True)
            alone.append(one[0])
        assert got == alone
        assert all(len(a) > 0 for a in got)
        # the single-conversation reference call sequence on the same context (conversation 0) gives the same first answer
        lib, ctx = srv.lib, srv.ctx
        lib.amd_select_conversation(ctx, 0)
        lib.minigpt4_reset_chat(ctx)
        lib.minigpt4_system_prompt(ctx)
        emb = lib.minigpt4_encode_image(ctx, ML.array_to_image_struct(reqs[0].image))
        lib.minigpt4_begin_chat_image(ctx, emb, reqs[0].prompt)
        ref = "".join(lib.minigpt4_end_chat_image(ctx, temp=0.0) for _ in range(8))
        lib.minigpt4_free_embedding(emb)
        assert ref == got[0]
        # stop rule: with EOS handling on, answers never contain the "###" terminator and are prefixes of the unrestricted ones
        stopped = srv.run(reqs, temp=0.0)
        for a, b in zip(stopped, got):
            assert "###" not in a and len(a) <= len(b)
    finally:
        srv.close()


def test_chatbot_uploads_a_file_through_the_native_image_path(gpu_lib, tmpdir_models):
    """MiniGPT4ChatBot.upload_image(path): decode + Pillow-exact preprocess inside the library; same answer as uploading the oracle-preprocessed array."""
    import refimage as RI
    from test_cpu_image import GOLD
    from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
    vp = os.path.join(tmpdir_models, "vision_serve.bin")
    lp = os.path.join(tmpdir_models, "llm_serve.bin")
    if not os.path.exists(vp):
        G.write_vision_file(vp, G.tiny_vision(n_embd_llm=4096), seed=31, std=0.05)
        G.write_llm_file(lp, G.tiny_llm(wtype="q4_0", n_embd=4096, n_layer=1, n_head=32, n_vocab=512, output_type="q6_k"), seed=4, std=0.02)
    bot = ML.MiniGPT4ChatBot(vp, lp, library=gpu_lib, n_ctx=512, n_batch=64)
    try:
        bot.upload_image(os.path.join(IMG_DIR, "png_llama_small.png"))
        a = [t for _, t in zip(range(6), bot.generate("what is it?", limit=6, temp=0.0, ignore_eos=True))]
        bot.upload_image(RI.preprocess(GOLD["decoded/png_llama_small.png"]))
        b = [t for _, t in zip(range(6), bot.generate("what is it?", limit=6, temp=0.0, ignore_eos=True))]
        bot.upload_image(open(os.path.join(IMG_DIR, "png_llama_small.png"), "rb").read())
        c = [t for _, t in zip(range(6), bot.generate("what is it?", limit=6, temp=0.0, ignore_eos=True))]
        assert a == b == c and len(a) == 6
    finally:
        bot.free()
