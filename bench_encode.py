#!/usr/bin/env python3
"""Image-encode micro-benchmark on the 13B-shaped vision file (synthetic); prints ms per encode (hipEvents)."""
import ctypes, os, sys, time
import _pkg
_pkg.load_package()
from minigpt4_cpp_amd import minigpt4_library as ML, modelgen as G
d = "/dev/shm/mg4_bench" if os.path.isdir("/dev/shm") else "/tmp/mg4_bench"
os.makedirs(d, exist_ok=True)
vp, lp = os.path.join(d, "vision_13b.bin"), os.path.join(d, "llm_enc_tiny.bin")
if not os.path.exists(vp + ".ok"):
    G.write_vision_file(vp, G.vision_13b(), unique_blocks=1, fast=True); open(vp + ".ok", "w").write("ok")
if not os.path.exists(lp):
    G.write_llm_file(lp, G.tiny_llm(wtype="q4_0", n_embd=256, n_layer=1, n_head=4, n_vocab=512))
lib = ML.load_library()
ctx = lib.minigpt4_model_load(vp, lp, verbosity=1, n_ctx=64, n_batch=32)
img = ML.array_to_image_struct(G.synth_image(1))
import zlib
ms, sig = [], 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):          # 0: no single-image encodes (a per-batch-size rocprofv3 pass wants ONE kind of launch)
    e = lib.minigpt4_encode_image(ctx, img); ms.append(lib.library.minigpt4_amd_last_encode_ms(ctx.ptr))
    sig = zlib.crc32(ctypes.string_at(e.data, e.n_embeddings * 4))        # identical embeddings <=> identical signature (bit-exactness of A/B arms)
    lib.minigpt4_free_embedding(e)
if ms:
    print("encode ms (device):", ["%.2f" % m for m in ms], "best %.2f" % min(ms), "sig %08x" % sig)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 0                          # optional: B images per encode_images call (one pass over the vision weights)
if nb > 1:
    imgs = [ML.array_to_image_struct(G.synth_image(1 + i)) for i in range(nb)]
    arr = (ML.MiniGPT4Image * nb)(*imgs)
    batch, outb = ML.MiniGPT4Images(arr, nb), ML.MiniGPT4Embeddings()
    msb = []
    for _ in range(5):
        assert lib.library.minigpt4_amd_encode_images(ctx.ptr, ctypes.byref(batch), ctypes.byref(outb), 0) == 0
        msb.append(lib.library.minigpt4_amd_last_encode_ms(ctx.ptr))
        sigb = zlib.crc32(b"".join(ctypes.string_at(outb.embeddings[i].data, outb.embeddings[i].n_embeddings * 4) for i in range(nb)))
        lib.library.minigpt4_amd_free_embeddings(ctypes.byref(outb))
    print("batched encode, %d images, ms (device):" % nb, ["%.2f" % m for m in msb], "best %.2f = %.2f per image" % (min(msb), min(msb) / nb), "sig %08x" % sigb)
lib.minigpt4_free(ctx)
