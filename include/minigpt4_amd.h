/*
 * minigpt4_amd.h -- ADDITIVE entry points of the MI355X build of libminigpt4.so.
 *
 * Nothing here changes the reference ABI (include/minigpt4.h).  These symbols exist for
 *   (1) measurement: device-resident decode loops timed with hipEvents, the per-launch-site table of the decode step, image-encode time;
 *   (2) token-level use of the language path (eval / logits / tokenize / sample) and the parity-mode switch;
 *   (3) batched / multi-GPU serving: the already-declared-but-unused MiniGPT4Images / MiniGPT4Embeddings carriers (reference minigpt4.h:80-90), several
 *       conversations per context, access to the weight arenas for a load-time RCCL broadcast.
 * Kernel-level test hooks, micro-benchmarks, hardware probes and host-only test helpers are NOT part of libminigpt4.so: they are declared in
 * minigpt4_amd_test.h and exported by libminigpt4_test.so (the same objects + csrc/test_hooks.cpp), which only tests/ and tools/ load.
 * Plain C types only (pointers + sizes); no torch / HIP types cross this boundary.
 */
#pragma once
#include "minigpt4.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- environment ------------------------------------------------------------------------------------------------ */
MINIGPT4_API int minigpt4_amd_device_count(void);                      /* 0 when no HIP device is usable */
MINIGPT4_API const char *minigpt4_amd_last_error(void);                /* thread-local text of the last failure */
MINIGPT4_API const char *minigpt4_amd_build_info(void);                /* "gfx950 ..." */

/* ---- language path, token level (parity tests; mirrors MiniGPT4::add_tokens / add_embedding / llama_get_logits) -- */
MINIGPT4_API int minigpt4_amd_n_vocab(struct MiniGPT4Context *ctx);
MINIGPT4_API int minigpt4_amd_n_embd(struct MiniGPT4Context *ctx);
MINIGPT4_API int minigpt4_amd_n_past(struct MiniGPT4Context *ctx);
MINIGPT4_API int minigpt4_amd_eval_tokens(struct MiniGPT4Context *ctx, const int32_t *tokens, int n);    /* 0 / 8 */
MINIGPT4_API int minigpt4_amd_eval_embd(struct MiniGPT4Context *ctx, const float *embd, int n_rows);      /* 0 / 10 */
MINIGPT4_API int minigpt4_amd_get_logits(struct MiniGPT4Context *ctx, float *out, size_t n);              /* last token's logits */
MINIGPT4_API int minigpt4_amd_tokenize(struct MiniGPT4Context *ctx, const char *text, int add_bos, int32_t *out, int cap); /* returns count */
MINIGPT4_API int minigpt4_amd_sample(struct MiniGPT4Context *ctx, int32_t *token_id, float temp, int32_t top_k, float top_p, float tfs_z, float typical_p,
                                     int mirostat, float mirostat_tau, float mirostat_eta);                 /* samples, does NOT eval */

/* Parity mode (also MINIGPT4_PARITY=1 in the environment at load): every fp32 accumulation of the language path in the CPU oracle's order (oracle/refcpu.c) -- logits and
 * greedy ids bit-identical to it; slow (one wavefront per output element).  Takes effect with the next evaluation; the KV cache and positions are kept.  0 / 1 */
MINIGPT4_API int minigpt4_amd_set_parity(struct MiniGPT4Context *ctx, int on);
MINIGPT4_API int minigpt4_amd_parity(struct MiniGPT4Context *ctx);                 /* 1 / 0; -1 without a context */

/* ---- measurement ---------------------------------------------------------------------------------------------------- */
/* `steps` greedy decode steps fed back on the device (no host round trip between steps); hipEvent time of steps 1..steps-1. */
MINIGPT4_API int minigpt4_amd_decode_loop(struct MiniGPT4Context *ctx, int steps, int32_t *tokens_out, float *ms_total);
/* hipEvent-bracketed timing of every quantised mat-vec launch over `steps` eager decode steps.
 * out_ms / out_bytes / out_launches are indexed by ggml type id (length 20); other_ms = everything else in the steps. */
MINIGPT4_API int minigpt4_amd_profile_sites(struct MiniGPT4Context *ctx, int steps, char *json_out, size_t capacity);   /* per-launch-site table of `steps` eager decode steps (the captured graph's launch set): JSON, see Engine::profile_sites; 2 = buffer too small */
MINIGPT4_API double minigpt4_amd_weight_bytes_per_token(struct MiniGPT4Context *ctx);
MINIGPT4_API float minigpt4_amd_last_encode_ms(struct MiniGPT4Context *ctx);     /* hipEvent time of the last minigpt4_encode_image */
MINIGPT4_API int minigpt4_amd_sync(struct MiniGPT4Context *ctx);

/* ---- batched image encode (data-parallel requests; carriers from reference minigpt4.h:80-90) --------------------- */
/* Encodes images->n_images images; allocates embeddings->embeddings[i].data like minigpt4_encode_image does. */
MINIGPT4_API int minigpt4_amd_encode_images(struct MiniGPT4Context *ctx, IN const struct MiniGPT4Images *images, OUT struct MiniGPT4Embeddings *embeddings, size_t n_threads);
MINIGPT4_API int minigpt4_amd_free_embeddings(struct MiniGPT4Embeddings *embeddings);
/* DEPRECATED aliases of the two functions above (their names until round 5): they live in the reference's own minigpt4_ namespace and would collide with an upstream
 * implementation of its declared-but-unused MiniGPT4Images API (reference minigpt4.h:80-90).  Forwarders for one more release; new callers use the minigpt4_amd_ names. */
MINIGPT4_API int minigpt4_encode_images(struct MiniGPT4Context *ctx, IN const struct MiniGPT4Images *images, OUT struct MiniGPT4Embeddings *embeddings, size_t n_threads);
MINIGPT4_API int minigpt4_free_embeddings(struct MiniGPT4Embeddings *embeddings);

/* ---- several conversations per context: batched decode ---------------------------------------------------------
 * The reference keeps ONE conversation per context (minigpt4.cpp:2513-2521).  Here a context can own n of them -- each with its own fp16 KV cache region,
 * position and prompt queue, all sharing one copy of the weights.  Every reference entry point (minigpt4_system_prompt, minigpt4_begin_chat(_image),
 * minigpt4_end_chat(_image), minigpt4_reset_chat) acts on the SELECTED conversation (0 after load), so existing callers see no change. */
MINIGPT4_API int minigpt4_amd_set_conversations(struct MiniGPT4Context *ctx, int n);        /* 1..64; reallocates the KV caches and resets every conversation; 0 / 1 */
MINIGPT4_API int minigpt4_amd_select_conversation(struct MiniGPT4Context *ctx, int slot);   /* 0 / 1 (out of range) */
MINIGPT4_API int minigpt4_amd_n_conversations(struct MiniGPT4Context *ctx);
/* One minigpt4_end_chat step for n DISTINCT conversations at once: each is sampled with the given parameters (temp <= 0: greedy) and the n sampled tokens are
 * evaluated in ONE pass over the weights.  tokens[i]: borrowed piece for slots[i], as minigpt4_end_chat returns it.  A conversation whose context is full is
 * sampled but not advanced.  0, or 1 on bad arguments / device error. */
MINIGPT4_API int minigpt4_amd_end_chat_batch(struct MiniGPT4Context *ctx, const int32_t *slots, int n, const char **tokens, float temp, int32_t top_k, float top_p, float tfs_z,
                                             float typical_p, int mirostat, float mirostat_tau, float mirostat_eta);
/* The same step with GIVEN next tokens (teacher forcing): conversation slots[i] is advanced by tokens[i] instead of the token its own logits choose; greedy_out[i] (may be
 * NULL) receives that own greedy choice.  Lets a test / bench leg compare every step's logits of B batched conversations with B independent oracle conversations fed the
 * same ids (reference behaviour: one independent conversation per context, minigpt4.cpp:2513-2521, 2704-2718).  0 / 1. */
MINIGPT4_API int minigpt4_amd_eval_batch(struct MiniGPT4Context *ctx, const int32_t *slots, int n, const int32_t *tokens, int32_t *greedy_out);
/* Launch kinds of the batched step as last built (eager or at graph capture): out = {rows, k_matvec_ri launches, k_matvec_ri_mix launches, k_matvec_ri launches that split K
 * over workgroups (w2), v_dot4 multi-row launches, v_dot4 mixed-type launches, per-matrix k_mul_mat launches, layers on the int8-MFMA set launches (B >= 5)}.  0 / 1. */
MINIGPT4_API int minigpt4_amd_batch_path(struct MiniGPT4Context *ctx, int32_t out[8]);

/* ---- weight arenas (load-time broadcast rank0 -> others over RCCL) ---------------------------- */
/* which: 0 = LLM arena, 1 = vision arena.  Returns the device pointer and size in bytes. */
MINIGPT4_API int minigpt4_amd_weight_arena(struct MiniGPT4Context *ctx, int which, void **device_ptr, size_t *bytes);
/* Multi-GPU load (replicas; the only exchange is the load-time broadcast of the two weight arenas from rank 0).  Rank 0 loads normally; the other ranks
 * set MINIGPT4_LOAD=recv before minigpt4_model_load: headers are parsed, both arenas are laid out and allocated exactly as rank 0's (compare minigpt4_amd_arena_plan), no
 * tensor data is read, uploaded or repacked; after the arenas have been received (minigpt4_amd_weight_arena gives the device ranges) minigpt4_amd_weights_received builds
 * what is derived from them on the device.  minigpt4_amd_plan_arenas computes the same layout from the files alone, without a GPU. */
/* The same exchange INSIDE minigpt4_model_load, for clients without Python / torch (csrc/dist.cpp): with MINIGPT4_WORLD_SIZE = N, MINIGPT4_RANK = r and
 * MINIGPT4_NCCL_ID_FILE = <a path every rank of the job can read, unique to the job> in the environment, rank 0 loads the files, writes the ncclUniqueId to that file and
 * broadcasts both arenas (librccl.so by dlopen, <= 1 GiB pieces); ranks 1..N-1 load headers only, wait for the id (MINIGPT4_DIST_TIMEOUT_S, default 120), receive, finish
 * the load; layout hashes before and arena checksums after must agree.  Any failure fails the load (NULL; text in minigpt4_amd_last_error) -- no fallback to the files.
 * One process per GPU: MINIGPT4_DEVICE (or LOCAL_RANK) selects it.  minigpt4_amd_dist_info: what this context's load did (bcast_ms = 0 for an ordinary load). */
MINIGPT4_API int minigpt4_amd_dist_info(struct MiniGPT4Context *ctx, int *world, int *rank, float *bcast_ms);
MINIGPT4_API int minigpt4_amd_plan_arenas(const char *vision_path, const char *llm_path, size_t *llm_bytes, size_t *vision_bytes, uint64_t *llm_hash, uint64_t *vision_hash);
MINIGPT4_API int minigpt4_amd_arena_plan(struct MiniGPT4Context *ctx, size_t *llm_bytes, size_t *vision_bytes, uint64_t *llm_hash, uint64_t *vision_hash);
MINIGPT4_API int minigpt4_amd_load_mode(struct MiniGPT4Context *ctx);              /* 0 full, 1 waiting for the arenas */
MINIGPT4_API int minigpt4_amd_weights_received(struct MiniGPT4Context *ctx);
MINIGPT4_API int minigpt4_amd_arena_checksum(struct MiniGPT4Context *ctx, int which, uint64_t *sum);   /* 64-bit sum of the arena's 32-bit words (device reduction) */

/* Image file decoding from memory (the host half of minigpt4_image_load_from_file; the request server takes images as bytes): PNG / JPEG / BMP / binary PNM bytes -> U8 HWC RGB
 * with cv::imread(IMREAD_COLOR)+BGR2RGB semantics.  The library allocates image->data; release with minigpt4_free_image.  0 or 5 (OpenImage). */
MINIGPT4_API int minigpt4_amd_decode_image(const void *bytes, size_t n, OUT struct MiniGPT4Image *image);

#ifdef __cplusplus
}
#endif
