/*
 * minigpt4_amd.h -- ADDITIVE entry points of the MI355X build of libminigpt4.so.
 *
 * Nothing here changes the reference ABI (include/minigpt4.h).  These symbols exist for
 *   (1) measurement: device-resident decode loops timed with hipEvents, per-kernel-class timing, image-encode time;
 *   (2) parity tests through the C-ABI: token-level eval / logits, single kernels (mat-mul, activation quantisation);
 *   (3) host logic that needs no GPU: file parsing, tokenizer, sampler (run by the CPU-only test tier);
 *   (4) batched / multi-GPU use: the already-declared-but-unused MiniGPT4Images / MiniGPT4Embeddings carriers
 *       (reference minigpt4.h:80-90) and access to the weight arenas for a load-time RCCL broadcast.
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

/* ---- weight arenas (load-time broadcast rank0 -> others over RCCL) ---------------------------- */
/* which: 0 = LLM arena, 1 = vision arena.  Returns the device pointer and size in bytes. */
MINIGPT4_API int minigpt4_amd_weight_arena(struct MiniGPT4Context *ctx, int which, void **device_ptr, size_t *bytes);
/* Multi-GPU load (replicas; the only exchange is the load-time broadcast of the two weight arenas from rank 0).  Rank 0 loads normally; the other ranks
 * set MINIGPT4_LOAD=recv before minigpt4_model_load: headers are parsed, both arenas are laid out and allocated exactly as rank 0's (compare minigpt4_amd_arena_plan), no
 * tensor data is read, uploaded or repacked; after the arenas have been received (minigpt4_amd_weight_arena gives the device ranges) minigpt4_amd_weights_received builds
 * what is derived from them on the device.  minigpt4_amd_plan_arenas computes the same layout from the files alone, without a GPU. */
MINIGPT4_API int minigpt4_amd_plan_arenas(const char *vision_path, const char *llm_path, size_t *llm_bytes, size_t *vision_bytes, uint64_t *llm_hash, uint64_t *vision_hash);
MINIGPT4_API int minigpt4_amd_arena_plan(struct MiniGPT4Context *ctx, size_t *llm_bytes, size_t *vision_bytes, uint64_t *llm_hash, uint64_t *vision_hash);
MINIGPT4_API int minigpt4_amd_load_mode(struct MiniGPT4Context *ctx);              /* 0 full, 1 waiting for the arenas */
MINIGPT4_API int minigpt4_amd_weights_received(struct MiniGPT4Context *ctx);
MINIGPT4_API int minigpt4_amd_copy_arenas(struct MiniGPT4Context *dst, struct MiniGPT4Context *src);   /* tests: device-to-device stand-in for the broadcast on one GPU */
MINIGPT4_API int minigpt4_amd_arena_checksum(struct MiniGPT4Context *ctx, int which, uint64_t *sum);   /* 64-bit sum of the arena's 32-bit words (device reduction) */

/* ---- single-kernel hooks for parity tests (need a GPU; allocate + free their own device memory) ------------------ */
/* y[N][n_out] = W . x with ggml's quantised-activation arithmetic.  raw_w: the tensor bytes exactly as stored in a model file. */
MINIGPT4_API int minigpt4_amd_test_mul_mat(int ggml_type, const void *raw_w, int64_t n_in, int64_t n_out, const float *x, int64_t N, float *y);
/* the same through the parity-mode kernel (k_mul_mat_ref): the per-block fp32 terms in the CPU oracle's order -- results bit-identical to oracle/refcpu.c */
MINIGPT4_API int minigpt4_amd_test_mul_mat_ref(int ggml_type, const void *raw_w, int64_t n_in, int64_t n_out, const float *x, int64_t N, float *y);
/* prefill launch as the engine issues it for N > 4 rows: n_mat (1..3) equally shaped k-quant matrices (raw blocks back to back) against N rows in ONE launch of the LDS-staged
 * int8-MFMA kernels; residual ([n_mat][N][n_out]) optional; ks > 1 forces that K split (0 = the launcher's choice).  y: [n_mat][N][n_out].  4 = shape refused. */
/* prefill mat-mul micro-benchmark: n_mat random matrices [rows][cols] against N random rows, average microseconds per (set) launch; generation 2 = mmq2_kernels.hip, 1 = round-1 kernels */
MINIGPT4_API int minigpt4_amd_bench_mmq(int ggml_type, int rows, int cols, int n_mat, int N, int iters, int ks, int generation, float *us_per_launch);
MINIGPT4_API int minigpt4_amd_test_mmq2(int ggml_type, const void *raw_w, int n_mat, int64_t n_in, int64_t n_out, const float *x, int64_t N, const float *residual, int ks, int generation, float *y);
/* The decode (N = 1) mat-vec launches as the engine issues them: n1 equally spaced matrices of type1 (raw1 = their file bytes back to back), optionally one more of
 * type2 in the same mixed-type launch; prep 1 = rms_norm(x) * x2, 2 = x, 3 = silu(x) * x2, run standalone (fuse = 0) or in the kernel prologue (fuse = 1);
 * epi = 1: y[g] = silu(W0[g] . a) * (W1[g] . a) (n1 == 2).  residual / y: (n1 + n2) * n_out floats.  Returns 4 when the shape is outside the kernel's range. */
MINIGPT4_API int minigpt4_amd_test_matvec(int type1, const void *raw1, int n1, int type2, const void *raw2, int n2, int64_t n_in, int64_t n_out, const float *x, const float *x2,
                                          int prep, int fuse, int epi, const float *residual, float *y);
/* The batched-decode mat-vec: N = 1..4 activation rows x[N][n_in] against n_mat (1..3) equally spaced matrices in one weight pass; y / residual: [n_mat][N][n_out].
 * Returns 4 when the shape / type is outside the kernel's range. */
MINIGPT4_API int minigpt4_amd_test_matvec_rows(int ggml_type, const void *raw_w, int n_mat, int64_t n_in, int64_t n_out, const float *x, int N, const float *residual, float *y);
/* Activation quantisation (optionally after rms_norm with weight w): returns Q8_K and Q8_0 images of x[N][K].
 * q8k: int8[N*K], dk: float[N*K/256], bsums: int16[N*K/16], q80: int8[N*K], d0: float[N*K/32] (fp16-rounded). Any may be NULL. */
MINIGPT4_API int minigpt4_amd_test_quantize(const float *x, const float *rms_w, int64_t N, int64_t K, int8_t *q8k, float *dk, int16_t *bsums, int8_t *q80, float *d0);
/* C[M][N] = A[M][K] . W[N][K]^T on the MFMA f16 path (inputs given as fp32, rounded to fp16 on the device) + optional bias/GELU */
MINIGPT4_API int minigpt4_amd_test_gemm_f16(const float *A, const float *W, const float *bias, int M, int N, int K, int gelu, float *C);

/* Micro-benchmark of the decode mat-vec kernels on synthetic weight planes (see bench_kernels.py). variant 0: one launch per matrix, 1: fused persistent-wave launch,
   2: the same with the rms-norm prologue, 3 / 4: the batched step's multi-row launch with 4 / 2 prepared rows, 5 / 6: 2 / 4 rows prepared inside the launch */
MINIGPT4_API int minigpt4_amd_bench_matvec(int ggml_type, int rows, int cols, int n_mat, int variant, int iters, int n_sets, int waves_per_cu, float *us_per_launch, double *bytes_per_launch);

/* Average latency (microseconds) of a device-wide barrier across n_blocks co-resident 512-thread workgroups (atomic counter + agent-scope fences); *errors
 * counts visibility failures of a neighbour-word check.  Measurement for DESIGN.md's launch-gap-vs-barrier analysis. */
MINIGPT4_API float minigpt4_amd_probe_grid_barrier(int n_blocks, int iters, unsigned *errors);
/* vector-ALU issue probe: ns per instruction and wave for one instruction kind (0 v_and, 1 v_dot4c_i32_i8, 2 v_mul_lo_u32, 3 v_mad_i32_i24, 4 v_fma_f32, 5 v_and_or,
   6 v_bfe_u32, 7 v_cvt_f32_i32, 8 v_dot4_i32_i8, 9 v_mad_u64_u32, 10 v_lshrrev) at 1..4 waves per SIMD; < 0 without a GPU */
MINIGPT4_API float minigpt4_amd_probe_valu(int op, int waves_per_simd, int iters);

/* ---- host-only logic (no GPU needed) -------------------------------------------------------------------------------- */
struct MiniGPT4Vocab;
MINIGPT4_API struct MiniGPT4Vocab *minigpt4_amd_vocab_load(const char *llm_path);            /* parses hparams + vocab of a GGJT v3 file */
MINIGPT4_API void minigpt4_amd_vocab_free(struct MiniGPT4Vocab *v);
MINIGPT4_API int minigpt4_amd_vocab_size(struct MiniGPT4Vocab *v);
MINIGPT4_API const char *minigpt4_amd_vocab_piece(struct MiniGPT4Vocab *v, int id, int *len);
MINIGPT4_API int minigpt4_amd_vocab_tokenize(struct MiniGPT4Vocab *v, const char *text, int add_bos, int32_t *out, int cap);
/* Parses both files without touching a GPU.  Returns a MiniGPT4Error; fills counts when non-NULL. */
MINIGPT4_API int minigpt4_amd_inspect_files(const char *vision_path, const char *llm_path, int *n_vision_tensors, int *n_llm_tensors, int64_t *llm_weight_bytes_per_token);
/* Image file decoding from memory (the host half of minigpt4_image_load_from_file): PNG / JPEG / BMP / binary PNM bytes -> U8 HWC RGB with
 * cv::imread(IMREAD_COLOR)+BGR2RGB semantics.  The library allocates image->data; release with minigpt4_free_image.  0 or 5 (OpenImage). */
MINIGPT4_API int minigpt4_amd_decode_image(const void *bytes, size_t n, OUT struct MiniGPT4Image *image);
/* Pillow's 8-bit bicubic resample tables for in_size -> out_size (what the preprocess kernels consume): first/count: int[out_size],
 * kk: int[out_size * ksize] (22-bit fixed point).  Call with kk = NULL to learn ksize.  0, -1 (bad sizes) or -2 (kk_cap too small). */
MINIGPT4_API int minigpt4_amd_resample_coeffs(int in_size, int out_size, int *ksize, int *first, int *count, int *kk, size_t kk_cap);
/* Digest (FNV-1a 64) of what the engine takes from an LLM file -- hyper-parameters, vocabulary, every tensor's name / type / shape (and bytes when
 * with_data != 0).  GGJT v3 and GGUF v2 / v3 files of the same model digest equally.  Returns a MiniGPT4Error. */
MINIGPT4_API int minigpt4_amd_llm_file_digest(const char *llm_path, uint64_t *digest, int with_data);
/* ggml's reference block quantisers as minigpt4_quantize_model applies them (ggml_quantize_chunk): n floats (a whole number of blocks) -> dst; returns the
 * bytes written, 0 for an unsupported type (supported: Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q4_K Q5_K Q6_K, ggml type ids) or a ragged n. */
MINIGPT4_API int64_t minigpt4_amd_quantize_chunk(int ggml_type, const float *x, void *dst, int64_t n);
/* Diagnostic builds only (-DMG4_TIMELINE): 8 x uint64 constant-clock (100 MHz) stamps per workgroup of the LAST decode mat-vec launch -- entry, first weight
 * request, activation row ready, first row group done, last row group done, results stored.  Returns the workgroups copied, 0 for a normal build, -1 on error. */
MINIGPT4_API int minigpt4_amd_timeline(unsigned long long *out, int max_workgroups);
/* load-time re-encoding of Q3_K super-blocks (110 B) as value-identical Q6_K super-blocks (210 B); host only.  0 / 1 */
MINIGPT4_API int minigpt4_amd_convert_q3k_q6k(const void *src, void *dst, int64_t n_blocks);
/* Host sampler with an explicit seed (fresh std::mt19937 per call). */
MINIGPT4_API int minigpt4_amd_sample_logits(const float *logits, int n_vocab, int seed, float temp, int32_t top_k, float top_p, float tfs_z, float typical_p,
                                            int mirostat, float mirostat_tau, float mirostat_eta);

#ifdef __cplusplus
}
#endif
