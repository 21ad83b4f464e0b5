/*
 * minigpt4_amd_test.h -- entry points of libminigpt4_test.so ONLY (csrc/test_hooks.cpp + csrc/probe_kernels.hip linked with the product's objects).
 *
 * Single-kernel parity hooks, micro-benchmarks, hardware probes and host-only helpers for the CPU test tier.  None of these symbols is exported by the product
 * library libminigpt4.so (tests/test_cpu_host.py checks both export lists).  Plain C types only.
 */
#pragma once
#include "minigpt4_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

MINIGPT4_API int minigpt4_amd_copy_arenas(struct MiniGPT4Context *dst, struct MiniGPT4Context *src);   /* tests: device-to-device stand-in for the broadcast on one GPU */

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
 * epi = 1: y[g] = silu(W0[g] . a) * (W1[g] . a) (n1 == 2); epi = 2: every fp32 accumulation in the CPU oracle's order (k-quants; bit-identical to oracle/refcpu.c).  residual / y: (n1 + n2) * n_out floats.  Returns 4 when the shape is outside the kernel's range. */
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
/* the same through the skinny-M kernel of the Q-Former (k_gemm_f16_skinny: N % 16 == 0, K % 32 == 0; 4 otherwise) */
MINIGPT4_API int minigpt4_amd_test_gemm_f16_skinny(const float *A, const float *W, const float *bias, int M, int N, int K, int gelu, float *C);

/* Micro-benchmark of the image path's fp16 GEMM on synthetic operands (tools/timeline_gemm.py). flags: 1 GELU, 2 residual, 4 fp16 output too; variant 0 = the dispatcher
 * (launch_gemm_f16), 1 = the skinny-M kernel, 2 + (slices << 8) = split K + k_splitk_reduce_ln; n_sets weight matrices are cycled (no Infinity-Cache repeats) */
MINIGPT4_API int minigpt4_amd_bench_gemm_f16(int M, int N, int K, int flags, int variant, int iters, int n_sets, float *us_per_launch);
/* Micro-benchmark of the ViT / Q-Former attention kernel on synthetic rows (tools/timeline_attn.py) */
MINIGPT4_API int minigpt4_amd_bench_attn_f32(int heads, int hd, int nq, int nk, int iters, float *us_per_launch);
/* the same with `batch` images per launch, computed exponentials (the engine's fast mode) and `qt` query tiles per workgroup forced (0 = the launcher's choice); nq == nk */
MINIGPT4_API int minigpt4_amd_bench_attn_f32_b(int heads, int hd, int nq, int nk, int batch, int qt, int iters, float *us_per_launch);
/* query tiles per workgroup of the ViT / Q-Former attention kernel for every later launch of this process (0 = the launcher's choice); bit-identical results for every value */
MINIGPT4_API void minigpt4_amd_test_set_attn_qt(int qt);
/* the F16 feed-forward pair launch: out_h[N][n_out] = fp16(silu_table(w1 x) * (w3 x)) (uint16 bit patterns), w = w1 then w3 as fp16 [n_out][n_in]; 4 = shape outside the path */
MINIGPT4_API int minigpt4_amd_test_f16_silu_pair(const float *x, const void *w_f16, int64_t N, int64_t n_in, int64_t n_out, unsigned short *out_h, float *out_f);
/* Micro-benchmark of the prompt-row attention on a synthetic fp16 K / V cache (tools/timeline_attn_prefill.py); _timeline_attn: its stamps in a -DMG4_TIMELINE build */
MINIGPT4_API int minigpt4_amd_bench_attn_prefill(int n_head, int hd, int N, int n_past, int iters, float *us_per_launch);
MINIGPT4_API int minigpt4_amd_timeline_attn(unsigned long long *out, int max_workgroups);
/* force one tile shape (an "arm" of launch_gemm_f16_arm in vision_kernels.hip; 0 = the launcher's own choice) for every small-M GEMM / split-K GEMM of this process */
MINIGPT4_API void minigpt4_amd_test_set_gemm_arm(int arm, int sk_arm);
/* 0: split-K GEMM slices in grid.z (the rounds 3-5 workgroup -> XCD mapping); 1 (default): the [slice][tile] work list dealt to the XCDs in contiguous ranges */
MINIGPT4_API void minigpt4_amd_test_set_splitk_xcd(int on);
/* diagnostic builds (-DMG4_TIMELINE): the 32 clock stamps per workgroup of the last image-path GEMM launch; 0 = built without */
MINIGPT4_API int minigpt4_amd_timeline_vision(unsigned long long *out, int max_workgroups);
/* Micro-benchmark of the decode mat-vec kernels on synthetic weight planes (see bench_kernels.py). variant 0: one launch per matrix, 1: fused persistent-wave launch,
   2: the same with the rms-norm prologue, 3 / 4: the batched step's multi-row launch with 4 / 2 prepared rows, 5 / 6: 2 / 4 rows prepared inside the launch */
MINIGPT4_API int minigpt4_amd_bench_matvec(int ggml_type, int rows, int cols, int n_mat, int variant, int iters, int n_sets, int waves_per_cu, float *us_per_launch, double *bytes_per_launch);

/* Average latency (microseconds) of a device-wide barrier across n_blocks co-resident 512-thread workgroups (atomic counter + agent-scope fences); *errors
 * counts visibility failures of a neighbour-word check.  Measurement for DESIGN.md's launch-gap-vs-barrier analysis. */
/* batched decode on the int8 matrix cores (csrc/ri_kernels.hip): ordinary planes -> row-interleaved image -> N = 1..4 prepared rows against n_mat equally shaped Q4_K / Q5_K /
 * Q6_K matrices in one launch; y [n_mat][N][n_out]; returns 4 when the type / shape is outside the kernel's range */
MINIGPT4_API int minigpt4_amd_test_matvec_ri(int ggml_type, const void *raw_w, int n_mat, int64_t n_in, int64_t n_out, const float *x, int N, const float *residual, const float *rms_w /* non-null: rows rms-normed with it inside the launch */, float *y);
MINIGPT4_API int minigpt4_amd_test_matvec_ri_mixed(int type_a, const void *raw_a, int n_a, int type_b, const void *raw_b, int n_b, int64_t n_in, int64_t n_out, const float *x, int N, float *y);   /* one launch over n_a matrices of Q4_K / Q5_K + n_b of Q6_K (a "more bits" layer's wq | wk + wv) */
MINIGPT4_API int minigpt4_amd_bench_matvec_ri(int ggml_type, int rows, int cols, int n_mat, int N, int iters, int n_sets, float *us_per_launch);   /* us per launch of the same, synthetic planes, rotating sets */
/* 1 when the test library was built by `make test-extras` (closed-direction kernels included: the generation-3 prompt mat-mul, the batched-decode MFMA probe), else 0 */
MINIGPT4_API int minigpt4_amd_test_extras(void);
MINIGPT4_API float minigpt4_amd_probe_grid_barrier(int n_blocks, int iters, unsigned *errors);
/* round 5 (csrc/tn_mfma_probe.hip, tools/batched_mfma_probe.py): the batched decode mat-vec on v_mfma_i32_4x4x4_16B_i8 over row-interleaved SYNTHETIC Q5_K planes -- microseconds per
 * launch for one rows x cols matrix set against TN = 1..4 prepared rows; check != 0 validates the launch against a scalar kernel over the same planes (relative difference out) */
MINIGPT4_API int minigpt4_amd_probe_tn_mfma(int rows, int cols, int TN, int iters, int n_sets, int check, float *us_per_launch, float *rel_diff);
/* what csrc/dist.cpp reads from MINIGPT4_WORLD_SIZE / MINIGPT4_RANK / MINIGPT4_NCCL_ID_FILE / MINIGPT4_DIST_TIMEOUT_S (host only): 0, or 1 with the reason in err */
MINIGPT4_API int minigpt4_amd_dist_env(int *world, int *rank, char *id_file, size_t cap, char *err, size_t err_cap);
/* LDS-DMA stream probe (csrc/probe_kernels.hip, tools/probe_dma.py): chip-wide GB/s of `waves` loader waves per CU keeping `depth` fills of `fill` bytes in flight into an LDS
 * ring; form 0 scalar base + instruction offsets, 1 per-lane addresses, 2 register loads, 3 register loads + ds_write; policy 0 nt, 1 default; deal 0 blocked, 1 round-robin */
MINIGPT4_API float minigpt4_amd_probe_dma(int form, int policy, int waves, int fill, int depth, int deal, double total_gb);
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
