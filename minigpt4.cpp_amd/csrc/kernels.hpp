// Host-callable launchers of the gfx950 kernels (llm_kernels.hip, vision_kernels.hip).
#pragma once
#include "common.hpp"

namespace mg4 {

struct Tables {               // fp16 lookup tables, 65536 entries each, indexed by the fp16 bit pattern of the argument
    const __half *gelu = nullptr, *silu = nullptr, *exp = nullptr;
    int exp_neg_n = 0;         // entries of exp[0x8000 ...] (arguments -0, -2^-24, ...) up to and including the last nonzero finite one, rounded up to a multiple of 2048:
                               // the part of the table k_attn_vit keeps in LDS (softmax arguments are <= 0)
};

// ---- load-time repack: raw ggml blocks (device copy of the file bytes) -> planes of a QWeight -------------------
// `dst` planes must already point into the arena (see plan_qweight).  raw/dst may not alias.
size_t plan_qweight(int type, int rows, int cols, QWeight &w, uint8_t *base);  // assigns plane pointers from `base`, returns bytes used
void launch_repack(const uint8_t *raw, const QWeight &w, hipStream_t s);
bool qweight_supported(int type);

// ---- activations -------------------------------------------------------------------------------------------------
// y = rms_norm(x) * w  (w == nullptr: y = x), quantised into the formats in `mask`; N rows of width K.
void launch_rms_quant(const float *x, const float *w, int N, int K, const ActQ &A, int mask, hipStream_t s, bool sequential_sum = false);   // sequential_sum: the sum of squares in element order by one thread (MINIGPT4_PARITY)
// y = silu(a) * b (fp16-table silu), quantised; or y = a when b == nullptr.
void launch_silu_mul_quant(const float *a, const float *b, int N, int K, const ActQ &A, int mask, const Tables &tb, hipStream_t s);

// ---- quantised mat-mul: y[t][r] = W[r] . act[t]  (+ residual[t][r]) ------------------------------------------------
void launch_mul_mat(const QWeight &W, const ActQ &A, int N, float *y, int ldy, const float *residual, hipStream_t s);

// MINIGPT4_PARITY=1: the same unit traits, the per-block fp32 terms added in the CPU oracle's order (one sequential chain per output) -- bit-identical to oracle/refcpu.c, slow
void launch_mul_mat_ref(const QWeight &W, const ActQ &A, int N, float *y, int ldy, const float *residual, hipStream_t s);
bool launch_mul_mat_ref_set(const QWeight *const *W, float *const *y, const float *const *res, int n, const ActQ &A, hipStream_t s);   // one row (decode) x 1..3 same-shape matrices: every unit of a row requested up front; false -> use launch_mul_mat_ref
bool matvec_prologue_supported(int type, int cols);   // the fused row-preparation variants of the decode mat-vec
void set_mmq_enabled(int v);
int mmq_enabled();
// second-generation prefill kernels (mmq2_kernels.hip): 1..3 same-type, same-shape k-quant matrices against N >= 1 prepared rows (A.bsq required), weights streamed once per
// chunk of <= 128 tokens, activations staged through LDS, optional K split (partial sums in A.ws) combined in fixed order.  false -> outside the kernels' range, nothing launched.
bool mmq2_supported(int type, int rows, int cols);
// A split-K mat-mul whose combine step is left to its consumer (round 3): matrix m's ks slabs of `stride` floats start m * ks * stride floats into ws; the value of an
// element is (slab_0 + ... + slab_{ks-1}) (+ res[m]) and belongs at y[m].  ks <= 1: nothing pending.
struct SlabSrc { const float *ws = nullptr; int ks = 0; long long stride = 0; int n = 0; float *y[3] = {nullptr, nullptr, nullptr}; const float *res[3] = {nullptr, nullptr, nullptr};
                 // general form (results of TWO launches pending together, e.g. wq|wk and a differently typed wv): matrix m = sum of mks[m] slabs from mbase[m]; mks[m] == 1 with
                 // mbase[m] == y[m] is a matrix that is already in place.  Set by the deferring launches as mbase[m] = ws + m * ks * stride, mks[m] = ks.
                 const float *mbase[3] = {nullptr, nullptr, nullptr}; int mks[3] = {0, 0, 0}; bool mixed = false; };
void launch_slab_flush(const SlabSrc &src, hipStream_t s);                                      // the combine as its own launch (k_mmq2_reduce_set)
void launch_rms_quant_slabs(const SlabSrc &src, const float *w, int N, int K, const ActQ &A, int mask, hipStream_t s);       // n == 1: y[0] = res[0] + slabs, then rms norm * w, quantised
void launch_silu_mul_quant_slabs(const SlabSrc &src, int N, int K, const ActQ &A, int mask, const Tables &tb, hipStream_t s);   // n == 2: silu(a) * b, quantised
void launch_rope_kv_slabs(const SlabSrc &src, int N, int n_head, int hd, const int *n_past, const float *cos_tab, const float *sin_tab, __half *kcache, __half *vcache, hipStream_t s);   // n == 3
// defer != nullptr: with a K split the combine launch is skipped and *defer describes the slabs (ks > 1); otherwise *defer is cleared
bool launch_mmq2_set(const QWeight *const *W, float *const *y, const float *const *residual, int n, const ActQ &A, int N, int ldy, hipStream_t s, SlabSrc *defer = nullptr,
                     int force_ks = 0);   // force_ks == 1: no K split = the CPU oracle's fp32 order (parity mode's prompt rows)
void set_mmq2_cus(int cus);
void set_mmqh(int v);   // 1: Q4_K / Q5_K prompt rows on the fp16 matrix cores with the sub-block scales folded into the weight operands (k_mmqh_q45k; measured slower, profiles/r05_prefill_fp16_scaled_operands.md); 0 (default): the int8 kernels
int mmqh_enabled();
// (test library only: measured in round 4, not adopted) prompt mat-mul on load-time digit planes (mmq3_kernels.hip): Q4_K / Q5_K, sub-block scale x quant stored as 128 hi + lo (1.5 B per weight) in MFMA-fragment order
bool mmq3_supported(int type, int rows, int cols);
size_t mmq3_plane_bytes(int rows, int cols, size_t *hi_off = nullptr);
void launch_mmq3_build(const QWeight &W, uint8_t *planes, hipStream_t s);
bool launch_mmq3_set(const QWeight *const *W, const uint8_t *const *planes, float *const *y, const float *const *residual, int n, const ActQ &A, int N, int ldy, hipStream_t s, SlabSrc *defer = nullptr);
int probe_tn_mfma(int rows, int cols, int TN, int iters, int n_sets, int check, int cus, float *us_per_launch, float *rel_diff);   // (test library with `make test-extras` only) tn_mfma_probe.hip
void set_mmq3_waves(int nw);                          // 8 (default): 128-row workgroups, one per CU; 4: 64-row workgroups, two per CU (slower)
void set_mmq3_tuning(int cus, int ks);                // CU count (<= 0: leave), forced K split (0 = the launcher's choice, < 0: leave)
void set_mmq2_tuning(int tt, int fill_pct, int ks);   // experiment knobs, 0 = the launcher's choice, < 0 = leave as it is (read from the environment once, by Engine::init)
void set_gemm_splitk_xcd(int on);   // 1 (default): a split-K GEMM's work list [slice][tile] is dealt to the XCDs in contiguous ranges; 0: slices in grid.z (rounds 3-5, A/B)
void set_gemm_tuning(int big_min_m, int f16_ks, int arm = -1, int sk_arm = -1);      // smallest M of the 128x128 GEMM (< 0: leave), forced K split of the F16 set launches (0 = choose)
void set_f16_gemm(int v);                             // F16 language-model weights at >= 16 rows on the MFMA GEMM (default 1)
void launch_slab_reduce(const float *slabs, int n_slabs, long long slab_stride, const float *residual, float *y, size_t n, hipStream_t s);   // y = (residual +) sum_z slab_z, fixed order
// F16 weights at prompt sizes: 1..3 equally spaced [N][K] matrices x M fp16 rows in one launch of the 128x128 LDS-DMA GEMM (split K when the tiles do not fill the chip);
// false -> outside this path, nothing launched
bool launch_gemm_f16_silu_pair(const __half *A, int lda, const __half *W1, const __half *W3, int M, int N, int K, const Tables &tb, float *out, __half *out_h, int ldo, int cus, hipStream_t s);   // fp16(silu(A.W1^T) * (A.W3^T)) in one launch (F16 w1|w3 at prompt sizes)
bool launch_gemm_f16_set(const __half *A, int lda, const __half *const *W, int n, int M, int N, int K, float *const *y, const float *const *residual, int ldo, float *ws,
                         size_t ws_floats, int cus, hipStream_t s, SlabSrc *defer = nullptr);

// decode (N = 1) persistent-wave mat-vec over 1..3 same-type, same-shape matrices (wq|wk|wv, w1|w3); false -> caller falls back to launch_mul_mat
// pro: 0 = activations come from `A` (prepared by launch_rms_quant / launch_silu_mul_quant); 1 = rms_norm(px) * pw, 2 = px, 3 = silu(px) * pw are
// prepared and quantised inside the kernel prologue (one launch less per use).
bool launch_matvec_set(const QWeight *const *W, float *const *y, const float *const *residual, int n, const ActQ &A, hipStream_t s, int pro = 0, const float *px = nullptr,
                       const float *pw = nullptr, const Tables *tb = nullptr, int epi = 0);   // epi 1: y[0][g] = silu(W0[g].x) * (W1[g].x) (n == 2); MATVEC_EPI_REF: the CPU oracle's fp32 order (k-quants; pro 0..2)
constexpr int MATVEC_EPI_REF = 2;
bool matvec_silu_pair_supported(int type, int cols);
// batched decode: N = 1..4 activation rows (prepared in `A`) against 1..3 same-type, same-shape, equally spaced matrices, weights streamed once;
// y[m][t * ldy + r] (+ residual[m][t * ldy + r]).  false -> outside the kernel's range, use launch_mul_mat.
bool launch_matvec_rows(const QWeight *const *W, float *const *y, const float *const *residual, int n, const ActQ &A, int N, int ldy, hipStream_t s, const float *px = nullptr,
                        const float *pw = nullptr, int ldx = 0);   // px != null: rows t of x (stride ldx) are prepared inside the launch (A unused): pw != null rms-normed with pw, pw == null taken as they are; then quantised
bool matvec_rows_prologue_ok(int type, int K);
// batched decode on v_mfma_i32_4x4x4_16B_i8 over the row-interleaved image (ri_kernels.hip, round 5): N = 1..4 PREPARED rows (Q8_K image incl. bsk / bsq) against 1..3 same-type,
// same-shape Q4_K / Q5_K / Q6_K matrices; false -> outside the kernel's range, nothing launched
bool ri_supported(int type, int rows, int cols);
size_t ri_plan(int type, int rows, int cols, RiPlanes &p, uint8_t *base);      // assigns the image's plane pointers from `base` (nullptr: sizes only), returns the bytes used
void launch_ri_build(const QWeight &w, const RiPlanes &p, hipStream_t s);      // ordinary planes -> row-interleaved image
// workspace of the K-split form (few row groups, long K: the 13B w2): slabs for the workgroups' partial sums and one arrival ticket per row group, ZEROED by the owner (the kernel
// leaves them zero); without it (all null) such sets run one workgroup per group.  Owned by the caller -- one per context (Engine::ri_ws_), passed with every launch: no
// process-global state, so two contexts decoding on different threads / streams never share slabs or tickets (round-5 advisor).
struct RiWorkspace { float *slabs = nullptr; size_t slab_floats = 0; unsigned *tickets = nullptr; int n_tickets = 0; };
bool launch_matvec_ri(const QWeight *const *W, const RiPlanes *const *ri, float *const *y, const float *const *residual, int n, const ActQ &A, int N, int ldy, hipStream_t s,
                      const float *px = nullptr, const float *pw = nullptr, int ldx = 0, const RiWorkspace &ws = RiWorkspace{});   // px != null: rows t of px (stride ldx) are rms-normed with pw and quantised inside the launch (A unused)
// a "more bits" layer's set in one launch: na matrices of Q4_K / Q5_K + nb of Q6_K (na + nb <= 3), same shape, prepared rows
bool launch_matvec_ri_mixed(const QWeight *const *Wa, const RiPlanes *const *ria, float *const *ya, int na, const QWeight *const *Wb, const RiPlanes *const *rib, float *const *yb, int nb,
                            const ActQ &A, int N, int ldy, hipStream_t s);
void set_ri_cus(int cus);
int ri_ksplit(int total_groups, int K, const RiWorkspace &ws);   // workgroups per row group the launch would use with this workspace (1 = no K split)
// two k-quant types (Q4_K|Q5_K + Q6_K) with the same K in one launch; pro: 0 or 1 (rms_norm prologue)
bool launch_matvec_mixed(const QWeight *const *W1, float *const *y1, int n1, const QWeight *const *W2, float *const *y2, int n2, const ActQ &A, hipStream_t s, int pro = 0,
                         const float *px = nullptr, const float *pw = nullptr, int epi = 0);
// the same for N = 1..4 rows of the batched step (K <= 6144); px != null: rows rms-normed with pw and quantised inside the launch
bool launch_matvec_rows_mixed(const QWeight *const *W1, float *const *y1, int n1, const QWeight *const *W2, float *const *y2, int n2, const ActQ &A, int N, int ldy, hipStream_t s,
                              const float *px = nullptr, const float *pw = nullptr, int ldx = 0);
void set_matvec_tuning(int waves_per_cu, int fat_threads, int cus);   // 0 = choose per launch
// measurement: while tracing is on, every launcher of llm_kernels.hip notes the kernel symbol it launched (as rocprofv3 prints it, without the argument list)
void kernel_name_tracing(bool on);
const char *last_kernel_name();          // "" when nothing was launched since the last reset
void reset_kernel_name();
size_t launch_probe_count();                       // launches noted (and probed with their own start / stop events) since tracing was switched on
float launch_probe_us(size_t first, size_t last);  // sum of the dispatch durations of probes [first, last) in microseconds (stream synchronised); < 0: not available
int read_matvec_timeline(unsigned long long *out, int max_workgroups);   // diagnostic builds (MG4_TIMELINE): stamps of the last decode mat-vec launch; 0 otherwise
int read_attn_timeline(unsigned long long *out, int max_workgroups);     // prompt attention (8 x u64 per workgroup, up to 2048 workgroups)
int read_vision_timeline(unsigned long long *out, int max_workgroups);   // the same for the image path's kernels (32 x u64 per workgroup)

// ---- token embedding gather (raw ggml rows, dequantised to f32) -----------------------------------------------------
void launch_get_rows(int type, const uint8_t *raw_table, int K, const int *tokens, int N, float *out, hipStream_t s);

// ---- RoPE + KV append, attention, argmax ----------------------------------------------------------------------------
// q,k,v: [N][E] f32.  Rotates q in place, writes rotated k and v as fp16 into the caches at position *n_past + t.
void launch_rope_kv(float *q, const float *k, const float *v, int N, int n_head, int hd, const int *n_past, const float *cos_tab,
                    const float *sin_tab, __half *kcache, __half *vcache, hipStream_t s);
// out[t][h*hd+i] = softmax(K q / sqrt(hd)) V over keys 0..*n_past+t.  caches: [n_ctx][E] fp16.
// fused (decode, N == 1): q,k,v are the raw projections; RoPE of q/k and the KV append happen inside the kernel.
// !fused: launch_rope_kv must have run (q rotated in place, caches appended).
void launch_attn_llm(float *q, const float *k, const float *v, __half *kcache, __half *vcache, int N, int n_head, int hd, const int *n_past, int n_ctx,
                     const float *cos_tab, const float *sin_tab, const Tables &tb, float *out, bool fused, hipStream_t s);
// key-split decode attention for long contexts (one row; RoPE + KV append fused): every head's keys are shared by `splits` workgroups in two launches (scores; softmax + P.V
// + deterministic combine by the last-arriving workgroup of the head).  ws: attn_split_workspace_bytes() bytes of device memory whose LAST 1 KiB + n_head words (the arrival
// counters) were zeroed once; the kernels leave them zero.
size_t attn_split_workspace_bytes(int n_head, int hd, int n_ctx, int splits);
int attn_split_count(int n_head, int cus);
void launch_attn_llm_split(float *q, const float *k, const float *v, __half *kcache, __half *vcache, int n_head, int hd, const int *n_past, int n_ctx, const float *cos_tab,
                           const float *sin_tab, const Tables &tb, float *out, void *ws, int splits, hipStream_t s);
// decode of B different conversations in one pass (RoPE + KV append fused): row t -> conversation row_slot[t] at position n_past[row_slot[t]], caches at
// kcache / vcache + row_slot[t] * seq_stride elements.
void launch_attn_llm_batched(float *q, const float *k, const float *v, __half *kcache, __half *vcache, int B, int n_head, int hd, const int *n_past, const int *row_slot,
                             size_t seq_stride, int n_ctx, const float *cos_tab, const float *sin_tab, const Tables &tb, float *out, hipStream_t s);
// per row r, slot = row_slot[r]: argmax[slot] = feed[slot] = argmax(logits[r]); slot_logits[slot] = logits[r]; n_past[slot] += 1
void launch_batch_finish(const float *logits, int n_vocab, int B, const int *row_slot, int *n_past, int *argmax, int *feed, float *slot_logits, hipStream_t s);
void launch_batch_begin(int *n_past, const int *row_slot, const int *row_pos, int B, hipStream_t s);   // n_past[row_slot[r]] = row_pos[r]
// prefill (N > 1 rows of one conversation, after launch_rope_kv): workgroup = (head, 16 queries), keys streamed through LDS in tiles, exact-f32 MFMA; t_max >= *n_past + N
// sizes the LDS score rows; false -> does not fit (the caller uses launch_attn_llm)
// out_h (optional): a kernel that can do so stores the fp16-rounded rows THERE instead of fp32 rows in `out` and sets *wrote_h (the F16 wo's input rows)
bool launch_attn_prefill(const float *q, const __half *kcache, const __half *vcache, int N, int n_head, int hd, const int *n_past, int t_max, const Tables &tb, float *out, hipStream_t s, __half *out_h = nullptr, bool *wrote_h = nullptr);
// MINIGPT4_PARITY=1: scores / softmax / P.V with every fp32 chain in the oracle's order (after launch_rope_kv); t_max >= *n_past + N
void launch_attn_ref(const float *q, const __half *kcache, const __half *vcache, int N, int n_head, int hd, const int *n_past, int t_max, const Tables &tb, float *out, hipStream_t s);
void launch_attn_ref_fused(const float *q, const float *k, const float *v, __half *kcache, __half *vcache, int n_head, int hd, const int *n_past, int t_max, const float *cos_tab, const float *sin_tab,
                           const Tables &tb, float *out, hipStream_t s);   // one query row: RoPE + cache append inside
void set_attn_prefill_f16(int v);
void set_attn_prefill_w8(int v);    // 8-wave loader / MFMA form of the fp16 prompt attention (MINIGPT4_ATTN_PREFILL_W8)   // 1 (default): prompt attention on the fp16 matrix cores, 0: the exact-f32 MFMA kernel
bool attn_head_size_supported(int hd);
void attn_ref_prepare();        // the oracle-order attention kernels' > 64 KiB LDS opt-in on the current device, checked (throws HipError); outside any stream capture
int attn_ref_max_ctx(int hd);   // the same bound for the oracle-order kernel of parity mode (smaller: it also stages value rows in LDS)
int attn_max_ctx(int hd);   // largest n_ctx whose score / probability rows fit the attention kernel's LDS
void launch_argmax(const float *logits, int n, int *out, void *scratch /*>= 512 bytes*/, hipStream_t s);
void launch_add_inplace(float *x, const float *y, size_t n, hipStream_t s);
void launch_set_int(int *p, int v, hipStream_t s);
void launch_stall(unsigned ms, hipStream_t s);   // test injection: a bounded (<= 5 s) busy kernel
uint64_t device_checksum(const void *p, size_t bytes, hipStream_t s);   // sum of the 32-bit words (bytes rounded down to 4), mod 2^64; synchronises the stream
void launch_fill_u16(void *p, size_t n, unsigned short v, hipStream_t s);
void launch_advance(int *n_past, int n, int *tok0, const int *argmax, hipStream_t s);
void launch_delay(int us, hipStream_t s);   // profiling gate: keeps the stream busy for `us` microseconds (see Engine::profile_sites)

// ---- vision tower -----------------------------------------------------------------------------------------------------
// C[M][N] = A[M][K](f16) . W[N][K](f16)^T + bias ; optional fp16-table GELU ; optional residual add (C = residual + C).
// Writes fp32 `out` (nullable) and/or fp16 `out_h` (nullable).  K must be a multiple of 16.
void launch_gemm_f16(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual,
                     bool gelu, const Tables &tb, float *out, __half *out_h, int ldo, hipStream_t s);
// fp32 output stored [N / D][D / hd][M][hd] (the ViT's qkv projection for k_attn_vit's head strides; vision_kernels.hip: struct HeadMajor)
void launch_gemm_f16_head_major(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const Tables &tb, float *out, int D, int hd, hipStream_t s);
// out = [residual +] gelu?(bias + y) over [rows][n] contiguous fp32 (y may alias out); writes fp32 (nullable) and / or fp16 (nullable)
// the same product for FEW rows (the Q-Former's 32 queries per image): N / 16 workgroups, K split across the waves of a workgroup, no LDS staging.  false -> shape outside
// the kernel's range (N % 16, K % 32), nothing launched
bool launch_gemm_f16_skinny(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                            float *out, __half *out_h, int ldo, hipStream_t s);
bool launch_gemm_f16_skinny_splitk(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, int slices, float *slabs, size_t slab_stride, int ldo, hipStream_t s);   // raw partial sums per K slice -> launch_splitk_reduce_ln
void launch_lin_epilogue(const float *y, const float *bias, const float *residual, bool gelu, const Tables &tb, int rows, int n, float *out, __half *out_h, hipStream_t s);
// LayerNorm (ggml_norm eps 1e-5, then w*x+b); rows x n; writes fp32 (nullable) and fp16 (nullable).
void launch_layernorm(const float *x, const float *w, const float *b, int rows, int n, float *out, __half *out_h, hipStream_t s, bool sequential_sums = false);   // sequential_sums: MINIGPT4_PARITY
// MINIGPT4_PARITY: ViT / BERT attention with every fp32 chain in the oracle's order (fp32 output); same argument meaning as launch_attn_f32
void launch_attn_vref(const float *q, int ldq, const float *k, const float *v, int ldk, int nq, int nk, int heads, int hd, float q_prescale, float score_div, const Tables &tb,
                      float *out, int ldo, hipStream_t s, int batch = 1);
// split-K GEMM (raw fp32 partial sums into `slices` slabs) + the deterministic reduce fused with bias / residual / the following LayerNorm
bool launch_gemm_f16_arm(int arm, const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                         float *out, __half *out_h, int ldo, hipStream_t s);   // micro-benchmark arms (other tile shapes of k_gemm_f16)
bool launch_gemm_f16_splitk_arm(int arm, const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, int slices, float *slabs, size_t slab_stride, int ldo, hipStream_t s);
int gemm_split_slices(int K, int want);   // slices actually used for a K (whole 128-wide k tiles per slice)
void launch_gemm_f16_splitk(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, int slices, float *slabs, size_t slab_stride, int ldo, hipStream_t s);
void launch_splitk_reduce_ln(const float *slabs, int n_slabs, size_t slab_stride, const float *bias, const float *residual, int rows, int n, float *x_out, const float *ln_w,
                             const float *ln_b, float *ln_out, __half *ln_out_h, hipStream_t s);
// f32 attention: q[nq][ldq], k/v[nk][ldk]; per head h the slice [h*hd, (h+1)*hd).  q_prescale != 0: q *= q_prescale first (ViT);
// score_div != 0: scores /= score_div (BERT).  Output fp32 (nullable) / fp16 (nullable) [nq][ldo].
// batch > 1: image z uses rows [z * nq, (z + 1) * nq) of q / out and rows [z * nk, (z + 1) * nk) of k / v.
void set_attn_vit_qt(int qt);   // query tiles per workgroup of k_attn_vit (0 = the launcher's choice); experiments / A-B
void launch_attn_f32(const float *q, int ldq, const float *k, const float *v, int ldk, int nq, int nk, int heads, int hd, float q_prescale,
                     float score_div, const Tables &tb, float *out, __half *out_h, int ldo, hipStream_t s, int batch = 1, int head_stride_q = 0, int head_stride_kv = 0)   /* head strides in floats; 0 = hd (heads side by side in a row) */;
// image CHW f32 [3][224][224] -> fp16 patches [256][ldp] (k = c*196 + kh*14 + kw, zero padded to ldp)
void launch_im2col(const float *image, __half *patches, int ldp, hipStream_t s, int batch = 1);   // batch images back to back -> [batch * 256][ldp]
// x[0] = cls + pos[0]; x[1+p] = pe[p] + pos[1+p]
void launch_assemble_embeddings(const float *cls, const float *pe, const float *pos, int D, float *x, hipStream_t s, int batch = 1);
void launch_f32_to_f16(const float *x, __half *y, size_t n, hipStream_t s);

// ---- image preprocess (image_kernels.hip): Pillow's 8-bit bicubic resample, horizontal then vertical pass, all pointers device ---------------
// dst[y][xx][c] = clip8(2^21 + sum_i src[y][first[xx] + i][c] * kk[xx * ksize + i] >> 22); src [H][W][3], dst [H][OW][3]
void launch_resample_h(const uint8_t *src, int W, int H, const int *first, const int *count, const int *kk, int ksize, uint8_t *dst, int OW, hipStream_t s);
// vertical pass over src [*][OW][3] fused with x * (1/255), (x - mean) / std and HWC -> CHW: out [3][OH][OW] f32
void launch_resample_v_norm(const uint8_t *src, int OW, const int *first, const int *count, const int *kk, int ksize, float *out, int OH, const float mean[3], const float std[3],
                            hipStream_t s);

}  // namespace mg4
