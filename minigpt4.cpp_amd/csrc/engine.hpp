// MI355X MiniGPT-4 engine: owns HBM arenas (weights repacked into planes, fp16 KV cache, activation buffers), the HIP stream
// and the decode hipGraph.  Mirrors the call surface of the reference's `class MiniGPT4` (minigpt4.cpp:1740-2522):
// init / encode_image / add_tokens / add_strings / add_embedding / sample_token / id_to_token / reset.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"
#include "formats.hpp"
#include "kernels.hpp"
#include "sampler.hpp"

namespace mg4 {

struct DeviceArena {
    uint8_t *base = nullptr;
    size_t cap = 0, used = 0;
    bool virt = false;                       // planning only (Engine::plan_arenas): no device memory, `base` is a fake non-null address that is never dereferenced
    uint64_t layout_hash = 1469598103934665603ull;   // FNV-1a over the (offset, bytes) sequence of take(): two loads with the same hash laid the arena out identically
    void alloc(size_t bytes);
    void release();
    uint8_t *take(size_t bytes, size_t align = 256);
    void abandon() { base = nullptr; cap = used = 0; }   // leak on purpose: hipFree synchronises with a stream that will never drain (Engine::stream_hung_)
    ~DeviceArena() { release(); }
};


class Engine {
public:
    Engine() = default;
    ~Engine();
    int init(const std::string &vision_path, const std::string &llm_path, int seed, int n_ctx, int n_batch);
    // ---- multi-GPU load: rank 0 loads the files (LOAD_FULL) and broadcasts its two weight arenas; the other ranks run LOAD_RECV:
    // parse the HEADERS only, lay the arenas out identically (same take() sequence -> same layout hash), allocate them, skip every file read / upload
    // / repack, receive the bytes, then call weights_received() for what is derived on the device (prefill planes, the replicated query tokens).
    // LOAD_PLAN does the layout without any device (CPU tests).
    enum LoadMode { LOAD_FULL = 0, LOAD_RECV = 1, LOAD_PLAN = 2 };
    int weights_received();
    struct ArenaPlan { size_t llm_bytes = 0, vision_bytes = 0; uint64_t llm_hash = 0, vision_hash = 0; };
    static int plan_arenas(const std::string &vision_path, const std::string &llm_path, ArenaPlan &out);   // host only
    ArenaPlan arena_plan() const { ArenaPlan p; p.llm_bytes = llm_arena_.used; p.vision_bytes = vis_arena_.used; p.llm_hash = llm_arena_.layout_hash; p.vision_hash = vis_arena_.layout_hash; return p; }
    LoadMode load_mode() const { return load_mode_; }
    // native broadcast (dist.hpp): what this context's load did -- world size, rank, milliseconds of the exchange (0 when the load was an ordinary
    // one)
    int dist_world() const { return dist_world_; }
    int dist_rank() const { return dist_rank_; }
    float dist_bcast_ms() const { return dist_bcast_ms_; }

    // ---- image path (reference encode_image, minigpt4.cpp:2094-2363)
    int encode_image(const float *chw, float *out);   // out: [32][proj_out()]
    int encode_images(const float *const *chw, int B, float *const *out);   // B <= VISION_BATCH_MAX images in one pass over the vision weights
    static constexpr int VISION_BATCH_MAX = 8;
    int proj_out() const { return v_out_; }
    int n_query() const { return v_nq_; }

    // ---- language path
    int add_tokens(const std::vector<int> &tokens, bool flush_now = false);   // queued; evaluated in chunks of n_batch (minigpt4.cpp:2365-2382)
    int flush();                                      // evaluate everything queued by add_tokens / add_embedding
    int add_string(const std::string &s);             // BOS + tokenize (minigpt4.cpp:2384-2397)
    int add_embedding(const float *data, int n_rows); // llama_eval_embd (minigpt4.cpp:2399-2422)
    int sample_token(const SampleParams &p);          // minigpt4.cpp:2425-2483
    const char *id_to_token(int id) const;            // minigpt4.cpp:2485-2497 (borrowed pointer)
    // minigpt4.cpp:2499-2502 (the selected conversation)
    void reset() { Conversation &c = conv_[(size_t)cur_]; c.pend_tok.clear(); c.pend_embd.clear(); c.n_past = 0; c.n_committed = 0; }
    void sync();
    hipStream_t stream() const { return stream_; }

    int n_vocab() const { return (int)llm_.n_vocab; }
    int n_embd() const { return (int)llm_.n_embd; }
    int n_past() const { return conv_[(size_t)cur_].n_past; }
    int n_ctx() const { return n_ctx_; }
    const float *logits_host();                       // syncs, copies the last logits to pinned memory
    const Tokenizer &tokenizer() const { return tok_; }
    size_t weight_bytes_per_token() const { return wbytes_token_; }
    size_t llm_arena_bytes() const { return llm_arena_.used; }
    size_t vision_arena_bytes() const { return vis_arena_.used; }
    uint8_t *llm_arena_ptr() { return llm_arena_.base; }
    uint8_t *vision_arena_ptr() { return vis_arena_.base; }

    // ---- several conversations per replica.  The reference holds ONE conversation per context (minigpt4.cpp:2513-2521); here a
    // context owns n >= 1 of them -- each with its own KV cache region, position and pending queue, sharing the weights -- so that a decode step of B
    // conversations streams the weights once instead of B times.  All reference entry points act on the selected conversation (0 by default).
    int set_conversations(int n);                      // (re)allocates the KV caches; every conversation is reset.  1 <= n <= MAX_CONVERSATIONS
    int select_conversation(int slot);
    int n_conversations() const { return (int)conv_.size(); }
    int current_conversation() const { return cur_; }
    // one decode step for `n` distinct conversations: sample each (like sample_token), evaluate the n sampled tokens in ONE weight pass. ids_out[i] =
    // the token sampled for slots[i]; a conversation whose context is full is sampled but not advanced (the reference discards the error too).
    // forced != null: the step evaluates forced[i] for slots[i] instead of the token it sampled (teacher forcing: tests / bench parity legs compare every step's logits with
    // an oracle conversation that is fed the same ids); ids_out still reports what the conversation's own logits chose
    int decode_batch(const int *slots, int n, const SampleParams &p, int *ids_out, const int *forced = nullptr);
    // Which launches the LAST BUILT batched step (forward_batch: eager, or the capture of a graph) took, per kind -- so that a test / the bench can assert that the
    // operating point it means to check (k_matvec_ri, k_matvec_ri_mix, the K-split w2 launch) is the one that ran, instead of a fallback with the same results
    struct BatchPath { int rows = 0, ri = 0, ri_mix = 0, ri_ksplit = 0, dot4 = 0, dot4_mix = 0, mul_mat = 0, sets = 0, ri_plain = 0; };
    const BatchPath &batch_path() const { return batch_path_; }
    static constexpr int MAX_CONVERSATIONS = 64;

    // ---- measurement hooks (bench / tests)
    // K greedy decode steps fed back on the device (no host round trip); returns ms per step via hipEvents.
    int decode_loop(int steps, int *tokens_out, float *ms_total);
    // Per-launch-site table of the decode step: `steps` eager decode steps issuing EXACTLY the launch set the captured hipGraph replays, a hipEvent
    // pair around every site; JSON array of {site, kernel (symbol as rocprofv3 prints it), calls_per_step, avg_us, bytes_per_call (algorithmic:
    // weight planes / KV rows the site reads)} + the whole-step time.  The events serialise nothing (one in-order stream) but add their own record
    // cost between launches, so the table is for attribution; the step time of record is the graph replay's.
    int profile_sites(int steps, std::string &json);
    float last_encode_ms() const { return last_encode_ms_; }
    void set_parity(bool on);                          // MINIGPT4_PARITY at run time (tests): drops the captured graphs, the next evaluation uses the other mode
    bool parity() const { return parity_; }

private:
    int load_llm(const std::string &path);
    int load_vision(const std::string &path);
    void alloc_buffers();
    int eval_chunk(const int *row_tok, int N, const float *embd);
    void forward(int N, bool from_tokens, hipStream_t s, bool feed = false);   // feed: N == 1 and the token comes from d_feed_ (decode)
    void forward_ref(int N, bool from_tokens, hipStream_t s, bool feed);   // parity mode: the oracle's accumulation order
    void forward_batch(int B, hipStream_t s);          // B decode rows of B conversations: tokens d_btok_[r], conversations d_bslot_[r]
    struct Prep { int kind; const float *x; const float *w; };   // 1: rms_norm(x)*w, 2: x, 3: silu(x)*w -- then quantised for the consumer's type
    void mul_mat(const QWeight &W, int N, float *y, int ldy, const float *residual, hipStream_t s, const Prep *prep, bool fuse, const char *site = "matmul", bool defer_ok = false, bool keep_pending = false);
    bool mul_mat_set(const QWeight *const *W, float *const *y, const float *const *res, int n, int N, int ldy, hipStream_t s, const Prep *prep, bool fuse, bool silu_pair = false, const char *site = "matmul", bool defer_ok = false, bool keep_pending = false);
    // prompt passes: the combine of a K-split mat-mul is left to the kernel that consumes the result (rope + cache append, the next norm +
    // quantisation, silu * mul); pend_ describes the slabs until then, flush_pending runs the combine as its own launch when the next consumer is not
    // one of those.  MINIGPT4_DEFER_COMBINE=0: always flush (A/B)
    SlabSrc pend_; bool defer_combine_ = true;
    const __half *xh_override_ = nullptr; int f16_pair_ = 7;   // F16 prompt pass: w1 | w3 + silu * mul in one launch, its fp16 rows feed w2 (MINIGPT4_F16_PAIR=0: A/B)
    // batched decode of > batch_rows_max_ conversations through the prompt pass's set launches (MINIGPT4_BATCH_SETS=0: one launch per matrix, A/B)
    bool batch_sets_ = true;
    void flush_pending(hipStream_t s);
    void prep_rms(const float *x, const float *w, int N, int K, int mask, hipStream_t s);
    void upload_qweight(const TensorMeta &t, const uint8_t *file_base, QWeight &w);
    template <typename T> T *upload_raw(DeviceArena &a, const void *src, size_t bytes);


    int native_broadcast(int world, int rank, const std::string &id_file, int timeout_s, int local_err = 0);
    int dist_world_ = 1, dist_rank_ = 0; float dist_bcast_ms_ = 0.0f;
    LoadMode load_mode_ = LOAD_FULL;
    bool moves_data() const { return load_mode_ == LOAD_FULL; }
    bool weights_missing() const;      // LOAD_RECV before weights_received(): sets the error text, true
    int device_ = 0;
    hipStream_t stream_ = nullptr;
    int n_ctx_ = 2048, n_batch_ = 512, max_rows_ = 512;
    struct Conversation {
        int n_past = 0;        // logical position: evaluated + queued rows
        int n_committed = 0;   // rows already evaluated on the device
        std::vector<int> pend_tok; std::vector<float> pend_embd;
        hipGraphExec_t graph = nullptr;   // decode step captured with this conversation's cache / position / token addresses
        bool graph_split = false;         // ... with the key-split attention launches (long context) or the one-workgroup-per-head kernel
    };
    std::vector<Conversation> conv_ = std::vector<Conversation>(1);
    int cur_ = 0;
    bool defer_ = true; int max_chunk_ = 512;
    void release_buffers();

    // LLM
    LLMFile llm_;
    Tokenizer tok_;
    Sampler sampler_;
    struct LayerW { float *attn_norm = nullptr, *ffn_norm = nullptr; QWeight wq, wk, wv, wo, w1, w2, w3; };
    bool mixed_qkv(const LayerW &L, hipStream_t s, bool fuse);
    std::vector<LayerW> layers_;
    float *norm_ = nullptr;
    QWeight output_;
    uint8_t *tok_raw_ = nullptr; int tok_type_ = -1;
    DeviceArena llm_arena_, vis_arena_, buf_arena_;
    // a collective of the native broadcast never completed (a peer died inside it): the stream will not drain, so the destructor must neither wait for it nor free
    // anything queued work may still touch -- the context's device memory is leaked, the load fails, the process lives
    bool stream_hung_ = false;
    uint8_t *stage_ = nullptr; size_t stage_cap_ = 0;
    size_t wbytes_token_ = 0;
    __half *kc_ = nullptr, *vc_ = nullptr;
    float *cos_ = nullptr, *sin_ = nullptr;
    Tables tabs_;
    // tabs_ = ggml's three fp16 tables (GELU, SiLU, exp) as the host libm evaluates them: what parity mode, every prompt pass and every GELU read.
    // tabs_dec_ = what the DECODE step's attention and SiLU launches get, tabs_vis_ = what the ViT / Q-Former attention gets: copies of tabs_ whose exp (and, for the decode step, silu)
    // pointers are NULL when MINIGPT4_COMPUTED_TABLES is on (the default) -- the kernels then compute the table VALUES (qtraits.hpp exp_h / silu_h: equal to the table's up to one fp16
    // ulp in ~1 of 10^4 values) instead of gathering them from the 128 KB tables; tabs_vis_.gelu is NULL too (computed_gelu_): the vision GEMMs' GELU epilogues compute (gelu_v)
    Tables tabs_dec_, tabs_vis_;
    // activations
    float *x_ = nullptr, *q_ = nullptr, *k_ = nullptr, *v_ = nullptr, *att_ = nullptr, *h1_ = nullptr, *h3_ = nullptr, *logits_ = nullptr;
    ActQ act_;
    // per conversation (indexed by slot): position, greedy token of the last evaluation, next input token; logits_ is [slots][n_vocab]
    int *d_npast_ = nullptr, *d_argmax_ = nullptr, *d_feed_ = nullptr;
    int *d_tokens_ = nullptr; void *d_scratch_ = nullptr;
    // batched decode: row tokens / conversations / positions (one 768-byte slab), [rows][n_vocab] logits
    int *d_btok_ = nullptr, *d_bslot_ = nullptr, *d_bpos_ = nullptr; float *blogits_ = nullptr;
    // [B]: the batched step for B rows (rows are described in device memory, so one graph serves any slot set)
    std::vector<hipGraphExec_t> batch_graph_;
    int *h_argmax_ = nullptr, *h_bstage_ = nullptr; float *h_logits_ = nullptr; int logits_host_slot_ = -1;
    bool use_graph_ = true, use_v2_ = true, attn_prefill_ = true, parity_ = false;
    FILE *trace_file_ = nullptr;       // MINIGPT4_PARITY_TRACE
    // key-split decode attention (llm_kernels.hip: k_attn_split_*)
    void *attn_ws_ = nullptr; int attn_splits_ = 6; int attn_split_t_ = 768; bool attn_split_now_ = false; int attn_splits_forced_ = 0;
    int batch_rows_max_ = 4; int batch_fuse_ = -1; int n_cus_ = 256;
    // round 5: B = 2..4 rows per weight pass on the int8 matrix cores over a row-interleaved second image of the k-quant matrices (ri_kernels.hip);
    // built by set_conversations(n > 1) -- a context with one conversation never pays the memory.  MINIGPT4_RI=0: the v_dot4 multi-row mat-vec of
    // rounds 2-4 (A/B)
    bool computed_tables_ = true;
    bool pair_silu_computed_ = true;   // the F16 model's w1 | w3 pair epilogue computes the SiLU table's values (with computed_tables_; MINIGPT4_PAIR_SILU_COMPUTED=0: gathers; round 6)
    bool computed_gelu_ = true;        // with computed_tables_: the vision GEMMs' GELU epilogues compute the table's values too (round 6)
    // B = 4: w2 (80 row groups x long K) on the K-split form of k_matvec_ri (+1.9 %; MINIGPT4_RI_W2=0: the v_dot4 launch)
    bool ri_w2_ = true;
    bool ri_wo_ = false;               // MINIGPT4_RI_WO (round 6 experiment): wo on k_matvec_ri with the plain row quantisation in its prologue
    // ri_fuse_: rows prepared inside the MFMA launches -- measured slower (profiles/r05_batched_decode_inengine.log), off
    bool use_ri_ = true, ri_ready_ = false, ri_fuse_ = false;
    DeviceArena ri_arena_;
    BatchPath batch_path_;
    RiWorkspace ri_ws_;                                     // this context's K-split workspace of k_matvec_ri (slabs + zeroed tickets; passed with every launch)
    std::vector<std::pair<const QWeight *, RiPlanes>> ri_map_;
    const RiPlanes *ri_of(const QWeight *w) const { for (const auto &e : ri_map_) if (e.first == w) return &e.second; return nullptr; }
    void build_ri_planes();
    // MINIGPT4_BATCH_MIX=0: wq|wk and wv of a mixed-type layer as two launches (the form before k_matvec_tn_mix)   // MINIGPT4_BATCH_ROWS_MAX:
    // batches up to this size use the multi-row mat-vec; 0 = never.  Measured (profiles/r02x_*): from 5 rows on the int8-MFMA kernels (single-wave
    // workgroups, one token tile) beat two passes of the 4-row mat-vec
    bool batch_mix_ = true;
    static constexpr int FUSE_DEFAULT = 87; int fuse_mask_ = FUSE_DEFAULT;
    // profiling (profile_sites)
    bool prof_on_ = false;
    struct SiteEv { hipEvent_t a, b; const char *site; std::string kernel; double bytes; size_t p0 = 0, p1 = 0; };   // [p0, p1): the launch probes of the site
    std::vector<SiteEv> site_events_;
    void site_begin(const char *site, double bytes, hipStream_t s);
    void site_end(hipStream_t s) noexcept;
    struct SiteScope { Engine *e; hipStream_t s; SiteScope(Engine *e_, const char *site, double bytes, hipStream_t s_) : e(e_), s(s_) { e->site_begin(site, bytes, s); } ~SiteScope() { e->site_end(s); } };

    // vision
    VisionFile vis_;
    int v_D_ = 0, v_depth_ = 0, v_M_ = 0, v_heads_ = 0, v_ql_ = 0, v_qi_ = 0, v_nq_ = 32, v_out_ = 0;
    struct VBlock { float *n1w, *n1b, *n2w, *n2b, *qkv_b, *proj_b, *fc1_b, *fc2_b; __half *qkv_w, *proj_w, *fc1_w, *fc2_w; };
    struct QAtt { __half *q_w = nullptr, *kv_w = nullptr, *dense_w = nullptr; float *q_b = nullptr, *kv_b = nullptr, *dense_b = nullptr, *ln_w = nullptr, *ln_b = nullptr; };
    struct QLayer { QAtt self, cross; bool has_cross = false; int cross_idx = -1; __half *inter_w, *out_w; float *inter_b, *out_b, *oln_w, *oln_b; };
    std::vector<VBlock> vblocks_;
    std::vector<QLayer> qlayers_;
    float *v_cls_ = nullptr, *v_pos_ = nullptr, *v_patch_b_ = nullptr, *v_lnv_w_ = nullptr, *v_lnv_b_ = nullptr, *v_qtok_ = nullptr, *v_qeln_w_ = nullptr, *v_qeln_b_ = nullptr, *v_proj_b_ = nullptr;
    __half *v_patch_w_ = nullptr, *v_proj_w_ = nullptr;
    // vision activations
    static constexpr int SPLITK_MAX = 12; int splitk_proj_ = 1, splitk_fc2_ = 4;   // MINIGPT4_SPLITK=proj,fc2 (1 = off)
    float *vi_slab_ = nullptr, *vi_qtok_rep_ = nullptr;
    float *vi_img_ = nullptr, *vi_pe_ = nullptr, *vi_x_ = nullptr, *vi_qkv_ = nullptr, *vi_hs_ = nullptr, *vi_a1_ = nullptr, *vi_a2_ = nullptr, *vi_d_ = nullptr, *vi_qq_ = nullptr, *vi_kv_ = nullptr, *vi_out_ = nullptr;
    __half *vi_patches_ = nullptr, *vi_ln_h_ = nullptr, *vi_att_h_ = nullptr, *vi_mlp_h_ = nullptr, *vi_img_h_ = nullptr, *vi_hs_h_ = nullptr, *vi_a1_h_ = nullptr, *vi_a2_h_ = nullptr, *vi_ctx_h_ = nullptr, *vi_im_h_ = nullptr;
    float last_encode_ms_ = 0;
    bool qf_skinny_ = true;            // Q-Former GEMMs on k_gemm_f16_skinny
    // round 5, launch count of the image path: (1) the cross-attention K | V projections of ALL cross layers are one weight block [n_cross * 1536][D]
    // -- they depend on the image features only (minigpt4.cpp:1148-1155), so one GEMM right after ln_vision replaces one per cross layer; (2) what
    // the Q-Former computes BEFORE it first looks at the image -- LayerNorm(query tokens), layer 0's self-attention block and its cross-attention
    // query projection -- does not depend on the image at all: evaluated once at load time by the same launches (fold_qformer_constants), kept for
    // every image of a batch.  MINIGPT4_QF_FOLD=0 / MINIGPT4_KV_HOIST=0: the round-4 form (A/B).
    __half *v_kv_all_w_ = nullptr; float *v_kv_all_b_ = nullptr; int v_ncross_ = 0;
    float *vi_c_a1_ = nullptr, *vi_c_qq_ = nullptr; __half *vi_c_a1_h_ = nullptr;
    bool qf_fold_ = true, qf_folded_ = false, kv_hoist_ = true;
    void fold_qformer_constants();
    bool qf_splitk_ = true;
    bool qkv_head_major_ = true;       // the ViT's qkv projection stores q | k | v head-major for k_attn_vit (round 6)
    void qf_dense_ln(const __half *A, int lda, const __half *W, int K, const float *bias, const float *residual, const float *ln_w, const float *ln_b, float *out, __half *out_h,
                     int rows, hipStream_t s);

    // ---- vision files whose Linear weights are not all F16 (an `--ftype f32` conversion, or a file written by minigpt4_quantize_model): every
    // Linear is a QWeight served by the LLM mat-mul kernels (activations quantised to the weight type's vec_dot_type, exactly ggml's mul_mat),
    // activations stay fp32.
    bool v_generic_ = false;
    struct GLin { QWeight w; };
    struct GBlock { GLin qkv, proj, fc1, fc2; };
    struct GAtt { GLin q, k, v, dense; };
    struct GQLayer { GAtt self, cross; GLin inter, out; };
    std::vector<GBlock> gblocks_; std::vector<GQLayer> gql_; GLin gproj_;
    DeviceArena vgen_arena_;
    ActQ vact_;
    float *vg_ln_ = nullptr, *vg_att_ = nullptr, *vg_mlp_ = nullptr, *vg_img_ = nullptr, *vg_tmp_ = nullptr, *vg_ctx_ = nullptr, *vg_im_ = nullptr;
    int load_vision_generic();
    void alloc_vision_generic();
    int encode_images_generic(const float *const *chw, int B, float *const *out);
    int encode_images_ref(const float *const *chw, int B, float *const *out);   // parity mode (engine_vision_generic.cpp)
    void build_vision_views();
    void glinear(const GLin &L, const float *x, int rows, const float *bias, bool gelu, const float *residual, float *out, hipStream_t s);
};

int device_count_noexcept();

// minigpt4_preprocess_image on the device (reference minigpt4.cpp:2597-2651): u8 HWC RGB of any size -> f32 [3][224][224], Pillow-bicubic resized and
// CLIP-normalised.  Host buffers in and out; the resample + normalisation run as HIP kernels on `s`.  Throws HipError.
void preprocess_image_device(hipStream_t s, const uint8_t *rgb, int w, int h, float *out_chw);

}  // namespace mg4
