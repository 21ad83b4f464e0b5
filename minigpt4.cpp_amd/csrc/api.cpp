// extern "C" boundary of libminigpt4.so: the reference's 18 entry points (include/minigpt4.h, reference minigpt4.cpp:2524-2986)
// plus the additive serving / measurement / multi-GPU hooks of include/minigpt4_amd.h.  Kernel-level test hooks, micro-benchmarks, probes and the host-only test
// helpers live in test_hooks.cpp and are linked into libminigpt4_test.so only.
#include <sys/stat.h>

#include <chrono>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "engine.hpp"
#include "api_internal.hpp"
#include "imageio.hpp"
#include "quantize.hpp"
#include "minigpt4_amd.h"

using namespace mg4;
using namespace mg4::apiutil;

namespace {
// Prompt constants (reference minigpt4.cpp:139, 2680, 2697, 2699, 2743, 2745)
const char *kSystemPrompt = "Give the following image: <Img>ImageContent</Img>. You will be able to see the image once I provide it to you. Please answer my questions.###";
constexpr size_t kEmb7B = 32 * 4096, kEmb13B = 32 * 5120;


const char *kErrNames[] = {"None", "LoadModelFileHeader", "LoadModelFileVersion", "LoadModelMiniGPT4DataType", "LoadLanguageModel", "OpenImage", "ImageSize", "MmapSupport",
                           "FailedToAddString", "LLamaProjectionEmbeddingInvalidSize", "FailedToAddEmbedding", "EosToken", "Eos", "ImageNot224_244_3", "ImageNotF32",
                           "ImageChannelsExpectedRGB", "ImageFormatExpectedU8", "PathDoesNotExist", "DumpModelFileOpen", "OpenCVNotLinked"};
}  // namespace


extern "C" {

// ======================================================================================================== reference ABI
struct MiniGPT4Context *minigpt4_model_load(const char *path, const char *llm_model, int verbosity, int seed, int n_ctx, int n_batch, bool /*numa*/) {
    g_verbosity = verbosity & 0xFF;
    auto t0 = std::chrono::steady_clock::now();
    if (!file_exists(path)) { MG4_ERR("%s does not exist", path ? path : "(null)"); return nullptr; }
    if (!file_exists(llm_model)) { MG4_ERR("%s does not exist", llm_model ? llm_model : "(null)"); return nullptr; }
    Engine *eng = new (std::nothrow) Engine();
    if (!eng) return nullptr;
    const int err = guarded((int)E_LoadLanguageModel, [&] { return eng->init(path, llm_model, seed, n_ctx, n_batch); });
    if (err) { MG4_ERR("Failed to initialize MiniGPT4: %s", minigpt4_error_code_to_string(err)); delete eng; return nullptr; }
    MG4_INFO("Load model from file took %lld ms to complete", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count());
    return reinterpret_cast<MiniGPT4Context *>(eng);
}

// Native replacement of the reference's OpenCV-only path (minigpt4.cpp:2576-2595: cv::imread(IMREAD_COLOR) + BGR->RGB, U8 HWC).  Host work, like imread.
int minigpt4_image_load_from_file(struct MiniGPT4Context *, const char *path, struct MiniGPT4Image *image, int /*flags*/) {
    if (!image) return E_OpenImage;
    return guarded((int)E_OpenImage, [&]() -> int {
        ImageRGB8 im;
        if (int err = load_image_file(path, im)) { MG4_ERR("%s", last_error().c_str()); return err; }
        uint8_t *data = im.px.release();                   // the decoder's own buffer (malloc family): no copy
        image->data = data; image->width = im.w; image->height = im.h; image->channels = 3; image->format = MINIGPT4_IMAGE_FORMAT_U8;
        return E_None;
    });
}

// minigpt4.cpp:2597-2651 (OpenCV build): Pillow-bicubic resize to 224x224, 1/255, CLIP mean/std, HWC -> CHW -- as HIP kernels (image_kernels.hip).
// Output geometry is the reference's: after its split/reshape/hconcat the matrix is 1 x 150528 single-channel, reported as width = rows = 1,
// height = cols = 150528, channels = 1 (:2639-2641); minigpt4_encode_image only checks the product (:2130).
int minigpt4_preprocess_image(struct MiniGPT4Context *ctx, const struct MiniGPT4Image *image, struct MiniGPT4Image *preprocessed_image, int /*flags*/) {
    if (!image || !preprocessed_image || !image->data) return E_ImageSize;
    if (image->channels != 3) { MG4_ERR("Image must have 3 channels"); return E_ImageChannelsExpectedRGB; }
    if (image->format != MINIGPT4_IMAGE_FORMAT_U8) { MG4_ERR("Image must be in U8 format"); return E_ImageFormatExpectedU8; }
    if (image->width <= 0 || image->height <= 0 || image->width > 65535 || image->height > 65535) return E_ImageSize;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device visible: image preprocessing runs on the GPU and has no CPU fallback"); MG4_ERR("%s", last_error().c_str()); return E_ImageSize; }
    return guarded((int)E_ImageSize, [&]() -> int {
        const size_t n = (size_t)3 * 224 * 224;
        float *out = static_cast<float *>(malloc(n * sizeof(float)));             // released by minigpt4_free_image (free)
        if (!out) return (int)E_ImageSize;
        try { preprocess_image_device(ctx ? E_(ctx)->stream() : nullptr, static_cast<const uint8_t *>(image->data), image->width, image->height, out); }
        catch (...) { free(out); throw; }
        preprocessed_image->data = out; preprocessed_image->width = 1; preprocessed_image->height = (int)n; preprocessed_image->channels = 1;
        preprocessed_image->format = MINIGPT4_IMAGE_FORMAT_F32;
        return E_None;
    });
}

int minigpt4_encode_image(struct MiniGPT4Context *ctx, struct MiniGPT4Image *image, struct MiniGPT4Embedding *embedding, size_t /*n_threads*/) {
    if (!ctx || !image || !embedding) return E_ImageSize;
    if ((long long)image->width * image->height * image->channels != 224LL * 224 * 3) return E_ImageNot224_244_3;
    if (image->format != MINIGPT4_IMAGE_FORMAT_F32) return E_ImageNotF32;
    Engine *e = E_(ctx);
    auto t0 = std::chrono::steady_clock::now();
    const size_t n = (size_t)e->n_query() * e->proj_out();
    float *out = new (std::nothrow) float[n];
    if (!out) return E_ImageSize;
    const int err = guarded((int)E_ImageSize, [&] { return e->encode_image(static_cast<const float *>(image->data), out); });
    if (err) { delete[] out; return err; }
    embedding->data = out; embedding->elements = n;
    MG4_INFO("Encoding image took %lld ms to complete", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count());
    return E_None;
}

int minigpt4_begin_chat_image(struct MiniGPT4Context *ctx, struct MiniGPT4Embedding *image_embedding, const char *s, size_t) {
    if (!ctx || !image_embedding || !s) return E_FailedToAddString;
    Engine *e = E_(ctx);
    return guarded((int)E_FailedToAddString, [&]() -> int {
        if (int err = e->add_string("Human: <Img>")) return err;
        if (image_embedding->elements != kEmb13B && image_embedding->elements != kEmb7B) { MG4_ERR("LLAMA projection image embedding size not equal %zu != %zu", image_embedding->elements, kEmb13B); return E_LLamaProjectionEmbeddingInvalidSize; }
        if (int err = e->add_embedding(image_embedding->data, 32)) return err;   // elements := 32 rows of llama_n_embd floats (minigpt4.cpp:2688-2695)
        if (int err = e->add_string("</Img> ")) return err;
        if (int err = e->add_string(s)) return err;
        return e->add_string("### Assistant:");
    });
}

int minigpt4_end_chat_image(struct MiniGPT4Context *ctx, const char **token, size_t, float temp, int32_t top_k, float top_p, float tfs_z, float typical_p, int32_t /*repeat_last_n*/,
                            float /*repeat_penalty*/, float /*alpha_presence*/, float /*alpha_frequency*/, int mirostat, float mirostat_tau, float mirostat_eta, int /*penalize_nl*/) {
    if (!ctx || !token) return E_None;
    Engine *e = E_(ctx);
    guarded(0, [&]() -> int {
        SampleParams p; p.temp = temp; p.top_k = top_k; p.top_p = top_p; p.tfs_z = tfs_z; p.typical_p = typical_p; p.mirostat = mirostat; p.mirostat_tau = mirostat_tau; p.mirostat_eta = mirostat_eta;
        const int id = e->sample_token(p);
        *token = e->id_to_token(id);
        (void)e->add_tokens({id}, /*flush_now=*/true);   // launched asynchronously; result discarded, as in the reference (minigpt4.cpp:2715)
        return 0;
    });
    return E_None;
}

int minigpt4_system_prompt(struct MiniGPT4Context *ctx, size_t) {
    if (!ctx) return E_FailedToAddString;
    return guarded((int)E_FailedToAddString, [&] { return E_(ctx)->add_string(kSystemPrompt); });
}

int minigpt4_begin_chat(struct MiniGPT4Context *ctx, const char *s, size_t) {
    if (!ctx || !s) return E_FailedToAddString;
    Engine *e = E_(ctx);
    return guarded((int)E_FailedToAddString, [&]() -> int {
        if (int err = e->add_string("Human: ")) return err;
        if (int err = e->add_string(s)) return err;
        return e->add_string("### Assistant:");
    });
}

int minigpt4_end_chat(struct MiniGPT4Context *ctx, const char **token, size_t n_threads, float temp, int32_t top_k, float top_p, float tfs_z, float typical_p, int32_t repeat_last_n,
                      float repeat_penalty, float alpha_presence, float alpha_frequency, int mirostat, float mirostat_tau, float mirostat_eta, int penalize_nl) {
    return minigpt4_end_chat_image(ctx, token, n_threads, temp, top_k, top_p, tfs_z, typical_p, repeat_last_n, repeat_penalty, alpha_presence, alpha_frequency, mirostat, mirostat_tau, mirostat_eta, penalize_nl);
}

int minigpt4_reset_chat(struct MiniGPT4Context *ctx) { if (ctx) E_(ctx)->reset(); return E_None; }
int minigpt4_contains_eos_token(const char *s) { return s && strcmp(s, "##") == 0 ? E_EosToken : E_None; }
int minigpt4_is_eos(const char *s) { if (!s) return E_None; const size_t n = strlen(s); return n >= 3 && memcmp(s + n - 3, "###", 3) == 0 ? E_Eos : E_None; }
int minigpt4_free(struct MiniGPT4Context *ctx) { delete E_(ctx); return E_None; }
int minigpt4_free_image(struct MiniGPT4Image *image) { if (image && image->data) { free(image->data); image->data = nullptr; } return E_None; }   // every image the library hands out is malloc-family memory
int minigpt4_free_embedding(struct MiniGPT4Embedding *embedding) { if (embedding && embedding->data) { delete[] embedding->data; embedding->data = nullptr; } return E_None; }
const char *minigpt4_error_code_to_string(int error_code) { return error_code >= 0 && error_code < 20 ? kErrNames[error_code] : ""; }
int minigpt4_quantize_model(const char *in_path, const char *out_path, int data_type) {
    if (!file_exists(in_path)) return E_PathDoesNotExist;                        // minigpt4.cpp:2823-2826
    try { return quantize_vision_file(in_path, out_path, data_type); }
    catch (const std::exception &e) { set_last_error(std::string("minigpt4_quantize_model: ") + e.what()); return E_DumpModelFileOpen; }
}
void minigpt4_set_verbosity(int verbosity) { g_verbosity = verbosity & 0xFF; }

// ======================================================================================================== additive API
int minigpt4_amd_device_count(void) { return device_count_noexcept(); }
const char *minigpt4_amd_last_error(void) { return last_error().c_str(); }
const char *minigpt4_amd_build_info(void) { return "minigpt4.cpp_amd: gfx950 (CDNA4) HIP kernels; v_dot4_i32_i8 fused-dequant mat-vec, MFMA f16 GEMM; api.o built " __DATE__ " " __TIME__; }
int minigpt4_amd_n_vocab(struct MiniGPT4Context *ctx) { return ctx ? E_(ctx)->n_vocab() : 0; }
int minigpt4_amd_n_embd(struct MiniGPT4Context *ctx) { return ctx ? E_(ctx)->n_embd() : 0; }
int minigpt4_amd_n_past(struct MiniGPT4Context *ctx) { return ctx ? E_(ctx)->n_past() : 0; }
int minigpt4_amd_eval_tokens(struct MiniGPT4Context *ctx, const int32_t *tokens, int n) {
    if (!ctx || !tokens || n < 0) return E_FailedToAddString;
    return guarded((int)E_FailedToAddString, [&] { return E_(ctx)->add_tokens(std::vector<int>(tokens, tokens + n)); });
}
int minigpt4_amd_eval_embd(struct MiniGPT4Context *ctx, const float *embd, int n_rows) {
    if (!ctx || !embd || n_rows <= 0) return E_FailedToAddEmbedding;
    return guarded((int)E_FailedToAddEmbedding, [&] { return E_(ctx)->add_embedding(embd, n_rows); });
}
int minigpt4_amd_get_logits(struct MiniGPT4Context *ctx, float *out, size_t n) {
    if (!ctx || !out) return 1;
    return guarded(1, [&] { const float *l = E_(ctx)->logits_host(); memcpy(out, l, std::min(n, (size_t)E_(ctx)->n_vocab()) * 4); return 0; });
}
int minigpt4_amd_tokenize(struct MiniGPT4Context *ctx, const char *text, int add_bos, int32_t *out, int cap) {
    if (!ctx || !text) return -1;
    const std::vector<int> t = E_(ctx)->tokenizer().tokenize(text, add_bos != 0);
    for (int i = 0; i < (int)t.size() && i < cap; i++) out[i] = t[(size_t)i];
    return (int)t.size();
}
int minigpt4_amd_sample(struct MiniGPT4Context *ctx, int32_t *token_id, float temp, int32_t top_k, float top_p, float tfs_z, float typical_p, int mirostat, float mirostat_tau, float mirostat_eta) {
    if (!ctx || !token_id) return 1;
    return guarded(1, [&] { SampleParams p; p.temp = temp; p.top_k = top_k; p.top_p = top_p; p.tfs_z = tfs_z; p.typical_p = typical_p; p.mirostat = mirostat; p.mirostat_tau = mirostat_tau; p.mirostat_eta = mirostat_eta;
        *token_id = E_(ctx)->sample_token(p); return 0; });
}
int minigpt4_amd_decode_loop(struct MiniGPT4Context *ctx, int steps, int32_t *tokens_out, float *ms_total) {
    if (!ctx) return 1;
    return guarded(1, [&] { return E_(ctx)->decode_loop(steps, tokens_out, ms_total); });
}
int minigpt4_amd_profile_sites(struct MiniGPT4Context *ctx, int steps, char *json_out, size_t capacity) {
    if (!ctx || !json_out || capacity < 2) return 1;
    return guarded(1, [&] { std::string js; const int rc = E_(ctx)->profile_sites(steps, js); if (rc) return rc; if (js.size() + 1 > capacity) return 2; memcpy(json_out, js.c_str(), js.size() + 1); return 0; });
}
double minigpt4_amd_weight_bytes_per_token(struct MiniGPT4Context *ctx) { return ctx ? (double)E_(ctx)->weight_bytes_per_token() : 0.0; }
float minigpt4_amd_last_encode_ms(struct MiniGPT4Context *ctx) { return ctx ? E_(ctx)->last_encode_ms() : 0.0f; }
int minigpt4_amd_sync(struct MiniGPT4Context *ctx) { if (!ctx) return 1; return guarded(1, [&] { E_(ctx)->sync(); return 0; }); }

int minigpt4_amd_encode_images(struct MiniGPT4Context *ctx, const struct MiniGPT4Images *images, struct MiniGPT4Embeddings *embeddings, size_t /*n_threads*/) {
    if (!ctx || !images || !embeddings) return E_ImageSize;
    embeddings->embeddings = new (std::nothrow) MiniGPT4Embedding[images->n_images ? images->n_images : 1]();
    embeddings->n_embeddings = 0;
    if (!embeddings->embeddings) return E_ImageSize;
    for (size_t i = 0; i < images->n_images; i++) {   // the checks of minigpt4_encode_image (minigpt4.cpp:2130-2138), before any work
        const MiniGPT4Image &im = images->images[i];
        const int err = !im.data ? (int)E_ImageSize : (long long)im.width * im.height * im.channels != 224LL * 224 * 3 ? (int)E_ImageNot224_244_3 : im.format != MINIGPT4_IMAGE_FORMAT_F32 ? (int)E_ImageNotF32 : 0;
        if (err) { minigpt4_amd_free_embeddings(embeddings); return err; }
    }
    Engine *e = E_(ctx);
    const size_t n = (size_t)e->n_query() * e->proj_out();
    const int err = guarded((int)E_ImageSize, [&]() -> int {
        for (size_t i0 = 0; i0 < images->n_images; i0 += Engine::VISION_BATCH_MAX) {   // passes of up to 8 images over the vision weights
            const int B = (int)std::min<size_t>(Engine::VISION_BATCH_MAX, images->n_images - i0);
            const float *in[Engine::VISION_BATCH_MAX]; float *out[Engine::VISION_BATCH_MAX];
            for (int b = 0; b < B; b++) {
                in[b] = static_cast<const float *>(images->images[i0 + (size_t)b].data);
                out[b] = new float[n];
                embeddings->embeddings[i0 + (size_t)b].data = out[b]; embeddings->embeddings[i0 + (size_t)b].elements = n;
                embeddings->n_embeddings = i0 + (size_t)b + 1;
            }
            if (int rc = e->encode_images(in, B, out)) return rc;
        }
        return E_None;
    });
    if (err) { minigpt4_amd_free_embeddings(embeddings); return err; }
    return E_None;
}
int minigpt4_amd_free_embeddings(struct MiniGPT4Embeddings *embeddings) {
    if (!embeddings || !embeddings->embeddings) return E_None;
    for (size_t i = 0; i < embeddings->n_embeddings; i++) minigpt4_free_embedding(&embeddings->embeddings[i]);
    delete[] embeddings->embeddings; embeddings->embeddings = nullptr; embeddings->n_embeddings = 0;
    return E_None;
}
// the round-2..5 names of the two entry points above, kept for one more round as forwarding definitions (include/minigpt4_amd.h: deprecated -- they sit in the reference's
// own minigpt4_ namespace and would collide if upstream ever implements its declared-but-unused MiniGPT4Images API, minigpt4.h:80-90)
int minigpt4_encode_images(struct MiniGPT4Context *ctx, const struct MiniGPT4Images *images, struct MiniGPT4Embeddings *embeddings, size_t n_threads) { return minigpt4_amd_encode_images(ctx, images, embeddings, n_threads); }
int minigpt4_free_embeddings(struct MiniGPT4Embeddings *embeddings) { return minigpt4_amd_free_embeddings(embeddings); }
// ---- several conversations per context ---------------------------------------------------------------------------
int minigpt4_amd_set_conversations(struct MiniGPT4Context *ctx, int n) {
    if (!ctx) return 1;
    return guarded(1, [&] { return E_(ctx)->set_conversations(n); });
}
int minigpt4_amd_select_conversation(struct MiniGPT4Context *ctx, int slot) { return ctx ? E_(ctx)->select_conversation(slot) : 1; }
int minigpt4_amd_n_conversations(struct MiniGPT4Context *ctx) { return ctx ? E_(ctx)->n_conversations() : 0; }
int minigpt4_amd_end_chat_batch(struct MiniGPT4Context *ctx, const int32_t *slots, int n, const char **tokens, float temp, int32_t top_k, float top_p, float tfs_z, float typical_p,
                                int mirostat, float mirostat_tau, float mirostat_eta) {
    if (!ctx || !slots || !tokens || n < 1 || n > Engine::MAX_CONVERSATIONS) return 1;
    Engine *e = E_(ctx);
    return guarded(1, [&]() -> int {
        SampleParams p; p.temp = temp; p.top_k = top_k; p.top_p = top_p; p.tfs_z = tfs_z; p.typical_p = typical_p; p.mirostat = mirostat; p.mirostat_tau = mirostat_tau; p.mirostat_eta = mirostat_eta;
        int ids[Engine::MAX_CONVERSATIONS];
        if (int rc = e->decode_batch(slots, n, p, ids)) return rc;
        for (int i = 0; i < n; i++) tokens[i] = e->id_to_token(ids[i]);
        return 0;
    });
}
int minigpt4_amd_eval_batch(struct MiniGPT4Context *ctx, const int32_t *slots, int n, const int32_t *tokens, int32_t *greedy_out) {
    if (!ctx || !slots || !tokens || n < 1 || n > Engine::MAX_CONVERSATIONS) return 1;
    Engine *e = E_(ctx);
    return guarded(1, [&]() -> int {
        SampleParams p; p.temp = 0.0f;
        int ids[Engine::MAX_CONVERSATIONS];
        if (int rc = e->decode_batch(slots, n, p, ids, tokens)) return rc;
        if (greedy_out) for (int i = 0; i < n; i++) greedy_out[i] = ids[i];
        return 0;
    });
}
int minigpt4_amd_batch_path(struct MiniGPT4Context *ctx, int32_t out[8]) {
    if (!ctx || !out) return 1;
    const Engine::BatchPath &b = E_(ctx)->batch_path();
    out[0] = b.rows; out[1] = b.ri; out[2] = b.ri_mix; out[3] = b.ri_ksplit; out[4] = b.dot4; out[5] = b.dot4_mix; out[6] = b.mul_mat; out[7] = b.sets;
    return 0;
}
int minigpt4_amd_weight_arena(struct MiniGPT4Context *ctx, int which, void **device_ptr, size_t *bytes) {
    if (!ctx || !device_ptr || !bytes) return 1;
    ClearStickyHipError clear_on_exit;   // the caller is about to hand this pointer to another HIP user (torch / RCCL): no stale error of ours may meet its launch checks
    Engine *e = E_(ctx);
    *device_ptr = which == 0 ? (void *)e->llm_arena_ptr() : (void *)e->vision_arena_ptr();
    *bytes = which == 0 ? e->llm_arena_bytes() : e->vision_arena_bytes();
    return 0;
}

// ---- multi-GPU load: arena layout plan (host only), receive-mode completion, arena checksum ------------------------------
int minigpt4_amd_plan_arenas(const char *vision_path, const char *llm_path, size_t *llm_bytes, size_t *vision_bytes, uint64_t *llm_hash, uint64_t *vision_hash) {
    if (!vision_path || !llm_path) return 1;
    return guarded(3, [&]() -> int {
        Engine::ArenaPlan p;
        if (int e = Engine::plan_arenas(vision_path, llm_path, p)) return e;
        if (llm_bytes) *llm_bytes = p.llm_bytes; if (vision_bytes) *vision_bytes = p.vision_bytes;
        if (llm_hash) *llm_hash = p.llm_hash; if (vision_hash) *vision_hash = p.vision_hash;
        return 0;
    });
}
int minigpt4_amd_arena_plan(struct MiniGPT4Context *ctx, size_t *llm_bytes, size_t *vision_bytes, uint64_t *llm_hash, uint64_t *vision_hash) {
    if (!ctx) return 1;
    const Engine::ArenaPlan p = E_(ctx)->arena_plan();
    if (llm_bytes) *llm_bytes = p.llm_bytes; if (vision_bytes) *vision_bytes = p.vision_bytes;
    if (llm_hash) *llm_hash = p.llm_hash; if (vision_hash) *vision_hash = p.vision_hash;
    return 0;
}
int minigpt4_amd_dist_info(struct MiniGPT4Context *ctx, int *world, int *rank, float *bcast_ms) {
    if (!ctx) return 1;
    if (world) *world = E_(ctx)->dist_world();
    if (rank) *rank = E_(ctx)->dist_rank();
    if (bcast_ms) *bcast_ms = E_(ctx)->dist_bcast_ms();
    return 0;
}
int minigpt4_amd_set_parity(struct MiniGPT4Context *ctx, int on) {
    if (!ctx) return 1;
    return guarded(1, [&]() -> int { E_(ctx)->set_parity(on != 0); return 0; });
}
int minigpt4_amd_parity(struct MiniGPT4Context *ctx) { return ctx ? (int)E_(ctx)->parity() : -1; }
int minigpt4_amd_load_mode(struct MiniGPT4Context *ctx) { return ctx ? (int)E_(ctx)->load_mode() : -1; }
int minigpt4_amd_weights_received(struct MiniGPT4Context *ctx) {
    if (!ctx) return 1;
    return guarded(3, [&] { return E_(ctx)->weights_received(); });
}
int minigpt4_amd_arena_checksum(struct MiniGPT4Context *ctx, int which, uint64_t *sum) {
    if (!ctx || !sum) return 1;
    return guarded(3, [&]() -> int {
        Engine *e = E_(ctx);
        const uint8_t *p = which == 0 ? e->llm_arena_ptr() : e->vision_arena_ptr();
        const size_t n = which == 0 ? e->llm_arena_bytes() : e->vision_arena_bytes();
        *sum = device_checksum(p, n, e->stream());
        return 0;
    });
}

int minigpt4_amd_decode_image(const void *bytes, size_t n, struct MiniGPT4Image *image) {
    if (!bytes || !image) return E_OpenImage;
    return guarded((int)E_OpenImage, [&]() -> int {
        ImageRGB8 im; std::string err;
        if (!decode_image(static_cast<const uint8_t *>(bytes), n, im, err)) { set_last_error(err); return E_OpenImage; }
        uint8_t *data = im.px.release();
        if (!data) return E_OpenImage;
        image->data = data; image->width = im.w; image->height = im.h; image->channels = 3; image->format = MINIGPT4_IMAGE_FORMAT_U8;
        return E_None;
    });
}
}  // extern "C"
