/*
 * minigpt4.h -- C ABI of the MI355X-native MiniGPT-4 engine (libminigpt4.so).
 *
 * This is the DROP-IN BOUNDARY: the 18 entry points, 4 POD structs and 4 enums below are
 * byte-compatible with the reference's public header (/root/reference/minigpt4.h:28-114), so the
 * reference's ctypes binding (/root/reference/minigpt4/minigpt4_library.py:94-227), its web UI and
 * its CLI example bind to this library unchanged.  Each declaration cites the reference declaration
 * and implementation it replaces.  Everything underneath (ggml / llama.cpp CPU graph) is replaced by
 * hand-written gfx950 HIP kernels; see DESIGN.md.
 *
 * Error codes returned as `int` follow the reference's MiniGPT4Error enum
 * (/root/reference/minigpt4.cpp:97-119); see minigpt4_error_code_to_string.
 */
#pragma once

#include <stdint.h>
#include <stdlib.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#if defined(MINIGPT4_SHARED) && !defined(_WIN32)
#define MINIGPT4_API __attribute__((visibility("default")))
#else
#define MINIGPT4_API
#endif

#define IN
#define OUT

#ifdef __cplusplus
extern "C" {
#endif

struct MiniGPT4Context; /* opaque; one context = one conversation, used from one thread at a time */

/* reference minigpt4.h:30-48 -- numbering used by the vision file and by minigpt4_quantize_model */
enum MiniGPT4DataType { F16, F32, I32, L64, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K };

/* reference minigpt4.h:50-56 */
enum MiniGPT4Verbosity { MINIGPT4_VERBOSITY_NONE, MINIGPT4_VERBOSITY_ERROR, MINIGPT4_VERBOSITY_INFO, MINIGPT4_VERBOSITY_DEBUG };

/* reference minigpt4.h:58-63 */
enum MiniGPT4ImageFormat { MINIGPT4_IMAGE_FORMAT_UNKNOWN, MINIGPT4_IMAGE_FORMAT_F32, MINIGPT4_IMAGE_FORMAT_U8 };

/* reference minigpt4.h:65-72 (24 bytes on x86-64; mirrored by minigpt4_library.py:56-63) */
struct MiniGPT4Image {
    void *data;
    int width;
    int height;
    int channels;
    enum MiniGPT4ImageFormat format;
};

/* reference minigpt4.h:74-78 (16 bytes) */
struct MiniGPT4Embedding {
    float *data;
    size_t elements;
};

/* reference minigpt4.h:80-84 / 86-90 (declared by the reference, used by the additive batched API) */
struct MiniGPT4Embeddings {
    struct MiniGPT4Embedding *embeddings;
    size_t n_embeddings;
};
struct MiniGPT4Images {
    struct MiniGPT4Image *images;
    size_t n_images;
};

/* reference minigpt4.h:92-95 */
enum MiniGPT4ImageLoadFlags { MINIGPT4_IMAGE_LOAD_FLAG_NONE };

/* minigpt4.h:97, impl minigpt4.cpp:2543-2574.  NULL when a path is missing, a file is malformed, or no
 * gfx950 device is usable (the library never falls back to a CPU path).  `numa` is ignored. */
MINIGPT4_API struct MiniGPT4Context *minigpt4_model_load(const char *path, const char *llm_model, int verbosity, int seed, int n_ctx, int n_batch, bool numa);
/* minigpt4.h:98, impl minigpt4.cpp:2576-2595 (OpenCV build: cv::imread(IMREAD_COLOR) + BGR2RGB).  Native here (no OpenCV): PNG / JPEG /
 * BMP / binary PNM -> U8 HWC RGB, library-allocated image->data (free with minigpt4_free_image).  0, 17 (missing file) or 5 (OpenImage).
 * ctx is not used (may be NULL).  Host work, as in the reference. */
MINIGPT4_API int minigpt4_image_load_from_file(struct MiniGPT4Context *ctx, const char *path, IN struct MiniGPT4Image *image, int flags);
/* minigpt4.h:99, impl minigpt4.cpp:2597-2651 (OpenCV build: PillowResize bicubic 224x224, 1/255, CLIP mean/std, HWC->CHW).  Native here, as
 * HIP kernels: needs a GPU (no CPU fallback; 6 when none).  Errors 15 (channels != 3) / 16 (format != U8).  Output as the reference reports it:
 * F32, width = 1, height = 150528, channels = 1, library-allocated (free with minigpt4_free_image).  ctx may be NULL (default stream). */
MINIGPT4_API int minigpt4_preprocess_image(struct MiniGPT4Context *ctx, IN const struct MiniGPT4Image *image, OUT struct MiniGPT4Image *preprocessed_image, int flags);
/* minigpt4.h:100, impl minigpt4.cpp:2653-2662 -> MiniGPT4::encode_image :2094-2363.  image: F32 CHW 3x224x224
 * (errors 13/14).  The library allocates embedding->data (32*n_embd floats); free with minigpt4_free_embedding. */
MINIGPT4_API int minigpt4_encode_image(struct MiniGPT4Context *ctx, IN struct MiniGPT4Image *image, OUT struct MiniGPT4Embedding *embedding, size_t n_threads);
/* minigpt4.h:101, impl minigpt4.cpp:2671-2702. */
MINIGPT4_API int minigpt4_begin_chat_image(struct MiniGPT4Context *ctx, IN struct MiniGPT4Embedding *image_embedding, const char *s, size_t n_threads);
/* minigpt4.h:102, impl minigpt4.cpp:2704-2718.  *token is borrowed (owned by ctx / static), never freed by the caller. */
MINIGPT4_API int minigpt4_end_chat_image(struct MiniGPT4Context *ctx, const char **token, size_t n_threads, float temp, int32_t top_k, float top_p, float tfs_z, float typical_p, int32_t repeat_last_n, float repeat_penalty, float alpha_presence, float alpha_frequency, int mirostat, float mirostat_tau, float mirostat_eta, int penalize_nl);
/* minigpt4.h:103, impl minigpt4.cpp:2720-2732. */
MINIGPT4_API int minigpt4_system_prompt(struct MiniGPT4Context *ctx, size_t n_threads);
/* minigpt4.h:104, impl minigpt4.cpp:2734-2748. */
MINIGPT4_API int minigpt4_begin_chat(struct MiniGPT4Context *ctx, const char *s, size_t n_threads);
/* minigpt4.h:105, impl minigpt4.cpp:2750-2753. */
MINIGPT4_API int minigpt4_end_chat(struct MiniGPT4Context *ctx, const char **token, size_t n_threads, float temp, int32_t top_k, float top_p, float tfs_z, float typical_p, int32_t repeat_last_n, float repeat_penalty, float alpha_presence, float alpha_frequency, int mirostat, float mirostat_tau, float mirostat_eta, int penalize_nl);
/* minigpt4.h:106, impl minigpt4.cpp:2755-2762. */
MINIGPT4_API int minigpt4_reset_chat(struct MiniGPT4Context *ctx);
/* minigpt4.h:107, impl minigpt4.cpp:2764-2772: 11 iff s == "##". */
MINIGPT4_API int minigpt4_contains_eos_token(const char *s);
/* minigpt4.h:108, impl minigpt4.cpp:2774-2782: 12 iff s ends with "###". */
MINIGPT4_API int minigpt4_is_eos(const char *s);
/* minigpt4.h:109-111, impl minigpt4.cpp:2784-2809. */
MINIGPT4_API int minigpt4_free(struct MiniGPT4Context *ctx);
MINIGPT4_API int minigpt4_free_image(struct MiniGPT4Image *image);
MINIGPT4_API int minigpt4_free_embedding(struct MiniGPT4Embedding *embedding);
/* minigpt4.h:112, impl minigpt4.cpp:2811-2815: the enum identifier as static text. */
MINIGPT4_API const char *minigpt4_error_code_to_string(int error_code);
/* minigpt4.h:113, impl minigpt4.cpp:2817-2982.  Host-only offline tool: rewrites a vision file with its eligible Linear weights re-quantised by ggml's
 * reference quantisers (data_type: MiniGPT4DataType Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q4_K Q5_K Q6_K; 3 for any other).  17 if the input is missing, 18 if the output
 * cannot be written.  Tensors whose row length is not a whole number of blocks keep their type (the reference would write a file ggml cannot load). */
MINIGPT4_API int minigpt4_quantize_model(const char *in_path, const char *out_path, int data_type);
/* minigpt4.h:114, impl minigpt4.cpp:2984-2986. */
MINIGPT4_API void minigpt4_set_verbosity(int verbosity);

#ifdef __cplusplus
}
#endif
