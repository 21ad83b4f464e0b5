/* TEST INFRASTRUCTURE: a C client of the drop-in library, compiled against include/minigpt4.h exactly as a user of the reference would compile against its
 * minigpt4.h.  Replays the call sequence of the reference's examples/main.cpp:207-293 (load -> image -> encode -> system prompt -> begin_chat_image -> end_chat_image loop
 * with the EOS checks -> free), with the float CHW image read from a raw file (the reference's OpenCV-less build cannot load images either).
 *   replay_main <vision.bin> <llm.bin> <image_f32_chw.raw> <n_tokens> <prompt>
 * prints one generated piece per line between "BEGIN" and "END"; exit code = first failing call's error code. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minigpt4.h"

#define CHECK(expr) do { int e_ = (expr); if (e_) { fprintf(stderr, "%s -> %d (%s)\n", #expr, e_, minigpt4_error_code_to_string(e_)); return e_ ? e_ : 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage\n"); return 100; }
    const int n_tokens = atoi(argv[4]);
    struct MiniGPT4Context *ctx = minigpt4_model_load(argv[1], argv[2], MINIGPT4_VERBOSITY_ERROR, 1337, 256, 64, 0);
    if (!ctx) { fprintf(stderr, "minigpt4_model_load failed\n"); return 101; }
    float *chw = (float *)malloc(3 * 224 * 224 * sizeof(float));
    FILE *f = fopen(argv[3], "rb");
    if (!f || fread(chw, sizeof(float), 3 * 224 * 224, f) != 3 * 224 * 224) { fprintf(stderr, "image read failed\n"); return 102; }
    fclose(f);
    struct MiniGPT4Image image;
    image.data = chw; image.width = 224; image.height = 224; image.channels = 3; image.format = MINIGPT4_IMAGE_FORMAT_F32;
    struct MiniGPT4Embedding embedding;
    CHECK(minigpt4_encode_image(ctx, &image, &embedding, 0));
    CHECK(minigpt4_system_prompt(ctx, 0));
    CHECK(minigpt4_begin_chat_image(ctx, &embedding, argv[5], 0));
    printf("BEGIN\n");
    char chat[1 << 16]; chat[0] = 0;
    for (int i = 0; i < n_tokens; i++) {
        const char *token = NULL;
        CHECK(minigpt4_end_chat_image(ctx, &token, 0, 0.0f, 40, 0.9f, 1.0f, 1.0f, 64, 1.1f, 1.0f, 1.0f, 0, 5.0f, 1.0f, 1));
        if (strlen(chat) + strlen(token) + 1 < sizeof(chat)) strcat(chat, token);
        printf("%s\n", token);
        /* the reference's loop `continue`s on contains_eos_token and stops on is_eos; the test model never produces either on purpose -- both are exercised for their codes */
        (void)minigpt4_contains_eos_token(token);
        (void)minigpt4_is_eos(chat);
    }
    printf("END\n");
    if (minigpt4_contains_eos_token("##") != 11 || minigpt4_is_eos("abc###") != 12 || minigpt4_is_eos("abc") != 0) { fprintf(stderr, "eos codes differ\n"); return 103; }
    CHECK(minigpt4_free_embedding(&embedding));
    CHECK(minigpt4_reset_chat(ctx));
    CHECK(minigpt4_free(ctx));
    free(chw);
    return 0;
}
