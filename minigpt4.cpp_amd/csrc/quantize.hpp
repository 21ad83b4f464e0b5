// minigpt4_quantize_model: ggml's reference block quantisers + the vision-file rewriter (host only).  See quantize.cpp.
#pragma once
#include <cstddef>
#include <cstdint>

namespace mg4 {

bool quantize_supported(int ggml_type);                                            // Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q4_K Q5_K Q6_K
// ggml_quantize_chunk: n floats (a whole number of blocks) -> blocks at dst; returns the bytes written (0: unsupported type / ragged n)
size_t quantize_chunk(int ggml_type, const float *x, uint8_t *dst, size_t n);
// reference minigpt4_quantize_model (minigpt4.cpp:2817-2982); returns a MiniGPT4Error
int quantize_vision_file(const char *in_path, const char *out_path, int mg4_data_type);
// Q3_K -> Q6_K, lossless (load time): both formats are d * scale_16 * q over sixteen 16-wide sub-blocks, so q6 = q3 (in [-4, 3]), int8 scale = scale6 - 32 and
// the same fp16 d reproduce every dequantised value and every integer block dot product of ggml's Q3_K arithmetic; the gfx950 kernels then run their Q6_K path.
// src: n_blocks x 110 bytes (block_q3_K), dst: n_blocks x 210 bytes (block_q6_K); spread over host threads.
void q3k_to_q6k(const uint8_t *src, uint8_t *dst, size_t n_blocks);

}  // namespace mg4
