// Host-side image front end: what the reference's OpenCV build does on the CPU before the GPU path starts --
// `cv::imread(path, IMREAD_COLOR)` + BGR->RGB (reference minigpt4.cpp:2576-2595) -- written from scratch (no OpenCV, libpng, libjpeg,
// zlib): PNG (all colour types / bit depths / Adam7), JPEG (baseline, extended-sequential and progressive Huffman; libjpeg's ISLOW IDCT,
// "fancy" chroma upsampling and fixed-point YCbCr->RGB so that the pixels equal libjpeg-turbo's; EXIF orientation applied as imread
// does), BMP and binary PPM/PGM.  Plus the coefficient tables of Pillow's 8-bit bicubic resample (PillowResize, reference :2620), which
// the HIP preprocess kernels consume.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mg4 {

struct ImageRGB8 {
    int w = 0, h = 0;
    std::vector<uint8_t> px;   // [h][w][3] RGB
};

// Decodes an in-memory file.  false + `err` on malformed / unsupported input.
bool decode_image(const uint8_t *data, size_t n, ImageRGB8 &out, std::string &err);
bool decode_png(const uint8_t *data, size_t n, ImageRGB8 &out, std::string &err);
bool decode_jpeg(const uint8_t *data, size_t n, ImageRGB8 &out, std::string &err);
// Reads and decodes a file: E_None, E_PathDoesNotExist or E_OpenImage (last_error() says why).
int load_image_file(const char *path, ImageRGB8 &out);

// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for a bicubic (a = -0.5, support 2) resize of `in_size` samples to
// `out_size`: out[xx] = clip8((2^21 + sum_{i < count[xx]} in[first[xx] + i] * kk[xx * ksize + i]) >> 22).
struct ResampleCoeffs {
    int in_size = 0, out_size = 0, ksize = 0;
    std::vector<int> first, count;   // [out_size]
    std::vector<int> kk;             // [out_size][ksize], zero padded
};
void precompute_bicubic_8bpc(int in_size, int out_size, ResampleCoeffs &c);

}  // namespace mg4
