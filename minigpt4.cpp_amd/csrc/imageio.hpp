// Host-side image front end: what the reference's OpenCV build does on the CPU before the GPU path starts --
// `cv::imread(path, IMREAD_COLOR)` + BGR->RGB (reference minigpt4.cpp:2576-2595) -- written from scratch (no OpenCV, libpng, libjpeg,
// zlib): PNG (all colour types / bit depths / Adam7), JPEG (baseline, extended-sequential and progressive Huffman; libjpeg's ISLOW IDCT,
// "fancy" chroma upsampling and fixed-point YCbCr->RGB so that the pixels equal libjpeg-turbo's; EXIF orientation applied as imread
// does), BMP and binary PPM/PGM.  Plus the coefficient tables of Pillow's 8-bit bicubic resample (PillowResize, reference :2620), which
// the HIP preprocess kernels consume.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace mg4 {

// Buffer of a decoded image / of the decoder's planes: malloc-family memory WITHOUT value-initialisation (a std::vector zero-fills: 36 MB written once more for a 12-megapixel
// photograph whose every byte the decoder writes anyway), on transparent huge pages from 4 MB on (2 MB-aligned + MADV_HUGEPAGE: a 12-megapixel decode touches ~130 MB of fresh
// memory -- 32 000 first-touch faults of 4 KB pages, which do not scale with threads, against ~65 of 2 MB), and handed to the C ABI's MiniGPT4Image without a copy (release();
// minigpt4_free_image calls free()).
template <typename T> class BigBuf {
public:
    BigBuf() = default;
    BigBuf(const BigBuf &) = delete;
    BigBuf &operator=(const BigBuf &) = delete;
    BigBuf(BigBuf &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    BigBuf &operator=(BigBuf &&o) noexcept { if (this != &o) { std::free(p_); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
    ~BigBuf() { std::free(p_); }
    void resize(size_t n);                                      // contents unspecified
    void assign(size_t n, T v) { resize(n); if (v == T()) std::memset(p_, 0, n * sizeof(T)); else std::fill(p_, p_ + n, v); }
    void clear() { std::free(p_); p_ = nullptr; n_ = 0; }
    T *data() { return p_; }
    const T *data() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
    void swap(BigBuf &o) { std::swap(p_, o.p_); std::swap(n_, o.n_); }
    T *release() { T *r = p_; p_ = nullptr; n_ = 0; return r; }  // the caller frees with free()
private:
    T *p_ = nullptr; size_t n_ = 0;
};
void *bigbuf_alloc(size_t bytes);                               // malloc, or 2 MB-aligned + MADV_HUGEPAGE from 4 MB on; throws std::bad_alloc
template <typename T> void BigBuf<T>::resize(size_t n) { clear(); p_ = static_cast<T *>(bigbuf_alloc((n ? n : 1) * sizeof(T))); n_ = n; }
using PixelBuf = BigBuf<uint8_t>;
struct ImageRGB8 {
    int w = 0, h = 0;
    PixelBuf px;   // [h][w][3] RGB
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
