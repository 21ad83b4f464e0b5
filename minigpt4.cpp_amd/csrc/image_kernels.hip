// gfx950 kernels of minigpt4_preprocess_image (reference minigpt4.cpp:2597-2651, OpenCV build): Pillow's two-pass 8-bit bicubic resample
// (PillowResize, :2620) + 1/255 scaling + CLIP mean/std normalisation + HWC -> CHW (:2621-2634).
//
// Both passes are integer arithmetic on Pillow's 22-bit fixed-point coefficients (imageio.cpp precompute_bicubic_8bpc), so the resized
// bytes are bit-identical to Pillow's; the float tail is three correctly rounded fp32 operations per value (no contraction).
// HBM-bound byte work: pass 1 reads the source image once ([H][W][3] u8, consecutive threads = consecutive output columns of one row, whose
// taps overlap and are served by L1/L2) and writes [H][224][3]; pass 2 reads that with fully coalesced rows and writes 3 x 224 x 224 floats.
#include "kernels.hpp"

namespace mg4 {

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;
__device__ __forceinline__ int clip8(int acc) { const int v = acc >> RS_PRECISION_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

// dst[y][xx][c] = clip8(2^21 + sum_i src[y][first[xx] + i][c] * kk[xx][i])
__global__ __launch_bounds__(256) void k_resample_h(const uint8_t *__restrict__ src, const int W, const int *__restrict__ first, const int *__restrict__ count,
                                                    const int *__restrict__ kk, const int ksize, uint8_t *__restrict__ dst, const int OW) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xx >= OW) return;
    const int x0 = first[xx], n = count[xx];
    const int *k = kk + (size_t)xx * ksize;
    const uint8_t *p = src + ((size_t)y * W + x0) * 3;
    int r = 1 << (RS_PRECISION_BITS - 1), g = r, b = r;
    for (int i = 0; i < n; i++, p += 3) { const int c = k[i]; r += (int)p[0] * c; g += (int)p[1] * c; b += (int)p[2] * c; }
    uint8_t *q = dst + ((size_t)y * OW + xx) * 3;
    q[0] = (uint8_t)clip8(r); q[1] = (uint8_t)clip8(g); q[2] = (uint8_t)clip8(b);
}

// v[yy][x][c] = clip8(2^21 + sum_i src[first[yy] + i][x][c] * kk[yy][i]);  out[c][yy][x] = (float(v) * (1/255) - mean[c]) / std[c]
__global__ __launch_bounds__(256) void k_resample_v_norm(const uint8_t *__restrict__ src, const int OW, const int *__restrict__ first, const int *__restrict__ count,
                                                         const int *__restrict__ kk, const int ksize, float *__restrict__ out, const int OH, const float m0, const float m1,
                                                         const float m2, const float s0, const float s1, const float s2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y;
    if (x >= OW) return;
    const int y0 = first[yy], n = count[yy];
    const int *k = kk + (size_t)yy * ksize;
    const uint8_t *p = src + ((size_t)y0 * OW + x) * 3;
    int r = 1 << (RS_PRECISION_BITS - 1), g = r, b = r;
    for (int i = 0; i < n; i++, p += (size_t)OW * 3) { const int c = k[i]; r += (int)p[0] * c; g += (int)p[1] * c; b += (int)p[2] * c; }
    const float a = 1.0f / 255.0f;
    const size_t plane = (size_t)OH * OW, o = (size_t)yy * OW + x;
    out[o] = __fdiv_rn(__fsub_rn(__fmul_rn((float)clip8(r), a), m0), s0);
    out[plane + o] = __fdiv_rn(__fsub_rn(__fmul_rn((float)clip8(g), a), m1), s1);
    out[2 * plane + o] = __fdiv_rn(__fsub_rn(__fmul_rn((float)clip8(b), a), m2), s2);
}

void launch_resample_h(const uint8_t *src, int W, int H, const int *first, const int *count, const int *kk, int ksize, uint8_t *dst, int OW, hipStream_t s) {
    hipLaunchKernelGGL(k_resample_h, dim3((unsigned)((OW + 255) / 256), (unsigned)H), dim3(256), 0, s, src, W, first, count, kk, ksize, dst, OW);
}
void launch_resample_v_norm(const uint8_t *src, int OW, const int *first, const int *count, const int *kk, int ksize, float *out, int OH, const float mean[3], const float std[3],
                            hipStream_t s) {
    hipLaunchKernelGGL(k_resample_v_norm, dim3((unsigned)((OW + 255) / 256), (unsigned)OH), dim3(256), 0, s, src, OW, first, count, kk, ksize, out, OH, mean[0], mean[1], mean[2], std[0],
                       std[1], std[2]);
}

}  // namespace mg4
