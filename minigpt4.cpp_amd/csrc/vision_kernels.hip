// gfx950 kernels for the image path (EVA ViT-g/14 -> ln_vision -> Q-Former -> llama_proj; reference minigpt4.cpp:2094-2363).
//   * f16 GEMM on MFMA (v_mfma_f32_32x32x16_f16): same arithmetic as ggml's f16 mul_mat -- activations rounded to fp16,
//     exact fp16 x fp16 products, fp32 accumulation -- with bias / fp16-table GELU / residual fused into the epilogue.
//   * LayerNorm with ggml_norm's double-precision statistics, eps 1e-5.
//   * fp32 attention (ViT 16 x 88, BERT 12 x 64) staged through LDS; exact-sum softmax through the fp16 exp table.
#include "kernels.hpp"

namespace mg4 {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned short f2h_bits_v(float f) { return __half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ float tab_v(const __half *t, float x) { return __half2float(t[f2h_bits_v(x)]); }

// =====================================================================================================================
// C[M][N] = A[M][K] . W[N][K]^T  (+bias, GELU, +residual).  64x64 tile per 256-thread workgroup, 4 waves of 32x32,
// BK = 32 staged through LDS (80-byte padded rows: conflict-free ds_read_b128), register prefetch of the next tile.
// =====================================================================================================================
constexpr int GB_M = 64, GB_N = 64, GB_K = 32, G_LD = GB_K + 8;

__global__ __launch_bounds__(256) void k_gemm_f16(const __half *__restrict__ A, int lda, const __half *__restrict__ W, int ldw, int M, int N, int K,
                                                  const float *__restrict__ bias, const float *residual, int gelu, const Tables tb,
                                                  float *out, __half *__restrict__ out_h, int ldo) {
    __shared__ __attribute__((aligned(16))) __half As[GB_M * G_LD];
    __shared__ __attribute__((aligned(16))) __half Ws[GB_N * G_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
    const int wm = wave >> 1, wn = wave & 1;
    // staging assignment: one 16-byte chunk of A and one of W per thread per k-tile
    const int srow = tid >> 2, schunk = tid & 3;
    const __half *a_src = A + (size_t)min(m0 + srow, M - 1) * lda + schunk * 8;
    const __half *w_src = W + (size_t)min(n0 + srow, N - 1) * ldw + schunk * 8;
    const int4 zero4 = make_int4(0, 0, 0, 0);
    int4 ra = (schunk * 8 < K) ? *reinterpret_cast<const int4 *>(a_src) : zero4;
    int4 rw = (schunk * 8 < K) ? *reinterpret_cast<const int4 *>(w_src) : zero4;
    float16_t acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    const int nk = (K + GB_K - 1) / GB_K;
    for (int kt = 0; kt < nk; kt++) {
        *reinterpret_cast<int4 *>(&As[srow * G_LD + schunk * 8]) = ra;
        *reinterpret_cast<int4 *>(&Ws[srow * G_LD + schunk * 8]) = rw;
        __syncthreads();
        if (kt + 1 < nk) {
            const int ko = (kt + 1) * GB_K + schunk * 8;
            ra = ko < K ? *reinterpret_cast<const int4 *>(a_src + (size_t)(kt + 1) * GB_K) : zero4;
            rw = ko < K ? *reinterpret_cast<const int4 *>(w_src + (size_t)(kt + 1) * GB_K) : zero4;
        }
#pragma unroll
        for (int ks = 0; ks < GB_K / 16; ks++) {
            const half8_t af = *reinterpret_cast<const half8_t *>(&As[(wm * 32 + (lane & 31)) * G_LD + ks * 16 + (lane >> 5) * 8]);
            const half8_t bf = *reinterpret_cast<const half8_t *>(&Ws[(wn * 32 + (lane & 31)) * G_LD + ks * 16 + (lane >> 5) * 8]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < N) {
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M) {
                float v = acc[r];
                if (bias) v = bv + v;
                if (gelu) v = tab_v(tb.gelu, v);
                const size_t o = (size_t)row * ldo + col;
                if (residual) v = residual[o] + v;
                if (out) out[o] = v;
                if (out_h) out_h[o] = __float2half_rn(v);
            }
        }
    }
}
void launch_gemm_f16(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                     float *out, __half *out_h, int ldo, hipStream_t s) {
    dim3 grid((unsigned)((N + GB_N - 1) / GB_N), (unsigned)((M + GB_M - 1) / GB_M));
    hipLaunchKernelGGL(k_gemm_f16, grid, dim3(256), 0, s, A, lda, W, ldw, M, N, K, bias, residual, gelu ? 1 : 0, tb, out, out_h, ldo);
}

// =====================================================================================================================
// LayerNorm: one workgroup per row
// =====================================================================================================================
__device__ __forceinline__ double block_sum_d(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b, int n, float *__restrict__ out,
                                                   __half *__restrict__ out_h) {
    __shared__ double red[4];
    const size_t row = blockIdx.x;
    const float *xr = x + row * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)xr[i];
    const float mean = (float)(block_sum_d(s, red) / (double)n);
    double s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const float v = xr[i] - mean; s2 += (double)(v * v); }
    const float variance = (float)(block_sum_d(s2, red) / (double)n);
    const float scale = 1.0f / sqrtf(variance + 1e-5f);
    for (int i = threadIdx.x; i < n; i += 256) {
        float v = (xr[i] - mean) * scale;
        v = w[i] * v;
        if (b) v = v + b[i];
        if (out) out[row * n + i] = v;
        if (out_h) out_h[row * n + i] = __float2half_rn(v);
    }
}
void launch_layernorm(const float *x, const float *w, const float *b, int rows, int n, float *out, __half *out_h, hipStream_t s) {
    hipLaunchKernelGGL(k_layernorm, dim3((unsigned)rows), dim3(256), 0, s, x, w, b, n, out, out_h);
}

// =====================================================================================================================
// fp32 attention.  Workgroup = (head, tile of 16 queries); thread = (query, 1 of 16 key/dim lanes).
// =====================================================================================================================
template <int HD>
__global__ __launch_bounds__(256) void k_attn_f32(const float *__restrict__ q, int ldq, const float *__restrict__ k, const float *__restrict__ v, int ldk, int nq, int nk,
                                                  float q_prescale, float score_div, const Tables tb, float *__restrict__ out, __half *__restrict__ out_h, int ldo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LDK = HD + 1;
    float *kv = reinterpret_cast<float *>(smem);            // [nk][HD+1]
    const int nkp = (nk + 3) & ~3;
    float *sc = kv + (size_t)nk * LDK;                        // [16][nkp]
    const int h = blockIdx.x, q0 = blockIdx.y * 16, tid = threadIdx.x;
    const int qi = tid >> 4, kl = tid & 15;
    const int qrow = min(q0 + qi, nq - 1);
    for (int e = tid; e < nk * HD; e += 256) { const int j = e / HD, i = e - j * HD; kv[j * LDK + i] = k[(size_t)j * ldk + h * HD + i]; }
    float qv[HD];
#pragma unroll
    for (int i = 0; i < HD; i++) { float t = q[(size_t)qrow * ldq + h * HD + i]; if (q_prescale != 0.0f) t *= q_prescale; qv[i] = t; }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = kl; j < nk; j += 16) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < HD; i++) s = fmaf(kv[j * LDK + i], qv[i], s);
        if (score_div != 0.0f) s = s / score_div;
        sc[qi * nkp + j] = s; mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    double sum = 0.0;
    for (int j = kl; j < nk; j += 16) { const float e = tab_v(tb.exp, sc[qi * nkp + j] - mx); sc[qi * nkp + j] = e; sum += (double)e; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = (float)(1.0 / sum);
    for (int j = kl; j < nk; j += 16) sc[qi * nkp + j] *= inv;
    __syncthreads();
    for (int e = tid; e < nk * HD; e += 256) { const int j = e / HD, i = e - j * HD; kv[j * LDK + i] = v[(size_t)j * ldk + h * HD + i]; }
    __syncthreads();
    if (q0 + qi < nq) {
        for (int d = kl; d < HD; d += 16) {
            float o = 0.0f;
            for (int j = 0; j < nk; j++) o = fmaf(kv[j * LDK + d], sc[qi * nkp + j], o);
            const size_t oo = (size_t)(q0 + qi) * ldo + h * HD + d;
            if (out) out[oo] = o;
            if (out_h) out_h[oo] = __float2half_rn(o);
        }
    }
}
void launch_attn_f32(const float *q, int ldq, const float *k, const float *v, int ldk, int nq, int nk, int heads, int hd, float q_prescale, float score_div,
                     const Tables &tb, float *out, __half *out_h, int ldo, hipStream_t s) {
    dim3 grid((unsigned)heads, (unsigned)((nq + 15) / 16));
    const size_t lds = ((size_t)nk * (hd + 1) + 16 * (size_t)((nk + 3) & ~3)) * 4;
    static bool attr_set = false;
    if (!attr_set) {   // > 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_attn_f32<88>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_attn_f32<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (hd == 88) hipLaunchKernelGGL((k_attn_f32<88>), grid, dim3(256), lds, s, q, ldq, k, v, ldk, nq, nk, q_prescale, score_div, tb, out, out_h, ldo);
    else if (hd == 64) hipLaunchKernelGGL((k_attn_f32<64>), grid, dim3(256), lds, s, q, ldq, k, v, ldk, nq, nk, q_prescale, score_div, tb, out, out_h, ldo);
    else throw HipError{hipErrorInvalidValue, "attn_f32: head size must be 88 or 64", __FILE__, __LINE__};
}

// =====================================================================================================================
// small data-movement kernels
// =====================================================================================================================
__global__ void k_im2col(const float *__restrict__ img, __half *__restrict__ patches, int ldp) {
    const int p = blockIdx.x, oh = p >> 4, ow = p & 15;
    for (int kk = threadIdx.x; kk < ldp; kk += blockDim.x) {
        float v = 0.0f;
        if (kk < 588) { const int c = kk / 196, r = kk - c * 196, kh = r / 14, kw = r - kh * 14; v = img[(size_t)c * 224 * 224 + (size_t)(oh * 14 + kh) * 224 + ow * 14 + kw]; }
        patches[(size_t)p * ldp + kk] = __float2half_rn(v);
    }
}
void launch_im2col(const float *image, __half *patches, int ldp, hipStream_t s) { hipLaunchKernelGGL(k_im2col, dim3(256), dim3(256), 0, s, image, patches, ldp); }

__global__ void k_assemble(const float *__restrict__ cls, const float *__restrict__ pe, const float *__restrict__ pos, int D, float *__restrict__ x) {
    const int r = blockIdx.x;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float base = r == 0 ? (0.0f + cls[i]) : (0.0f + pe[(size_t)(r - 1) * D + i]);
        x[(size_t)r * D + i] = base + pos[(size_t)r * D + i];
    }
}
void launch_assemble_embeddings(const float *cls, const float *pe, const float *pos, int D, float *x, hipStream_t s) {
    hipLaunchKernelGGL(k_assemble, dim3(257), dim3(256), 0, s, cls, pe, pos, D, x);
}
__global__ void k_f32_to_f16(const float *__restrict__ x, __half *__restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __float2half_rn(x[i]);
}
void launch_f32_to_f16(const float *x, __half *y, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_f32_to_f16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n); }

}  // namespace mg4
