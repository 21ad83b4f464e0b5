// Prefill mat-mul on the int8 matrix cores (v_mfma_i32_32x32x32_i8) for the weight types of the BASELINE configurations
// (Q4_K / Q5_K / Q6_K / Q4_0).  Same arithmetic as the decode path and as ggml (reference minigpt4.cpp:2373, 2412 -> llama_eval):
// activations quantised to Q8_K / Q8_0, exact int32 block dots -- one MFMA yields the 32-weight sub-block dots of 32 weight rows x 32
// tokens -- then the integer sub-block scales and the fp32 super-block scales, accumulated block by block in fp32.
//
// Operand mapping (activations are the MFMA "A" side so that a lane's 16 accumulators belong to ONE weight row):
//   A: lane l -> token t0 + (l & 31), 16 consecutive int8 activations at k = base + 16 (l >> 5)
//   B: lane l -> weight row r0 + (l & 31), the 16 weights at the same k: exactly the low- or high-nibble half of one 16-byte unit of
//      the repacked main plane (DESIGN.md "HBM layout"), so one dwordx4 load feeds two MFMAs
//   C: col (l & 31) = weight row, reg r -> token (r & 3) + 8 (r >> 2) + 4 (l >> 5)
#include "kernels.hpp"
#include "devutil.hpp"

namespace mg4 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float h2f_b(unsigned short h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ v4i ldv4(const void *p) { return *reinterpret_cast<const v4i *>(p); }
__device__ __forceinline__ v16i zero16() { v16i z; for (int i = 0; i < 16; i++) z[i] = 0; return z; }
__device__ __forceinline__ v4i bcast_byte(int b) { const int w = b * 0x01010101; v4i r = {w, w, w, w}; return r; }
__device__ __forceinline__ int tok_of(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

constexpr int MMQ_TT = 2;           // token tiles (of 32) per wave
#define MMQ_BOUNDS __launch_bounds__(256)

// Combine the 4 K-slices of a workgroup (fixed order: deterministic) and store.  Wave w finalises accumulator registers 4w..4w+3.
__device__ __forceinline__ void mmq_reduce_store(float (&acc)[MMQ_TT][16], int wv, int lane, int r0, int t0, int rows, int N, float *y, int ldy, const float *residual) {
    __shared__ float part[4][MMQ_TT][16][64];
#pragma unroll
    for (int tt = 0; tt < MMQ_TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) part[wv][tt][r][lane] = acc[tt][r];
    __syncthreads();
    const int hh = lane >> 5, orow = r0 + (lane & 31);
#pragma unroll
    for (int tt = 0; tt < MMQ_TT; tt++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int r = 4 * wv + rr;
            const float v = ((part[0][tt][r][lane] + part[1][tt][r][lane]) + part[2][tt][r][lane]) + part[3][tt][r][lane];
            const int tok = t0 + tt * 32 + tok_of(r, hh);
            if (orow < rows && tok < N) { const size_t o = (size_t)tok * ldy + orow; y[o] = residual ? v + residual[o] : v; }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Q4_K / Q5_K
// ---------------------------------------------------------------------------------------------------------------------
template <bool Q5>
__global__ MMQ_BOUNDS void k_mmq_q45k(const QWeight W, const ActQ A, const int N, float *y, const int ldy, const float *residual) {
    const int lane = threadIdx.x & 63, hh = lane >> 5;
    const int wv = threadIdx.x >> 6;               // the 4 waves of a workgroup split K; partial sums are combined through LDS
    const int r0 = blockIdx.x * 32;
    const int t0 = blockIdx.y * 32 * MMQ_TT;
    const int K = W.cols, U = K / 32, NSB = K / 256;
    const int row = min(r0 + (lane & 31), W.rows - 1);
    int tokc[MMQ_TT];
#pragma unroll
    for (int tt = 0; tt < MMQ_TT; tt++) tokc[tt] = min(t0 + tt * 32 + (lane & 31), N - 1);
    float acc[MMQ_TT][16];
#pragma unroll
    for (int tt = 0; tt < MMQ_TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tt][r] = 0.0f;
    struct Wsb { v4i q[4]; unsigned P[4]; v4i h; };
    auto fetch = [&](int sb, Wsb &w) {
        sb = min(sb, NSB - 1);
        const size_t g0 = (size_t)row * U + (size_t)sb * 8 + hh;
#pragma unroll
        for (int jp = 0; jp < 4; jp++) { w.q[jp] = ldv4(W.qs + (g0 + 2 * jp) * 16); w.P[jp] = Q5 ? *reinterpret_cast<const unsigned *>(W.qh + (g0 + 2 * jp) * 4) : 0u; }
        w.h = ldv4(W.sc + ((size_t)row * NSB + sb) * 16);
    };
    // the activation super-block scales of this workgroup's 64 tokens: one LDS image instead of 32 dependent global loads per super-block
    extern __shared__ float dkl[];                  // [64][NSB]
    for (int e = threadIdx.x; e < 64 * NSB; e += 256) { const int t = e / NSB, b = e - t * NSB; dkl[e] = A.dk[(size_t)min(t0 + t, N - 1) * NSB + b]; }
    __syncthreads();
    Wsb cur, nxt;
    fetch(wv, cur);
    for (int sb = wv; sb < NSB; sb += 4) {
        fetch(sb + 4, nxt);
        // activation fragments of this super-block (L2-resident): per token tile 4 x {lo, hi}
        v4i alo[MMQ_TT][4], ahi[MMQ_TT][4];
#pragma unroll
        for (int tt = 0; tt < MMQ_TT; tt++)
#pragma unroll
            for (int jp = 0; jp < 4; jp++) { const int8_t *p = A.q8k + (size_t)tokc[tt] * K + (size_t)sb * 256 + 64 * jp + 16 * hh; alo[tt][jp] = ldv4(p); ahi[tt][jp] = ldv4(p + 32); }
        // 6-bit scales / mins of this lane's weight row
        const unsigned s0 = (unsigned)cur.h[1], s1 = (unsigned)cur.h[2], s2 = (unsigned)cur.h[3];
        const unsigned scw[2] = {s0 & 0x3f3f3f3fu, (s2 & 0x0f0f0f0fu) | (((s0 >> 6) & 0x03030303u) << 4)};
        const unsigned mw[2] = {s1 & 0x3f3f3f3fu, ((s2 >> 4) & 0x0f0f0f0fu) | (((s1 >> 6) & 0x03030303u) << 4)};
        v16i isum[MMQ_TT], msum[MMQ_TT];
#pragma unroll
        for (int tt = 0; tt < MMQ_TT; tt++) { isum[tt] = zero16(); msum[tt] = zero16(); }
#pragma unroll
        for (int jp = 0; jp < 4; jp++) {
            const int sc0 = (scw[jp >> 1] >> (16 * (jp & 1))) & 0xFF, sc1 = (scw[jp >> 1] >> (16 * (jp & 1) + 8)) & 0xFF;
            const int m0 = (mw[jp >> 1] >> (16 * (jp & 1))) & 0xFF, m1 = (mw[jp >> 1] >> (16 * (jp & 1) + 8)) & 0xFF;
            const v4i q = cur.q[jp]; const unsigned P = cur.P[jp];
            v4i wlo, whi;
            wlo[0] = (q[0] & 0x0F0F0F0F) | (int)((P << 4) & 0x10101010u); wlo[1] = (q[1] & 0x0F0F0F0F) | (int)((P << 3) & 0x10101010u);
            wlo[2] = (q[2] & 0x0F0F0F0F) | (int)((P << 2) & 0x10101010u); wlo[3] = (q[3] & 0x0F0F0F0F) | (int)((P << 1) & 0x10101010u);
            whi[0] = ((q[0] >> 4) & 0x0F0F0F0F) | (int)(P & 0x10101010u); whi[1] = ((q[1] >> 4) & 0x0F0F0F0F) | (int)((P >> 1) & 0x10101010u);
            whi[2] = ((q[2] >> 4) & 0x0F0F0F0F) | (int)((P >> 2) & 0x10101010u); whi[3] = ((q[3] >> 4) & 0x0F0F0F0F) | (int)((P >> 3) & 0x10101010u);
            const v4i bm0 = bcast_byte(m0), bm1 = bcast_byte(m1);
#pragma unroll
            for (int tt = 0; tt < MMQ_TT; tt++) {
                const v16i d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo[tt][jp], wlo, zero16(), 0, 0, 0);
                const v16i d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi[tt][jp], whi, zero16(), 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) isum[tt][r] += __mul24(d0[r], sc0) + __mul24(d1[r], sc1);
                msum[tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo[tt][jp], bm0, msum[tt], 0, 0, 0);
                msum[tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi[tt][jp], bm1, msum[tt], 0, 0, 0);
            }
        }
        const float d = h2f_b((unsigned)cur.h[0] & 0xFFFF), dmin = h2f_b((unsigned)cur.h[0] >> 16);
#pragma unroll
        for (int tt = 0; tt < MMQ_TT; tt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float da = dkl[(tt * 32 + tok_of(r, hh)) * NSB + sb];
                acc[tt][r] = fmaf(d * da, (float)isum[tt][r], acc[tt][r]);
                acc[tt][r] = fmaf(-(dmin * da), (float)msum[tt][r], acc[tt][r]);
            }
        cur = nxt;
    }
    mmq_reduce_store(acc, wv, lane, r0, t0, W.rows, N, y, ldy, residual);
}

// ---------------------------------------------------------------------------------------------------------------------
// Q6_K: 16-weight sub-blocks with int8 scales -> the two 16-wide halves of an MFMA's K = 32 carry different scales, so each unit pair is
// multiplied twice with the other half zeroed.  Weights are sign-extended to q - 32 in int8.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sext6(int x) { const int yv = x ^ 0x20202020; const int s = yv & 0x20202020; return yv | (s << 1) | (s << 2); }

__global__ MMQ_BOUNDS void k_mmq_q6k(const QWeight W, const ActQ A, const int N, float *y, const int ldy, const float *residual) {
    const int lane = threadIdx.x & 63, hh = lane >> 5;
    const int wv = threadIdx.x >> 6;               // the 4 waves of a workgroup split K; partial sums are combined through LDS
    const int r0 = blockIdx.x * 32;
    const int t0 = blockIdx.y * 32 * MMQ_TT;
    const int K = W.cols, U = K / 32, NSB = K / 256;
    const int row = min(r0 + (lane & 31), W.rows - 1);
    int tokc[MMQ_TT];
#pragma unroll
    for (int tt = 0; tt < MMQ_TT; tt++) tokc[tt] = min(t0 + tt * 32 + (lane & 31), N - 1);
    float acc[MMQ_TT][16];
#pragma unroll
    for (int tt = 0; tt < MMQ_TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tt][r] = 0.0f;
    // unit index within the super-block for pair p = 2n + c: 4n + 2c + h; this lane reads h = hh and the scales of both h
    struct Wsb { v4i q[4]; unsigned Plo[4], Phi[4]; unsigned short sc_a[4], sc_b[4]; unsigned short d; };
    auto fetch = [&](int sb, Wsb &w) {
        sb = min(sb, NSB - 1);
        const size_t g0 = (size_t)row * U + (size_t)sb * 8;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const size_t g = g0 + 2 * p + hh;
            w.q[p] = ldv4(W.qs + g * 16);
            const uint2 ph = *reinterpret_cast<const uint2 *>(W.qh + g * 8); w.Plo[p] = ph.x; w.Phi[p] = ph.y;
            w.sc_a[p] = *reinterpret_cast<const unsigned short *>(W.sc + (g0 + 2 * p) * 2);        // unit h = 0: {lo scale, hi scale}
            w.sc_b[p] = *reinterpret_cast<const unsigned short *>(W.sc + (g0 + 2 * p + 1) * 2);    // unit h = 1
        }
        w.d = *reinterpret_cast<const unsigned short *>(W.d + ((size_t)row * NSB + sb) * 2);
    };
    extern __shared__ float dkl[];                  // [64][NSB] activation super-block scales of this workgroup's tokens
    for (int e = threadIdx.x; e < 64 * NSB; e += 256) { const int t = e / NSB, b = e - t * NSB; dkl[e] = A.dk[(size_t)min(t0 + t, N - 1) * NSB + b]; }
    __syncthreads();
    Wsb cur, nxt;
    fetch(wv, cur);
    const v4i z4 = {0, 0, 0, 0};
    for (int sb = wv; sb < NSB; sb += 4) {
        fetch(sb + 4, nxt);
        v4i alo[MMQ_TT][4], ahi[MMQ_TT][4];
#pragma unroll
        for (int tt = 0; tt < MMQ_TT; tt++)
#pragma unroll
            for (int p = 0; p < 4; p++) { const int n = p >> 1, c = p & 1; const int8_t *ap = A.q8k + (size_t)tokc[tt] * K + (size_t)sb * 256 + 128 * n + 32 * c + 16 * hh; alo[tt][p] = ldv4(ap); ahi[tt][p] = ldv4(ap + 64); }
        v16i isum[MMQ_TT];
#pragma unroll
        for (int tt = 0; tt < MMQ_TT; tt++) isum[tt] = zero16();
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const v4i q = cur.q[p]; const unsigned L = cur.Plo[p], H = cur.Phi[p];
            v4i wlo, whi;
            wlo[0] = sext6((q[0] & 0x0F0F0F0F) | (int)((L << 4) & 0x30303030u)); wlo[1] = sext6((q[1] & 0x0F0F0F0F) | (int)((L << 2) & 0x30303030u));
            wlo[2] = sext6((q[2] & 0x0F0F0F0F) | (int)(L & 0x30303030u)); wlo[3] = sext6((q[3] & 0x0F0F0F0F) | (int)((L >> 2) & 0x30303030u));
            whi[0] = sext6(((q[0] >> 4) & 0x0F0F0F0F) | (int)((H << 4) & 0x30303030u)); whi[1] = sext6(((q[1] >> 4) & 0x0F0F0F0F) | (int)((H << 2) & 0x30303030u));
            whi[2] = sext6(((q[2] >> 4) & 0x0F0F0F0F) | (int)(H & 0x30303030u)); whi[3] = sext6(((q[3] >> 4) & 0x0F0F0F0F) | (int)((H >> 2) & 0x30303030u));
            // halves: lanes hh == 0 carry sub-block (.., h = 0), lanes hh == 1 carry (.., h = 1)
            const v4i wlo0 = hh ? z4 : wlo, wlo1 = hh ? wlo : z4, whi0 = hh ? z4 : whi, whi1 = hh ? whi : z4;
            const int s_lo0 = (int)(signed char)(cur.sc_a[p] & 0xFF), s_hi0 = (int)(signed char)(cur.sc_a[p] >> 8);
            const int s_lo1 = (int)(signed char)(cur.sc_b[p] & 0xFF), s_hi1 = (int)(signed char)(cur.sc_b[p] >> 8);
#pragma unroll
            for (int tt = 0; tt < MMQ_TT; tt++) {
                const v16i a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo[tt][p], wlo0, zero16(), 0, 0, 0);
                const v16i a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo[tt][p], wlo1, zero16(), 0, 0, 0);
                const v16i b0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi[tt][p], whi0, zero16(), 0, 0, 0);
                const v16i b1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi[tt][p], whi1, zero16(), 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) isum[tt][r] += __mul24(a0[r], s_lo0) + __mul24(a1[r], s_lo1) + __mul24(b0[r], s_hi0) + __mul24(b1[r], s_hi1);
            }
        }
        const float d = h2f_b(cur.d);
#pragma unroll
        for (int tt = 0; tt < MMQ_TT; tt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                acc[tt][r] = fmaf(d * dkl[(tt * 32 + tok_of(r, hh)) * NSB + sb], (float)isum[tt][r], acc[tt][r]);
            }
        cur = nxt;
    }
    mmq_reduce_store(acc, wv, lane, r0, t0, W.rows, N, y, ldy, residual);
}

// ---------------------------------------------------------------------------------------------------------------------
// Q4_0: one MFMA per 32-weight block (lanes hh = 0 / 1 take the low / high nibbles of the same unit), fp16 block scale x fp16-rounded
// activation scale applied per block.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sext4m8(int x) { const int yv = x ^ 0x08080808; return yv | ((yv & 0x08080808) * 30); }   // nibble - 8 as int8

__global__ MMQ_BOUNDS void k_mmq_q40(const QWeight W, const ActQ A, const int N, float *y, const int ldy, const float *residual) {
    const int lane = threadIdx.x & 63, hh = lane >> 5;
    const int wv = threadIdx.x >> 6;               // the 4 waves of a workgroup split K; partial sums are combined through LDS
    const int r0 = blockIdx.x * 32;
    const int t0 = blockIdx.y * 32 * MMQ_TT;
    const int K = W.cols, NB = K / 32;
    const int row = min(r0 + (lane & 31), W.rows - 1);
    int tokc[MMQ_TT];
#pragma unroll
    for (int tt = 0; tt < MMQ_TT; tt++) tokc[tt] = min(t0 + tt * 32 + (lane & 31), N - 1);
    float acc[MMQ_TT][16];
#pragma unroll
    for (int tt = 0; tt < MMQ_TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tt][r] = 0.0f;
    constexpr int BB = 4;   // blocks per iteration
    struct Wb { v4i q[BB]; unsigned short d[BB]; };
    auto fetch = [&](int b0, Wb &w) {
#pragma unroll
        for (int i = 0; i < BB; i++) { const size_t g = (size_t)row * NB + min(b0 + i, NB - 1); w.q[i] = ldv4(W.qs + g * 16); w.d[i] = *reinterpret_cast<const unsigned short *>(W.sc + g * 2); }
    };
    Wb cur, nxt;
    fetch(wv * BB, cur);
    for (int b0 = wv * BB; b0 < NB; b0 += 4 * BB) {
        fetch(b0 + 4 * BB, nxt);
#pragma unroll
        for (int i = 0; i < BB; i++) {
            const int b = b0 + i;
            const bool live = b < NB;
            const int bc = live ? b : NB - 1;
            const v4i q = cur.q[i];
            v4i wq;
#pragma unroll
            for (int e = 0; e < 4; e++) wq[e] = sext4m8(hh ? ((q[e] >> 4) & 0x0F0F0F0F) : (q[e] & 0x0F0F0F0F));
            const float dw = live ? h2f_b(cur.d[i]) : 0.0f;
#pragma unroll
            for (int tt = 0; tt < MMQ_TT; tt++) {
                const v4i a = ldv4(A.q80 + (size_t)tokc[tt] * K + (size_t)bc * 32 + 16 * hh);
                const v16i dd = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, wq, zero16(), 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int tok = min(t0 + tt * 32 + tok_of(r, hh), N - 1);
                    acc[tt][r] = fmaf(dw * A.d0[(size_t)tok * NB + bc], (float)dd[r], acc[tt][r]);
                }
            }
        }
        cur = nxt;
    }
    mmq_reduce_store(acc, wv, lane, r0, t0, W.rows, N, y, ldy, residual);
}

bool mmq_supported(int type) { return type == GT_Q4_K || type == GT_Q5_K || type == GT_Q6_K || type == GT_Q4_0; }

void launch_mmq(const QWeight &W, const ActQ &A, int N, float *y, int ldy, const float *residual, hipStream_t s) {
    const dim3 grid((unsigned)((W.rows + 31) / 32), (unsigned)((N + 32 * MMQ_TT - 1) / (32 * MMQ_TT))), block(256);
    const size_t dk_lds = (size_t)64 * (W.cols / 256) * 4;   // 32 * MMQ_TT tokens x super-blocks
    switch (W.type) {
    case GT_Q4_K: hipLaunchKernelGGL((k_mmq_q45k<false>), grid, block, dk_lds, s, W, A, N, y, ldy, residual); break;
    case GT_Q5_K: hipLaunchKernelGGL((k_mmq_q45k<true>), grid, block, dk_lds, s, W, A, N, y, ldy, residual); break;
    case GT_Q6_K: hipLaunchKernelGGL(k_mmq_q6k, grid, block, dk_lds, s, W, A, N, y, ldy, residual); break;
    case GT_Q4_0: hipLaunchKernelGGL(k_mmq_q40, grid, block, 0, s, W, A, N, y, ldy, residual); break;
    default: throw HipError{hipErrorInvalidValue, "mmq: unsupported type", __FILE__, __LINE__};
    }
}

}  // namespace mg4
