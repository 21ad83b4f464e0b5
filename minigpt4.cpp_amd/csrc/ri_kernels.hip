// Batched decode (B = 2..4 conversations per weight pass, SURVEY.md 8f-1 / BASELINE.json configs[3]: 4 requests per replica) on the int8 matrix cores
// -- round 5.
//
// The multi-row mat-vec of rounds 2-4 (k_matvec_tn, llm_kernels.hip) keeps the single-row kernel's lane map -- 64 lanes share ONE weight row, a lane
// owns a 32-weight unit -- and multiplies every unit against each of the B activation rows with v_dot4_i32_i8: 8 dot instructions per row and unit,
// vector-issue bound at B = 4 (84 % VALU busy, weights at 2 TB/s).  v_mfma_i32_4x4x4_16B_i8 computes 16 independent 4 x 4 x 4 products per
// instruction; with the TOKENS as the 4 rows of the A operand (the same in all 16 blocks) and 64 different WEIGHT ROWS as the columns of the B
// operand, one instruction multiplies 4 consecutive weights of 64 rows against up to 4 tokens -- the work of four v_dot4 per lane -- and accumulates
// it in the lane's own four registers (register r = token r).  That needs lane = weight ROW, i.e. loads of 16 bytes per row at a stride of one row;
// so the k-quant matrices get a second, ROW-INTERLEAVED image (built on the device from the ordinary planes when a context is given more than one
// conversation: Engine::build_ri_planes): for every group of 64 rows and every unit the 64 rows' 16-byte pieces back to back (1 KiB per wave load),
// the same for the high-bit words and the per-super-block headers.  Micro-benchmark on the 13B w1|w3 set (profiles/r05_batched_decode_mfma.log): 22
// us per launch at B = 2, 3 and 4 against 25.6 / 29.7 / 33.8 us for k_matvec_tn.
//
// Arithmetic: ggml's (reference minigpt4.cpp:2373 -> llama_eval -> ggml_mul_mat, k-quant x Q8_K): exact int32 sub-block dots (the MFMA's integer
// accumulation), integer sub-block scales, one fp32 update per super-block with d_w * d_a.  Per output the super-blocks of a K quarter are added in
// order and the four quarters (the four waves of a workgroup) in wave order: a fixed order, but not k_matvec_tn's -- results differ from it in the
// last bits like any two summation orders.
//   Q4_K / Q5_K: min term sum_j m_j * bsum_j as four more MFMAs per super-block on the digit split bsum = 128 hi + lo (the quantiser's bsq plane).
//   Q6_K: weights enter as their unsigned 6-bit codes; the -32 offset is sum_g sc_g * bsum16_g, eight MFMAs per super-block on the digit split of the
//   16-element sums.
#include "kernels.hpp"
#include "devutil.hpp"
#include "qtraits.hpp"

#include <algorithm>

namespace mg4 {

typedef int v4i_r __attribute__((ext_vector_type(4)));

// ---- the row-interleaved image ---------------------------------------------------------------------------------------------------------------------
// G = rows / 64 groups, U = K / 32 units, NSB = K / 256 super-blocks:
//   qs [G][U][64][16]                       every type
//   qh [G][U][64][4] (Q5_K: pack_hb1 word)  /  [G][U][64][8] (Q6_K: Plo, Phi)
//   sc [G][NSB][64][16]                     Q4_K / Q5_K: {d, dmin, 12 packed scale / min bytes};  Q6_K: the super-block's 8 x {lo, hi} int8 scales
//   d  [G][NSB][64][2]                      Q6_K: fp16 d
bool ri_supported(int type, int rows, int cols) { return (type == GT_Q4_K || type == GT_Q5_K || type == GT_Q6_K) && rows % 64 == 0 && cols % 256 == 0 && cols <= 16384; }
size_t ri_plan(int type, int rows, int cols, RiPlanes &p, uint8_t *base) {
    p = RiPlanes{};
    if (!ri_supported(type, rows, cols)) return 0;
    const size_t G = (size_t)rows / 64, U = (size_t)cols / 32, NSB = (size_t)cols / 256;
    size_t off = 0;
    auto take = [&](size_t bytes) { const uint8_t *q = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return q; };
    p.qs = take(G * U * 1024);
    if (type == GT_Q5_K) p.qh = take(G * U * 256);
    if (type == GT_Q6_K) p.qh = take(G * U * 512);
    p.sc = take(G * NSB * 1024);
    if (type == GT_Q6_K) p.d = take(G * NSB * 128);
    return off;
}
__global__ __launch_bounds__(256) void k_ri_build(const QWeight w, const RiPlanes p, const size_t n_units) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;              // (row, unit) of the ordinary planes
    if (i >= n_units) return;
    const int U = w.cols / 32, NSB = w.cols / 256;
    const size_t row = i / U; const int u = (int)(i - row * U);
    const size_t g = row >> 6; const int l = (int)(row & 63);
    uint8_t *qs = const_cast<uint8_t *>(p.qs), *qh = const_cast<uint8_t *>(p.qh), *sc = const_cast<uint8_t *>(p.sc), *dd = const_cast<uint8_t *>(p.d);
    *reinterpret_cast<v4i_r *>(qs + ((g * U + u) * 64 + l) * 16) = *reinterpret_cast<const v4i_r *>(w.qs + i * 16);
    if (w.type == GT_Q5_K) *reinterpret_cast<unsigned *>(qh + ((g * U + u) * 64 + l) * 4) = *reinterpret_cast<const unsigned *>(w.qh + i * 4);
    if (w.type == GT_Q6_K) *reinterpret_cast<uint2 *>(qh + ((g * U + u) * 64 + l) * 8) = *reinterpret_cast<const uint2 *>(w.qh + i * 8);
    if ((u & 7) == 0) {
        const int sb = u >> 3;
        if (w.type == GT_Q6_K) {
            *reinterpret_cast<v4i_r *>(sc + ((g * NSB + sb) * 64 + l) * 16) = *reinterpret_cast<const v4i_r *>(w.sc + i * 2);      // 8 units x 2 int8, contiguous in the ordinary plane
            *reinterpret_cast<unsigned short *>(dd + ((g * NSB + sb) * 64 + l) * 2) = *reinterpret_cast<const unsigned short *>(w.d + (i >> 3) * 2);
        } else *reinterpret_cast<v4i_r *>(sc + ((g * NSB + sb) * 64 + l) * 16) = *reinterpret_cast<const v4i_r *>(w.sc + (i >> 3) * 16);
    }
}
void launch_ri_build(const QWeight &w, const RiPlanes &p, hipStream_t s) {
    const size_t n = (size_t)w.rows * (w.cols / 32);
    hipLaunchKernelGGL(k_ri_build, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, p, n);
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------------------------------------
struct RiMat { RiPlanes p; float *y; const float *res; };
struct RiArgs { RiMat m[3]; int n_mat, groups_each, rows_each, K, N, ldy; const float *px, *pw; int ldx;
                int n_a;            // k_matvec_ri_mix: the first n_a matrices are of type TA, the others of type TB
                // ksplit > 1 (matrices with few row groups and a long K: the 13B w2 has 80 groups x 54 super-blocks): `ksplit` workgroups share a row
                // group, each a contiguous K range; their partial sums go to `slabs` [group][part][4][64] and the LAST one to arrive (ticket per
                // group, self-resetting) adds them in part order
                int ksplit; float *slabs; unsigned *tickets; };   // px: rows prepared inside the launch (PRO): rms_norm(px_t) * pw, quantised

__device__ __forceinline__ void ri_scale_min_words(const v4i_r &h, unsigned &scw0, unsigned &scw1, unsigned &mw0, unsigned &mw1) {
    const unsigned s0 = (unsigned)h[1], s1 = (unsigned)h[2], s2 = (unsigned)h[3];
    scw0 = s0 & 0x3f3f3f3fu; scw1 = (s2 & 0x0f0f0f0fu) | (((s0 >> 6) & 0x03030303u) << 4);
    mw0 = s1 & 0x3f3f3f3fu; mw1 = ((s2 >> 4) & 0x0f0f0f0fu) | (((s1 >> 6) & 0x03030303u) << 4);
}
__device__ __forceinline__ float ri_h2f(unsigned short h) { return __half2float(__ushort_as_half(h)); }

// One super-block of one row group as a lane holds it: 8 units of ITS row (16 B each), the high bits, the 16-byte scale header (Q6_K: 16 int8 scales)
// and Q6_K's fp16 d (one type for every format -- only the words a format uses are ever live -- so that the mixed launch can carry ONE prefetched
// image across its prologue)
struct RiRaw { v4i_r q[8]; unsigned p[16]; v4i_r h; unsigned short d; };
template <int T>
__device__ __forceinline__ void ri_fetch(const uint8_t *pq, const uint8_t *pp, const uint8_t *ph, const uint8_t *pd, int sb, int NSB, RiRaw &r) {
    constexpr bool Q6 = T == GT_Q6_K, Q5 = T == GT_Q5_K;
    const int sbc = min(sb, NSB - 1);               // every load unconditional (clamped super-block): counted waits
#pragma unroll
    for (int u = 0; u < 8; u++) {
        r.q[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i_r *>(pq + (size_t)(sbc * 8 + u) * 1024));
        if (Q5) r.p[u] = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(pp + (size_t)(sbc * 8 + u) * 256));
        if (Q6) { typedef unsigned v2u_r __attribute__((ext_vector_type(2)));
            const v2u_r w2 = __builtin_nontemporal_load(reinterpret_cast<const v2u_r *>(pp + (size_t)(sbc * 8 + u) * 512)); r.p[Q6 ? 2 * u : 0] = w2.x; r.p[Q6 ? 2 * u + 1 : 0] = w2.y; }
    }
    r.h = __builtin_nontemporal_load(reinterpret_cast<const v4i_r *>(ph + (size_t)sbc * 1024));
    if (Q6) r.d = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(pd + (size_t)sbc * 128));
}
// acc[t] += (this lane's weight row, super-block sb) . (token row t); qa / da: the LDS image of the token row this lane feeds the A operand from (row
// lane & 3)
template <int T>
__device__ __forceinline__ void ri_consume(const int8_t *qa, const int8_t *da, const float *dk, int NSB, int sb, const RiRaw &r, float (&acc)[4]) {
    constexpr bool Q6 = T == GT_Q6_K, Q5 = T == GT_Q5_K;
    int isum[4] = {0, 0, 0, 0};
    unsigned scw0 = 0, scw1 = 0, mw0 = 0, mw1 = 0;
    if (!Q6) ri_scale_min_words(r.h, scw0, scw1, mw0, mw1);
#pragma unroll
    for (int u = 0; u < 8; u++) {
        // activation bytes of the unit's low / high nibbles (Tr<T>::loada): Q4_K / Q5_K: 64 j + 16 h and + 32;  Q6_K: 128 n + 32 c + 16 h and + 64
        const int off = Q6 ? 128 * (u >> 2) + 32 * ((u >> 1) & 1) + 16 * (u & 1) : 64 * (u >> 1) + 16 * (u & 1);
        const v4i_r alo = *reinterpret_cast<const v4i_r *>(qa + sb * 256 + off), ahi = *reinterpret_cast<const v4i_r *>(qa + sb * 256 + off + (Q6 ? 64 : 32));
        v4i_r D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const unsigned q = (unsigned)r.q[u][d];
            unsigned wlo = q & 0x0F0F0F0Fu, whi = (q >> 4) & 0x0F0F0F0Fu;
            if (Q5) { const unsigned P = r.p[Q5 ? u : 0];
                wlo |= (d == 0 ? P << 4 : d == 1 ? P << 3 : d == 2 ? P << 2 : P << 1) & 0x10101010u; whi |= (d == 0 ? P : d == 1 ? P >> 1 : d == 2 ? P >> 2 : P >> 3) & 0x10101010u; }
            if (Q6) { const unsigned L = r.p[Q6 ? 2 * u : 0], H = r.p[Q6 ? 2 * u + 1 : 0];
                wlo |= (d == 0 ? L << 4 : d == 1 ? L << 2 : d == 2 ? L : L >> 2) & 0x30303030u; whi |= (d == 0 ? H << 4 : d == 1 ? H << 2 : d == 2 ? H : H >> 2) & 0x30303030u; }
            D0 = __builtin_amdgcn_mfma_i32_4x4x4i8(alo[d], (int)wlo, D0, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_i32_4x4x4i8(ahi[d], (int)whi, D1, 0, 0, 0);
        }
        int sc0, sc1;
        if (Q6) { const unsigned w16 = ((unsigned)r.h[u >> 1] >> (16 * (u & 1))) & 0xFFFFu; sc0 = (int)(signed char)(w16 & 0xFF); sc1 = (int)(signed char)(w16 >> 8); }
        else { const int j = u >> 1; const unsigned scw = (j & 2) ? scw1 : scw0; sc0 = (int)(scw >> (16 * (j & 1))) & 0xFF; sc1 = (int)(scw >> (16 * (j & 1) + 8)) & 0xFF; }
#pragma unroll
        for (int t = 0; t < 4; t++) isum[t] += __mul24(sc0, D0[t]) + __mul24(sc1, D1[t]);
    }
    if (!Q6) {
        // min term: sum_j m_j * bsum_j on the digit split of the per-32 sums
        const v4i_r dgt = *reinterpret_cast<const v4i_r *>(da + sb * 16);
        v4i_r Ml = {0, 0, 0, 0}, Mh = {0, 0, 0, 0};
        Ml = __builtin_amdgcn_mfma_i32_4x4x4i8(dgt[0], (int)mw0, Ml, 0, 0, 0); Ml = __builtin_amdgcn_mfma_i32_4x4x4i8(dgt[1], (int)mw1, Ml, 0, 0, 0);
        Mh = __builtin_amdgcn_mfma_i32_4x4x4i8(dgt[2], (int)mw0, Mh, 0, 0, 0); Mh = __builtin_amdgcn_mfma_i32_4x4x4i8(dgt[3], (int)mw1, Mh, 0, 0, 0);
        const float d = ri_h2f((unsigned short)((unsigned)r.h[0] & 0xFFFF)), dmin = ri_h2f((unsigned short)((unsigned)r.h[0] >> 16));
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float dkt = dk[t * NSB + sb];
            acc[t] = fmaf(d * dkt, (float)isum[t], acc[t]);
            acc[t] = fmaf(-(dmin * dkt), (float)(Mh[t] * 128 + Ml[t]), acc[t]);
        }
    } else {
        // the -32 offset of the 6-bit codes: sum over the 16 scale groups of sc_g * bsum16_g, digit split (scale bytes and digit bytes are in the
        // same order)
        const v4i_r dl = *reinterpret_cast<const v4i_r *>(da + sb * 32), dh = *reinterpret_cast<const v4i_r *>(da + sb * 32 + 16);
        v4i_r Cl = {0, 0, 0, 0}, Ch = {0, 0, 0, 0};
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) { Cl = __builtin_amdgcn_mfma_i32_4x4x4i8(dl[k4], r.h[k4], Cl, 0, 0, 0); Ch = __builtin_amdgcn_mfma_i32_4x4x4i8(dh[k4], r.h[k4], Ch, 0, 0, 0); }
        const float d = ri_h2f(r.d);
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = fmaf(d * dk[t * NSB + sb], (float)(isum[t] - 32 * (Ch[t] * 128 + Cl[t])), acc[t]);
    }
}
// super-blocks [sb0, sb1) of row group gl of one matrix, two-stage register pipeline; `cur` already holds super-block sb0 (ri_fetch issued by the
// caller: for a workgroup's first task BEFORE it stages the activation image, so that the first weight bytes are in flight during the prologue)
struct RiPtr { const uint8_t *pq, *pp, *ph, *pd; };
template <int T>
__device__ __forceinline__ RiPtr ri_ptrs(const RiPlanes &P, int gl, int U, int NSB, int lane) {
    constexpr bool Q6 = T == GT_Q6_K;
    return RiPtr{P.qs + (size_t)gl * U * 1024 + lane * 16, P.qh + (size_t)gl * U * (Q6 ? 512 : 256) + lane * (Q6 ? 8 : 4), P.sc + (size_t)gl * NSB * 1024 + lane * 16, P.d + (size_t)gl * NSB * 128 + lane * 2};
}
template <int T>
__device__ __forceinline__ void ri_stream(const RiPtr &p, int NSB, int sb0, int sb1, const int8_t *qa, const int8_t *da, const float *dk, float (&acc)[4], RiRaw &cur) {
    RiRaw nxt;
    for (int sb = sb0; sb < sb1;) {
        ri_fetch<T>(p.pq, p.pp, p.ph, p.pd, sb + 1, NSB, nxt);
        __builtin_amdgcn_sched_barrier(0);
        ri_consume<T>(qa, da, dk, NSB, sb, cur, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (++sb >= sb1) break;
        ri_fetch<T>(p.pq, p.pp, p.ph, p.pd, sb + 1, NSB, cur);
        __builtin_amdgcn_sched_barrier(0);
        ri_consume<T>(qa, da, dk, NSB, sb, nxt, acc);
        __builtin_amdgcn_sched_barrier(0);
        ++sb;
    }
}
// the <= 4 quantised rows -> LDS, K range [c_sb0, c_sb1) super-blocks; the loads of a batch are all issued before its first LDS store (one memory
// round trip per batch instead of one per 16 bytes and thread)
__device__ __forceinline__ void ri_stage_rows(const ActQ &A, int8_t *q8, int N, int K, int c_sb0, int c_sb1, int NT) {
    constexpr int SB = 6;
    const int cK = (c_sb1 - c_sb0) * 256, lim = 4 * cK, step = NT * 16;
    for (int base = threadIdx.x * 16; base < lim; base += step * SB) {
        v4i_r v[SB];
#pragma unroll
        for (int j = 0; j < SB; j++) {
            const int i = base + j * step;
            const v4i_r z = {0, 0, 0, 0}; v[j] = z;
            if (i < lim) { const int t = i / cK, e = c_sb0 * 256 + (i - t * cK); if (t < N) v[j] = *reinterpret_cast<const v4i_r *>(A.q8k + (size_t)t * K + e); }
        }
#pragma unroll
        for (int j = 0; j < SB; j++) {
            const int i = base + j * step;
            if (i < lim) { const int t = i / cK, e = c_sb0 * 256 + (i - t * cK); *reinterpret_cast<v4i_r *>(q8 + (size_t)t * K + e) = v[j]; }
        }
    }
}
// the digit image of the rows' block sums for type T (header below), super-blocks [c_sb0, c_sb1); dk != null: the rows' Q8_K scales too.  NT =
// threads of the workgroup
template <int T>
__device__ __forceinline__ void ri_stage_digits(const ActQ &A, int8_t *dg, float *dk, int N, int NSB, int K, int c_sb0, int c_sb1, int NT) {
    constexpr bool Q6 = T == GT_Q6_K;
    for (int i0 = threadIdx.x; i0 < 4 * (c_sb1 - c_sb0); i0 += NT) {
        const int t = i0 / (c_sb1 - c_sb0), sb = c_sb0 + (i0 - t * (c_sb1 - c_sb0)), i = t * NSB + sb;
        float d = 0.0f;
        if (!Q6) {
            v4i_r v = {0, 0, 0, 0};
            if (t < N) { v = *reinterpret_cast<const v4i_r *>(A.bsq + ((size_t)t * NSB + sb) * 16); d = A.dk[(size_t)t * NSB + sb]; }
            *reinterpret_cast<v4i_r *>(dg + (size_t)i * 16) = v;
        } else {
            int8_t *o = dg + (size_t)i * 32;
            if (t < N) d = A.dk[(size_t)t * NSB + sb];
#pragma unroll
            for (int ui = 0; ui < 8; ui++) {
                const int n = ui >> 2, c = (ui >> 1) & 1, h = ui & 1, grp = 8 * n + 2 * c + h;
                const int s_lo = t < N ? (int)A.bsk[(size_t)t * (K / 16) + sb * 16 + grp] : 0, s_hi = t < N ? (int)A.bsk[(size_t)t * (K / 16) + sb * 16 + grp + 4] : 0;
                o[2 * ui] = (int8_t)(s_lo & 127); o[2 * ui + 1] = (int8_t)(s_hi & 127); o[16 + 2 * ui] = (int8_t)(s_lo >> 7); o[16 + 2 * ui + 1] = (int8_t)(s_hi >> 7);
            }
        }
        if (dk) dk[i] = d;
    }
}

// LDS image of the <= 4 activation rows (rows >= N are zero):  q8 [4][K]  |  dg [4][NSB][DG]  |  dk [4][NSB]  |  red [WPB][4][64]
//   DG = 16 (Q4_K / Q5_K): the quantiser's digit-split per-32 sums (bytes 0..7 low digits of sub-blocks 0..7, 8..15 high digits)
//   DG = 32 (Q6_K): the 16-element sums of the super-block in the ORDER OF THE SCALE BYTES (unit i = (n, c, h): byte 2 i <-> group 8 n + 2 c + h,
//   byte 2 i + 1 <-> that + 4),
//                   bytes 0..15 low digits (s & 127), 16..31 high digits (s >> 7) PRO: the rows are prepared inside the launch -- rms_norm(x_t) * w
//                   (ggml_rms_norm: fp32 squares summed in double, eps 1e-6) and ggml's Q8_K quantisation, the arithmetic of k_rms_quant -- by every
//                   workgroup into its own LDS image (the image is needed there anyway): one standalone preparation launch less in front of wq|wk|wv,
//                   w1|w3 and the output matrix (~5 us each against ~1 us of redundant work per workgroup).
template <int T, int WPB, bool PRO>
__global__ __launch_bounds__(64 * WPB, 2) void k_matvec_ri(const RiArgs a, const ActQ A) {
    constexpr bool Q6 = T == GT_Q6_K;
    constexpr int DG = Q6 ? 32 : 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ri[];
    const int K = a.K, U = K / 32, NSB = K / 256, N = a.N;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, t4 = lane & 3;
    int8_t *q8 = reinterpret_cast<int8_t *>(smem_ri);
    int8_t *dg = q8 + 4 * K;
    float *dk = reinterpret_cast<float *>(dg + 4 * NSB * DG);
    float *red = dk + 4 * NSB;
    int16_t *bsl = reinterpret_cast<int16_t *>(red + WPB * 4 * 64);       // PRO: the rows' 16-element sums [4][K / 16] (quantiser output; Q6_K's digit image is derived from them)
    const int S = a.ksplit > 1 ? a.ksplit : 1;
    const int total_groups = a.n_mat * a.groups_each;
    // task -> (row group g of matrix m, K part, this wave's super-block range)
    struct Task { int g, m, gl, sb0, sb1; };
    auto task_of = [&](int task) {
        Task k; k.g = task / S; const int part = task - k.g * S;
        const int psb0 = (int)((long long)NSB * part / S), psb1 = (int)((long long)NSB * (part + 1) / S);       // this workgroup's K range
        const int sb_per = (psb1 - psb0 + WPB - 1) / WPB; k.sb0 = psb0 + wv * sb_per; k.sb1 = min(psb1, k.sb0 + sb_per);
        k.m = k.g / a.groups_each; k.gl = k.g - k.m * a.groups_each;
        return k;
    };
    RiRaw cur;
    { const Task k = task_of((int)blockIdx.x); const RiPtr p = ri_ptrs<T>(a.m[k.m].p, k.gl, U, NSB, lane); ri_fetch<T>(p.pq, p.pp, p.ph, p.pd, k.sb0, NSB, cur); }   // (grid <= tasks)
    if constexpr (PRO) {
        constexpr int NT = 64 * WPB;
        ActQ L{}; L.q8k = q8; L.dk = dk; L.bsk = bsl; L.bsq = Q6 ? nullptr : dg;
        double *redd = reinterpret_cast<double *>(red);
        for (int t = 0; t < 4; t++) {
            if (t < N) {                                                      // workgroup-uniform
                const float *xr = a.px + (size_t)t * a.ldx;
                const bool norm = a.pw != nullptr;                            // workgroup-uniform.  false (round 6): the rows are taken as they are and only quantised (the attention
                float scale = 1.0f;                                           // output in front of wo: no norm, no double-precision sum)
                if (norm) {
                    double sum = 0.0;
                    for (int i = threadIdx.x * 4; i < K; i += NT * 4) { const float4 v = *reinterpret_cast<const float4 *>(xr + i);
                        double q = 0.0; q += (double)(v.x * v.x); q += (double)(v.y * v.y); q += (double)(v.z * v.z); q += (double)(v.w * v.w); sum += q; }
                    sum = wave_sum_d(sum);
                    if (lane == 0) redd[wv] = sum;
                    __syncthreads();
                    double tot = 0.0;
                    for (int w = 0; w < WPB; w++) tot += redd[w];
                    __syncthreads();
                    scale = 1.0f / sqrtf((float)(tot / (double)K) + 1e-6f);
                }
                for (int i0 = 0; i0 < K; i0 += NT * 4) {                      // whole waves per 256-block (K is a multiple of 256): the Q8_K emission is a wave-level operation
                    const int i = i0 + threadIdx.x * 4; const bool in = i < K; const int ic = in ? i : 0;
                    const float4 xv = *reinterpret_cast<const float4 *>(xr + ic);
                    float4 wv4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                    if (norm) wv4 = *reinterpret_cast<const float4 *>(a.pw + ic);
                    float v[4] = {xv.x, xv.y, xv.z, xv.w};
                    if (norm) { v[0] = (xv.x * scale) * wv4.x; v[1] = (xv.y * scale) * wv4.y; v[2] = (xv.z * scale) * wv4.z; v[3] = (xv.w * scale) * wv4.w; }
                    if (!in) { v[0] = 0.0f; v[1] = 0.0f; v[2] = 0.0f; v[3] = 0.0f; }
                    if (i0 + (int)(threadIdx.x & ~63) * 4 < K) quant_emit4(v, in, i, (size_t)t, K, L, ACT_Q8K);   // (wave-uniform condition: a wave entirely past the row skips)
                }
            } else {
                for (int i = threadIdx.x * 16; i < K; i += NT * 16) { const v4i_r z = {0, 0, 0, 0}; *reinterpret_cast<v4i_r *>(q8 + (size_t)t * K + i) = z; }
                for (int i = threadIdx.x; i < NSB; i += NT) { dk[t * NSB + i] = 0.0f; if (!Q6) { const v4i_r z = {0, 0, 0, 0}; *reinterpret_cast<v4i_r *>(dg + (size_t)(t * NSB + i) * 16) = z; } }
                for (int i = threadIdx.x; i < K / 16; i += NT) bsl[t * (K / 16) + i] = 0;
            }
        }
        __syncthreads();
        if (Q6) {
            for (int i = threadIdx.x; i < 4 * NSB; i += NT) {
                const int t = i / NSB, sb = i - t * NSB;
                int8_t *o = dg + (size_t)i * 32;
#pragma unroll
                for (int ui = 0; ui < 8; ui++) {
                    const int n = ui >> 2, c = (ui >> 1) & 1, h = ui & 1, grp = 8 * n + 2 * c + h;
                    const int s_lo = (int)bsl[t * (K / 16) + sb * 16 + grp], s_hi = (int)bsl[t * (K / 16) + sb * 16 + grp + 4];
                    o[2 * ui] = (int8_t)(s_lo & 127); o[2 * ui + 1] = (int8_t)(s_hi & 127); o[16 + 2 * ui] = (int8_t)(s_lo >> 7); o[16 + 2 * ui + 1] = (int8_t)(s_hi >> 7);
                }
            }
        }
    } else {
    // a workgroup that serves exactly ONE task of the K-split form copies only that task's K range (320 workgroups each copying the 55 KB image of a
    // K = 13824 row set would move more bytes than a third of the weights)
    int c_sb0 = 0, c_sb1 = NSB;
    if (a.ksplit > 1 && (int)gridDim.x >= a.n_mat * a.groups_each * a.ksplit) { const int part = (int)blockIdx.x % a.ksplit; c_sb0 = (int)((long long)NSB * part / a.ksplit); c_sb1 = (int)((long long)NSB * (part + 1) / a.ksplit); }
    ri_stage_rows(A, q8, N, K, c_sb0, c_sb1, 64 * WPB);
    ri_stage_digits<T>(A, dg, dk, N, NSB, K, c_sb0, c_sb1, 64 * WPB);
    }
    __syncthreads();
    __shared__ int s_last;
    const int8_t *qa = q8 + (size_t)t4 * K;
    const int8_t *da = dg + (size_t)t4 * NSB * DG;
    for (int task = blockIdx.x; task < total_groups * S; task += gridDim.x) {
        const Task tk = task_of(task);
        const int g = tk.g, gl = tk.gl;
        const RiMat &M = a.m[tk.m];
        const RiPtr wp = ri_ptrs<T>(M.p, gl, U, NSB, lane);
        if (task != (int)blockIdx.x) ri_fetch<T>(wp.pq, wp.pp, wp.ph, wp.pd, tk.sb0, NSB, cur);
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        ri_stream<T>(wp, NSB, tk.sb0, tk.sb1, qa, da, dk, acc, cur);
        // the K ranges of the WPB waves, combined in wave order
#pragma unroll
        for (int t = 0; t < 4; t++) red[(wv * 4 + t) * 64 + lane] = acc[t];
        __syncthreads();
        float s = 0.0f;
        if (wv < N) {
            s = red[wv * 64 + lane];
#pragma unroll
            for (int w = 1; w < WPB; w++) s += red[(w * 4 + wv) * 64 + lane];
        }
        if (S == 1) {
            if (wv < N) { const size_t o = (size_t)wv * a.ldy + (size_t)gl * 64 + lane; M.y[o] = M.res ? s + M.res[o] : s; }
        } else {
            // hand-off without cache maintenance (a release / acquire FENCE writes back and invalidates the whole L2 of the XCD -- measured: 78 us
            // instead of 22 for the 13B w2, the weight stream of every other workgroup loses its lines): the partial sums are agent-scope
            // (write-through, sc1) stores, drained with vmcnt(0) before the ticket; the last arriver reads them with agent-scope (cache-bypassing)
            // loads.  This is the guide's valid form for gfx942 / gfx950 (MI355X_MICROARCH.md "Valid forms": sc1 payload + drain replaces release / acquire), NOT a statement
            // of the HIP memory model -- Engine::init refuses any device that is not gfx950
            if (wv < N) __hip_atomic_store(a.slabs + ((size_t)task * 4 + wv) * 64 + lane, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(a.tickets + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(S - 1);
            __syncthreads();
            if (s_last) {                                      // workgroup-uniform: the last of the group's S workgroups adds the parts in part order (deterministic)
                if (wv < N) {
                    float tot = 0.0f;
                    for (int p = 0; p < S; p++) tot += __hip_atomic_load(a.slabs + (((size_t)g * S + p) * 4 + wv) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const size_t o = (size_t)wv * a.ldy + (size_t)gl * 64 + lane;
                    M.y[o] = M.res ? tot + M.res[o] : tot;
                }
                if (threadIdx.x == 0) __hip_atomic_store(a.tickets + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
            }
        }
        __syncthreads();
    }
}

// A "more bits" layer's wq | wk (Q4_K / Q5_K) + wv (Q6_K) in ONE launch (the single-row step's k_matvec_mix, the v_dot4 batched step's
// k_matvec_tn_mix): the same row image and the same per-group stream as k_matvec_ri, the digit image of the block sums staged once per type; row
// groups of the first n_a matrices stream as TA, the others as TB. No K split, rows prepared by the caller.
template <int TA, int TB, int WPB>
__global__ __launch_bounds__(64 * WPB, 2) void k_matvec_ri_mix(const RiArgs a, const ActQ A) {
    constexpr int DGA = TA == GT_Q6_K ? 32 : 16, DGB = TB == GT_Q6_K ? 32 : 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ri[];
    const int K = a.K, U = K / 32, NSB = K / 256, N = a.N;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, t4 = lane & 3;
    int8_t *q8 = reinterpret_cast<int8_t *>(smem_ri);
    int8_t *dga = q8 + 4 * K, *dgb = dga + 4 * NSB * DGA;
    float *dk = reinterpret_cast<float *>(dgb + 4 * NSB * DGB);
    float *red = dk + 4 * NSB;
    const int sb_per = (NSB + WPB - 1) / WPB, sb0 = wv * sb_per, sb1 = min(NSB, sb0 + sb_per);
    // (no prefetch across the prologue here: one register image that either type may have filled stays live in full through both branches -- 324-396
    // B of scratch)
    ri_stage_rows(A, q8, N, K, 0, NSB, 64 * WPB);
    ri_stage_digits<TA>(A, dga, dk, N, NSB, K, 0, NSB, 64 * WPB);
    ri_stage_digits<TB>(A, dgb, nullptr, N, NSB, K, 0, NSB, 64 * WPB);
    __syncthreads();
    const int8_t *qa = q8 + (size_t)t4 * K;
    for (int g = blockIdx.x; g < a.n_mat * a.groups_each; g += gridDim.x) {
        const int m = g / a.groups_each, gl = g - m * a.groups_each;
        const RiMat &M = a.m[m];
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (m < a.n_a) {                                                                                          // workgroup-uniform branch
            const RiPtr p = ri_ptrs<TA>(M.p, gl, U, NSB, lane);
            RiRaw cur; ri_fetch<TA>(p.pq, p.pp, p.ph, p.pd, sb0, NSB, cur);
            ri_stream<TA>(p, NSB, sb0, sb1, qa, dga + (size_t)t4 * NSB * DGA, dk, acc, cur);
        } else {
            const RiPtr p = ri_ptrs<TB>(M.p, gl, U, NSB, lane);
            RiRaw cur; ri_fetch<TB>(p.pq, p.pp, p.ph, p.pd, sb0, NSB, cur);
            ri_stream<TB>(p, NSB, sb0, sb1, qa, dgb + (size_t)t4 * NSB * DGB, dk, acc, cur);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) red[(wv * 4 + t) * 64 + lane] = acc[t];
        __syncthreads();
        if (wv < N) {
            float s = red[wv * 64 + lane];
#pragma unroll
            for (int w = 1; w < WPB; w++) s += red[(w * 4 + wv) * 64 + lane];
            const size_t o = (size_t)wv * a.ldy + (size_t)gl * 64 + lane;
            M.y[o] = M.res ? s + M.res[o] : s;
        }
        __syncthreads();
    }
}

static int g_ri_cus = 256;
void set_ri_cus(int cus) { if (cus > 0) g_ri_cus = cus; }
static size_t ri_lds(int type, int K, int wpb, bool pro) { const int NSB = K / 256; return (size_t)4 * K + (size_t)4 * NSB * (type == GT_Q6_K ? 32 : 16) + (size_t)4 * NSB * 4 + (size_t)wpb * 4 * 64 * 4 + (pro ? (size_t)4 * (K / 16) * 2 : 0); }
template <int T, int WPB, bool PRO>
static void launch_ri_k(const RiArgs &a, const ActQ &A, unsigned blocks, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_matvec_ri<T, WPB, PRO>)); attr = true; }
    hipLaunchKernelGGL((k_matvec_ri<T, WPB, PRO>), dim3(blocks), dim3(64 * WPB), lds, s, a, A);
}
template <int T>
static bool launch_ri_t(const RiArgs &a, const ActQ &A, hipStream_t s) {
    const int total = a.n_mat * a.groups_each;
    // 4 waves per workgroup (each a K quarter), two workgroups per CU; matrices with fewer 64-row groups than CUs (wo, w2: 80 groups at the 13B
    // width) split K over 8 waves
    const bool wide = total < g_ri_cus && a.ksplit <= 1, pro = a.px != nullptr;
    const size_t lds = ri_lds(T, a.K, wide ? 8 : 4, pro);
    if (lds > (wide ? 150u : 78u) * 1024u) return false;
    if (wide) { const unsigned nb = (unsigned)std::min(total, g_ri_cus); if (pro) launch_ri_k<T, 8, true>(a, A, nb, lds, s); else launch_ri_k<T, 8, false>(a, A, nb, lds, s); }
    else { const unsigned nb = (unsigned)std::min(total * std::max(1, a.ksplit), 2 * g_ri_cus); if (pro) launch_ri_k<T, 4, true>(a, A, nb, lds, s); else launch_ri_k<T, 4, false>(a, A, nb, lds, s); }
    return true;
}
// y[m][t * ldy + r] = W_m[r] . act[t] (+ residual[m][t * ldy + r]) for N = 1..4 prepared rows (A: Q8_K image incl. bsq) against 1..3 same-type,
// same-shape k-quant matrices that carry their row-interleaved image (ri[k]); false -> outside this kernel's range, nothing launched
// K split over workgroups for a set with few row groups and a long K (ri_kernels.hip header): parts of >= 8 super-blocks (two per wave), at most 4,
// only with a workspace
int ri_ksplit(int total_groups, int K, const RiWorkspace &ws) {
    const int NSB = K / 256;
    if (!ws.slabs || !ws.tickets || total_groups >= g_ri_cus / 2 || total_groups > ws.n_tickets) return 1;
    int S = std::min(4, std::min(NSB / 8, 2 * g_ri_cus / std::max(1, total_groups)));
    while (S > 1 && (size_t)total_groups * S * 256 > ws.slab_floats) S--;
    return std::max(1, S);
}
bool launch_matvec_ri(const QWeight *const *W, const RiPlanes *const *ri, float *const *y, const float *const *residual, int n, const ActQ &A, int N, int ldy, hipStream_t s,
                      const float *px, const float *pw, int ldx, const RiWorkspace &ws) {
    if (n < 1 || n > 3 || N < 1 || N > 4) return false;
    if (px ? ldx < W[0]->cols : (!A.q8k || !A.dk || !A.bsk || !A.bsq)) return false;      // px with pw == null: rows quantised as they are (no norm)
    RiArgs a{};
    a.px = px; a.pw = pw; a.ldx = ldx;
    for (int i = 0; i < n; i++) {
        if (!ri[i] || !ri[i]->qs || W[i]->type != W[0]->type || W[i]->rows != W[0]->rows || W[i]->cols != W[0]->cols) return false;
        a.m[i].p = *ri[i]; a.m[i].y = y[i]; a.m[i].res = residual ? residual[i] : nullptr;
    }
    if (!ri_supported(W[0]->type, W[0]->rows, W[0]->cols)) return false;
    a.n_mat = n; a.groups_each = W[0]->rows / 64; a.rows_each = W[0]->rows; a.K = W[0]->cols; a.N = N; a.ldy = ldy;
    a.ksplit = ri_ksplit(n * a.groups_each, a.K, ws); a.slabs = ws.slabs; a.tickets = ws.tickets;
    switch (W[0]->type) {
    case GT_Q4_K: return launch_ri_t<GT_Q4_K>(a, A, s);
    case GT_Q5_K: return launch_ri_t<GT_Q5_K>(a, A, s);
    case GT_Q6_K: return launch_ri_t<GT_Q6_K>(a, A, s);
    default: return false;
    }
}

template <int TA, int TB, int WPB>
static void launch_ri_mix_k(const RiArgs &a, const ActQ &A, unsigned blocks, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_matvec_ri_mix<TA, TB, WPB>)); attr = true; }
    hipLaunchKernelGGL((k_matvec_ri_mix<TA, TB, WPB>), dim3(blocks), dim3(64 * WPB), lds, s, a, A);
}
// wq | wk of one type (Q4_K / Q5_K) and wv of another (Q6_K), same shape, every one with its row-interleaved image: y_k[t * ldy + r] = W_k[r] .
// act[t] in one launch
bool launch_matvec_ri_mixed(const QWeight *const *Wa, const RiPlanes *const *ria, float *const *ya, int na, const QWeight *const *Wb, const RiPlanes *const *rib, float *const *yb, int nb,
                            const ActQ &A, int N, int ldy, hipStream_t s) {
    if (na < 1 || nb < 1 || na + nb > 3 || N < 1 || N > 4 || !A.q8k || !A.dk || !A.bsk || !A.bsq) return false;
    const int ta = Wa[0]->type, tb = Wb[0]->type;
    if (!((ta == GT_Q4_K || ta == GT_Q5_K) && tb == GT_Q6_K) || !ri_supported(ta, Wa[0]->rows, Wa[0]->cols)) return false;
    RiArgs a{};
    for (int i = 0; i < na + nb; i++) {
        const QWeight *w = i < na ? Wa[i] : Wb[i - na]; const RiPlanes *r = i < na ? ria[i] : rib[i - na];
        if (!r || !r->qs || w->type != (i < na ? ta : tb) || w->rows != Wa[0]->rows || w->cols != Wa[0]->cols) return false;
        a.m[i].p = *r; a.m[i].y = i < na ? ya[i] : yb[i - na]; a.m[i].res = nullptr;
    }
    a.n_mat = na + nb; a.n_a = na; a.groups_each = Wa[0]->rows / 64; a.rows_each = Wa[0]->rows; a.K = Wa[0]->cols; a.N = N; a.ldy = ldy; a.ksplit = 1;
    const int total = a.n_mat * a.groups_each, NSB = a.K / 256;
    const bool wide = total < g_ri_cus;
    const int wpb = wide ? 8 : 4;
    const size_t lds = (size_t)4 * a.K + (size_t)4 * NSB * (16 + 32) + (size_t)4 * NSB * 4 + (size_t)wpb * 4 * 64 * 4;
    if (lds > (wide ? 150u : 78u) * 1024u) return false;
    const unsigned nbk = (unsigned)std::min(total, wide ? g_ri_cus : 2 * g_ri_cus);
    if (ta == GT_Q4_K) { if (wide) launch_ri_mix_k<GT_Q4_K, GT_Q6_K, 8>(a, A, nbk, lds, s); else launch_ri_mix_k<GT_Q4_K, GT_Q6_K, 4>(a, A, nbk, lds, s); }
    else { if (wide) launch_ri_mix_k<GT_Q5_K, GT_Q6_K, 8>(a, A, nbk, lds, s); else launch_ri_mix_k<GT_Q5_K, GT_Q6_K, 4>(a, A, nbk, lds, s); }
    return true;
}

}  // namespace mg4
