// TEST LIBRARY ONLY (libminigpt4_test.so; `make`'s TEST_OBJS) -- round-5 micro-benchmark for the batched decode step (verdict item: "B = 2 ... 4 on the matrix cores: measure,
// do not argue").  The product's multi-row mat-vec (k_matvec_tn, llm_kernels.hip) multiplies every weight unit against each of the B activation rows with v_dot4_i32_i8:
// 8 dot instructions per row and 32 weights, vector-issue bound at B = 4 (84 % VALU busy).  v_mfma_i32_4x4x4_16B_i8 does 16 independent 4 x 4 x 4 products per
// instruction: with tokens as the M side (4 rows of the A operand, the same in all 16 blocks) and 64 different WEIGHT ROWS as the N side (lane = weight row), ONE
// instruction multiplies 4 consecutive weights of 64 rows against 4 tokens -- what takes 4 v_dot4 per lane.  That needs a lane to own a weight ROW (the product's planes give
// a lane a UNIT of one row shared by the wave), i.e. a row-interleaved plane layout: [64-row group][unit][64 rows][16 B].  This file measures that form on Q5_K with
// SYNTHETIC planes written directly in that layout (random quants / scales, sane fp16 d / dmin), validated against a scalar reference kernel over the same planes:
//   * k_probe_tn_mfma: workgroup = one 64-row group, its 4 waves split K (contiguous super-block ranges, partial sums combined through LDS in wave order);
//     the B <= 4 activation rows (Q8_K values, digit-split per-32 sums, scales: what k_rms_quant writes) sit in LDS, every lane reads token (lane & 3)'s bytes;
//     per unit and lane: the Q5_K unpack of the product kernels (28 VALU), 8 MFMAs, 8 integer scale multiply-adds (4 tokens x 2 sub-blocks); per super-block 4 more MFMAs
//     for the min term on the digit split, and one fp32 fma pair per token.
// Result and decision: profiles/r05_batched_decode_mfma.log.
#include "kernels.hpp"
#include "devutil.hpp"

#include <algorithm>
#include <vector>

namespace mg4 {

typedef int v4i_p __attribute__((ext_vector_type(4)));

struct TnPlanes { const uint8_t *qs, *qh, *sc; int rows, K; };   // qs [G][U][64][16], qh [G][U][64][4] (pack_hb1 words), sc [G][NSB][64][16] ({d, dmin} fp16, 12 packed 6-bit scale / min bytes)

__device__ __forceinline__ void scale_min_words(const v4i_p &h, unsigned &scw0, unsigned &scw1, unsigned &mw0, unsigned &mw1) {
    const unsigned s0 = (unsigned)h[1], s1 = (unsigned)h[2], s2 = (unsigned)h[3];
    scw0 = s0 & 0x3f3f3f3fu; scw1 = (s2 & 0x0f0f0f0fu) | (((s0 >> 6) & 0x03030303u) << 4);
    mw0 = s1 & 0x3f3f3f3fu; mw1 = ((s2 >> 4) & 0x0f0f0f0fu) | (((s1 >> 6) & 0x03030303u) << 4);
}

template <int TN>
__global__ __launch_bounds__(256, 2) void k_probe_tn_mfma(const TnPlanes P, const ActQ A, float *__restrict__ y, const int n_groups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tn_probe[];
    const int K = P.K, U = K / 32, NSB = K / 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, t4 = lane & 3;
    // LDS image of the activation rows: q8 [4][K], digits [4][NSB][16], scales [4][NSB]; rows >= TN are zero
    int8_t *q8 = reinterpret_cast<int8_t *>(smem_tn_probe);
    int8_t *dg = q8 + 4 * K;
    float *dk = reinterpret_cast<float *>(dg + 4 * NSB * 16);
    float *red = dk + 4 * NSB;                                             // [4 waves][4 tokens][64 lanes]
    for (int i = threadIdx.x * 16; i < 4 * K; i += 256 * 16) {
        const int t = i / K;
        v4i_p v = {0, 0, 0, 0};
        if (t < TN) v = *reinterpret_cast<const v4i_p *>(A.q8k + (size_t)t * K + (i - t * K));
        *reinterpret_cast<v4i_p *>(q8 + i) = v;
    }
    for (int i = threadIdx.x; i < 4 * NSB; i += 256) {
        const int t = i / NSB, sb = i - t * NSB;
        v4i_p v = {0, 0, 0, 0};
        float d = 0.0f;
        if (t < TN) { v = *reinterpret_cast<const v4i_p *>(A.bsq + ((size_t)t * NSB + sb) * 16); d = A.dk[(size_t)t * NSB + sb]; }
        *reinterpret_cast<v4i_p *>(dg + (size_t)i * 16) = v; dk[i] = d;
    }
    __syncthreads();
    const int sb_per = (NSB + 3) / 4, sb0 = wv * sb_per, sb1 = min(NSB, sb0 + sb_per);
    const int8_t *qa = q8 + (size_t)t4 * K;
    const int8_t *da = dg + (size_t)t4 * NSB * 16;
    struct Raw { v4i_p q[8]; unsigned p[8]; v4i_p h; };
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint8_t *pq = P.qs + (size_t)g * U * 1024 + lane * 16, *pp = P.qh + (size_t)g * U * 256 + lane * 4, *ph = P.sc + (size_t)g * NSB * 1024 + lane * 16;
        auto fetch = [&](int sb, Raw &r) {
            const int sbc = min(sb, NSB - 1);
#pragma unroll
            for (int u = 0; u < 8; u++) { r.q[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i_p *>(pq + (size_t)(sbc * 8 + u) * 1024)); r.p[u] = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(pp + (size_t)(sbc * 8 + u) * 256)); }
            r.h = __builtin_nontemporal_load(reinterpret_cast<const v4i_p *>(ph + (size_t)sbc * 1024));
        };
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        auto consume = [&](int sb, const Raw &r) {
            unsigned scw0, scw1, mw0, mw1; scale_min_words(r.h, scw0, scw1, mw0, mw1);
            int isum[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = u >> 1, hs = u & 1;
                const v4i_p alo = *reinterpret_cast<const v4i_p *>(qa + sb * 256 + 64 * j + 16 * hs), ahi = *reinterpret_cast<const v4i_p *>(qa + sb * 256 + 64 * j + 16 * hs + 32);
                v4i_p D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0};
                const unsigned Pw = r.p[u];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned q = (unsigned)r.q[u][d];
                    const unsigned wlo = (q & 0x0F0F0F0Fu) | ((d == 0 ? Pw << 4 : d == 1 ? Pw << 3 : d == 2 ? Pw << 2 : Pw << 1) & 0x10101010u);
                    const unsigned whi = ((q >> 4) & 0x0F0F0F0Fu) | ((d == 0 ? Pw : d == 1 ? Pw >> 1 : d == 2 ? Pw >> 2 : Pw >> 3) & 0x10101010u);
                    D0 = __builtin_amdgcn_mfma_i32_4x4x4i8(alo[d], (int)wlo, D0, 0, 0, 0);
                    D1 = __builtin_amdgcn_mfma_i32_4x4x4i8(ahi[d], (int)whi, D1, 0, 0, 0);
                }
                const unsigned scw = (j & 2) ? scw1 : scw0;
                const int sc0 = (int)(scw >> (16 * (j & 1))) & 0xFF, sc1 = (int)(scw >> (16 * (j & 1) + 8)) & 0xFF;
#pragma unroll
                for (int t = 0; t < 4; t++) isum[t] += __mul24(sc0, D0[t]) + __mul24(sc1, D1[t]);
            }
            // min term: sum_j m_j * bsum_j on the digit split bsum = 128 hi + lo (bytes 0..7 low digits of sub-blocks 0..7, 8..15 high digits)
            const v4i_p dgt = *reinterpret_cast<const v4i_p *>(da + sb * 16);
            v4i_p Ml = {0, 0, 0, 0}, Mh = {0, 0, 0, 0};
            Ml = __builtin_amdgcn_mfma_i32_4x4x4i8(dgt[0], (int)mw0, Ml, 0, 0, 0); Ml = __builtin_amdgcn_mfma_i32_4x4x4i8(dgt[1], (int)mw1, Ml, 0, 0, 0);
            Mh = __builtin_amdgcn_mfma_i32_4x4x4i8(dgt[2], (int)mw0, Mh, 0, 0, 0); Mh = __builtin_amdgcn_mfma_i32_4x4x4i8(dgt[3], (int)mw1, Mh, 0, 0, 0);
            const float d = __half2float(__ushort_as_half((unsigned short)((unsigned)r.h[0] & 0xFFFF))), dmin = __half2float(__ushort_as_half((unsigned short)((unsigned)r.h[0] >> 16)));
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float dkt = dk[t * NSB + sb];
                acc[t] = fmaf(d * dkt, (float)isum[t], acc[t]);
                acc[t] = fmaf(-(dmin * dkt), (float)(Mh[t] * 128 + Ml[t]), acc[t]);
            }
        };
        Raw cur, nxt;
        fetch(sb0, cur);
        for (int sb = sb0; sb < sb1;) {
            fetch(sb + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            consume(sb, cur);
            __builtin_amdgcn_sched_barrier(0);
            if (++sb >= sb1) break;
            fetch(sb + 1, cur);
            __builtin_amdgcn_sched_barrier(0);
            consume(sb, nxt);
            __builtin_amdgcn_sched_barrier(0);
            ++sb;
        }
        // K quarters of the 4 waves, combined in wave order
#pragma unroll
        for (int t = 0; t < 4; t++) red[(wv * 4 + t) * 64 + lane] = acc[t];
        __syncthreads();
        if (wv < TN) {
            const int t = wv;
            const float s = ((red[(0 * 4 + t) * 64 + lane] + red[(1 * 4 + t) * 64 + lane]) + red[(2 * 4 + t) * 64 + lane]) + red[(3 * 4 + t) * 64 + lane];
            y[(size_t)t * P.rows + (size_t)g * 64 + lane] = s;
        }
        __syncthreads();
    }
}

// scalar reference over the same planes: one thread per (row, token)
__global__ void k_probe_tn_ref(const TnPlanes P, const ActQ A, float *__restrict__ y, const int TN) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (row >= P.rows || t >= TN) return;
    const int K = P.K, U = K / 32, NSB = K / 256, g = row >> 6, l = row & 63;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};                               // the four K quarters, combined like the kernel above
    const int sb_per = (NSB + 3) / 4;
    for (int sb = 0; sb < NSB; sb++) {
        const v4i_p h = *reinterpret_cast<const v4i_p *>(P.sc + ((size_t)g * NSB + sb) * 1024 + l * 16);
        unsigned scw0, scw1, mw0, mw1; scale_min_words(h, scw0, scw1, mw0, mw1);
        int isum = 0, msum = 0;
        for (int u = 0; u < 8; u++) {
            const int j = u >> 1, hs = u & 1;
            const uint8_t *q = P.qs + ((size_t)g * U + sb * 8 + u) * 1024 + l * 16;
            const unsigned Pw = *reinterpret_cast<const unsigned *>(P.qh + ((size_t)g * U + sb * 8 + u) * 256 + l * 4);
            const int8_t *a = A.q8k + (size_t)t * K + sb * 256 + 64 * j + 16 * hs;
            int s0 = 0, s1 = 0;
            for (int e = 0; e < 16; e++) {
                const int k = e >> 2, i = e & 3;
                const int lo = (q[e] & 15) | (int)(((Pw >> (8 * i + k)) & 1u) << 4), hi = (q[e] >> 4) | (int)(((Pw >> (8 * i + 4 + k)) & 1u) << 4);
                s0 += lo * a[e]; s1 += hi * a[e + 32];
            }
            const unsigned scw = (j & 2) ? scw1 : scw0;
            isum += (int)((scw >> (16 * (j & 1))) & 0xFF) * s0 + (int)((scw >> (16 * (j & 1) + 8)) & 0xFF) * s1;
        }
        for (int jb = 0; jb < 8; jb++) {
            const int m = (int)(((jb & 4) ? mw1 : mw0) >> (8 * (jb & 3))) & 0xFF;
            int bs = 0; for (int e = 0; e < 32; e++) bs += A.q8k[(size_t)t * K + sb * 256 + 32 * jb + e];
            msum += m * bs;
        }
        const float d = __half2float(__ushort_as_half((unsigned short)((unsigned)h[0] & 0xFFFF))), dmin = __half2float(__ushort_as_half((unsigned short)((unsigned)h[0] >> 16)));
        const float dkt = A.dk[(size_t)t * NSB + sb];
        float &a4 = acc[min(sb / sb_per, 3)];
        a4 = fmaf(d * dkt, (float)isum, a4);
        a4 = fmaf(-(dmin * dkt), (float)msum, a4);
    }
    y[(size_t)t * P.rows + row] = ((acc[0] + acc[1]) + acc[2]) + acc[3];
}

__global__ void k_probe_fill_headers(uint8_t *sc, size_t n_headers) {   // {d, dmin} = two small positive fp16 values; the 12 scale / min bytes stay random
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_headers) { unsigned short *p = reinterpret_cast<unsigned short *>(sc + i * 16); p[0] = (unsigned short)(0x1C00 + (i * 37 % 256)); p[1] = (unsigned short)(0x1800 + (i * 11 % 256)); }
}

void launch_fill_random(void *p, size_t bytes, unsigned seed, hipStream_t s);   // probe_kernels.hip

// us per launch of the MFMA form for one Q5_K matrix SET of `rows` x `cols` (rows % 64 == 0, cols % 256 == 0) against TN = 1..4 prepared rows; `n_sets` plane sets are
// rotated so that no launch finds its weights in the caches.  check != 0: also runs the scalar reference on the first set and reports the largest |difference| / max |value|.
int probe_tn_mfma(int rows, int cols, int TN, int iters, int n_sets, int check, int cus, float *us_per_launch, float *rel_diff) {
    if (rows % 64 || cols % 256 || TN < 1 || TN > 4 || iters < 1 || n_sets < 1) return 1;
    const int K = cols, U = K / 32, NSB = K / 256, G = rows / 64;
    const size_t qs_b = (size_t)G * U * 1024, qh_b = (size_t)G * U * 256, sc_b = (size_t)G * NSB * 1024;
    std::vector<uint8_t *> bufs;
    std::vector<TnPlanes> sets((size_t)n_sets);
    for (int i = 0; i < n_sets; i++) {
        uint8_t *p = nullptr; HIP_CHECK(hipMalloc((void **)&p, qs_b + qh_b + sc_b)); bufs.push_back(p);
        launch_fill_random(p, qs_b + qh_b + sc_b, (unsigned)(i * 7919 + 29), nullptr);
        hipLaunchKernelGGL(k_probe_fill_headers, dim3((unsigned)(((size_t)G * NSB * 64 + 255) / 256)), dim3(256), 0, nullptr, p + qs_b + qh_b, (size_t)G * NSB * 64);
        sets[(size_t)i] = TnPlanes{p, p + qs_b, p + qs_b + qh_b, rows, K};
    }
    // activation rows through the product's own quantiser
    float *dx = nullptr; HIP_CHECK(hipMalloc((void **)&dx, (size_t)4 * K * 4));
    { std::vector<float> hx((size_t)4 * K); for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)((int)(i * 37 % 201) - 100) / 64.0f; HIP_CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); }
    ActQ A{};
    uint8_t *abuf = nullptr; const size_t ab = (size_t)4 * (K + (K / 256) * 4 + (K / 16) * 2 + (K / 16)) + 4096;
    HIP_CHECK(hipMalloc((void **)&abuf, ab));
    A.q8k = reinterpret_cast<int8_t *>(abuf); A.dk = reinterpret_cast<float *>(abuf + (size_t)4 * K); A.bsk = reinterpret_cast<int16_t *>(abuf + (size_t)4 * K + (size_t)4 * (K / 256) * 4 + 256);
    A.bsq = reinterpret_cast<int8_t *>(abuf + (size_t)4 * K + (size_t)4 * (K / 256) * 4 + 256 + (size_t)4 * (K / 16) * 2 + 256);
    launch_rms_quant(dx, nullptr, TN, K, A, ACT_Q8K, nullptr);
    float *dy = nullptr, *dref = nullptr; HIP_CHECK(hipMalloc((void **)&dy, (size_t)4 * rows * 4)); HIP_CHECK(hipMalloc((void **)&dref, (size_t)4 * rows * 4));
    const size_t lds = (size_t)4 * K + (size_t)4 * NSB * 16 + (size_t)4 * NSB * 4 + (size_t)4 * 4 * 64 * 4;
    const dim3 grid((unsigned)std::min(G, 2 * cus));
    auto run = [&](int set) {
        switch (TN) {
        case 1: hipLaunchKernelGGL(k_probe_tn_mfma<1>, grid, dim3(256), lds, nullptr, sets[(size_t)set], A, dy, G); break;
        case 2: hipLaunchKernelGGL(k_probe_tn_mfma<2>, grid, dim3(256), lds, nullptr, sets[(size_t)set], A, dy, G); break;
        case 3: hipLaunchKernelGGL(k_probe_tn_mfma<3>, grid, dim3(256), lds, nullptr, sets[(size_t)set], A, dy, G); break;
        default: hipLaunchKernelGGL(k_probe_tn_mfma<4>, grid, dim3(256), lds, nullptr, sets[(size_t)set], A, dy, G); break;
        }
    };
    static bool attr = false;
    if (!attr) { HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe_tn_mfma<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe_tn_mfma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe_tn_mfma<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe_tn_mfma<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); attr = true; }
    run(0);
    HIP_CHECK(hipDeviceSynchronize());
    if (rel_diff) *rel_diff = -1.0f;
    if (check) {
        hipLaunchKernelGGL(k_probe_tn_ref, dim3((unsigned)((rows + 63) / 64), (unsigned)TN), dim3(64), 0, nullptr, sets[0], A, dref, TN);
        HIP_CHECK(hipDeviceSynchronize());
        std::vector<float> a((size_t)TN * rows), b((size_t)TN * rows);
        HIP_CHECK(hipMemcpy(a.data(), dy, a.size() * 4, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(b.data(), dref, b.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0.0, big = 0.0;
        for (size_t i = 0; i < a.size(); i++) { worst = std::max(worst, (double)fabsf(a[i] - b[i])); big = std::max(big, (double)fabsf(b[i])); }
        if (rel_diff) *rel_diff = big > 0.0 ? (float)(worst / big) : (worst == 0.0 ? 0.0f : 1.0f);
    }
    for (int i = 0; i < std::min(n_sets, 3); i++) run(i);
    HIP_CHECK(hipDeviceSynchronize());
    hipEvent_t ea, eb; HIP_CHECK(hipEventCreate(&ea)); HIP_CHECK(hipEventCreate(&eb));
    HIP_CHECK(hipEventRecord(ea, nullptr));
    for (int i = 0; i < iters; i++) run(i % n_sets);
    HIP_CHECK(hipEventRecord(eb, nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    float ms = 0.0f; HIP_CHECK(hipEventElapsedTime(&ms, ea, eb));
    if (us_per_launch) *us_per_launch = ms * 1e3f / (float)iters;
    HIP_IGNORE(hipEventDestroy(ea)); HIP_IGNORE(hipEventDestroy(eb));
    for (uint8_t *p : bufs) HIP_IGNORE(hipFree(p));
    HIP_IGNORE(hipFree(dx)); HIP_IGNORE(hipFree(abuf)); HIP_IGNORE(hipFree(dy)); HIP_IGNORE(hipFree(dref));
    return 0;
}

}  // namespace mg4
