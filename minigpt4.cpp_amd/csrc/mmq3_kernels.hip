// TEST LIBRARY ONLY -- measured in round 4 and NOT adopted: in the engine it is at parity with k_mmq2_q45k (142-row image turn 10.94 -> 10.82 ms) for 1.5 bytes of HBM per weight
// (profiles/r04_prefill_digit_planes.log has every version, the leave-one-out timings and what bounds it: instruction issue, ~620 per wave and super-block).
// Prompt mat-mul on pre-scaled digit planes (round 4): y[t][r] = W[r] . act[t] for Q4_K / Q5_K weights at prompt sizes, the arithmetic of mmq2_kernels.hip
// (reference minigpt4.cpp:2373 / 2412 -> llama_eval -> ggml_mul_mat: Q8_K activations, exact int32 sums, fp32 super-block scales accumulated super-block by super-block),
// bit-identical to k_mmq2_q45k for the same K split.
//
// k_mmq2_q45k is bound by the integer sub-block scale multiply-adds (one per output element per 32 weights: vector pipe 84 % busy, matrix pipe 9 %).  Here the 6-bit
// sub-block scale is multiplied into the quant at LOAD time: p = sc_j * q (<= 63 * 31: 11 bits) is stored as two digits p = 128 hi + lo (lo: 1 byte, hi: 4 bits), so that
// sum_k x_k p_k over a whole super-block runs inside the MFMA accumulation (exact integers, two K = 256 chains) and the vector pipe touches an output element once per 256
// weights instead of once per 32.  1.5 bytes per weight instead of 0.69; round 2 measured this idea with register-staged loads and <= 96-token chunks and lost to the byte
// count (mmq2_kernels.hip header); what is new:
//   * every token of a <= 144-row prompt in ONE pass (v_mfma_i32_16x16x64_i8: 9 token tiles of 16 per wave -- 142 rows waste 1.4 %), so the planes are streamed once;
//   * both planes are stored in MFMA-fragment order ([row tile of 16][k step of 64][lane][16 B]), so a stage is a handful of contiguous 1-KiB pieces that travel
//     global -> LDS by LDS-DMA (no staging registers, no transpose, no unpack beyond two shift / and pairs for the 4-bit digit) and the fragment reads are lane-linear
//     (conflict-free by construction);
//   * a 3-deep ring of K = 128 stages with counted vmcnt waits and one raw barrier per stage (k_gemm_dma's scheme): two stages (48 KiB per CU) are always in flight.
// Workgroup = 8 waves = 128 weight rows (wave: one row tile of 16) x up to 9 token tiles x one K range (grid.z); 1 workgroup per CU (145 KiB of LDS).
#include "kernels.hpp"
#include "devutil.hpp"

namespace mg4 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {
typedef __attribute__((address_space(3))) void *lds3_t;
typedef const __attribute__((address_space(1))) void *glb1_t;
__device__ __forceinline__ float h2f_u(unsigned h) { return __half2float(__ushort_as_half((unsigned short)h)); }
__device__ __forceinline__ long pk64(int lo, int hi) { return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo); }
__device__ __forceinline__ v4i z4() { return v4i{0, 0, 0, 0}; }

// LDS image of a workgroup of NW waves (= NW row tiles of 16 weight rows).  Stage (K = 128 = two k steps): W_LO [NW row tiles][2 k steps][1 KiB fragment], W_HI [NW][2][512 B],
// ACT [NT token tiles][2 k steps][1 KiB]; S stages.  Side data of one super-block (S copies): HDR [16 NW rows][16 B] {d, dmin, scales12}, BSQ [192 tokens][16 B] digit-split
// per-32 sums, DK [192 tokens] fp32.
//   NW = 8, S = 3: one workgroup per CU (145 KiB), two stages in flight.
//   NW = 4, S = 2: TWO workgroups per CU (73 KiB each): the two waves of a SIMD belong to different workgroups and are not phase-locked by each other's barriers, so one's
//                  super-block epilogue (vector pipe) overlaps the other's MFMAs; one stage in flight per workgroup.
template <int NT, int NW> struct L3 {
    static constexpr int S = NW == 8 ? 3 : 2;
    static constexpr int W_LO = 0, W_HI = NW * 2048, ACT = NW * 3072, STAGE = ACT + NT * 2048;
    static constexpr int SIDE0 = S * STAGE, HDR = 0, BSQ = 2048, DK = 5120, SIDE = 6144;
    static constexpr int DUMP = SIDE0 + S * SIDE, TOTAL = DUMP + 1024;
    static constexpr int APW = (2 * NT + NW - 1) / NW;                 // activation pieces per wave and stage (the last ones may be dummies: the wait counts stay uniform)
    static constexpr int SPW = 8 / NW;                                 // side pieces per wave and super-block (8 slots: HDR x NW / 4, BSQ x 3, DK x 3, the rest dummies)
    static constexpr int N_ODD = 3 + APW, N_EVEN = N_ODD + SPW;        // DMA instructions per wave: odd stage; even stage
    static_assert((size_t)16 * NT * 16 * NW * 4 <= (size_t)S * STAGE, "the result tile is staged through the ring");
};
}  // namespace

struct Mmq3Args {
    QWeight w[3];
    const uint8_t *plo[3], *phi[3];
    float *y[3];
    const float *res[3];
    int n_mat, groups_each, N, ldy, tiles_per_chunk, sb_per_split;
    long long slab_stride;
};

// wave = one row tile of 16 weight rows x NT token tiles
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void k_mmq3_q45k(const Mmq3Args a, const ActQ A) {
    using L = L3<NT, NW>;
    constexpr int S = L::S, RW = 16 * NW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem3[];
    const int lane = threadIdx.x & 63, l15 = lane & 15, kg = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x / a.groups_each, g = blockIdx.x - m * a.groups_each;
    const QWeight W = a.w[m];
    const int K = W.cols, KS = K / 64, NSB = K / 256, N = a.N;
    const int r0 = g * RW, t0 = blockIdx.y * a.tiles_per_chunk * 16;
    const int sb0 = blockIdx.z * a.sb_per_split, nsb = min(NSB, sb0 + a.sb_per_split) - sb0;

    // ---- DMA sources of this lane (stage 2 i + h relative to sb0; everything advances linearly with the stage / super-block index)
    const size_t rt = (size_t)(g * NW + wv);
    const uint8_t *s_lo = a.plo[m] + (rt * KS + (size_t)sb0 * 4) * 1024 + lane * 16;
    const uint8_t *s_hi = a.phi[m] + (rt * KS + (size_t)sb0 * 4) * 512 + lane * 16;
    const int8_t *s_act[L::APW]; unsigned d_act[L::APW];
#pragma unroll
    for (int u = 0; u < L::APW; u++) {
        const int p = wv * L::APW + u, live = p < 2 * NT, pp = live ? p : 0;
        const int tok = min(t0 + 16 * (pp >> 1) + l15, N - 1);
        s_act[u] = A.q8k + (size_t)tok * K + (size_t)sb0 * 256 + (pp & 1) * 64 + kg * 16;
        d_act[u] = live ? (unsigned)(L::ACT + p * 1024) : 0xFFFFFFFFu;
    }
    // side pieces: slot q = wv * SPW + j of 8.  NW = 8: q 0, 1 HDR (rows 64 q ..), 2..4 BSQ (tokens 64 (q - 2) ..), 5..7 DK.  NW = 4: q 0 HDR, 1..3 BSQ, 4..6 DK, 7 dummy.
    const uint8_t *s_side[L::SPW]; unsigned d_side[L::SPW]; bool side16[L::SPW];
#pragma unroll
    for (int j = 0; j < L::SPW; j++) {
        const int q = wv * L::SPW + j;
        constexpr int NH = NW / 4;                                     // HDR pieces
        if (q < NH) { s_side[j] = W.sc + ((size_t)min(r0 + 64 * q + lane, W.rows - 1) * NSB + sb0) * 16; d_side[j] = (unsigned)(L::HDR + q * 1024); side16[j] = true; }
        else if (q < NH + 3) { s_side[j] = reinterpret_cast<const uint8_t *>(A.bsq) + ((size_t)min(t0 + 64 * (q - NH) + lane, N - 1) * NSB + sb0) * 16; d_side[j] = (unsigned)(L::BSQ + (q - NH) * 1024); side16[j] = true; }
        else { const int c = min(q - NH - 3, 2); s_side[j] = reinterpret_cast<const uint8_t *>(A.dk) + ((size_t)min(t0 + 64 * c + lane, N - 1) * NSB + sb0) * 4; d_side[j] = q - NH - 3 < 3 ? (unsigned)(L::DK + c * 256) : 0xFFFFFFFFu; side16[j] = false; }
    }

    auto issue = [&](int i, int h, int slot, int copy) {   // stage 2 i + h of this K range into ring slot `slot`; h == 0 also brings super-block i's side data into side copy `copy`
        unsigned char *st = smem3 + slot * L::STAGE;
        int so_ = 2 * i + h;
        asm volatile("" : "+s"(so_));                    // opaque: otherwise every piece of every call site becomes its own 64-bit induction pointer (20 register pairs, scratch)
        const size_t so = (size_t)so_;
        unsigned char *dl = st + L::W_LO + wv * 2048;
        __builtin_amdgcn_global_load_lds((glb1_t)(s_lo + so * 2048), (lds3_t)dl, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb1_t)(s_lo + so * 2048), (lds3_t)dl, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((glb1_t)(s_hi + so * 1024), (lds3_t)(st + L::W_HI + wv * 1024), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < L::APW; u++) {
            unsigned char *d = d_act[u] == 0xFFFFFFFFu ? smem3 + L::DUMP : st + d_act[u];
            __builtin_amdgcn_global_load_lds((glb1_t)(s_act[u] + so * 128), (lds3_t)d, 16, 0, 0);
        }
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < L::SPW; j++) {
                unsigned char *sd = d_side[j] == 0xFFFFFFFFu ? smem3 + L::DUMP : smem3 + L::SIDE0 + copy * L::SIDE + d_side[j];
                if (side16[j]) __builtin_amdgcn_global_load_lds((glb1_t)(s_side[j] + (size_t)i * 16), (lds3_t)sd, 16, 0, 0);
                else __builtin_amdgcn_global_load_lds((glb1_t)(s_side[j] + (size_t)i * 4), (lds3_t)sd, 4, 0, 0);
            }
        }
    };

    v4i alo[NT], ahi[NT];
    v4f acc[NT];
#pragma unroll
    for (int tt = 0; tt < NT; tt++) { acc[tt] = v4f{0.f, 0.f, 0.f, 0.f}; alo[tt] = z4(); ahi[tt] = z4(); }

    // One k step (K = 64) of a stage = one batch: the weight tile's lo / packed hi fragment + NT token fragments.  The loop is software-pipelined by one batch inside a wave:
    // while batch b's MFMAs issue, the token fragment a tile has just consumed is reloaded (same registers) from batch b + 1, and batch b + 1's weight fragments are requested
    // at the top of batch b -- also across a stage barrier (a wave arrives at the barrier that publishes stage s + 1 with batch (s, 1) in registers, and reloads from stage
    // s + 1 while it multiplies that batch).
    struct WB { v4i blo; v2i hp; };
    v4i X[NT];
    auto rdw = [&](const unsigned char *st, int ks, WB &w) {
        w.blo = *reinterpret_cast<const v4i *>(st + L::W_LO + (wv * 2 + ks) * 1024 + lane * 16);
        w.hp = *reinterpret_cast<const v2i *>(st + L::W_HI + wv * 1024 + ks * 512 + lane * 8);
    };
    auto mm = [&](const WB &w, bool zero, bool reload, const unsigned char *nst, int nks, WB &wn) {   // reload: the next batch is k step nks of stage nst
        if (reload) rdw(nst, nks, wn);
        const v4i bhi = v4i{w.hp[0] & 0x0F0F0F0F, w.hp[1] & 0x0F0F0F0F, (w.hp[0] >> 4) & 0x0F0F0F0F, (w.hp[1] >> 4) & 0x0F0F0F0F};
#pragma unroll
        for (int tt = 0; tt < NT; tt++) {
            alo[tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(X[tt], w.blo, zero ? z4() : alo[tt], 0, 0, 0);
            ahi[tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(X[tt], bhi, zero ? z4() : ahi[tt], 0, 0, 0);
            if (reload) X[tt] = *reinterpret_cast<const v4i *>(nst + L::ACT + (tt * 2 + nks) * 1024 + lane * 16);
        }
    };
    // end of a super-block: min term on the matrix cores (K = 8 sub-blocks, digit split of the per-32 sums), then the two fp32 updates of mmq2 in mmq2's order
    auto finish = [&](const unsigned char *sd) {
        const v4i h = *reinterpret_cast<const v4i *>(sd + L::HDR + (wv * 16 + l15) * 16);
        const float dw = h2f_u((unsigned)h[0] & 0xFFFFu), ndmin = -h2f_u((unsigned)h[0] >> 16);
        const unsigned s1 = (unsigned)h[2], s2 = (unsigned)h[3];
        const unsigned mw0 = s1 & 0x3f3f3f3fu, mw1 = ((s2 >> 4) & 0x0f0f0f0fu) | (((s1 >> 6) & 0x03030303u) << 4);
        const long bm = kg == 0 ? pk64((int)mw0, (int)mw1) : 0L;
#pragma unroll
        for (int tt = 0; tt < NT; tt++) {
            const v4i bs = *reinterpret_cast<const v4i *>(sd + L::BSQ + (tt * 16 + l15) * 16);
            const v4f da = *reinterpret_cast<const v4f *>(sd + L::DK + (tt * 16 + 4 * kg) * 4);
            const v4i slo = __builtin_amdgcn_mfma_i32_16x16x32_i8(pk64(bs[0], bs[1]), bm, z4(), 0, 0, 0);
            const v4i shi = __builtin_amdgcn_mfma_i32_16x16x32_i8(pk64(bs[2], bs[3]), bm, z4(), 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int P = (ahi[tt][e] << 7) + alo[tt][e];
                float v = fmaf(dw * da[e], (float)P, acc[tt][e]);
                v = fmaf(ndmin * da[e], (float)(shi[e] * 128 + slo[e]), v);
                acc[tt][e] = v;
            }
        }
    };

    // Ring: stage s lives in slot s % S; the barrier that publishes stage s (everybody's pieces landed, everybody's LDS reads of stage s - 1 done) is followed by the request
    // for stage s + S - 1 into the slot stage s - 1 has just left.  Super-block i's side data travels with stage 2 i into side copy i % S and is used one stage late.
    WB w0, w1;
    issue(0, 0, 0, 0);
    if (S == 3) issue(0, 1, 1, 0);
    int slot = 0, copy = 0;                                 // ring slot of stage 2 i; side copy of super-block i
#pragma clang loop unroll(disable)
    for (int i = 0; i < nsb; i++) {
        asm volatile("" : "+s"(slot));                  // S == 2: keeps the slot a run-time scalar -- as constants hipcc hoists every fragment address into its own register (492 B of scratch)
        const int s1 = slot + 1 == S ? 0 : slot + 1, s2 = s1 + 1 == S ? 0 : s1 + 1, inext = min(i + 1, nsb - 1);
        const int c1 = copy + 1 == S ? 0 : copy + 1, cprev = copy == 0 ? S - 1 : copy - 1;
        const unsigned char *se = smem3 + slot * L::STAGE, *so = smem3 + s1 * L::STAGE;
        // ---- stage 2 i published
        if (S == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(L::N_ODD) : "memory");   // own pieces of stage 2 i (+ side data) landed (stage 2 i + 1 may be in flight); own LDS reads of stage 2 i - 1 done
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (S == 3) issue(inext, 0, s2, c1);                                            // stage 2 i + 2 (past the end: the last even stage again, into the free slot / copy)
        else issue(i, 1, s1, 0);                                                        // stage 2 i + 1
        if (i > 0) {
            mm(w1, false, true, se, 0, w0);                                             // batch (2 i - 1, 1) closes super-block i - 1
            finish(smem3 + L::SIDE0 + cprev * L::SIDE);
        } else {
            rdw(se, 0, w0);
#pragma unroll
            for (int tt = 0; tt < NT; tt++) X[tt] = *reinterpret_cast<const v4i *>(se + L::ACT + (tt * 2) * 1024 + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        mm(w0, true, true, se, 1, w1);                                                  // batch (2 i, 0)
        __builtin_amdgcn_sched_barrier(0);
        // ---- stage 2 i + 1 published
        if (S == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(L::N_EVEN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (S == 3) issue(inext, 1, slot, 0);                                           // stage 2 i + 3 -> the slot stage 2 i just left
        else issue(inext, 0, slot, c1);                                                 // stage 2 i + 2 (+ the next super-block's side data)
        mm(w1, false, true, so, 0, w0);                                                 // batch (2 i, 1)
        __builtin_amdgcn_sched_barrier(0);
        mm(w0, false, true, so, 1, w1);                                                 // batch (2 i + 1, 0)
        __builtin_amdgcn_sched_barrier(0);
        slot = S == 3 ? s2 : slot; copy = c1;
    }
    mm(w1, false, false, smem3, 0, w0);                                                 // batch (2 nsb - 1, 1)
    finish(smem3 + L::SIDE0 + (copy == 0 ? S - 1 : copy - 1) * L::SIDE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    // the trailing dummy pieces

    // ---- results: through LDS (the ring is free now), so that a token's RW outputs leave as one run instead of 64-byte pieces from NW waves
    __builtin_amdgcn_s_barrier();
    float *T = reinterpret_cast<float *>(smem3);                                        // [16 NT tokens][RW rows]
#pragma unroll
    for (int tt = 0; tt < NT; tt++)
#pragma unroll
        for (int e = 0; e < 4; e++) T[(tt * 16 + 4 * kg + e) * RW + wv * 16 + l15] = acc[tt][e];
    __syncthreads();
    float *y = a.y[m] + (size_t)blockIdx.z * a.slab_stride;
    const float *res = (gridDim.z == 1) ? a.res[m] : nullptr;
    constexpr int C4 = RW / 4, TR = 64 * NW / C4;                                       // float4 columns of a token's run; tokens per sweep of the workgroup (16)
    const int c4 = threadIdx.x % C4, tr = threadIdx.x / C4, row = r0 + 4 * c4;
    const bool vec = (a.ldy & 3) == 0 && (W.rows & 3) == 0;
#pragma unroll
    for (int j = 0; j < 16 * NT / TR; j++) {
        const int tl = j * TR + tr, tok = t0 + tl;
        if (tl < 16 * a.tiles_per_chunk && tok < N && row < W.rows) {
            v4f v = *reinterpret_cast<const v4f *>(T + tl * RW + 4 * c4);
            const size_t o = (size_t)tok * a.ldy + row;
            if (vec) {
                if (res) { const v4f r = *reinterpret_cast<const v4f *>(res + o); v += r; }
                *reinterpret_cast<v4f *>(y + o) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) if (row + e < W.rows) y[o + e] = res ? v[e] + res[o + e] : v[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Plane builder: thread = (row, unit u).  Unit u of a super-block (16 bytes of the repacked main plane): low nibbles = elements 64 (u >> 1) + 16 (u & 1) + i of sub-block
// 2 (u >> 1), high nibbles = the same elements + 32 (sub-block 2 (u >> 1) + 1); Q5_K: bit w of byte b of the unit's high-bit word = bit 4 of low-part element 4 w + b,
// bit 4 + w = the high part's (llm_kernels.hip k_repack).  Destination: fragment (row tile rt = row / 16, k step) of 64 lanes x 16 bytes, lane = (row % 16) + 16 * (k % 64) / 16.
// ---------------------------------------------------------------------------------------------------------------------
template <bool Q5>
__global__ __launch_bounds__(256) void k_mmq3_build(const QWeight W, uint8_t *__restrict__ plo, uint8_t *__restrict__ phi) {
    const int U = W.cols / 32, NSB = W.cols / 256, KS = W.cols / 64;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)W.rows * U) return;
    const int row = (int)(idx / U), u = (int)(idx - (size_t)row * U), sb = u >> 3, u8 = u & 7;
    const v4i q = *reinterpret_cast<const v4i *>(W.qs + idx * 16);
    const unsigned P = Q5 ? *reinterpret_cast<const unsigned *>(W.qh + idx * 4) : 0u;
    const v4i h = *reinterpret_cast<const v4i *>(W.sc + ((size_t)row * NSB + sb) * 16);
    const unsigned s0 = (unsigned)h[1], s2 = (unsigned)h[3];
    const unsigned scw0 = s0 & 0x3f3f3f3fu, scw1 = (s2 & 0x0f0f0f0fu) | (((s0 >> 6) & 0x03030303u) << 4);
    const int jl = 2 * (u8 >> 1), jh = jl + 1;
    const int scl = (int)(((jl & 4) ? scw1 : scw0) >> (8 * (jl & 3))) & 0xFF, sch = (int)(((jh & 4) ? scw1 : scw0) >> (8 * (jh & 3))) & 0xFF;
    unsigned lo_l[4] = {0, 0, 0, 0}, lo_h[4] = {0, 0, 0, 0};     // lo digit bytes of the low / high part, 16 elements each
    unsigned hi_l[2] = {0, 0}, hi_h[2] = {0, 0};                 // hi digits packed: byte b = digit(e = b) | digit(e = b + 8) << 4
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int w = i >> 2, b = i & 3;
        const unsigned byte = ((unsigned)q[w] >> (8 * b)) & 0xFFu;
        const int ql = (int)(byte & 15u) | (int)(((P >> (8 * b + w)) & 1u) << 4), qh = (int)(byte >> 4) | (int)(((P >> (8 * b + 4 + w)) & 1u) << 4);
        const int pl = scl * ql, ph = sch * qh;
        lo_l[w] |= (unsigned)(pl & 127) << (8 * b); lo_h[w] |= (unsigned)(ph & 127) << (8 * b);
        const int e8 = i & 7, sh = 8 * (e8 & 3) + 4 * (i >> 3);
        hi_l[e8 >> 2] |= (unsigned)(pl >> 7) << sh; hi_h[e8 >> 2] |= (unsigned)(ph >> 7) << sh;
    }
    const size_t frag = (size_t)(row >> 4) * KS + (size_t)(4 * sb + (u8 >> 1));
    const int ln_l = (row & 15) + 16 * (u8 & 1), ln_h = ln_l + 32;
    *reinterpret_cast<v4i *>(plo + frag * 1024 + ln_l * 16) = v4i{(int)lo_l[0], (int)lo_l[1], (int)lo_l[2], (int)lo_l[3]};
    *reinterpret_cast<v4i *>(plo + frag * 1024 + ln_h * 16) = v4i{(int)lo_h[0], (int)lo_h[1], (int)lo_h[2], (int)lo_h[3]};
    *reinterpret_cast<v2i *>(phi + frag * 512 + ln_l * 8) = v2i{(int)hi_l[0], (int)hi_l[1]};
    *reinterpret_cast<v2i *>(phi + frag * 512 + ln_h * 8) = v2i{(int)hi_h[0], (int)hi_h[1]};
}

bool mmq3_supported(int type, int rows, int cols) { return (type == GT_Q4_K || type == GT_Q5_K) && cols % 256 == 0 && rows >= 32; }
// bytes of the two planes of one matrix (rows padded to whole 128-row workgroup tiles); lo plane first, hi plane at *hi_off
size_t mmq3_plane_bytes(int rows, int cols, size_t *hi_off) {
    const size_t rp = (size_t)((rows + 127) / 128) * 128, lo = rp * (size_t)cols, hi = lo / 2;
    if (hi_off) *hi_off = lo;
    return lo + hi;
}
void launch_mmq3_build(const QWeight &W, uint8_t *planes, hipStream_t s) {
    size_t hi_off; const size_t total = mmq3_plane_bytes(W.rows, W.cols, &hi_off);
    if (W.rows % 128) HIP_CHECK(hipMemsetAsync(planes, 0, total, s));
    const size_t n = (size_t)W.rows * (W.cols / 32);
    if (W.type == GT_Q5_K) hipLaunchKernelGGL(k_mmq3_build<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, planes, planes + hi_off);
    else hipLaunchKernelGGL(k_mmq3_build<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, planes, planes + hi_off);
}

static int g_mmq3_cus = 256, g_mmq3_ks = 0, g_mmq3_nw = 8;
void set_mmq3_tuning(int cus, int ks) { if (cus > 0) g_mmq3_cus = cus; if (ks >= 0) g_mmq3_ks = ks; }
void set_mmq3_waves(int nw) { g_mmq3_nw = nw == 4 ? 4 : 8; }

template <int NT, int NW>
static void mmq3_launch_nt(dim3 grid, hipStream_t s, const Mmq3Args &a, const ActQ &A) {
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_mmq3_q45k<NT, NW>)); attr = true; }
    hipLaunchKernelGGL((k_mmq3_q45k<NT, NW>), grid, dim3(64 * NW), (size_t)(L3<NT, NW>::TOTAL), s, a, A);
}
template <int NW>
static void mmq3_launch(int tiles, dim3 grid, hipStream_t s, const Mmq3Args &a, const ActQ &A) {
    switch (tiles) {
    case 1: case 2: mmq3_launch_nt<2, NW>(grid, s, a, A); break;
    case 3: case 4: mmq3_launch_nt<4, NW>(grid, s, a, A); break;
    case 5: case 6: mmq3_launch_nt<6, NW>(grid, s, a, A); break;
    case 7: case 8: mmq3_launch_nt<8, NW>(grid, s, a, A); break;
    default: mmq3_launch_nt<9, NW>(grid, s, a, A); break;
    }
}

// 1..3 same-type, same-shape Q4_K / Q5_K matrices with planes (planes[i]: launch_mmq3_build's output) against the N prepared rows in one launch.  Same contract as
// launch_mmq2_set (K split into A.ws, combine deferred on request); false -> outside the kernel's range, nothing launched.
bool launch_mmq3_set(const QWeight *const *W, const uint8_t *const *planes, float *const *y, const float *const *residual, int n, const ActQ &A, int N, int ldy, hipStream_t s, SlabSrc *defer) {
    if (defer) *defer = SlabSrc{};
    if (n < 1 || n > 3 || N < 1 || !A.bsq || !A.q8k || !A.dk) return false;
    for (int i = 0; i < n; i++) if (!planes[i] || !mmq3_supported(W[i]->type, W[i]->rows, W[i]->cols) || W[i]->type != W[0]->type || W[i]->rows != W[0]->rows || W[i]->cols != W[0]->cols) return false;
    Mmq3Args a{};
    size_t hi_off; mmq3_plane_bytes(W[0]->rows, W[0]->cols, &hi_off);
    for (int i = 0; i < n; i++) { a.w[i] = *W[i]; a.plo[i] = planes[i]; a.phi[i] = planes[i] + hi_off; a.y[i] = y[i]; a.res[i] = residual ? residual[i] : nullptr; }
    const int tiles = (N + 15) / 16, n_chunks = (tiles + 8) / 9;
    a.tiles_per_chunk = (tiles + n_chunks - 1) / n_chunks;
    const int nw = g_mmq3_nw, rw = 16 * nw, slots = g_mmq3_cus * (nw == 8 ? 1 : 2);   // workgroups the chip holds at once
    a.n_mat = n; a.groups_each = (W[0]->rows + rw - 1) / rw; a.N = N; a.ldy = ldy;
    const int NSB = W[0]->cols / 256, wgs = n * a.groups_each * n_chunks;
    const size_t out_floats = (size_t)N * ldy;
    // K split: split until the launch just fills the chip; at least 4 super-blocks (8 stages) per slice
    int ks = std::max(1, slots / wgs);
    while (ks > 1 && (NSB / ks < 4 || !A.ws || (size_t)ks * out_floats * n > A.ws_floats)) ks--;
    if (g_mmq3_ks > 0) ks = std::max(1, std::min(g_mmq3_ks, std::min(NSB, A.ws ? (int)(A.ws_floats / std::max<size_t>(1, out_floats * n)) : 1)));
    a.sb_per_split = (NSB + ks - 1) / ks;
    ks = (NSB + a.sb_per_split - 1) / a.sb_per_split;
    if (ks > 1) {
        if ((size_t)ldy % 4 || out_floats % 4) return false;
        a.slab_stride = (long long)out_floats;
        for (int i = 0; i < n; i++) { a.y[i] = A.ws + (size_t)i * ks * out_floats; a.res[i] = nullptr; }
    }
    const dim3 grid((unsigned)(n * a.groups_each), (unsigned)n_chunks, (unsigned)ks);
    if (nw == 8) mmq3_launch<8>(a.tiles_per_chunk, grid, s, a, A); else mmq3_launch<4>(a.tiles_per_chunk, grid, s, a, A);
    if (ks > 1) {
        SlabSrc src; src.ws = A.ws; src.ks = ks; src.stride = (long long)out_floats; src.n = n;
        for (int i = 0; i < n; i++) { src.y[i] = y[i]; src.res[i] = residual ? residual[i] : nullptr; src.mbase[i] = A.ws + (size_t)i * ks * out_floats; src.mks[i] = ks; }
        if (defer) *defer = src; else launch_slab_flush(src, s);
    }
    return true;
}

}  // namespace mg4
