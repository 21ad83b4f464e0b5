// Prefill mat-mul on the int8 matrix cores, second generation (round 2): y[t][r] = W[r] . act[t] for N >= 5 activation rows against k-quant weights,
// with ggml's arithmetic (reference minigpt4.cpp:2373 / 2412 -> llama_eval -> ggml_mul_mat: activations in Q8_K, exact int32 sub-block dots, integer
// sub-block scales, fp32 super-block scales accumulated super-block by super-block).
//
// Structure (what round 1's k_mmq_* lacked -- weights re-read once per 64 tokens, one wave per SIMD, activation fragments fetched from L2 with the full latency
// exposed at the top of every super-block):
//   * workgroup = 4 waves = 4 row tiles of 32 weight rows (128 rows) x one chunk of up to 4 token tiles of 32 tokens x one K range (grid.z);
//     the weights of a row tile are loaded ONCE per chunk, unpacked ONCE per super-block into MFMA B operands and reused for every token tile;
//   * the chunk's activation super-block (int8 values, fp32 scale, the per-32 sums as two int8 digits) is staged through LDS with global_load_lds
//     (LDS-DMA, 16 bytes per lane, no VGPR staging), double buffered, shared by the 4 waves; the 16-byte chunks of a token's 256-byte row are XOR-swizzled
//     on the SOURCE side (the DMA destination is lane-linear) so that the A-fragment ds_read_b128 of 32 consecutive tokens is bank-conflict free;
//   * one barrier per super-block: [wait own DMA + weight loads] barrier [unpack weights] [request next weights + next stage] [MFMAs on the current stage];
//   * v_mfma_i32_32x32x32_i8: activations are the A side (lane = token), weights the B side (lane = weight row), so a lane's 16 accumulators belong to ONE
//     weight row and its sub-block scales are per-lane scalars;  the Q4_K / Q5_K min term sum_j m_j * bsum_j runs as two MFMAs on the digit split
//     bsum = 128 hi + lo (exact) instead of round 1's eight MFMAs against a broadcast byte;
//   * <= 256 VGPRs: two workgroups (8 waves) per CU, so one wave's scale arithmetic (VALU) overlaps another's MFMAs.
// Per (row tile, super-block, token tile): 8 + 2 MFMAs (Q4_K / Q5_K) or 16 (Q6_K) and ~300 VALU instructions per wave -- the kernel is VALU-bound by the integer
// sub-block scale multiply-adds ggml's format requires (one per output element per 32 weights: 1 VALU operation per 32 MFMA multiply-accumulates on a chip whose matrix
// pipe is 64 x wider than its vector pipe), not by the matrix cores (PMC, profiles/r02e_pmc_mmq2.txt: two waves per SIMD issue 84 % of the time, MFMA 9 % busy).
// Measured and removed, second attempt at the same bound (profiles/r02l_prefill_gen3_scales_in_operands_microbench.log): the 6-bit sub-block scales multiplied into
// the int8 OPERANDS at unpack time as two 3-bit digits (sc = 8 hi + lo; digit x quant <= 217 fits a byte, one v_pk_mul_lo_u16 scales four weights; Q5_K bytes stored
// minus 128 and corrected through the staged per-32 sums), so that a token tile is two K = 256 accumulation chains and the per-(row, token, sub-block) integer
// multiply-adds disappear.  Bit-identical to this kernel on every test shape, but 64 operand registers + 5-tile chunks meant one wave per SIMD, and the launches
// were 1.4-1.8x SLOWER (13B layer at 142 rows: qkv 118 vs 64 us, w1|w3 169 vs 101, w2 93 vs 61; whole prefill 18.3 vs 13.2 ms).  The kernel is latency-bound at
// one wave per SIMD long before the VALU work it saved matters.  Third attempt (profiles/r02p_prefill_scaled_operands_pair_outer_microbench.log): the same arithmetic
// with the sub-block PAIRS as the outer loop and the token tiles inside, so that one pair's four scaled operands (16 registers) are live at a time and two waves per
// SIMD fit (256 VGPRs, <= 236 B of scratch at 3 tiles): bit-identical again, and 2.4x SLOWER than this kernel (qkv 154 vs 64 us, w1|w3 247 vs 99 at 142 rows; 123 / 199
// with two tiles per chunk).  Folding the scales into the operands doubles the MFMAs per token tile (two digit chains + the offset correction: 20 vs 10) and makes them
// dependent accumulation chains; the multiply-adds it removes were overlapping with the MFMAs of the other resident waves anyway.  The scale multiply-adds stay.
// Measured and removed (profiles/r02g_prefill_generations_microbench.log): pre-scaled "prefill planes" -- sub-block scale x quant stored as two int8 digits,
// 2 bytes per weight, so that the scales ride inside the MFMA accumulation (exact, bit-identical results, 41 % fewer VALU instructions) -- lost on Q4_K / Q5_K
// (w1|w3 at 142 rows: 155 vs 100 us): 2.8 x the weight bytes per chunk of <= 96 tokens turned the kernel into a latency-bound HBM stream; it won only on Q6_K.
#include "kernels.hpp"
#include "devutil.hpp"

namespace mg4 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ float h2f_b(unsigned short h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ v4i ldg16(const void *p) { return *reinterpret_cast<const v4i *>(p); }
__device__ __forceinline__ v16i zero16() { v16i z; for (int i = 0; i < 16; i++) z[i] = 0; return z; }
__device__ __forceinline__ int tok_of(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }   // MFMA 32x32 C layout: row (= token) of accumulator register `reg`
__device__ __forceinline__ long pack64(int lo, int hi) { return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo); }
__device__ __forceinline__ int sext6(int x) { const int yv = x ^ 0x20202020; const int s = yv & 0x20202020; return yv | (s << 1) | (s << 2); }   // 4 x (6-bit q - 32) as int8

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;
__device__ __forceinline__ void dma16(const void *src, unsigned char *lds_wave_base) {   // LDS[lds_wave_base + lane * 16 .. +16) <- src (per lane)
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void *src, unsigned char *lds_wave_base) {    // LDS[lds_wave_base + lane * 4 .. +4) <- src
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 4, 0, 0);
}
}  // namespace

// LDS image of one activation stage (TT token tiles of one super-block):
//   q8 [TT * 32 tokens][16 chunks of 16 B], chunk c of token t stored at slot c ^ (t & 15)
//   bs [TTP * 32][16 B]  digit-split per-32 sums (TTP = TT rounded up to 2: one DMA instruction covers 64 tokens)
//   dk [TTP * 32] fp32   activation super-block scale
template <int TT> struct Mmq2Stage {
    static constexpr int TTP = (TT + 1) & ~1;
    static constexpr int Q8 = TT * 32 * 256, BS = TTP * 32 * 16, DK = TTP * 32 * 4;
    static constexpr int BYTES = Q8 + BS + DK;
};

struct Mmq2Args {
    QWeight w[3];                 // 1..3 matrices of one type and one shape; row group g belongs to matrix g / groups_each
    float *y[3];                  // outputs, [N][ldy] each (K split: slab z at y + z * slab_stride)
    const float *res[3];          // optional residual inputs (only without a K split)
    int n_mat, groups_each;       // row groups (of 128 rows) per matrix
    int N, ldy;
    int n_tiles, tiles_per_chunk; // token tiles of 32 in total / per grid.y chunk
    int sb_per_split;             // super-blocks per grid.z slice
    long long slab_stride;        // floats between K-split slabs
};

// Requests stage `sb` of this chunk into LDS buffer `st` (all WPB waves of the workgroup take part; every request is unconditional, indices clamped).  n_q8: q8 wave-instructions
// (4 tokens each) that carry live tokens -- the single-wave form (batched decode: a few rows) skips the rest; their LDS rows keep stale bytes that only ever reach
// accumulator registers of tokens >= N, which are never stored.
template <int TT, int WPB>
__device__ __forceinline__ void mmq2_stage_load(const ActQ &A, int K, int NSB, int N, int t0, int sb, unsigned char *st, int wv, int lane, int n_q8) {
    using S = Mmq2Stage<TT>;
    // q8: TT * 8 wave-instructions of 4 tokens x 16 chunks; wave wv issues instructions wv * (8 TT / WPB) .. + 8 TT / WPB - 1
#pragma unroll
    for (int k = 0; k < 8 * TT / WPB; k++) {
        const int ii = wv * (8 * TT / WPB) + k;
        const int tl = 4 * ii + (lane >> 4);                 // token within the chunk
        const int c = (lane & 15) ^ (tl & 15);               // logical chunk that lands in slot (lane & 15)
        const int tok = min(t0 + tl, N - 1);
        if (WPB == 4 || ii < n_q8) dma16(A.q8k + (size_t)tok * K + (size_t)sb * 256 + c * 16, st + ii * 1024);
    }
    if (WPB == 1 || wv == 0) {
#pragma unroll
        for (int j = 0; j < S::TTP / 2; j++) { const int tok = min(t0 + 64 * j + lane, N - 1); dma16(A.bsq + ((size_t)tok * NSB + sb) * 16, st + S::Q8 + j * 1024); }
    }
    if (WPB == 1 || wv == 1) {
#pragma unroll
        for (int j = 0; j < S::TTP / 2; j++) { const int tok = min(t0 + 64 * j + lane, N - 1); dma4(A.dk + (size_t)tok * NSB + sb, st + S::Q8 + S::BS + j * 256); }
    }
}

// Result store: lane = weight row, accumulator register = token -> 32 consecutive floats per (register, lane half).  The residual values of a token tile are
// requested together (clamped addresses, no branch around a load) before any of them is used.
template <int TT>
__device__ __forceinline__ void mmq2_store(const float (&acc)[TT][16], const Mmq2Args &a, int m, int orow, int rows, int t0, int my_tiles, int hh) {
    float *y = a.y[m] + (size_t)blockIdx.z * a.slab_stride;
    const bool has_res = gridDim.z == 1 && a.res[m] != nullptr;
    const float *rb = has_res ? a.res[m] : y;
    const int rowc = min(orow, rows - 1);
#pragma unroll
    for (int tt = 0; tt < TT; tt++) {
        if (tt < my_tiles) {
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) rv[r] = rb[(size_t)min(t0 + tt * 32 + tok_of(r, hh), a.N - 1) * a.ldy + rowc];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int tok = t0 + tt * 32 + tok_of(r, hh);
                if (tok < a.N && orow < rows) y[(size_t)tok * a.ldy + orow] = has_res ? acc[tt][r] + rv[r] : acc[tt][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Q4_K / Q5_K.  Unit u of a super-block (16 bytes of the repacked main plane): low nibbles = elements 64 (u >> 1) + 16 (u & 1) + i of sub-block 2 (u >> 1),
// high nibbles = the same elements of sub-block 2 (u >> 1) + 1.  Pair jp: lanes hh = 0 / 1 hold units 2 jp / 2 jp + 1 -> one K = 32 MFMA per sub-block.
// ---------------------------------------------------------------------------------------------------------------------
template <bool Q5, int TT, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, 2) void k_mmq2_q45k(const Mmq2Args a, const ActQ A) {
    using S = Mmq2Stage<TT>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_mmq2[];   // [2 activation stages][4 waves x 4 KiB weight transpose scratch]
    const int lane = threadIdx.x & 63, hh = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x / a.groups_each, g = blockIdx.x - m * a.groups_each;
    const QWeight W = a.w[m];
    const int K = W.cols, U = K / 32, NSB = K / 256, N = a.N;
    const int r0 = (g * WPB + wv) * 32;
    const int row = min(r0 + l31, W.rows - 1);
    const int tile0 = blockIdx.y * a.tiles_per_chunk, my_tiles = min(TT, a.n_tiles - tile0), t0 = tile0 * 32;
    const int sb0 = blockIdx.z * a.sb_per_split, sb1 = min(NSB, sb0 + a.sb_per_split);
    unsigned char *scratch = smem_mmq2 + 2 * S::BYTES + wv * 4096;
    const int n_q8 = (min(N - t0, 32 * TT) + 3) / 4;             // q8 staging instructions that carry live tokens (single-wave form)

    float acc[TT][16];
#pragma unroll
    for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tt][r] = 0.0f;

    // Weight requests of one super-block, COALESCED (a lane-per-row gather costs 32 cache-line requests per instruction and the address path, not HBM, becomes the
    // bound -- measured: the staging loop alone took 58 % of the round-2a kernel): main plane: instruction n reads rows 8 n .. 8 n + 7 x the super-block's 8 units
    // (128 contiguous bytes per row); high-bit plane: lane (row, hh) reads the 16 bytes of units 4 hh .. 4 hh + 3; header: 16 bytes per row.
    struct Raw { v4i q[4]; v4i p; v4i h; };
    const unsigned char *wq[4];
#pragma unroll
    for (int n = 0; n < 4; n++) wq[n] = W.qs + ((size_t)min(r0 + 8 * n + (lane >> 3), W.rows - 1) * U + (lane & 7)) * 16;
    const unsigned char *wp = W.qh + (size_t)row * U * 4 + hh * 16, *wh = W.sc + (size_t)row * NSB * 16;
    auto fetch = [&](int sb, Raw &w) {
#pragma unroll
        for (int n = 0; n < 4; n++) w.q[n] = ldg16(wq[n] + (size_t)sb * 128);
        if (Q5) w.p = ldg16(wp + (size_t)sb * 32);
        w.h = ldg16(wh + (size_t)sb * 16);
    };
    // transpose scratch: [32 rows][8 units x 16 B], unit u of row r at slot u ^ ((r >> 1) & 7) (conflict-free for the row-per-lane fragment reads)
    unsigned sw_addr[4], sr_addr[4];
#pragma unroll
    for (int n = 0; n < 4; n++) sw_addr[n] = (unsigned)((8 * n + (lane >> 3)) * 128 + (((lane & 7) ^ ((4 * n + (lane >> 4)) & 7)) << 4));
#pragma unroll
    for (int jp = 0; jp < 4; jp++) sr_addr[jp] = (unsigned)(l31 * 128 + (((2 * jp + hh) ^ ((l31 >> 1) & 7)) << 4));
    // per-lane LDS read addresses (stage 0): A fragment of chunk C = 4 jp + 2 x (x = 0: low-nibble sub-block, 1: high) for token l31 of tile 0
    const int v = hh ^ (lane & 15);
    unsigned a_addr[8];
#pragma unroll
    for (int c8 = 0; c8 < 8; c8++) a_addr[c8] = (unsigned)(l31 * 256 + (((2 * c8) ^ v) << 4));
    const unsigned bs_addr = (unsigned)(S::Q8 + l31 * 16), dk_addr = (unsigned)(S::Q8 + S::BS + 16 * hh);

    Raw raw;
    fetch(sb0, raw);
    mmq2_stage_load<TT, WPB>(A, K, NSB, N, t0, sb0, smem_mmq2, wv, lane, n_q8);
    for (int sb = sb0; sb < sb1; sb++) {
        const int buf = (sb - sb0) & 1;
        unsigned char *st = smem_mmq2 + buf * S::BYTES;
        __syncthreads();                                           // own DMA + weight loads done (vmcnt(0) is part of the barrier's fence), then everybody's
        // ---- this super-block's weights: transpose through LDS (same wave writes and reads: LDS operations of a wave execute in order), unpack into MFMA B operands
#pragma unroll
        for (int n = 0; n < 4; n++) *reinterpret_cast<v4i *>(scratch + sw_addr[n]) = raw.q[n];
        unsigned P[4] = {0u, 0u, 0u, 0u};
        if (Q5) {   // lanes hh = 0 hold the high bits of units 0..3, hh = 1 of units 4..7: lane (row, hh) needs units 2 jp + hh
            const auto s01 = __builtin_amdgcn_permlane32_swap((unsigned)raw.p[0], (unsigned)raw.p[1], false, false);
            const auto s23 = __builtin_amdgcn_permlane32_swap((unsigned)raw.p[2], (unsigned)raw.p[3], false, false);
            P[0] = s01[0]; P[2] = s01[1]; P[1] = s23[0]; P[3] = s23[1];
        }
        v4i wlo[4], whi[4];
#pragma unroll
        for (int jp = 0; jp < 4; jp++) {
            const v4i q = *reinterpret_cast<const v4i *>(scratch + sr_addr[jp]); const unsigned Pj = P[jp];
            wlo[jp][0] = (q[0] & 0x0F0F0F0F) | (int)((Pj << 4) & 0x10101010u); wlo[jp][1] = (q[1] & 0x0F0F0F0F) | (int)((Pj << 3) & 0x10101010u);
            wlo[jp][2] = (q[2] & 0x0F0F0F0F) | (int)((Pj << 2) & 0x10101010u); wlo[jp][3] = (q[3] & 0x0F0F0F0F) | (int)((Pj << 1) & 0x10101010u);
            whi[jp][0] = ((q[0] >> 4) & 0x0F0F0F0F) | (int)(Pj & 0x10101010u); whi[jp][1] = ((q[1] >> 4) & 0x0F0F0F0F) | (int)((Pj >> 1) & 0x10101010u);
            whi[jp][2] = ((q[2] >> 4) & 0x0F0F0F0F) | (int)((Pj >> 2) & 0x10101010u); whi[jp][3] = ((q[3] >> 4) & 0x0F0F0F0F) | (int)((Pj >> 3) & 0x10101010u);
        }
        const unsigned s0 = (unsigned)raw.h[1], s1 = (unsigned)raw.h[2], s2 = (unsigned)raw.h[3];
        const unsigned scw0 = s0 & 0x3f3f3f3fu, scw1 = (s2 & 0x0f0f0f0fu) | (((s0 >> 6) & 0x03030303u) << 4);          // scales of sub-blocks 0..3 / 4..7, one byte each
        const unsigned mw0 = s1 & 0x3f3f3f3fu, mw1 = ((s2 >> 4) & 0x0f0f0f0fu) | (((s1 >> 6) & 0x03030303u) << 4);     // mins
        const float dw = h2f_b((unsigned)raw.h[0] & 0xFFFF), ndmin = -h2f_b((unsigned)raw.h[0] >> 16);
        // min term operands: A bytes = {lo_0..7, hi_0..7} of the token (lanes hh = 0), B = {m_0..7, 0} resp. {0, m_0..7}; lanes hh = 1 contribute nothing
        const v4i bm_lo = {hh ? 0 : (int)mw0, hh ? 0 : (int)mw1, 0, 0}, bm_hi = {0, 0, hh ? 0 : (int)mw0, hh ? 0 : (int)mw1};
        int sc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) sc[j] = (int)(((j & 4) ? scw1 : scw0) >> (8 * (j & 3))) & 0xFF;
        __builtin_amdgcn_sched_barrier(0);                         // the raw registers are dead from here: the next super-block's loads reuse them (no second register stage)
        // ---- request the next super-block (weights into the now free raw registers, activations into the other LDS buffer: every wave has passed the barrier,
        // so nobody still reads it)
        {
            const int sbn = min(sb + 1, sb1 - 1);
            fetch(sbn, raw);
#ifndef MG4_MMQ2_NOSTAGE   // diagnostic build only (make nostage; tools/mmq2_bench.py): what the launch would cost if the activation staging were free -- results are wrong
            mmq2_stage_load<TT, WPB>(A, K, NSB, N, t0, sbn, smem_mmq2 + (buf ^ 1) * S::BYTES, wv, lane, n_q8);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- token tiles.  Within a tile the five MFMA pairs (4 sub-block pairs + the min term) run one step ahead of the integer scale multiply-adds that
        // consume them (two result sets): an in-order wave otherwise sits out every MFMA's result latency before its 32 dependent VALU operations.
#pragma unroll
        for (int tt = 0; tt < TT; tt++) {
            if (tt < my_tiles) {
                const unsigned char *sq = st + tt * 8192;
                v16i isum = zero16();
                v16i D[2][2];
#define MMQ2_ISSUE(jp, b) { const v4i alo_ = *reinterpret_cast<const v4i *>(sq + a_addr[2 * (jp)]), ahi_ = *reinterpret_cast<const v4i *>(sq + a_addr[2 * (jp) + 1]);     \
                            D[b][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo_, wlo[jp], zero16(), 0, 0, 0); D[b][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi_, whi[jp], zero16(), 0, 0, 0); }
#define MMQ2_MADS(jp, b) { v16i t_;                                                                                                                                  \
                           _Pragma("unroll") for (int r = 0; r < 16; r++) t_[r] = __mul24(D[b][0][r], sc[2 * (jp)]) + isum[r];                                        \
                           asm volatile("" : "+v"(t_));      /* one v_mad_i32_i24 per element: without the pin LLVM emits mul, mul, add3 (3 instructions for 2) */      \
                           _Pragma("unroll") for (int r = 0; r < 16; r++) isum[r] = __mul24(D[b][1][r], sc[2 * (jp) + 1]) + t_[r];                                    \
                           asm volatile("" : "+v"(isum)); }
                MMQ2_ISSUE(0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE(1, 1)
                MMQ2_MADS(0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE(2, 0)
                MMQ2_MADS(1, 1)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE(3, 1)
                MMQ2_MADS(2, 0)
                __builtin_amdgcn_sched_barrier(0);
                {
                    const v4i abs_ = *reinterpret_cast<const v4i *>(st + bs_addr + tt * 512);
                    D[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(abs_, bm_lo, zero16(), 0, 0, 0);
                    D[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(abs_, bm_hi, zero16(), 0, 0, 0);
                }
                MMQ2_MADS(3, 1)
                __builtin_amdgcn_sched_barrier(0);
#undef MMQ2_ISSUE
#undef MMQ2_MADS
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const v4f da = *reinterpret_cast<const v4f *>(st + dk_addr + tt * 128 + q4 * 32);   // tokens 8 q4 + 4 hh + 0..3 = accumulator registers 4 q4 .. 4 q4 + 3
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * q4 + e;
                        acc[tt][r] = fmaf(dw * da[e], (float)isum[r], acc[tt][r]);
                        acc[tt][r] = fmaf(ndmin * da[e], (float)(D[0][1][r] * 128 + D[0][0][r]), acc[tt][r]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    mmq2_store<TT>(acc, a, m, r0 + l31, W.rows, t0, my_tiles, hh);
}

// ---------------------------------------------------------------------------------------------------------------------
// Q6_K.  Unit u = 4 n + 2 c + h: low nibbles (+ 2 high bits) = elements 128 n + 32 c + 16 h + i, high nibbles = the same + 64; int8 scale per 16 elements.
// Pair p = 2 n + c: lanes hh = 0 / 1 load units (.., h = 0) / (.., h = 1).  A scale group is 16 weights, so the products run on v_mfma_i32_32x32x16_i8 (K = 16: one group per
// instruction, 8 bytes per lane; round 1 multiplied K = 32 operands twice with one lane half zeroed, which costs two 4-register operand copies per pair and token tile
// budget).  Weights are q - 32 in int8; no min term.
// ---------------------------------------------------------------------------------------------------------------------
template <int TT, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, 2) void k_mmq2_q6k(const Mmq2Args a, const ActQ A) {
    using S = Mmq2Stage<TT>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_mmq2[];   // [2 activation stages][4 waves x 4 KiB weight transpose scratch]
    const int lane = threadIdx.x & 63, hh = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x / a.groups_each, g = blockIdx.x - m * a.groups_each;
    const QWeight W = a.w[m];
    const int K = W.cols, U = K / 32, NSB = K / 256, N = a.N;
    const int r0 = (g * WPB + wv) * 32;
    const int row = min(r0 + l31, W.rows - 1);
    const int tile0 = blockIdx.y * a.tiles_per_chunk, my_tiles = min(TT, a.n_tiles - tile0), t0 = tile0 * 32;
    const int sb0 = blockIdx.z * a.sb_per_split, sb1 = min(NSB, sb0 + a.sb_per_split);
    unsigned char *scratch = smem_mmq2 + 2 * S::BYTES + wv * 4096;
    const int n_q8 = (min(N - t0, 32 * TT) + 3) / 4;             // q8 staging instructions that carry live tokens (single-wave form)

    float acc[TT][16];
#pragma unroll
    for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tt][r] = 0.0f;

    // coalesced weight requests (see k_mmq2_q45k): main plane 8 rows x 128 B per instruction; high-bit plane (8 B per unit): lane (row, hh) reads the 32 bytes of units
    // 4 hh .. 4 hh + 3; scale plane: the row's 8 x {lo, hi} int8 scales of the super-block in one 16-byte load
    struct Raw { v4i q[4]; v4i p[2]; v4i sc; unsigned short d; };
    const unsigned char *wq[4];
#pragma unroll
    for (int n = 0; n < 4; n++) wq[n] = W.qs + ((size_t)min(r0 + 8 * n + (lane >> 3), W.rows - 1) * U + (lane & 7)) * 16;
    const unsigned char *wp = W.qh + (size_t)row * U * 8 + hh * 32, *ws = W.sc + (size_t)row * U * 2, *wd = W.d + (size_t)row * NSB * 2;
    auto fetch = [&](int sb, Raw &w) {
#pragma unroll
        for (int n = 0; n < 4; n++) w.q[n] = ldg16(wq[n] + (size_t)sb * 128);
        w.p[0] = ldg16(wp + (size_t)sb * 64); w.p[1] = ldg16(wp + (size_t)sb * 64 + 16);
        w.sc = ldg16(ws + (size_t)sb * 16);
        w.d = *reinterpret_cast<const unsigned short *>(wd + (size_t)sb * 2);
    };
    unsigned sw_addr[4], sr_addr[4];
#pragma unroll
    for (int n = 0; n < 4; n++) sw_addr[n] = (unsigned)((8 * n + (lane >> 3)) * 128 + (((lane & 7) ^ ((4 * n + (lane >> 4)) & 7)) << 4));
#pragma unroll
    for (int p = 0; p < 4; p++) sr_addr[p] = (unsigned)(l31 * 128 + (((2 * p + hh) ^ ((l31 >> 1) & 7)) << 4));
    // A fragments (8 bytes per lane): group gsel of the low (x = 0) / high (x = 1) part of pair p = 2 n + c = elements 128 n + 32 c + 64 x + 16 gsel + 8 hh .. + 7
    // -> chunk 8 n + 2 c + 4 x + gsel of the token's row (stored at slot chunk ^ (token & 15)), byte 8 hh within it
    const int tk = lane & 15;
    unsigned a_addr[16];                                       // [4 p + 2 x + gsel]
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
            for (int gs = 0; gs < 2; gs++) a_addr[4 * p + 2 * x + gs] = (unsigned)(l31 * 256 + (((8 * (p >> 1) + 2 * (p & 1) + 4 * x + gs) ^ tk) << 4) + 8 * hh);
    const unsigned dk_addr = (unsigned)(S::Q8 + S::BS + 16 * hh);

    Raw raw;
    fetch(sb0, raw);
    mmq2_stage_load<TT, WPB>(A, K, NSB, N, t0, sb0, smem_mmq2, wv, lane, n_q8);
    for (int sb = sb0; sb < sb1; sb++) {
        const int buf = (sb - sb0) & 1;
        unsigned char *st = smem_mmq2 + buf * S::BYTES;
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 4; n++) *reinterpret_cast<v4i *>(scratch + sw_addr[n]) = raw.q[n];
        // high bits: lanes hh = 0 hold units 0..3 as {lo0, hi0, lo1, hi1 | lo2, hi2, lo3, hi3}, lanes hh = 1 units 4..7; lane (row, hh) needs units 2 p + hh
        unsigned PL[4], PH[4];
        {
            const auto l01 = __builtin_amdgcn_permlane32_swap((unsigned)raw.p[0][0], (unsigned)raw.p[0][2], false, false);   // lo of units (0|4) x (1|5) -> (0|1), (4|5)
            const auto h01 = __builtin_amdgcn_permlane32_swap((unsigned)raw.p[0][1], (unsigned)raw.p[0][3], false, false);
            const auto l23 = __builtin_amdgcn_permlane32_swap((unsigned)raw.p[1][0], (unsigned)raw.p[1][2], false, false);   // (2|6) x (3|7) -> (2|3), (6|7)
            const auto h23 = __builtin_amdgcn_permlane32_swap((unsigned)raw.p[1][1], (unsigned)raw.p[1][3], false, false);
            PL[0] = l01[0]; PL[2] = l01[1]; PL[1] = l23[0]; PL[3] = l23[1];
            PH[0] = h01[0]; PH[2] = h01[1]; PH[1] = h23[0]; PH[3] = h23[1];
        }
        // MFMA B operands (v_mfma_i32_32x32x16_i8: 8 bytes per lane, one 16-weight scale group per instruction): after the unpack lane (row, h) holds the 16 weights of
        // group h; one permlane32_swap per dword pair leaves lane (row, 0) with elements 0..7 and lane (row, 1) with elements 8..15 of BOTH groups:
        // wl[p] = {group h = 0: 2 dwords | group h = 1: 2 dwords} of the low-nibble part, wh[p] of the high-nibble part
        v4i wl[4], wh[4];
        unsigned scp[4];                                          // per pair: {unit h = 0: lo, hi | unit h = 1: lo, hi} int8 scales, extracted where they are used (registers)
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const v4i q = *reinterpret_cast<const v4i *>(scratch + sr_addr[p]); const unsigned L = PL[p], H = PH[p];
            v4i wlo, whi;
            wlo[0] = sext6((q[0] & 0x0F0F0F0F) | (int)((L << 4) & 0x30303030u)); wlo[1] = sext6((q[1] & 0x0F0F0F0F) | (int)((L << 2) & 0x30303030u));
            wlo[2] = sext6((q[2] & 0x0F0F0F0F) | (int)(L & 0x30303030u)); wlo[3] = sext6((q[3] & 0x0F0F0F0F) | (int)((L >> 2) & 0x30303030u));
            whi[0] = sext6(((q[0] >> 4) & 0x0F0F0F0F) | (int)((H << 4) & 0x30303030u)); whi[1] = sext6(((q[1] >> 4) & 0x0F0F0F0F) | (int)((H << 2) & 0x30303030u));
            whi[2] = sext6(((q[2] >> 4) & 0x0F0F0F0F) | (int)(H & 0x30303030u)); whi[3] = sext6(((q[3] >> 4) & 0x0F0F0F0F) | (int)((H >> 2) & 0x30303030u));
            {   // permlane32_swap(X, Y) exchanges X's lanes 32..63 with Y's lanes 0..31.  X = w[0] = (hh = 0: g0[0..3] | hh = 1: g1[0..3]), Y = w[2] = (g0[8..11] | g1[8..11])
                // -> X' = (g0[0..3] | g0[8..11]) = group 0, Y' = (g1[0..3] | g1[8..11]) = group 1; likewise w[1] / w[3] for elements 4..7 / 12..15
                const auto l0 = __builtin_amdgcn_permlane32_swap((unsigned)wlo[0], (unsigned)wlo[2], false, false);
                const auto l1 = __builtin_amdgcn_permlane32_swap((unsigned)wlo[1], (unsigned)wlo[3], false, false);
                const auto h0 = __builtin_amdgcn_permlane32_swap((unsigned)whi[0], (unsigned)whi[2], false, false);
                const auto h1 = __builtin_amdgcn_permlane32_swap((unsigned)whi[1], (unsigned)whi[3], false, false);
                wl[p] = v4i{(int)l0[0], (int)l1[0], (int)l0[1], (int)l1[1]};
                wh[p] = v4i{(int)h0[0], (int)h1[0], (int)h0[1], (int)h1[1]};
            }
            scp[p] = (unsigned)raw.sc[p];                             // units 2 p, 2 p + 1: {lo, hi} int8 scales each
        }
        const float dw = h2f_b(raw.d);
        __builtin_amdgcn_sched_barrier(0);
        {
            const int sbn = min(sb + 1, sb1 - 1);
            fetch(sbn, raw);
#ifndef MG4_MMQ2_NOSTAGE   // diagnostic build only (make nostage; tools/mmq2_bench.py): what the launch would cost if the activation staging were free -- results are wrong
            mmq2_stage_load<TT, WPB>(A, K, NSB, N, t0, sbn, smem_mmq2 + (buf ^ 1) * S::BYTES, wv, lane, n_q8);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        // eight half-pair steps per tile (low / high nibble part of pair p), MFMAs one step ahead of the scale multiply-adds
#pragma unroll
        for (int tt = 0; tt < TT; tt++) {
            if (tt < my_tiles) {
                const unsigned char *sq = st + tt * 8192;
                v16i isum = zero16();
                v16i D[2][2];
#define MMQ2_ISSUE6(p, x, b) { const long a0_ = *reinterpret_cast<const long *>(sq + a_addr[4 * (p) + 2 * (x)]), a1_ = *reinterpret_cast<const long *>(sq + a_addr[4 * (p) + 2 * (x) + 1]); \
                               const v4i w_ = (x) ? wh[p] : wl[p];                                                                                               \
                               D[b][0] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a0_, pack64(w_[0], w_[1]), zero16(), 0, 0, 0);                                    \
                               D[b][1] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a1_, pack64(w_[2], w_[3]), zero16(), 0, 0, 0); }
#define MMQ2_MADS6(p, x, b) { const int sa_ = (int)(signed char)((scp[p] >> (8 * (x))) & 0xFF), sb_ = (int)(signed char)((scp[p] >> (16 + 8 * (x))) & 0xFF);                                                   \
                              v16i t_;                                                                                                                        \
                              _Pragma("unroll") for (int r = 0; r < 16; r++) t_[r] = __mul24(D[b][0][r], sa_) + isum[r];                                            \
                              asm volatile("" : "+v"(t_));                                                                                                          \
                              _Pragma("unroll") for (int r = 0; r < 16; r++) isum[r] = __mul24(D[b][1][r], sb_) + t_[r];                                            \
                              asm volatile("" : "+v"(isum)); }
                MMQ2_ISSUE6(0, 0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE6(0, 1, 1) MMQ2_MADS6(0, 0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE6(1, 0, 0) MMQ2_MADS6(0, 1, 1)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE6(1, 1, 1) MMQ2_MADS6(1, 0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE6(2, 0, 0) MMQ2_MADS6(1, 1, 1)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE6(2, 1, 1) MMQ2_MADS6(2, 0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE6(3, 0, 0) MMQ2_MADS6(2, 1, 1)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE6(3, 1, 1) MMQ2_MADS6(3, 0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_MADS6(3, 1, 1)
#undef MMQ2_ISSUE6
#undef MMQ2_MADS6
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const v4f da = *reinterpret_cast<const v4f *>(st + dk_addr + tt * 128 + q4 * 32);
#pragma unroll
                    for (int e = 0; e < 4; e++) { const int r = 4 * q4 + e; acc[tt][r] = fmaf(dw * da[e], (float)isum[r], acc[tt][r]); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    mmq2_store<TT>(acc, a, m, r0 + l31, W.rows, t0, my_tiles, hh);
}

// ---------------------------------------------------------------------------------------------------------------------
// Q4_0 (the 7B file of BASELINE configs[1..2]; round 1's k_mmq_q40 took 24.6 ms for the 142-row image turn).  Block = 32 weights = one 16-byte unit (low nibbles =
// elements 0..15, high nibbles = 16..31) + one fp16 scale; activations are Q8_0 (int8 + one fp32 scale per 32).  Same skeleton as the k-quant kernels: 8 blocks
// ("super-chunk" of 256 weights, 128 contiguous bytes per row) per stage, weights read coalesced and transposed through the per-wave LDS scratch, activations and
// their scales by LDS-DMA; one K = 32 MFMA per block, nibbles sign-extended to q - 8 at unpack time (no per-32 sum correction), and per block
// acc = fma(d_w * d_a, (float)isum, acc) exactly as the decode kernel (Tr<GT_Q4_0>::dot) and ggml's AVX2 path do it.
// ---------------------------------------------------------------------------------------------------------------------
template <int TT> struct Mmq2Stage80 {
    static constexpr int TTP = (TT + 1) & ~1;
    static constexpr int Q8 = TT * 32 * 256, DS = 8 * TTP * 32 * 4;      // q8 as Mmq2Stage; scales [8 blocks][TTP * 32 tokens] fp32 (token-contiguous: 4 accumulator registers = 16 bytes)
    static constexpr int BYTES = Q8 + DS;
};
template <int TT, int WPB>
__device__ __forceinline__ void mmq2_stage_load80(const ActQ &A, int K, int N, int t0, int sb, unsigned char *st, int wv, int lane, const unsigned (&tokoff)[Mmq2Stage80<TT>::TTP / 2], int n_q8) {
    using S = Mmq2Stage80<TT>;
#pragma unroll
    for (int k = 0; k < 8 * TT / WPB; k++) {
        const int ii = wv * (8 * TT / WPB) + k;
        const int tl = 4 * ii + (lane >> 4);
        const int c = (lane & 15) ^ (tl & 15);
        const int tok = min(t0 + tl, N - 1);
        if (WPB == 4 || ii < n_q8) dma16(A.q80 + (size_t)tok * K + (size_t)sb * 256 + c * 16, st + ii * 1024);
    }
    // scales: instruction (b, j) = block b of the super-chunk, tokens 64 j + lane; wave wv takes blocks wv and wv + 4 (the single wave: all 8).  tokoff[j] (this lane's row
    // offset into d0) is computed once by the caller: rebuilding 64-bit addresses per instruction is cheaper than the 16 hoisted pointer pairs hipcc otherwise keeps (and spills)
    constexpr int HJ = S::TTP / 2;
#pragma unroll
    for (int j = 0; j < HJ; j++) {
        if (WPB == 4) {
            const float *pj = A.d0 + (size_t)tokoff[j] + (size_t)sb * 8 + wv;
            dma4(pj, st + S::Q8 + (wv * S::TTP * 32 + 64 * j) * 4);
            dma4(pj + 4, st + S::Q8 + ((wv + 4) * S::TTP * 32 + 64 * j) * 4);
        } else {
            const float *pj = A.d0 + (size_t)tokoff[j] + (size_t)sb * 8;
#pragma unroll
            for (int b = 0; b < 8; b++) dma4(pj + b, st + S::Q8 + (b * S::TTP * 32 + 64 * j) * 4);
        }
    }
}
template <int TT, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, 2) void k_mmq2_q40(const Mmq2Args a, const ActQ A) {
    using S = Mmq2Stage80<TT>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_mmq2[];   // [2 activation stages][4 waves x 4 KiB weight transpose scratch]
    const int lane = threadIdx.x & 63, hh = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x / a.groups_each, g = blockIdx.x - m * a.groups_each;
    const QWeight W = a.w[m];
    const int K = W.cols, U = K / 32, NSB = K / 256, N = a.N;
    const int r0 = (g * WPB + wv) * 32;
    const int row = min(r0 + l31, W.rows - 1);
    const int tile0 = blockIdx.y * a.tiles_per_chunk, my_tiles = min(TT, a.n_tiles - tile0), t0 = tile0 * 32;
    const int sb0 = blockIdx.z * a.sb_per_split, sb1 = min(NSB, sb0 + a.sb_per_split);
    unsigned char *scratch = smem_mmq2 + 2 * S::BYTES + wv * 4096;
    const int n_q8 = (min(N - t0, 32 * TT) + 3) / 4;             // q8 staging instructions that carry live tokens (single-wave form)

    typedef float v16f_t __attribute__((ext_vector_type(16)));
    v16f_t acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tt][r] = 0.0f;

    struct Raw { v4i q[4]; v4i h; };
    const unsigned char *wq[4];
#pragma unroll
    for (int n = 0; n < 4; n++) wq[n] = W.qs + ((size_t)min(r0 + 8 * n + (lane >> 3), W.rows - 1) * U + (lane & 7)) * 16;
    const unsigned char *wh = W.sc + (size_t)row * U * 2;
    auto fetch = [&](int sb, Raw &w) {
#pragma unroll
        for (int n = 0; n < 4; n++) w.q[n] = ldg16(wq[n] + (size_t)sb * 128);
        w.h = ldg16(wh + (size_t)sb * 16);
    };
    unsigned sw_addr[4], sr_addr[8], a_addr[8];
#pragma unroll
    for (int n = 0; n < 4; n++) sw_addr[n] = (unsigned)((8 * n + (lane >> 3)) * 128 + (((lane & 7) ^ ((4 * n + (lane >> 4)) & 7)) << 4));
#pragma unroll
    for (int b = 0; b < 8; b++) sr_addr[b] = (unsigned)(l31 * 128 + ((b ^ ((l31 >> 1) & 7)) << 4));     // both lane halves read unit b of their row (hh selects the nibble)
    const int v = hh ^ (lane & 15);
#pragma unroll
    for (int b = 0; b < 8; b++) a_addr[b] = (unsigned)(l31 * 256 + (((2 * b) ^ v) << 4));                // elements 32 b + 16 hh .. + 15 of token l31
    const unsigned ds_addr = (unsigned)(S::Q8 + 16 * hh);

    unsigned tokoff[S::TTP / 2];
#pragma unroll
    for (int j = 0; j < S::TTP / 2; j++) tokoff[j] = (unsigned)min(t0 + 64 * j + lane, N - 1) * (unsigned)(K / 32);
    Raw raw;
    fetch(sb0, raw);
    mmq2_stage_load80<TT, WPB>(A, K, N, t0, sb0, smem_mmq2, wv, lane, tokoff, n_q8);
    for (int sb = sb0; sb < sb1; sb++) {
        const int buf = (sb - sb0) & 1;
        unsigned char *st = smem_mmq2 + buf * S::BYTES;
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 4; n++) *reinterpret_cast<v4i *>(scratch + sw_addr[n]) = raw.q[n];
        float dw[8];
#pragma unroll
        for (int b = 0; b < 8; b++) dw[b] = h2f_b(((unsigned)raw.h[b >> 1] >> (16 * (b & 1))) & 0xFFFF);
        v4i wb[8];
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const v4i q = *reinterpret_cast<const v4i *>(scratch + sr_addr[b]);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const unsigned x = hh ? ((unsigned)q[w] >> 4) & 0x0F0F0F0Fu : (unsigned)q[w] & 0x0F0F0F0Fu;
                wb[b][w] = (int)(((x | 0x80808080u) - 0x08080808u) ^ 0x80808080u);   // per byte x - 8 as int8: bit 7 set keeps the subtraction from borrowing across bytes
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const int sbn = min(sb + 1, sb1 - 1);
            fetch(sbn, raw);
            mmq2_stage_load80<TT, WPB>(A, K, N, t0, sbn, smem_mmq2 + (buf ^ 1) * S::BYTES, wv, lane, tokoff, n_q8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < TT; tt++) {
            if (tt < my_tiles) {
                const unsigned char *sq = st + tt * 8192;
                v16i D[2];
#define MMQ2_ISSUE40(b, k) { const v4i af_ = *reinterpret_cast<const v4i *>(sq + a_addr[b]); D[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af_, wb[b], zero16(), 0, 0, 0); }
#define MMQ2_SCALE40(b, k) { _Pragma("unroll") for (int q4 = 0; q4 < 4; q4++) {                                                                                   \
                               const v4f da = *reinterpret_cast<const v4f *>(st + ds_addr + ((b) * S::TTP * 32 + tt * 32 + 8 * q4) * 4);                            \
                               _Pragma("unroll") for (int e = 0; e < 4; e++) { const int r = 4 * q4 + e; acc[tt][r] = fmaf(dw[b] * da[e], (float)D[k][r], acc[tt][r]); } }      \
                             asm volatile("" : "+v"(acc[tt])); }   /* pins the block's scale arithmetic here: without it LLVM sinks all 8 blocks' conversions and fmas behind the 8 MFMAs (8 result sets + 128 scales live -> scratch) */
                MMQ2_ISSUE40(0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE40(1, 1) MMQ2_SCALE40(0, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE40(2, 0) MMQ2_SCALE40(1, 1)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE40(3, 1) MMQ2_SCALE40(2, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE40(4, 0) MMQ2_SCALE40(3, 1)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE40(5, 1) MMQ2_SCALE40(4, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE40(6, 0) MMQ2_SCALE40(5, 1)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_ISSUE40(7, 1) MMQ2_SCALE40(6, 0)
                __builtin_amdgcn_sched_barrier(0);
                MMQ2_SCALE40(7, 1)
                __builtin_amdgcn_sched_barrier(0);
#undef MMQ2_ISSUE40
#undef MMQ2_SCALE40
            }
        }
    }
    float accf[TT][16];
#pragma unroll
    for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) accf[tt][r] = acc[tt][r];
    mmq2_store<TT>(accf, a, m, r0 + l31, W.rows, t0, my_tiles, hh);
}

// ---------------------------------------------------------------------------------------------------------------------
// Q4_K / Q5_K prompt rows on the FP16 matrix cores, sub-block scales folded into the weight operands (round 5; north_star: "block dequant fused into the ... matmul, MFMA
// bf16/fp16 tiles for ... Vicuna prefill").  The int8 kernels above are bound by vector issue: one integer multiply-add per output element per 32 weights for the 6-bit
// sub-block scales (8 per 256 weights, ~300 VALU instructions per token tile and super-block beside 10 MFMAs).  Here the scale rides INSIDE the matrix product:
//   * B operand = sc_j * q as fp16 -- an integer <= 63 * 31 = 1953 < 2048, EXACT in fp16 -- built once per super-block and row tile (mask + v_pk_fma_f16 per two weights:
//     (1024 + q) * sc - 1024 * sc) and reused for every token tile;
//   * A operand = the activation row's Q8_K values (ggml's quantisation, unchanged) as fp16, from the plane the activation quantiser writes next to the int8 one (ActQ::q16,
//     already in fragment order): |q8| <= 127 exact;
//   * v_mfma_f32_32x32x16_f16 accumulates the super-block's 256 products (<= 1953 * 127, exact in the fp32 accumulator up to 2^24, beyond that rounded at 6e-8 relative)
//     -> ONE fma per output element per super-block with d_w * d_a;  the min term sum_j m_j * bsum_j is one more MFMA on the digit split bsum = 128 hi + lo (exact).
// 17 MFMAs + 48 VALU operations per token tile and super-block instead of 10 + ~300; same ggml arithmetic (Q8_K activations, integer sub-block products, fp32 super-block
// scales) up to the fp32 rounding of the in-block sum, i.e. it differs from k_mmq2_q45k like one summation order from another.  Parity mode keeps k_mmq2_q45k (force_ks == 1).
// Stages are HALF super-blocks (128 weights = units 4 h .. 4 h + 3): the fp16 activation image is twice the int8 one, and half stages keep the LDS footprint of the int8
// kernel (two workgroups per CU).  Lane (row, hh) owns pair p = 2 h + hh of the half: units 2 p, 2 p + 1 = low nibbles -> sub-block 2 p, high nibbles -> sub-block 2 p + 1;
// MFMA (uu, d) of a half multiplies dword d of unit 2 p + uu: k order [lo e0, e2, e1, e3 | hi e0, e2, e1, e3] (what two masks of one dword give), the activation plane is
// stored in that order.
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int TT> struct MmqhStage {
    static constexpr int TTP = (TT + 1) & ~1;
    static constexpr int Q = TT * 32 * 256;                  // half super-block of fp16 values: 16 chunks of 16 B per token, chunk c at slot c ^ (token & 15)
    static constexpr int BS = 2 * TTP * 32 * 16, DK = TTP * 32 * 4;   // per SUPER-block: digit halves [lo | hi][token][16 B], fp32 scale
    static constexpr int BYTES = 2 * Q + 2 * (BS + DK);      // two half stages + two super-block side buffers
};
__device__ __forceinline__ unsigned pkfma(unsigned x, unsigned s2, unsigned c2) {   // v_pk_fma_f16 on bit patterns
    const h2_t r = __builtin_elementwise_fma(__builtin_bit_cast(h2_t, x), __builtin_bit_cast(h2_t, s2), __builtin_bit_cast(h2_t, c2));
    return __builtin_bit_cast(unsigned, r);
}
template <bool Q5, int TT>
__global__ __launch_bounds__(256, 2) void k_mmqh_q45k(const Mmq2Args a, const ActQ A) {
    using S = MmqhStage<TT>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_mmq2[];   // [2 half stages][2 x (digits, scales)][4 waves x 4 KiB weight transpose scratch]
    const int lane = threadIdx.x & 63, hh = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x / a.groups_each, g = blockIdx.x - m * a.groups_each;
    const QWeight W = a.w[m];
    const int K = W.cols, U = K / 32, NSB = K / 256, N = a.N;
    const int r0 = (g * 4 + wv) * 32;
    const int row = min(r0 + l31, W.rows - 1);
    const int tile0 = blockIdx.y * a.tiles_per_chunk, my_tiles = min(TT, a.n_tiles - tile0), t0 = tile0 * 32;
    const int sb0 = blockIdx.z * a.sb_per_split, sb1 = min(NSB, sb0 + a.sb_per_split);
    unsigned char *const side0 = smem_mmq2 + 2 * S::Q;
    unsigned char *scratch = smem_mmq2 + S::BYTES + wv * 4096;

    v16f acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tt][r] = 0.0f;

    struct Raw { v4i q[4]; v4i p; v4i h; };
    const unsigned char *wq[4];
#pragma unroll
    for (int n = 0; n < 4; n++) wq[n] = W.qs + ((size_t)min(r0 + 8 * n + (lane >> 3), W.rows - 1) * U + (lane & 7)) * 16;
    const unsigned char *wp = W.qh + (size_t)row * U * 4 + hh * 16, *wh = W.sc + (size_t)row * NSB * 16;
    auto fetch = [&](int sb, Raw &w) {
#pragma unroll
        for (int n = 0; n < 4; n++) w.q[n] = ldg16(wq[n] + (size_t)sb * 128);
        if (Q5) w.p = ldg16(wp + (size_t)sb * 32);
        w.h = ldg16(wh + (size_t)sb * 16);
    };
    // half-stage requests: TT * 8 wave-instructions of 4 tokens x 16 chunks (the half's 256 bytes per token); wave wv issues 2 TT of them
    // (per-lane byte offsets once, 32 bits: the rows of a prompt pass span < 2^31 bytes; per step only a wave-uniform offset is added)
    unsigned qoff[2 * TT], soff[S::TTP / 2];
#pragma unroll
    for (int k = 0; k < 2 * TT; k++) {
        const int tl = 4 * (wv * (2 * TT) + k) + (lane >> 4);
        qoff[k] = (unsigned)min(t0 + tl, N - 1) * (unsigned)K * 2u + (unsigned)(((lane & 15) ^ (tl & 15)) * 16);
    }
#pragma unroll
    for (int j = 0; j < S::TTP / 2; j++) soff[j] = (unsigned)min(t0 + 64 * j + lane, N - 1) * (unsigned)NSB;
    const unsigned char *const q16b = reinterpret_cast<const unsigned char *>(A.q16), *const bs16b = reinterpret_cast<const unsigned char *>(A.bs16);
    auto stage_half = [&](int sb, int h, unsigned char *st) {
        const unsigned step = (unsigned)sb * 512u + (unsigned)h * 256u;
#pragma unroll
        for (int k = 0; k < 2 * TT; k++) dma16(q16b + (size_t)(qoff[k] + step), st + (wv * (2 * TT) + k) * 1024);
    };
    auto stage_side = [&](int sb, unsigned char *sd) {      // the super-block's digit sums (wave 0: low digits, wave 1: high digits) and scales (wave 2)
#pragma unroll
        for (int j = 0; j < S::TTP / 2; j++) {
            if (wv < 2) dma16(bs16b + (size_t)((soff[j] + (unsigned)sb) * 32u + (unsigned)wv * 16u), sd + wv * (S::TTP * 32 * 16) + j * 1024);
            if (wv == 2) dma4(A.dk + (size_t)(soff[j] + (unsigned)sb), sd + S::BS + j * 256);
        }
    };
    unsigned sw_addr[4];
#pragma unroll
    for (int n = 0; n < 4; n++) sw_addr[n] = (unsigned)((8 * n + (lane >> 3)) * 128 + (((lane & 7) ^ ((4 * n + (lane >> 4)) & 7)) << 4));
    // A fragment of MFMA (uu, d): chunk 8 hh + 4 uu + d of the half's 16, token l31 of tile 0
    unsigned a_addr[8];
#pragma unroll
    for (int c8 = 0; c8 < 8; c8++) a_addr[c8] = (unsigned)(l31 * 256 + (((8 * hh + c8) ^ (l31 & 15)) << 4));
    const unsigned bs_addr = (unsigned)(hh * (S::TTP * 32 * 16) + l31 * 16), dk_addr = (unsigned)(S::BS + 16 * hh);

    Raw raw;
    fetch(sb0, raw);
    stage_half(sb0, 0, smem_mmq2);
    stage_side(sb0, side0);
    unsigned Pn[2][2] = {{0u, 0u}, {0u, 0u}};
    unsigned scw[2] = {0u, 0u};
    float dw = 0.0f, ndmin = 0.0f;
    v4i bmin = {0, 0, 0, 0};
    for (int sb = sb0; sb < sb1; sb++) {
        unsigned char *const sd = side0 + ((sb - sb0) & 1) * (S::BS + S::DK);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            unsigned char *const st = smem_mmq2 + h * S::Q;
            __syncthreads();                                       // own DMA + weight loads landed (vmcnt(0) is part of the barrier's fence), then everybody's
            if (h == 0) {
                // this super-block's weights: transpose through LDS (a wave's LDS operations execute in order), scales / mins / high bits into registers
#pragma unroll
                for (int n = 0; n < 4; n++) *reinterpret_cast<v4i *>(scratch + sw_addr[n]) = raw.q[n];
                if (Q5) {   // lanes hh = 0 hold the high-bit words of units 0..3, hh = 1 of units 4..7; lane (row, hh) needs units 4 h + 2 hh + uu
                    const auto s02 = __builtin_amdgcn_permlane32_swap((unsigned)raw.p[0], (unsigned)raw.p[2], false, false);
                    const auto s13 = __builtin_amdgcn_permlane32_swap((unsigned)raw.p[1], (unsigned)raw.p[3], false, false);
                    Pn[0][0] = s02[0]; Pn[1][0] = s02[1]; Pn[0][1] = s13[0]; Pn[1][1] = s13[1];
                }
                const unsigned s0 = (unsigned)raw.h[1], s1 = (unsigned)raw.h[2], s2 = (unsigned)raw.h[3];
                scw[0] = s0 & 0x3f3f3f3fu; scw[1] = (s2 & 0x0f0f0f0fu) | (((s0 >> 6) & 0x03030303u) << 4);                    // scales of sub-blocks 0..3 / 4..7, one byte each
                const unsigned mw0 = s1 & 0x3f3f3f3fu, mw1 = ((s2 >> 4) & 0x0f0f0f0fu) | (((s1 >> 6) & 0x03030303u) << 4);   // mins
                dw = h2f_b((unsigned)raw.h[0] & 0xFFFF); ndmin = -h2f_b((unsigned)raw.h[0] >> 16);
                // min-term B operand: k = 8 hh + j -> lanes hh = 0 carry m_j (against the low digits), hh = 1 carry 128 m_j (against the high digits); fp16 integers, exact
                const unsigned mult = hh ? 0x58005800u : 0x3C003C00u;                                                         // half2(128) : half2(1)
                const unsigned mm[4] = {(mw0 & 0xFFu) | ((mw0 & 0xFF00u) << 8), ((mw0 >> 16) & 0xFFu) | ((mw0 >> 8) & 0xFF0000u), (mw1 & 0xFFu) | ((mw1 & 0xFF00u) << 8), ((mw1 >> 16) & 0xFFu) | ((mw1 >> 8) & 0xFF0000u)};
#pragma unroll
                for (int i = 0; i < 4; i++) bmin[i] = (int)pkfma(pkfma(mm[i] | 0x64006400u, 0x3C003C00u, 0xE400E400u), mult, 0u);   // ((1024 + m) - 1024) * mult: m, resp. 128 m <= 8064
            }
            __builtin_amdgcn_sched_barrier(0);
            // request the next half stage (the other buffer: every wave has passed the barrier, nobody still reads it); at h = 1 also the next super-block's weights and side data
            if (h == 0) stage_half(sb, 1, smem_mmq2 + S::Q);
            else {
                const int sbn = min(sb + 1, sb1 - 1);
                fetch(sbn, raw);
                stage_half(sbn, 0, smem_mmq2);
                stage_side(sbn, side0 + ((sb + 1 - sb0) & 1) * (S::BS + S::DK));
            }
            __builtin_amdgcn_sched_barrier(0);
            // B operands of this half: units 2 p, 2 p + 1 of pair p = 2 h + hh, scaled by sc[2 p] (low nibbles) / sc[2 p + 1] (high nibbles)
            const unsigned scb = scw[h] >> (16 * hh);
            const unsigned short slo_h = __half_as_ushort(__int2half_rn((int)(scb & 0xFFu))), shi_h = __half_as_ushort(__int2half_rn((int)((scb >> 8) & 0xFFu)));
            const unsigned slo2 = (unsigned)slo_h * 0x10001u, shi2 = (unsigned)shi_h * 0x10001u;
            const unsigned clo2 = (unsigned)__half_as_ushort(__float2half_rn(-1024.0f * (float)(scb & 0xFFu))) * 0x10001u, chi2 = (unsigned)__half_as_ushort(__float2half_rn(-1024.0f * (float)((scb >> 8) & 0xFFu))) * 0x10001u;
            v4i wb[8];
#pragma unroll
            for (int uu = 0; uu < 2; uu++) {
                const v4i q = *reinterpret_cast<const v4i *>(scratch + (unsigned)(l31 * 128 + (((4 * h + 2 * hh + uu) ^ ((l31 >> 1) & 7)) << 4)));
                const unsigned P = Pn[h][uu];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    unsigned lo = (unsigned)q[d], hi = (unsigned)q[d] >> 4;
                    unsigned mask = 0x000F000Fu;
                    if (Q5) {   // bit 4 of every byte <- the element's high bit (pack_hb1 layout); bits 5..7 keep garbage that the 0x001F001F masks drop
                        lo = (lo & ~0x10101010u) | ((d == 0 ? P << 4 : d == 1 ? P << 3 : d == 2 ? P << 2 : P << 1) & 0x10101010u);
                        hi = (hi & ~0x10101010u) | ((d == 0 ? P : d == 1 ? P >> 1 : d == 2 ? P >> 2 : P >> 3) & 0x10101010u);
                        mask = 0x001F001Fu;
                    }
                    wb[4 * uu + d][0] = (int)pkfma((lo & mask) | 0x64006400u, slo2, clo2);
                    wb[4 * uu + d][1] = (int)pkfma(((lo >> 8) & mask) | 0x64006400u, slo2, clo2);
                    wb[4 * uu + d][2] = (int)pkfma((hi & mask) | 0x64006400u, shi2, chi2);
                    wb[4 * uu + d][3] = (int)pkfma(((hi >> 8) & mask) | 0x64006400u, shi2, chi2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < TT; tt++) {
                if (tt < my_tiles) {
                    const unsigned char *sq = st + tt * 8192;
                    v16f c; for (int r = 0; r < 16; r++) c[r] = 0.0f;
#pragma unroll
                    for (int s8 = 0; s8 < 8; s8++) {
                        const h8_t af = *reinterpret_cast<const h8_t *>(sq + a_addr[s8]);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(h8_t, wb[s8]), c, 0, 0, 0);
                    }
                    // the half's 128 products, scaled and added at once (a super-block's two halves as two fmas: the half sums are exact integers < 2^24, so the only
                    // difference from one fma on their sum is one fp32 rounding); h = 1 also adds the super-block's min term
                    v16f cm;
                    if (h == 1) {
                        const h8_t ab = *reinterpret_cast<const h8_t *>(sd + bs_addr + tt * 512);
                        v16f z; for (int r = 0; r < 16; r++) z[r] = 0.0f;
                        cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab, __builtin_bit_cast(h8_t, bmin), z, 0, 0, 0);
                    }
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        const v4f da = *reinterpret_cast<const v4f *>(sd + dk_addr + tt * 128 + q4 * 32);   // tokens 8 q4 + 4 hh + 0..3 = accumulator registers 4 q4 .. 4 q4 + 3
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int r = 4 * q4 + e;
                            acc[tt][r] = fmaf(dw * da[e], c[r], acc[tt][r]);
                            if (h == 1) acc[tt][r] = fmaf(ndmin * da[e], cm[r], acc[tt][r]);
                        }
                    }
                    asm volatile("" : "+v"(acc[tt]));   // keeps this tile's scale arithmetic here (see k_mmq2_q40: LLVM otherwise sinks every tile's fmas behind all the MFMAs and spills)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float accf[TT][16];
#pragma unroll
    for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) accf[tt][r] = acc[tt][r];
    mmq2_store<TT>(accf, a, m, r0 + l31, W.rows, t0, my_tiles, hh);
}

// y[t][r] = (residual[t][r] +) sum_z slab_z[t][r], z in fixed order (deterministic); rows x cols floats per slab
__global__ __launch_bounds__(256) void k_mmq2_reduce(const float *__restrict__ slabs, int n_slabs, long long slab_stride, const float *__restrict__ residual, float *__restrict__ y, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = reinterpret_cast<const float4 *>(slabs)[i];
    for (int z = 1; z < n_slabs; z++) { const float4 t = reinterpret_cast<const float4 *>(slabs + (size_t)z * slab_stride)[i]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    if (residual) { const float4 t = reinterpret_cast<const float4 *>(residual)[i]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    reinterpret_cast<float4 *>(y)[i] = s;
}
// the same for the 1..3 matrices of a set in ONE launch (blockIdx.y = matrix; its slabs start m * n_slabs * slab_stride floats into the workspace): a prompt pass has
// three sets per layer, and a 5 us launch per matrix was 11 % of the 142-row image turn
struct ReduceSet { float *y[3]; const float *res[3]; };
__global__ __launch_bounds__(256) void k_mmq2_reduce_set(const float *__restrict__ slabs, int n_slabs, long long slab_stride, const ReduceSet rs, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int m = blockIdx.y;
    const float *sl = slabs + (size_t)m * n_slabs * slab_stride;
    const float *residual = m == 0 ? rs.res[0] : (m == 1 ? rs.res[1] : rs.res[2]);
    float *y = m == 0 ? rs.y[0] : (m == 1 ? rs.y[1] : rs.y[2]);
    float4 s = reinterpret_cast<const float4 *>(sl)[i];
    for (int z = 1; z < n_slabs; z++) { const float4 t = reinterpret_cast<const float4 *>(sl + (size_t)z * slab_stride)[i]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    if (residual) { const float4 t = reinterpret_cast<const float4 *>(residual)[i]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    reinterpret_cast<float4 *>(y)[i] = s;
}

// y = (residual +) sum_z slab_z in fixed order, n floats (a multiple of 4) per slab: also the combine step of the split-K F16 GEMM (vision_kernels.hip)
void launch_slab_reduce(const float *slabs, int n_slabs, long long slab_stride, const float *residual, float *y, size_t n, hipStream_t s) {
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(k_mmq2_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, slabs, n_slabs, slab_stride, residual, y, n4);
}

bool mmq2_supported(int type, int rows, int cols) { return (type == GT_Q4_K || type == GT_Q5_K || type == GT_Q6_K || type == GT_Q4_0) && cols % 256 == 0 && rows >= 32; }

static int g_mmq2_cus = 256;
void set_mmq2_cus(int cus) { if (cus > 0) g_mmq2_cus = cus; }
// experiment knobs (MINIGPT4_MMQ2_TT / _FILL / _KS, read ONCE by Engine::init; the test hooks set the K split directly): 0 = the launcher's own choice
static int g_mmq2_tt = 0, g_mmq2_fill = 0, g_mmq2_ks = 0;
void set_mmq2_tuning(int tt, int fill_pct, int ks) { if (tt >= 0) g_mmq2_tt = tt; if (fill_pct >= 0) g_mmq2_fill = fill_pct ? std::max(50, fill_pct) : 0; if (ks >= 0) g_mmq2_ks = ks; }

template <typename KernelT>
static void mmq2_launch_kernel(KernelT kernel, bool &attr_done, dim3 grid, int threads, size_t lds, hipStream_t s, const Mmq2Args &a, const ActQ &A) {
    if (!attr_done) { HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); attr_done = true; }
    hipLaunchKernelGGL(kernel, grid, dim3((unsigned)threads), lds, s, a, A);
}
static int g_mmqh = 0;   // measured in round 5 and NOT adopted (profiles/r05_prefill_fp16_scaled_operands.md): 10-17 % slower per launch than the int8 kernels -- both forms are bound by the CU's load path, and the fp16 activation image doubles the staged bytes
void set_mmqh(int v) { g_mmqh = v != 0; }
int mmqh_enabled() { return g_mmqh; }
template <int TT>
static void mmq2_launch_tt(int type, dim3 grid, size_t lds, hipStream_t s, const Mmq2Args &a, const ActQ &A, bool fp16_form = false) {
    static bool attr[6] = {false, false, false, false, false, false};
    if constexpr (TT <= 3) {
        if (fp16_form && type == GT_Q4_K) { mmq2_launch_kernel(&k_mmqh_q45k<false, TT>, attr[4], grid, 256, MmqhStage<TT>::BYTES + 16384, s, a, A); return; }
        if (fp16_form && type == GT_Q5_K) { mmq2_launch_kernel(&k_mmqh_q45k<true, TT>, attr[5], grid, 256, MmqhStage<TT>::BYTES + 16384, s, a, A); return; }
    }
    if constexpr (TT <= 3) {
        if (type == GT_Q4_0) { mmq2_launch_kernel(&k_mmq2_q40<TT>, attr[3], grid, 256, lds, s, a, A); return; }
        if (type == GT_Q4_K) { mmq2_launch_kernel(&k_mmq2_q45k<false, TT>, attr[0], grid, 256, lds, s, a, A); return; }
        if (type == GT_Q5_K) { mmq2_launch_kernel(&k_mmq2_q45k<true, TT>, attr[1], grid, 256, lds, s, a, A); return; }
    }
    if constexpr (TT <= 2) {
        if (type == GT_Q6_K) { mmq2_launch_kernel(&k_mmq2_q6k<TT>, attr[2], grid, 256, lds, s, a, A); return; }
    }
    throw HipError{hipErrorInvalidValue, "mmq2: token tiles per chunk outside the kernel's register budget", __FILE__, __LINE__};
}
// (Single-wave workgroups -- 32 weight rows x the whole K range, 160 .. 864 workgroups per launch without a K split -- were built for one-tile launches (batched decode of
// 5..32 conversations), bit-identical, and measured SLOWER than the 4-wave form with its K split: 13B layer at 8 rows qkv 25.3 vs 18.9 us, w2 25.9 vs 21.0, w2 Q6_K 38.2 vs
// 28.3 (profiles/r02x_prefill_single_wave_workgroups_microbench.log).  The WPB template parameter stays at 4.)

// 1..3 same-type, same-shape matrices against the N prepared activation rows in one launch.  y[m][t * ldy + r] (+ residual[m][..]).  false -> shape outside the
// kernel's range (nothing launched).  force_ks == 1: no K split -- every output is then the CPU oracle's value bit for bit (exact integer block sums, the per-block fp32 updates
// in block order: tests/test_gpu_paritymode.py), which is how parity mode runs its prompt rows.
void launch_slab_flush(const SlabSrc &src, hipStream_t s) {
    if (src.ks <= 1) return;
    if (src.mixed) {   // matrices of two launches: one combine per matrix that was split
        for (int i = 0; i < src.n; i++) if (src.mks[i] > 1) launch_slab_reduce(src.mbase[i], src.mks[i], src.stride, src.res[i], src.y[i], (size_t)src.stride, s);
        return;
    }
    const size_t n4 = (size_t)src.stride / 4;
    ReduceSet rs{};
    for (int i = 0; i < src.n; i++) { rs.y[i] = src.y[i]; rs.res[i] = src.res[i]; }
    hipLaunchKernelGGL(k_mmq2_reduce_set, dim3((unsigned)((n4 + 255) / 256), (unsigned)src.n), dim3(256), 0, s, src.ws, src.ks, src.stride, rs, n4);
}
bool launch_mmq2_set(const QWeight *const *W, float *const *y, const float *const *residual, int n, const ActQ &A, int N, int ldy, hipStream_t s, SlabSrc *defer, int force_ks) {
    if (defer) *defer = SlabSrc{};
    if (n < 1 || n > 3 || N < 1 || (W[0]->type == GT_Q4_0 ? !(A.q80 && A.d0) : !A.bsq)) return false;
    for (int i = 0; i < n; i++) if (!mmq2_supported(W[i]->type, W[i]->rows, W[i]->cols) || W[i]->type != W[0]->type || W[i]->rows != W[0]->rows || W[i]->cols != W[0]->cols) return false;
    Mmq2Args a{};
    for (int i = 0; i < n; i++) { a.w[i] = *W[i]; a.y[i] = y[i]; a.res[i] = residual ? residual[i] : nullptr; }
    a.n_tiles = (N + 31) / 32;
    a.n_mat = n; a.groups_each = (W[0]->rows + 127) / 128; a.N = N; a.ldy = ldy;
    int max_tt = W[0]->type == GT_Q6_K ? 2 : 3;          // token tiles per chunk the 256-register budget (two waves per SIMD) admits: Q6_K keeps four half-masked operand sets per pair
    if (g_mmq2_tt > 0) max_tt = std::min(max_tt, g_mmq2_tt);   // experiments
    const int n_chunks = (a.n_tiles + max_tt - 1) / max_tt;
    a.tiles_per_chunk = (a.n_tiles + n_chunks - 1) / n_chunks;
    const int NSB = W[0]->cols / 256;
    // K split: enough workgroups for 1.5 per CU, at least 4 super-blocks per slice, slabs must fit the workspace; the slices are combined in fixed order by k_mmq2_reduce
    const int wgs = n * a.groups_each * n_chunks;
    int ks = 1;
    const size_t out_floats = (size_t)N * ldy;
    // workgroups per CU (x 100) below which another K slice is added.  Measured on the 142-row image turn (profiles/r02r_prefill_ksplit_fill_sweep.log): 13B (5120-wide
    // matrices) 13.1 ms at 200, 11.7 at 150, 12.6 at 125 / 100 -- w1|w3's 432 workgroups are better left unsplit; 7B (4096 x 4096 wq / wo ...) 7.7 at 200, 8.2 at 150,
    // 8.0 at 125: the narrower model wants the deeper split.  MINIGPT4_MMQ2_FILL overrides.
    const int fill_env = g_mmq2_fill;
    // by the model's width (the smaller matrix dimension): rule = 0 / 1 lines of the sweep log are two other rules that lost (by weights per matrix; by workgroup count)
    const int fill_pct = fill_env ? fill_env : (std::min(W[0]->rows, W[0]->cols) >= 5120 ? 150 : 200);
    while (wgs * ks * 100 < fill_pct * g_mmq2_cus && NSB / (ks + 1) >= 4 && A.ws && (size_t)(ks + 1) * out_floats * n <= A.ws_floats) ks++;
    if (g_mmq2_ks > 0) ks = std::max(1, std::min(g_mmq2_ks, std::min(NSB, A.ws ? (int)(A.ws_floats / std::max<size_t>(1, out_floats * n)) : 1)));
    if (force_ks == 1) ks = 1;   // parity mode: without a K split the kernels add a row's per-block terms block after block -- the CPU oracle's order, bit for bit
    a.sb_per_split = (NSB + ks - 1) / ks;
    ks = (NSB + a.sb_per_split - 1) / a.sb_per_split;
    if (ks > 1) {
        if ((size_t)ldy % 4 || out_floats % 4) return false;
        a.slab_stride = (long long)out_floats;
        for (int i = 0; i < n; i++) { a.y[i] = A.ws + (size_t)i * ks * out_floats; a.res[i] = nullptr; }
    }
    const dim3 grid((unsigned)(n * a.groups_each), (unsigned)n_chunks, (unsigned)ks);
    const int type = W[0]->type;
    const bool q40 = type == GT_Q4_0;
    // fast mode, Q4_K / Q5_K: the fp16-MFMA form (k_mmqh_q45k) whenever the activation rows carry their fp16 image; force_ks == 1 (parity mode: the oracle's fp32 order) keeps the int8 kernels
    const bool h16 = g_mmqh && force_ks != 1 && (type == GT_Q4_K || type == GT_Q5_K) && A.q16 && A.bs16;
    switch (a.tiles_per_chunk) {
    case 1: mmq2_launch_tt<1>(type, grid, 2 * (q40 ? Mmq2Stage80<1>::BYTES : Mmq2Stage<1>::BYTES) + 16384, s, a, A, h16); break;
    case 2: mmq2_launch_tt<2>(type, grid, 2 * (q40 ? Mmq2Stage80<2>::BYTES : Mmq2Stage<2>::BYTES) + 16384, s, a, A, h16); break;
    case 3: mmq2_launch_tt<3>(type, grid, 2 * (q40 ? Mmq2Stage80<3>::BYTES : Mmq2Stage<3>::BYTES) + 16384, s, a, A, h16); break;
    default: throw HipError{hipErrorInvalidValue, "mmq2: bad chunking", __FILE__, __LINE__};
    }
    if (ks > 1) {
        SlabSrc src; src.ws = A.ws; src.ks = ks; src.stride = (long long)out_floats; src.n = n;
        for (int i = 0; i < n; i++) { src.y[i] = y[i]; src.res[i] = residual ? residual[i] : nullptr; src.mbase[i] = A.ws + (size_t)i * ks * out_floats; src.mks[i] = ks; }
        if (defer) *defer = src; else launch_slab_flush(src, s);
    }
    return true;
}

}  // namespace mg4
