// gfx950 kernels for the image path (EVA ViT-g/14 -> ln_vision -> Q-Former -> llama_proj; reference minigpt4.cpp:2094-2363).
//   * f16 GEMM on MFMA (v_mfma_f32_32x32x16_f16): same arithmetic as ggml's f16 mul_mat -- activations rounded to fp16,
//     exact fp16 x fp16 products, fp32 accumulation -- with bias / fp16-table GELU / residual fused into the epilogue.
//   * LayerNorm with ggml_norm's double-precision statistics, eps 1e-5.
//   * fp32 attention (ViT 16 x 88, BERT 12 x 64) staged through LDS; exact-sum softmax through the fp16 exp table.
#include "kernels.hpp"
#include "devutil.hpp"

#include <algorithm>

namespace mg4 {

static int g_gemm_arm = 0, g_gemm_sk_arm = 0;   // experiments: force one tile shape for every small-M GEMM / split-K GEMM (MINIGPT4_GEMM_ARM / _SK_ARM, read once by Engine::init); 0 = choose, -2 = the 64x64 launch always, -3 = round 2's kernels everywhere (64x64 tiles, 128x128 large-M kernel)
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef unsigned v4u_g __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short f2h_bits_v(float f) { return __half_as_ushort(f2h_rn(f)); }
__device__ __forceinline__ float tab_v(const __half *t, float x) { return __half2float(t[f2h_bits_v(x)]); }
// SiLU likewise (the F16 language model's w1 | w3 pair epilogue): table[x] = fp16(x / (1 + expf(-x))) on the fp16-rounded argument (qtraits.hpp silu_h)
__device__ __forceinline__ float silu_v(const __half *t, float x) {
    if (t) return tab_v(t, x);
    const float xh = __half2float(f2h_rn(x));
    return __half2float(f2h_rn(xh / (1.0f + __expf(-xh))));
}
// GELU through ggml's fp16 table, or (t == null: fast mode, round 6) the table's VALUE computed: table[x] = fp16(0.5 x (1 + tanhf(sqrt(2 / pi) x (1 + 0.044715 x^2)))) on the fp16-rounded
// argument, with tanh(u) = 1 - 2 / (exp(2 u) + 1) on the device's exp -- the host table's entry except within ~1e-7 of an fp16 rounding boundary (as exp_h / silu_h, qtraits.hpp).
// A GEMM tile's epilogue gathers 16-64 table entries per lane from a 128 KB table: at four images that is 7 us of the 39 us fc1 launch.
__device__ __forceinline__ float gelu_v(const __half *t, float x) {
    if (t) return tab_v(t, x);
    const float xh = __half2float(f2h_rn(x));
    const float u = 0.79788456080286535587989211986876f * xh * (1.0f + 0.044715f * xh * xh);
    const float th = 1.0f - 2.0f / (__expf(2.0f * u) + 1.0f);
    return __half2float(f2h_rn(0.5f * xh * (1.0f + th)));
}

// In-kernel timeline of the image path's kernels (diagnostic builds only, as the mat-vec's: make EXTRA=-DMG4_TIMELINE OUT=../libminigpt4_tl.so OBJ=build_tl).  Thread 0 of
// every workgroup stamps the 100 MHz constant clock into 32 slots; the last launch wins; read with minigpt4_amd_timeline_vision (tools/timeline_gemm.py).
#ifdef MG4_TIMELINE
__device__ unsigned long long g_tlv[1024 * 32];
#define MG4_TLV(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.z == 0) g_tlv[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MG4_TLA(i) do { const unsigned lb_ = blockIdx.x + gridDim.x * blockIdx.y; if (threadIdx.x == 0 && lb_ < 1024 && blockIdx.z == 0) g_tlv[lb_ * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MG4_TLV(i) do {} while (0)
#define MG4_TLA(i) do {} while (0)
#endif
int read_vision_timeline(unsigned long long *out, int max_workgroups) {
#ifdef MG4_TIMELINE
    const int n = std::min(max_workgroups, 1024);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tlv), (size_t)n * 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return n;
#else
    (void)out; (void)max_workgroups; return 0;
#endif
}

// Head-major output (round 6): the ViT's qkv projection stores its [rows][3 x heads x hd] result as [3][head][rows][hd] -- every head's Q, K and V rows contiguous (rows x hd floats)
// instead of hd-float pieces 3 x D floats apart -- because that is what the attention kernel wants to stream: 16-key tiles become one contiguous 5.6 KB block (45 cache lines instead
// of 64 half-used ones): k_attn_vit 16.4 -> 14.6 us at one image, 39.3 -> 35.3 at four (profiles/r06_attn_query_tiles.log, run 26).  Same values, same arithmetic: only the address
// of an output element changes (column c of row r -> ((c / D) * heads + (c % D) / hd) * rows * hd + r * hd + c % hd).  rows == 0: the plain [row][ldo] layout.
struct HeadMajor { int rows, D, hd; };
static thread_local HeadMajor t_hm{0, 0, 0};       // set around ONE dispatcher call by launch_gemm_f16_head_major (per thread: two contexts on two threads never see each other's)
__device__ __forceinline__ void gemm_out_map(const HeadMajor &hm, int col, int ldo, int &row_mul, unsigned &col_off) {
    if (hm.rows > 0) {
        const int which = col / hm.D, c = col - which * hm.D, head = c / hm.hd, d = c - head * hm.hd;
        row_mul = hm.hd; col_off = (unsigned)((which * (hm.D / hm.hd) + head) * hm.rows * hm.hd + d);
    } else { row_mul = ldo; col_off = (unsigned)col; }
}
// Workgroup -> (tile, K slice).  Blocks are dealt round-robin to the 8 XCDs (XCD = blockIdx.x & 7), each with its own 4 MB L2.  The work list of a launch is
// [K slice][column tile][row tile] (row tile fastest) and XCD x owns the CONTIGUOUS range [x * per, (x + 1) * per) of it, per = ceil(slices * tiles / 8): the row tiles
// that share a weight tile sit on one XCD, every XCD gets the same number of workgroups +- 1, and -- round 6 -- an XCD walks ONE K slice (or two neighbouring ones), not
// all of them.  Rounds 3-5 put the slices in grid.z with the tile list dealt to the XCDs per slice, so every XCD walked every K slice and therefore fetched ALL of A:
// for the ViT's fc2 (K = 6144, 4 slices) 8 x 3.2 MB at one image and 8 x 12.6 MB (three times an L2) at four -- FETCH_SIZE 47 MB for 20 MB of operands at B = 1,
// 180 MB for 30 MB at B = 4 (profiles/r06_encode_kernel_table_b1.txt / _b4.txt).  xs = 0 keeps that form (A/B: MINIGPT4_SPLITK_XCD=0).  Which workgroup multiplies a
// (tile, slice) never changes what is computed (slab z = columns [z * k_per_slice, ...)), so results are bit-identical under both mappings.
static int g_splitk_xcd = 1;                       // 0: split-K slices in grid.z as in rounds 3-5 (read once by Engine::init)
void set_gemm_splitk_xcd(int on) { g_splitk_xcd = on != 0; }
__device__ __forceinline__ bool gemm_tile_of_block(int tile_n, int xs, int &tile_id, int &z) {
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    if (xs > 1) {
        const int total = xs * tile_n, per = (total + 7) >> 3, w = xcd * per + slot;
        z = w / tile_n; tile_id = w - z * tile_n;
        return slot < per && w < total;
    }
    z = (int)blockIdx.z;
    const int chunk = (tile_n + 7) >> 3;
    tile_id = xcd * chunk + slot;
    return slot < chunk && tile_id < tile_n;
}
static inline bool splitk_on_xcds(int slices) { return g_splitk_xcd && slices > 1; }
static inline unsigned gemm_grid_x(int tile_n, int slices, bool on_xcds) { return (unsigned)(((on_xcds ? slices : 1) * tile_n + 7) / 8 * 8); }
// =====================================================================================================================
// C[M][N] = A[M][K] . W[N][K]^T  (+bias, GELU, +residual).  64x64 tile per 256-thread workgroup, 4 waves of 32x32,
// BK = 32 staged through LDS (80-byte padded rows: conflict-free ds_read_b128), register prefetch of the next tile.
// =====================================================================================================================
// Deep K tiles (BK = 64/128): with M = 257 a launch has ~1 workgroup per CU, so the exposed global-load latency per k-iteration dominates; fewer, fatter
// iterations amortise it.  A wave owns TM x TN MFMA tiles (32x32 each): with 1 x 1 every MFMA costs two 1 KB fragment reads and the kernel is bound by the LDS
// (round-3 timeline: 0.5 us per 64x64x128 step against 0.21 us of MFMA); 2 x 1 / 1 x 2 share a fragment between two MFMAs.
// Round-3 findings that shaped this form (tools/timeline_gemm.py, tools/isa_waits.py):
//   * operands by raw buffer loads whose out-of-range chunks come back as zeros -- zeroing the loaded registers made hipcc drain the pipeline (vmcnt(0)) every sextuple;
//   * scheduling barriers keep the prologue's three stages in issue order (one s_waitcnt serves the prologue and the back edge);
//   * the epilogue stores through buffer descriptors too (rows / columns past the edge and absent outputs are dropped by the bounds check): the branchy form
//     waited for vmcnt(0) in front of every store, i.e. 16 serial store round trips = 2 us per workgroup.
template <int BK, int GB_M, int BN, int TM, int TN, bool GELU, bool RES>
__global__ __launch_bounds__(GB_M / (32 * TM) * (BN / (32 * TN)) * 64) void k_gemm_f16(const __half *__restrict__ A, int lda, const __half *__restrict__ W, int ldw, int M, int N, int K,
                                                  const float *__restrict__ bias, const float *residual, const Tables tb,
                                                  float *out, __half *__restrict__ out_h, int ldo, int k_per_slice, size_t slab_stride, int xs, const HeadMajor hm) {
    constexpr int LD = BK + 8;                 // +16 bytes per row: conflict-free ds_read_b128 for BK = 32/64/128/256
    constexpr int CPR = BK / 8;                // 16-byte chunks per row
    constexpr int WNN = BN / (32 * TN), NT = GB_M / (32 * TM) * WNN * 64;
    constexpr int NCA = GB_M * CPR / NT, NCW = BN * CPR / NT;   // 16-byte chunks per thread: A tile, W tile
    static_assert(NCA * NT == GB_M * CPR && NCW * NT == BN * CPR, "tile chunks must divide over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
    __half *As = reinterpret_cast<__half *>(smem_g), *Ws = As + GB_M * LD, *As1 = Ws + BN * LD, *Ws1 = As1 + GB_M * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order (blocks are dealt round-robin to the 8 XCDs, each with its own L2): XCD x owns the tiles [x * chunk, (x + 1) * chunk) of the column-major
    // tile list, so the row tiles that share a weight tile sit on ONE XCD (the weight tile is fetched from HBM once instead of once per XCD: 4x re-fetch measured
    // without) AND every XCD gets the same number of tiles +- 1.  (Until round 3 XCD x owned the column tiles x, x + 8, ...: 33 column tiles x 7 row tiles put 35
    // workgroups on XCD 0's 32 CUs and 28 on the others -- 35 us instead of ~20 for the 3-image qkv launch.)
    const int ntx = (N + BN - 1) / BN, rt = (M + GB_M - 1) / GB_M;
    int tile_id, kz;
    const bool owns = gemm_tile_of_block(ntx * rt, xs, tile_id, kz);                 // XCD-aware, balanced (above)
    const int bx = tile_id / rt, by = tile_id - bx * rt;
    MG4_TLV(0);
    if (!owns) return;
    const int m0 = by * GB_M, n0 = bx * BN;
    const int wm = wave / WNN, wn = wave % WNN;
    if (k_per_slice > 0) {   // split-K: slice z multiplies columns [z * k_per_slice, ...) and writes its raw fp32 partial sums into slab z
        const int k0 = kz * k_per_slice;
        A += k0; W += k0; K = min(K - k0, k_per_slice);
        out += (size_t)kz * slab_stride;
    }
    const int nk = (K + BK - 1) / BK;
    const __amdgpu_buffer_rsrc_t ab = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(A), 0, (int)(((size_t)(M - 1) * lda + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wb = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(W), 0, (int)(((size_t)(N - 1) * ldw + K) * 2), 0x00020000);
    int aoff[NCA], woff[NCW]; int lofa[NCA], kofa[NCA], lofw[NCW], kofw[NCW];
#pragma unroll
    for (int i = 0; i < NCA; i++) { const int c = tid + NT * i, row = c / CPR; kofa[i] = (c % CPR) * 8; lofa[i] = row * LD + kofa[i]; aoff[i] = (min(m0 + row, M - 1) * lda + kofa[i]) * 2; }
#pragma unroll
    for (int i = 0; i < NCW; i++) { const int c = tid + NT * i, row = c / CPR; kofw[i] = (c % CPR) * 8; lofw[i] = row * LD + kofw[i]; woff[i] = (min(n0 + row, N - 1) * ldw + kofw[i]) * 2; }
    float16_t acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.0f;
    // Three statically named register stages: while tile k is multiplied out of LDS, the loads of tiles k+1 and k+2 are in flight and tile
    // k+3 is issued.  No lambdas / structs / register copies around the stages (each of those made hipcc spill to scratch or wait on the
    // in-flight loads); all loads unconditional.  K is a multiple of 8, so a 16-byte chunk is entirely in or out.
    v4u_g ra0[NCA], rw0[NCW], ra1[NCA], rw1[NCW], ra2[NCA], rw2[NCW];
#define MG4_GLOAD(kt, RA, RW)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NCA; i++)                                                             \
        RA[i] = __builtin_amdgcn_raw_buffer_load_b128(ab, (kt) * BK + kofa[i] < K ? aoff[i] + (kt) * (BK * 2) : (int)0x80000000, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < NCW; i++)                                                             \
        RW[i] = __builtin_amdgcn_raw_buffer_load_b128(wb, (kt) * BK + kofw[i] < K ? woff[i] + (kt) * (BK * 2) : (int)0x80000000, 0, 0);
#define MG4_STEP(kt, RA, RW, AS, WS)                                                                            \
    {   _Pragma("unroll") for (int i = 0; i < NCA; i++) *reinterpret_cast<v4u_g *>(&AS[lofa[i]]) = RA[i];       \
        _Pragma("unroll") for (int i = 0; i < NCW; i++) *reinterpret_cast<v4u_g *>(&WS[lofw[i]]) = RW[i];       \
        __syncthreads();                                                                                        \
        if ((kt) < 26) MG4_TLV(2 + (kt));                                                                       \
        MG4_GLOAD((kt) + 3, RA, RW)      /* past the end: zeros */                                              \
        _Pragma("unroll") for (int ks = 0; ks < BK / 16; ks++) {                                                \
            half8_t af[TM], bf[TN];                                                                             \
            _Pragma("unroll") for (int a = 0; a < TM; a++) af[a] = *reinterpret_cast<const half8_t *>(&AS[((wm * TM + a) * 32 + (lane & 31)) * LD + ks * 16 + (lane >> 5) * 8]); \
            _Pragma("unroll") for (int b = 0; b < TN; b++) bf[b] = *reinterpret_cast<const half8_t *>(&WS[((wn * TN + b) * 32 + (lane & 31)) * LD + ks * 16 + (lane >> 5) * 8]); \
            _Pragma("unroll") for (int a = 0; a < TM; a++)                                                      \
                _Pragma("unroll") for (int b = 0; b < TN; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0); } }
    // Two LDS buffers, ONE barrier per k tile: tile k+1 is written into the other buffer while slower waves still multiply tile k; a wave can only
    // reach the barrier of step k+1 after its own reads of step k, so the buffer written in step k+2 is free.  (3 register stages x 2 LDS buffers
    // -> the loop body is unrolled 6 times with statically named stages and buffers.)
    MG4_GLOAD(0, ra0, rw0)
    __builtin_amdgcn_sched_barrier(0);
    MG4_GLOAD(1, ra1, rw1)
    __builtin_amdgcn_sched_barrier(0);
    MG4_GLOAD(2, ra2, rw2)
    __builtin_amdgcn_sched_barrier(0);
    MG4_TLV(1);
    // epilogue operands requested now, behind the three stages: bias (and, for one tile per wave, the residual) would otherwise be a memory round trip after the loop
    const int lcol = lane & 31, hh = lane >> 5;
    float bv[TN];
#pragma unroll
    for (int b = 0; b < TN; b++) bv[b] = bias ? bias[min(n0 + (wn * TN + b) * 32 + lcol, N - 1)] : 0.0f;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(RES ? residual : bias), 0, RES ? (int)(((size_t)(M - 1) * ldo + N) * 4) : 0, 0x00020000);
    constexpr bool RES_EARLY = RES && TM * TN == 1;
    float rr0[16];
    if (RES_EARLY) {
        const int col = n0 + wn * 32 + lcol;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            rr0[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, row < M && col < N ? (row * ldo + col) * 4 : (int)0x80000000, 0, 0));
        }
    }
    for (int kt = 0; kt < nk; kt += 6) {      // whole sextuples: steps past nk multiply zero tiles (accumulators unchanged).  (Uniform exits between the steps were tried: the
        MG4_STEP(kt, ra0, rw0, As, Ws)        //  merged control flow brought vmcnt(0) waits back into the body.)
        MG4_STEP(kt + 1, ra1, rw1, As1, Ws1)
        MG4_STEP(kt + 2, ra2, rw2, As, Ws)
        MG4_STEP(kt + 3, ra0, rw0, As1, Ws1)
        MG4_STEP(kt + 4, ra1, rw1, As, Ws)
        MG4_STEP(kt + 5, ra2, rw2, As1, Ws1)
    }
#undef MG4_STEP
#undef MG4_GLOAD
    MG4_TLV(29);
    // epilogue, branch-free: every store goes through a buffer descriptor (an absent output has zero records), rows >= M / columns >= N get an out-of-range offset
    const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? (int)(((size_t)(M - 1) * ldo + N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t hb = __builtin_amdgcn_make_buffer_rsrc(out_h, 0, out_h ? (int)(((size_t)(M - 1) * ldo + N) * 2) : 0, 0x00020000);
#pragma unroll
    for (int b = 0; b < TN; b++) {
        const int col = n0 + (wn * TN + b) * 32 + lcol;
        int rmul; unsigned coff; gemm_out_map(hm, min(col, N - 1), ldo, rmul, coff);
#pragma unroll
        for (int a = 0; a < TM; a++) {
            float v[16]; unsigned o[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                o[r] = row < M && col < N ? (unsigned)(row * rmul) + coff : 0x20000000u;   // x 4 = 0x80000000, x 2 = 0x40000000: both past any tensor of this path
                v[r] = bias ? bv[b] + acc[a][b][r] : acc[a][b][r];
            }
            if (GELU) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = gelu_v(tb.gelu, v[r]);
            }
            if (RES) {
                float rr[16];
#pragma unroll
                for (int r = 0; r < 16; r++) rr[r] = RES_EARLY ? rr0[r] : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (int)(o[r] * 4u), 0, 0));
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = rr[r] + v[r];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ob, (int)(o[r] * 4u), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(__half_as_ushort(f2h_rn(v[r])), hb, (int)(o[r] * 2u), 0, 0);
            }
        }
    }
#ifdef MG4_TIMELINE
    MG4_TLV(30);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG4_TLV(31);
#endif
}
template <int BK, int BM, int BN, int TM = 1, int TN = 1>
static void launch_gemm_t(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                          float *out, __half *out_h, int ldo, hipStream_t s, int slices = 1, size_t slab_stride = 0) {
    const int k_per_slice = slices > 1 ? ((K + BK - 1) / BK + slices - 1) / slices * BK : 0;
    const int ntx = (N + BN - 1) / BN, rt = (M + BM - 1) / BM;
    const bool sx = splitk_on_xcds(slices);
    const int xs = sx ? slices : 0;
    dim3 grid(gemm_grid_x(ntx * rt, slices, sx), 1, sx ? 1u : (unsigned)slices), block(BM / (32 * TM) * (BN / (32 * TN)) * 64);
    const size_t lds = (size_t)2 * (BM + BN) * (BK + 8) * 2;
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_gemm_f16<BK, BM, BN, TM, TN, true, true>));
        HIP_IGNORE(lds_optin_max(&k_gemm_f16<BK, BM, BN, TM, TN, true, false>));
        HIP_IGNORE(lds_optin_max(&k_gemm_f16<BK, BM, BN, TM, TN, false, true>));
        HIP_IGNORE(lds_optin_max(&k_gemm_f16<BK, BM, BN, TM, TN, false, false>)); attr = true; }
    if (gelu && residual) hipLaunchKernelGGL((k_gemm_f16<BK, BM, BN, TM, TN, true, true>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, k_per_slice, slab_stride, xs, t_hm);
    else if (gelu) hipLaunchKernelGGL((k_gemm_f16<BK, BM, BN, TM, TN, true, false>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, k_per_slice, slab_stride, xs, t_hm);
    else if (residual) hipLaunchKernelGGL((k_gemm_f16<BK, BM, BN, TM, TN, false, true>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, k_per_slice, slab_stride, xs, t_hm);
    else hipLaunchKernelGGL((k_gemm_f16<BK, BM, BN, TM, TN, false, false>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, k_per_slice, slab_stride, xs, t_hm);
}
// =====================================================================================================================
// LDS-DMA ring form of the small-M GEMM (round 3).  The timeline of k_gemm_f16 put the k loop at 0.5 us per 64x64x128 step against 0.21 us of MFMA: the
// register-staged tile costs 32 ds_write_b128 per step (13 cycles each through the VGPR -> LDS path, MI355X_MICROARCH.md LDS table) on the same waves that issue
// the MFMAs.  Here the tiles travel global -> LDS by global_load_lds (no staging registers, no ds_write), S buffers deep, with counted waits:
//     [s_waitcnt vmcnt((S - 2) * PW): own pieces of tile k landed]  [raw s_barrier: everybody's pieces landed, everybody left buffer k - 1]
//     [request tile k + S - 1 into buffer k - 1]  [fragment reads + MFMAs on tile k]
// (raw barrier + explicit counts: __syncthreads() would drain the DMA queue -- cdna_hip_programming.md "Pipelining across barriers").  A stage holds KT
// sub-tiles of 64 k each: rows of 128 bytes, the 16-byte chunks XOR-swizzled by ((row >> 1) & 7) on the SOURCE side (the DMA destination is lane-linear), as in
// k_gemm_f16_big.  Same arithmetic as k_gemm_f16 (exact fp16 products, fp32 accumulation in k order per 16-wide MFMA step), same epilogue.
// =====================================================================================================================
typedef __attribute__((address_space(3))) void *g_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *g_glb_ptr_t;
// several equally spaced, equally shaped weight matrices in one launch (column tile -> matrix), and / or a K range split over grid.z with one fp32 slab per slice
struct GemmSet { int n_per_mat; long long w_mat_stride, out_mat_stride; int k_per_slice; long long slab_stride; int xs; HeadMajor hm; };   // all zero: one matrix, whole K; xs: gemm_tile_of_block
// PAIR (TN = 2, no GELU / residual): the feed-forward pair of an F16 language model in one tile -- the BN tile columns are 32-column blocks taken alternately from W
// (w1) and W + gs.w_mat_stride (w3), rows n0 .. of both, so a wave's two column tiles hold (w1 x)[r][c] and (w3 x)[r][c] for the SAME (r, c) and the epilogue stores
// fp16(silu_table(w1 x) * (w3 x)) -- the row w2 multiplies -- instead of the two fp32 products (56 MB written and read back per 512-row layer, and a launch).
template <int BM, int BN, int TM, int TN, int KT, int S, bool GELU, bool RES, bool PAIR = false>
__global__ __launch_bounds__(BM / (32 * TM) * (BN / (32 * TN)) * 64) void k_gemm_dma(const __half *__restrict__ A, int lda, const __half *__restrict__ W, int ldw, int M, int N, int K,
                                                  const float *__restrict__ bias, const float *residual, const Tables tb,
                                                  float *out, __half *__restrict__ out_h, int ldo, const GemmSet gs) {
    constexpr int WNN = BN / (32 * TN), NW = BM / (32 * TM) * WNN;
    constexpr int SUB = (BM + BN) * 128, STAGE = KT * SUB;          // bytes: one 64-wide pair of tiles [A rows | W rows]; one stage
    constexpr int RI = (BM + BN) / 8, PW = RI * KT / NW;            // DMA instructions (8 rows x 128 bytes each) per sub-tile pair; per wave per stage
    static_assert(PW * NW == RI * KT && (S - 2) * PW <= 63 && S >= 2, "stage pieces must divide over the waves; vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_d[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    static_assert(!PAIR || (TN == 2 && !GELU && !RES), "pair epilogue: two column tiles per wave");

    constexpr int BNO = PAIR ? BN / 2 : BN;                         // output columns per tile
    const int ntx = (N + BNO - 1) / BNO, rt = (M + BM - 1) / BM;
    int tile_id, kz;
    const bool owns = gemm_tile_of_block(ntx * rt, gs.xs, tile_id, kz);              // XCD-aware, balanced
    const int bx = tile_id / rt, by = tile_id - bx * rt;
    MG4_TLV(0);
    if (!owns) return;
    const int m0 = by * BM;
    int n0 = bx * BNO;
    const int wm = wave / WNN, wn = wave % WNN;
    if (!PAIR && gs.n_per_mat > 0) {   // the F16 language model's wq|wk|wv and w1|w3 in one launch: column tile -> (matrix, local column); n_per_mat is a multiple of BN
        const int mat = n0 / gs.n_per_mat;
        n0 -= mat * gs.n_per_mat; N = gs.n_per_mat;
        W += (size_t)mat * gs.w_mat_stride;
        if (out) out += (size_t)mat * gs.out_mat_stride;
        if (RES) residual += (size_t)mat * gs.out_mat_stride;
    }
    if (gs.k_per_slice > 0) {
        const int k0 = kz * gs.k_per_slice;
        A += k0; W += k0; K = min(K - k0, gs.k_per_slice);
        out += (size_t)kz * gs.slab_stride;
    }
    const int nk = K > 0 ? K / (64 * KT) : 0;                       // the launcher guarantees K % (64 * KT) == 0
    // epilogue operands first: an ordinary load whose result is used while DMAs are in flight would make hipcc wait for vmcnt(0)
    const int lcol = lane & 31, hh = lane >> 5;
    float bv[TN];
#pragma unroll
    for (int b = 0; b < TN; b++) bv[b] = bias ? bias[min(n0 + (wn * TN + b) * 32 + lcol, N - 1)] : 0.0f;
    // DMA sources of this lane: piece q = wave * PW + u -> (sub-tile q / RI, rows 8 j .. 8 j + 7 of [A rows | W rows], j = q % RI); lane -> (row = 8 j + lane / 8,
    // slot = lane % 8), source chunk = slot ^ ((row >> 1) & 7)
    const __half *src[PW]; unsigned dst[PW];
#pragma unroll
    for (int u = 0; u < PW; u++) {
        const int q = wave * PW + u, sub = q / RI, j = q % RI;
        const bool isA = j < BM / 8;
        const int row = 8 * (isA ? j : j - BM / 8) + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        // W tile row -> weight row: as it is, or (PAIR) 32-row blocks alternately from the two matrices
        const size_t wrow = PAIR ? (size_t)((row >> 5) & 1) * (size_t)gs.w_mat_stride + (size_t)min(n0 + ((row >> 6) << 5) + (row & 31), N - 1) * ldw : (size_t)min(n0 + row, N - 1) * ldw;
        src[u] = (isA ? A + (size_t)min(m0 + row, M - 1) * lda : W + wrow) + 8 * c + 64 * sub;
        dst[u] = (unsigned)(sub * SUB + j * 1024);
    }
    auto stage = [&](int kt, int buf) {
        unsigned char *base = smem_d + buf * STAGE;
#pragma unroll
        for (int u = 0; u < PW; u++) {
            const __half *g = kt < nk ? src[u] + (size_t)kt * (64 * KT) : A;     // past the end: one harmless line into a free buffer (the wait counts stay uniform)
            __builtin_amdgcn_global_load_lds((g_glb_ptr_t)g, (g_lds_ptr_t)(base + dst[u]), 16, 0, 0);
        }
    };
    const int l31 = lane & 31, key = (l31 >> 1) & 7;
    unsigned fa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) fa[ks] = (unsigned)(l31 * 128 + (((2 * ks + hh) ^ key) << 4));
    float16_t acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.0f;
#pragma unroll
    for (int t = 0; t < S - 1; t++) stage(t, t);
    MG4_TLV(1);
    int buf = 0;
    for (int kt = 0; kt < nk; kt++) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * PW) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt < 26) MG4_TLV(2 + kt);
        stage(kt + S - 1, buf == 0 ? S - 1 : buf - 1);
        const unsigned char *st = smem_d + buf * STAGE;
#pragma unroll
        for (int sub = 0; sub < KT; sub++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                half8_t af[TM], bf[TN];
#pragma unroll
                for (int a = 0; a < TM; a++) af[a] = *reinterpret_cast<const half8_t *>(st + sub * SUB + (wm * TM + a) * 32 * 128 + fa[ks]);
#pragma unroll
                for (int b = 0; b < TN; b++) bf[b] = *reinterpret_cast<const half8_t *>(st + sub * SUB + BM * 128 + (wn * TN + b) * 32 * 128 + fa[ks]);
#pragma unroll
                for (int a = 0; a < TM; a++)
#pragma unroll
                    for (int b = 0; b < TN; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == S ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the trailing dummy pieces
    MG4_TLV(29);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(RES ? residual : bias), 0, RES ? (int)(((size_t)(M - 1) * ldo + N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? (int)(((size_t)(M - 1) * ldo + N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t hb = __builtin_amdgcn_make_buffer_rsrc(out_h, 0, out_h ? (int)(((size_t)(M - 1) * ldo + N) * 2) : 0, 0x00020000);
    if constexpr (PAIR) {
        // all 16 table gathers of a tile are requested together; neighbouring lanes hold neighbouring columns, so the even lane stores the fp16 PAIR (4 bytes: a store
        // instruction covers 128 contiguous bytes per token row instead of 64)
        const int col = n0 + wn * 32 + lcol;
        const bool even = !(lcol & 1);
#pragma unroll
        for (int a = 0; a < TM; a++) {
            float sv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) sv[r] = silu_v(tb.silu, acc[a][0][r]);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float v = sv[r] * acc[a][1][r], v1 = __shfl_xor(v, 1);
                const bool ok = row < M && col < N;
                if (out && ok) out[(size_t)row * ldo + col] = v;
                if (out_h && ok) {
                    if (col + 1 < N || !even) { if (even) *reinterpret_cast<__half2 *>(out_h + (size_t)row * ldo + col) = __halves2half2(f2h_rn(v), f2h_rn(v1)); }
                    else out_h[(size_t)row * ldo + col] = f2h_rn(v);           // odd N: the last column alone
                }
            }
        }
    } else
#pragma unroll
    for (int b = 0; b < TN; b++) {
        const int col = n0 + (wn * TN + b) * 32 + lcol;
        int rmul; unsigned coff; gemm_out_map(gs.hm, min(col, N - 1), ldo, rmul, coff);
#pragma unroll
        for (int a = 0; a < TM; a++) {
            float v[16]; unsigned o[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                o[r] = row < M && col < N ? (unsigned)(row * rmul) + coff : 0x20000000u;
                v[r] = bias ? bv[b] + acc[a][b][r] : acc[a][b][r];
            }
            if (GELU) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = gelu_v(tb.gelu, v[r]);
            }
            if (RES) {
                float rr[16];
#pragma unroll
                for (int r = 0; r < 16; r++) rr[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (int)(o[r] * 4u), 0, 0));
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = rr[r] + v[r];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ob, (int)(o[r] * 4u), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(__half_as_ushort(f2h_rn(v[r])), hb, (int)(o[r] * 2u), 0, 0);
            }
        }
    }
#ifdef MG4_TIMELINE
    MG4_TLV(30);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG4_TLV(31);
#endif
}
template <int BM, int BN, int TM, int TN, int KT, int S>
static bool launch_gemm_dma_t(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                              float *out, __half *out_h, int ldo, hipStream_t s, int slices = 1, size_t slab_stride = 0, GemmSet gs = GemmSet{0, 0, 0, 0, 0, 0}) {
    constexpr int BKS = 64 * KT;
    if (K % BKS || lda % 8 || ldw % 8 || (reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) % 16 || (gs.n_per_mat > 0 && gs.n_per_mat % BN)) return false;
    gs.k_per_slice = slices > 1 ? (K / BKS + slices - 1) / slices * BKS : 0;
    gs.slab_stride = (long long)slab_stride;
    const int ntx = (N + BN - 1) / BN, rt = (M + BM - 1) / BM;
    const bool sx = splitk_on_xcds(slices) && gs.n_per_mat == 0;
    gs.xs = sx ? slices : 0;
    gs.hm = t_hm;
    dim3 grid(gemm_grid_x(ntx * rt, slices, sx), 1, sx ? 1u : (unsigned)slices), block(BM / (32 * TM) * (BN / (32 * TN)) * 64);
    const size_t lds = (size_t)S * KT * (BM + BN) * 128;
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_gemm_dma<BM, BN, TM, TN, KT, S, true, true>));
        HIP_IGNORE(lds_optin_max(&k_gemm_dma<BM, BN, TM, TN, KT, S, true, false>));
        HIP_IGNORE(lds_optin_max(&k_gemm_dma<BM, BN, TM, TN, KT, S, false, true>));
        HIP_IGNORE(lds_optin_max(&k_gemm_dma<BM, BN, TM, TN, KT, S, false, false>)); attr = true; }
    if (gelu && residual) hipLaunchKernelGGL((k_gemm_dma<BM, BN, TM, TN, KT, S, true, true>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, gs);
    else if (gelu) hipLaunchKernelGGL((k_gemm_dma<BM, BN, TM, TN, KT, S, true, false>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, gs);
    else if (residual) hipLaunchKernelGGL((k_gemm_dma<BM, BN, TM, TN, KT, S, false, true>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, gs);
    else hipLaunchKernelGGL((k_gemm_dma<BM, BN, TM, TN, KT, S, false, false>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, gs);
    return true;
}
template <int BM, int BN, int TM, int KT, int S>
static bool launch_gemm_dma_pair_t(const __half *A, int lda, const __half *W1, long long w3_minus_w1, int ldw, int M, int N, int K, const Tables &tb, float *out, __half *out_h, int ldo, hipStream_t s) {
    if (K % (64 * KT) || lda % 8 || ldw % 8 || N % 32 || (reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W1)) % 16 || w3_minus_w1 % 8) return false;
    GemmSet gs{0, w3_minus_w1, 0, 0, 0, 0};
    const int ntx = (N + BN / 2 - 1) / (BN / 2), rt = (M + BM - 1) / BM;
    dim3 grid((unsigned)((ntx * rt + 7) / 8 * 8), 1, 1), block(BM / (32 * TM) * (BN / 64) * 64);
    const size_t lds = (size_t)S * KT * (BM + BN) * 128;
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_gemm_dma<BM, BN, TM, 2, KT, S, false, false, true>)); attr = true; }
    hipLaunchKernelGGL((k_gemm_dma<BM, BN, TM, 2, KT, S, false, false, true>), grid, block, lds, s, A, lda, W1, ldw, M, N, K, nullptr, nullptr, tb, out, out_h, ldo, gs);
    return true;
}
// The feed-forward pair of an F16 language model at prompt sizes: out_h[M][ldo] = fp16(silu_table(A . W1^T) * (A . W3^T)), N columns (rows of W1 / W3), one launch; `out`
// (optional) receives the fp32 product before the rounding.  false: shape outside this path.
bool launch_gemm_f16_silu_pair(const __half *A, int lda, const __half *W1, const __half *W3, int M, int N, int K, const Tables &tb, float *out, __half *out_h, int ldo, int cus, hipStream_t s) {
    if (M < 256 || g_gemm_arm == -3 || W3 <= W1) return false;       // (tb.silu == null: the epilogue computes the table's values)
    const long long d = W3 - W1;
    const int wgs128 = (((M + 255) / 256) * ((N + 63) / 64) + 7) / 8 * 8;          // 256x128 tiles carry 64 pair columns
    if ((wgs128 * 2 > cus * 3 || N % 64) && launch_gemm_dma_pair_t<256, 256, 4, 1, 2>(A, lda, W1, d, K, M, N, K, tb, out, out_h, ldo, s)) return true;
    return launch_gemm_dma_pair_t<256, 128, 2, 1, 3>(A, lda, W1, d, K, M, N, K, tb, out, out_h, ldo, s);
}
// =====================================================================================================================
// Large-M form (M >= 512: the unquantised LLM's prompt rows -- BASELINE.json configs[4] -- and batched image encodes): 128x128 tile per 256-thread workgroup,
// wave = 64x64 = 2x2 MFMA tiles (four MFMAs per four 16-byte LDS fragment reads), BK = 64.  Both operands are K-contiguous fp16 rows, staged with
// global_load_lds (LDS-DMA: no staging registers, no ds_write pass) into two LDS buffers; the 16-byte chunks of a 128-byte tile row are XOR-swizzled by
// ((row >> 1) & 7) on the SOURCE side (the DMA destination is lane-linear), which makes the fragment ds_read_b128 of 32 consecutive rows bank-conflict free.
// One barrier per k tile: [own DMA landed] barrier [request tile k + 1 into the other buffer] [16 MFMAs on tile k].
// Same arithmetic as k_gemm_f16: exact fp16 products, fp32 accumulation over k in index order per 16-wide MFMA step.
// =====================================================================================================================
template <bool GELU, bool RES>
__global__ __launch_bounds__(256, 2) void k_gemm_f16_big(const __half *__restrict__ A, int lda, const __half *__restrict__ W, int ldw, int M, int N, int K, const float *__restrict__ bias,
                                                         const float *residual, const Tables tb, float *out, __half *__restrict__ out_h, int ldo, const GemmSet gs) {
    constexpr int BM = 128, BN = 128, BK = 64, TILE = BM * BK * 2;       // 16 KiB per operand tile
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_gb[];   // [2 buffers][A tile | W tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: the row tiles that share a weight tile run on ONE XCD (blocks are dealt round-robin to the 8 XCDs, each with its own L2)
    const int ntx = (N + BN - 1) / BN, rt = (M + BM - 1) / BM;
    const int tile_n = ntx * rt, tile_chunk = (tile_n + 7) >> 3, tile_slot = blockIdx.x >> 3, tile_id = (int)(blockIdx.x & 7) * tile_chunk + tile_slot;   // XCD-aware, balanced (below)
    const int bx = tile_id / rt, by = tile_id - bx * rt;
    if (tile_slot >= tile_chunk || tile_id >= tile_n) return;
    const int m0 = by * BM;
    int n0 = bx * BN;
    // several equally spaced, equally shaped weight matrices in one launch (the F16 language model's wq|wk|wv and w1|w3: one launch with 3x / 2x the workgroups instead
    // of launches that fill 160 of 256 CUs): column tile -> (matrix, local column); N becomes the matrix's own column count
    if (gs.n_per_mat > 0) {
        const int mat = n0 / gs.n_per_mat;
        n0 -= mat * gs.n_per_mat; N = gs.n_per_mat;
        W += (size_t)mat * gs.w_mat_stride;
        if (out) out += (size_t)mat * gs.out_mat_stride;
        if (RES) residual += (size_t)mat * gs.out_mat_stride;
    }
    // split K (grid.y slices, each writes raw partial sums into its own slab; combined in fixed order by launch_slab_reduce): wo / w2 have 160 column x row tiles
    int k_begin = 0, k_end = K;
    if (gs.k_per_slice > 0) { k_begin = blockIdx.y * gs.k_per_slice; k_end = min(K, k_begin + gs.k_per_slice); if (out) out += (size_t)blockIdx.y * gs.slab_stride; }
    const int wm = wave >> 1, wn = wave & 1;
    // DMA sources: instruction j (0..15) of a tile covers rows 8 j .. 8 j + 7; wave w issues j = 4 w .. 4 w + 3 for each operand.  lane -> (row = 8 j + lane / 8, slot = lane % 8),
    // source chunk = slot ^ ((row >> 1) & 7)
    const __half *asrc[4], *wsrc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int row = 8 * (4 * wave + u) + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        asrc[u] = A + (size_t)min(m0 + row, M - 1) * lda + 8 * c + k_begin;
        wsrc[u] = W + (size_t)min(n0 + row, N - 1) * ldw + 8 * c + k_begin;
    }
    auto stage = [&](int kt, int buf) {
        unsigned char *base = smem_gb + buf * 2 * TILE;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            __builtin_amdgcn_global_load_lds((g_glb_ptr_t)(asrc[u] + (size_t)kt * BK), (g_lds_ptr_t)(base + (4 * wave + u) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((g_glb_ptr_t)(wsrc[u] + (size_t)kt * BK), (g_lds_ptr_t)(base + TILE + (4 * wave + u) * 1024), 16, 0, 0);
        }
    };
    // fragment read offsets: row r = (lane & 31) (+ 32 for the second tile), chunk c = 2 ks + (lane >> 5) at slot c ^ ((r >> 1) & 7); r + 32 has the same swizzle key
    const int l31 = lane & 31, key = (l31 >> 1) & 7, hh = lane >> 5;
    unsigned fa[4];                                                   // per ks: byte offset of this lane's chunk within a 32-row band
#pragma unroll
    for (int ks = 0; ks < 4; ks++) fa[ks] = (unsigned)(l31 * 128 + (((2 * ks + hh) ^ key) << 4));
    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    const int nk = (k_end - k_begin) / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        __syncthreads();                                              // tile kt landed (the barrier's fence waits for this wave's DMA), everybody left the other buffer
        stage(min(kt + 1, nk - 1), buf ^ 1);
        const unsigned char *at = smem_gb + buf * 2 * TILE + (wm * 64) * 128, *wt = smem_gb + buf * 2 * TILE + TILE + (wn * 64) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const half8_t a0 = *reinterpret_cast<const half8_t *>(at + fa[ks]), a1 = *reinterpret_cast<const half8_t *>(at + 32 * 128 + fa[ks]);
            const half8_t b0 = *reinterpret_cast<const half8_t *>(wt + fa[ks]), b1 = *reinterpret_cast<const half8_t *>(wt + 32 * 128 + fa[ks]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // epilogue (as k_gemm_f16): branch-free, every store through a buffer descriptor (rows >= M / columns >= N get an out-of-range offset, an absent output has
    // zero records) -- the branchy form put s_waitcnt vmcnt(0) in front of each of the 64 stores of a wave
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(RES ? residual : bias), 0, RES ? (int)(((size_t)(M - 1) * ldo + N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? (int)(((size_t)(M - 1) * ldo + N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t hb = __builtin_amdgcn_make_buffer_rsrc(out_h, 0, out_h ? (int)(((size_t)(M - 1) * ldo + N) * 2) : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int col = n0 + wn * 64 + j * 32 + l31, colc = min(col, N - 1);
        const float bv = bias ? bias[colc] : 0.0f;
        int rmul; unsigned coff; gemm_out_map(gs.hm, colc, ldo, rmul, coff);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float v[16]; unsigned o[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                o[r] = row < M && col < N ? (unsigned)(row * rmul) + coff : 0x20000000u;
                v[r] = bias ? bv + acc[i][j][r] : acc[i][j][r];
            }
            if (GELU) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = gelu_v(tb.gelu, v[r]);
            }
            if (RES) {
                float rr[16];
#pragma unroll
                for (int r = 0; r < 16; r++) rr[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (int)(o[r] * 4u), 0, 0));
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = rr[r] + v[r];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ob, (int)(o[r] * 4u), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(__half_as_ushort(f2h_rn(v[r])), hb, (int)(o[r] * 2u), 0, 0);
            }
        }
    }
}
static int g_gemm_big_min_m = 512;   // smallest M that takes the 128x128 kernel (0 = never); MINIGPT4_GEMM_BIG_M, read once by Engine::init
static int g_f16_ks = 0;             // forced K split of the F16 set launches (0 = choose); MINIGPT4_F16_KS, read once by Engine::init
void set_gemm_tuning(int big_min_m, int f16_ks, int arm, int sk_arm) {
    if (big_min_m >= 0) g_gemm_big_min_m = big_min_m;
    g_f16_ks = std::max(0, std::min(f16_ks, 8));
    if (arm >= 0 || arm == -2 || arm == -3) g_gemm_arm = arm;
    if (sk_arm >= 0) g_gemm_sk_arm = sk_arm;
}
static bool launch_gemm_big(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                            float *out, __half *out_h, int ldo, hipStream_t s, const GemmSet gs_in = GemmSet{0, 0, 0, 0, 0}, int slices = 1) {
    GemmSet gs = gs_in; gs.hm = t_hm;
    static bool init = false;
    if (!init) { init = true;
        HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16_big<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16_big<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16_big<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16_big<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); }
    if (g_gemm_big_min_m <= 0 || M < g_gemm_big_min_m || K % 64 || K < 64 || lda % 8 || ldw % 8 || (reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) % 16) return false;
    const int ntx = (N + 127) / 128, rt = (M + 127) / 128;
    const dim3 grid((unsigned)((ntx * rt + 7) / 8 * 8), (unsigned)slices), block(256);
    const size_t lds = 64 * 1024;
    if (gelu && residual) hipLaunchKernelGGL((k_gemm_f16_big<true, true>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, gs);
    else if (gelu) hipLaunchKernelGGL((k_gemm_f16_big<true, false>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, gs);
    else if (residual) hipLaunchKernelGGL((k_gemm_f16_big<false, true>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, gs);
    else hipLaunchKernelGGL((k_gemm_f16_big<false, false>), grid, block, lds, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, gs);
    return true;
}

// F16 language-model weights at prompt sizes (M >= the big kernel's threshold): n = 1..3 equally spaced, equally shaped matrices [N][K] against the M fp16 rows of A in
// ONE launch; y[m] are [M][ldo] fp32 (+ residual[m]).  With fewer column x row tiles than 2 per CU the K range is split (partial slabs in `ws`, combined in fixed
// order by launch_slab_reduce).  false: shape outside this path (nothing launched) -- the caller multiplies matrix by matrix.
bool launch_gemm_f16_set(const __half *A, int lda, const __half *const *W, int n, int M, int N, int K, float *const *y, const float *const *residual, int ldo, float *ws,
                         size_t ws_floats, int cus, hipStream_t s, SlabSrc *defer) {
    if (defer) *defer = SlabSrc{};
    if (n < 1 || n > 3 || N % 128 || K % 64) return false;
    const long long wstride = n > 1 ? W[1] - W[0] : 0, ystride = n > 1 ? y[1] - y[0] : 0;
    for (int i = 2; i < n; i++) if (W[i] - W[i - 1] != wstride || y[i] - y[i - 1] != ystride) return false;
    if (n > 1 && (wstride <= 0 || ystride <= 0)) return false;
    if (residual) for (int i = 0; i < n; i++) if (!residual[i] || residual[i] - residual[0] != (long long)i * ystride) return false;
    Tables tb{};
    const size_t out_floats = (size_t)M * ldo;
    // Round 3: 256x128 tiles of the LDS-DMA ring kernel (8 waves of 64x64, three stages, counted waits), with the K range split until the launch has about one
    // workgroup per CU -- 13B at 512 rows: wq|wk|wv 240 workgroups 104 -> 83 us, w1|w3 432 workgroups 199 -> 161 us, wo 80 x 3 slices 63 -> 45 us, w2 139 -> 85 us
    // (profiles/r03b_gemm_f16_set_512rows.log).  MINIGPT4_GEMM_ARM=-3: the 128x128 kernel below (A/B).
    if (g_gemm_arm != -3 && M >= 256) {
        const int wgs = (((M + 255) / 256) * (n * N / 128) + 7) / 8 * 8;   // the tiles are dealt to the 8 XCDs in equal shares
        int ks = 1;
        if (n == 1 && out_floats % 4 == 0)
            while (wgs * (ks + 1) <= cus && (K / 64) / (ks + 1) >= 16 && ws && (size_t)(ks + 1) * out_floats <= ws_floats && ks < 4) ks++;
        if (g_f16_ks > 0 && n == 1 && out_floats % 4 == 0 && ws && (size_t)g_f16_ks * out_floats <= ws_floats && (K / 64) / g_f16_ks >= 8) ks = g_f16_ks;   // experiments (MINIGPT4_F16_KS)
        const GemmSet gs{n > 1 ? N : 0, wstride, ystride, 0, 0};
        if (ks > 1) {
            if (launch_gemm_dma_t<256, 128, 2, 2, 1, 3>(A, lda, W[0], K, M, N, K, nullptr, nullptr, false, tb, ws, nullptr, ldo, s, ks, out_floats, gs)) {
                if (defer) { defer->ws = ws; defer->ks = ks; defer->stride = (long long)out_floats; defer->n = 1; defer->y[0] = y[0]; defer->res[0] = residual ? residual[0] : nullptr; }   // combined by the consumer
                else launch_slab_reduce(ws, ks, (long long)out_floats, residual ? residual[0] : nullptr, y[0], out_floats, s);
                return true;
            }
        } else {
            // far more 256x128 tiles than CUs (w1|w3: 432): 256x256 tiles, 8 waves of 128x64, two stages (216 workgroups: 159 -> 146 us)
            if (wgs * 2 > cus * 3 && N % 256 == 0 &&
                launch_gemm_dma_t<256, 256, 4, 2, 1, 2>(A, lda, W[0], K, M, n * N, K, nullptr, residual ? residual[0] : nullptr, false, tb, y[0], nullptr, ldo, s, 1, 0, gs)) return true;
            if (launch_gemm_dma_t<256, 128, 2, 2, 1, 3>(A, lda, W[0], K, M, n * N, K, nullptr, residual ? residual[0] : nullptr, false, tb, y[0], nullptr, ldo, s, 1, 0, gs)) return true;
        }
    }
    const int tiles = ((M + 127) / 128) * (n * N / 128);
    int ks = 1;
    while (tiles * ks < 2 * cus && (K / 64) / (ks + 1) >= 16 && ws && (size_t)(ks + 1) * out_floats * n <= ws_floats && ks < 4) ks++;
    if (g_f16_ks > 0) ks = g_f16_ks;
    if (ks > 1 && (!ws || (size_t)ks * out_floats * n > ws_floats || n > 1 || out_floats % 4)) ks = 1;   // split launches are single-matrix (wo, w2): the sets have tiles enough
    GemmSet gs{n > 1 ? N : 0, wstride, ystride, 0, 0};
    if (ks > 1) {
        const int kt = K / 64, per = (kt + ks - 1) / ks;
        gs.k_per_slice = per * 64; gs.slab_stride = (long long)out_floats;
        ks = (kt + per - 1) / per;
        if (!launch_gemm_big(A, lda, W[0], K, M, N, K, nullptr, nullptr, false, tb, ws, nullptr, ldo, s, gs, ks)) return false;
        launch_slab_reduce(ws, ks, (long long)out_floats, residual ? residual[0] : nullptr, y[0], out_floats, s);
        return true;
    }
    return launch_gemm_big(A, lda, W[0], K, M, n * N, K, nullptr, residual ? residual[0] : nullptr, false, tb, y[0], nullptr, ldo, s, gs, 1);
}

// Tile shape experiments at M = 257 (profiles/r01i_ab_encode.log): 128x64 (8 waves) and 128x128 (16 waves) tiles halve the bytes moved per flop but are
// 16 % / 28 % SLOWER end to end than 64x64; 64x32 tiles (more workgroups) are slower too; an XCD-aware tile order removes a 4x weight re-fetch from
// HBM (PMC FETCH_SIZE) without changing the time; 64x128 tiles with two accumulators per wave (fewer LDS reads per MFMA) are 26 % slower.  Every
// variant with fewer workgroups loses: the launches are bound by the serial per-k-tile chain of each workgroup (global -> register -> LDS ->
// MFMA, two barriers per tile), not by traffic, LDS bandwidth or the matrix cores -- the next step is an LDS-DMA ring per workgroup.
// =====================================================================================================================
// Skinny-M form (the Q-Former: 32 query rows per image against 768- / 3072-wide layers).  With 64x64 tiles a [32 x 768] x [768 x 768] product is 12 workgroups on a
// 256-CU chip, each walking its K alone: 8-12 us per GEMM for 1.2 MB of weights (round 2: the Q-Former ran at ~40x its HBM floor).  Here a workgroup owns 16 output
// COLUMNS (N / 16 workgroups: 48 ... 320) and 32 or 64 rows; its eight waves split K (k-steps of 32, interleaved), every lane fetches its MFMA fragments straight from
// global memory (16-byte loads, up to 12 k-steps in flight = ONE round trip for K <= 3072 -- no LDS staging: nothing is reused inside a workgroup),
// v_mfma_f32_16x16x32_f16, and the eight partial tiles are added in wave order through LDS (deterministic).  Same arithmetic as k_gemm_f16 (exact fp16 products, fp32 accumulation); per output element the order of additions
// does not depend on M, so a batched encode still equals the single-image encode bit for bit.
// =====================================================================================================================
typedef float float4v_t __attribute__((ext_vector_type(4)));
constexpr int SK_WAVES = 4;                                       // waves per workgroup = K slices (fixed: the order of additions per output element must not depend on M)
// k_per_slice > 0 (round 6, the Q-Former's 768-wide dense / output layers): grid.z workgroups share an output tile, each a contiguous K range, and store their RAW partial
// sums into slab z (no bias / GELU / residual); k_splitk_reduce_ln adds the slabs in slab order with bias + residual and applies the LayerNorm that follows in the graph.
// With N = 768 the whole-K form is 48 workgroups on 256 CUs, each pulling A [32][K] + W [16][K] (294 KB at K = 3072) through ONE CU's load path.
template <int MT, bool GELU, bool RES>
__global__ __launch_bounds__(64 * SK_WAVES) void k_gemm_f16_skinny(const __half *__restrict__ A, int lda, const __half *__restrict__ W, int ldw, int M, int N, int K,
                                                                   const float *__restrict__ bias, const float *residual, const Tables tb, float *out, __half *__restrict__ out_h, int ldo,
                                                                   int k_per_slice, size_t slab_stride) {
    constexpr int U = MT <= 2 ? 12 : 8;                           // k-steps in flight per wave: U x (1 + MT) 16-byte loads per lane (36 / 40)
    if (k_per_slice > 0) { const int k0 = (int)blockIdx.z * k_per_slice; A += k0; W += k0; K = min(K - k0, k_per_slice); out += (size_t)blockIdx.z * slab_stride; }
    __shared__ float red[SK_WAVES][MT][64][4];                    // [wave][m tile][lane][acc register]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MT);
    const int r16 = lane & 15, kc = lane >> 4;
    const float bv = bias ? bias[min(n0 + r16, N - 1)] : 0.0f;   // requested with the operands (it was a memory round trip after the reduction barrier)
    const half8_t *wp = reinterpret_cast<const half8_t *>(W + (size_t)min(n0 + r16, N - 1) * ldw + 8 * kc);
    const half8_t *ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) ap[mt] = reinterpret_cast<const half8_t *>(A + (size_t)min(m0 + 16 * mt + r16, M - 1) * lda + 8 * kc);
    const int mt_n = min(MT, (M - m0 + 15) / 16);                 // live M tiles (workgroup-uniform)
    float4v_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = float4v_t{0.0f, 0.0f, 0.0f, 0.0f};
    const int nks = K / 32;                                       // k-steps; wave w takes w, w + 8, ... (K = 768: 3 each, 3072: 12 each -- one round trip)
    for (int ks0 = wave; ks0 < nks; ks0 += SK_WAVES * U) {
        half8_t wf[U], af[U][MT];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int ks = min(ks0 + SK_WAVES * u, nks - 1);      // clamped: the loads never branch; a step past the end is not multiplied
            wf[u] = wp[ks * 4];                                   // 32 halfs = 4 chunks of 8 per k-step
#pragma unroll
            for (int mt = 0; mt < MT; mt++) af[u][mt] = ap[mt][ks * 4];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (ks0 + SK_WAVES * u < nks) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) if (mt < mt_n) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u][mt], wf[u], acc[mt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave][mt][lane][r] = acc[mt][r];
    __syncthreads();
    // wave w finishes M tile w: the eight K partials in wave order, then the epilogue of k_gemm_f16 (bias, fp16-table GELU, residual)
    const int mt = wave;
    if (mt >= mt_n) return;
    const int col = n0 + r16;
    // branch-free (as k_gemm_f16's epilogue): the four residual values are requested together through a buffer descriptor, the stores go through descriptors whose
    // bounds check drops rows >= M / columns >= N and absent outputs -- the branchy form waited for vmcnt(0) around every element (11.2 vs 7.7 us for the launches with a residual)
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(RES ? residual : bias), 0, RES ? (int)(((size_t)(M - 1) * ldo + N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? (int)(((size_t)(M - 1) * ldo + N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t hb = __builtin_amdgcn_make_buffer_rsrc(out_h, 0, out_h ? (int)(((size_t)(M - 1) * ldo + N) * 2) : 0, 0x00020000);
    unsigned o[4]; float rr[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = m0 + 16 * mt + 4 * kc + r;
        o[r] = row < M && col < N ? (unsigned)(row * ldo + col) : 0x20000000u;
        rr[r] = RES ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (int)(o[r] * 4u), 0, 0)) : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float v = red[0][mt][lane][r];
#pragma unroll
        for (int w = 1; w < SK_WAVES; w++) v += red[w][mt][lane][r];
        if (bias) v = bv + v;
        if (GELU) v = gelu_v(tb.gelu, v);
        if (RES) v = rr[r] + v;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ob, (int)(o[r] * 4u), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b16(__half_as_ushort(f2h_rn(v)), hb, (int)(o[r] * 2u), 0, 0);
    }
}
template <int MT>
static void launch_skinny_mt(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                             float *out, __half *out_h, int ldo, hipStream_t s) {
    const dim3 grid((unsigned)(N / 16), (unsigned)((M + 16 * MT - 1) / (16 * MT))), block(64 * SK_WAVES);
    if (gelu && residual) hipLaunchKernelGGL((k_gemm_f16_skinny<MT, true, true>), grid, block, 0, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, 0, (size_t)0);
    else if (gelu) hipLaunchKernelGGL((k_gemm_f16_skinny<MT, true, false>), grid, block, 0, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, 0, (size_t)0);
    else if (residual) hipLaunchKernelGGL((k_gemm_f16_skinny<MT, false, true>), grid, block, 0, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, 0, (size_t)0);
    else hipLaunchKernelGGL((k_gemm_f16_skinny<MT, false, false>), grid, block, 0, s, A, lda, W, ldw, M, N, K, bias, residual, tb, out, out_h, ldo, 0, (size_t)0);
}
// Split-K form of the skinny-M kernel: `slices` workgroups per output tile, raw partial sums into slabs [slice][M][ldo] (k_gemm_f16_skinny's header).  The slice count
// must not depend on M (image b of a batch equals the image encoded alone, bit for bit).  false: shape outside this path, nothing launched.
bool launch_gemm_f16_skinny_splitk(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, int slices, float *slabs, size_t slab_stride, int ldo, hipStream_t s) {
    if (N % 16 || K % 32 || M < 1 || (lda % 8) || (ldw % 8) || slices < 2 || slices > 12 || !slabs) return false;
    const int nks = K / 32, per = (nks + slices - 1) / slices;
    if ((nks + per - 1) / per != slices) return false;           // every slice multiplies something
    Tables tb{};
    if (M <= 32) { const dim3 grid((unsigned)(N / 16), (unsigned)((M + 31) / 32), (unsigned)slices);
        hipLaunchKernelGGL((k_gemm_f16_skinny<2, false, false>), grid, dim3(64 * SK_WAVES), 0, s, A, lda, W, ldw, M, N, K, nullptr, nullptr, tb, slabs, nullptr, ldo, per * 32, slab_stride); }
    else { const dim3 grid((unsigned)(N / 16), (unsigned)((M + 63) / 64), (unsigned)slices);
        hipLaunchKernelGGL((k_gemm_f16_skinny<4, false, false>), grid, dim3(64 * SK_WAVES), 0, s, A, lda, W, ldw, M, N, K, nullptr, nullptr, tb, slabs, nullptr, ldo, per * 32, slab_stride); }
    return true;
}
bool launch_gemm_f16_skinny(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                            float *out, __half *out_h, int ldo, hipStream_t s) {
    if (N % 16 || K % 32 || M < 1 || (lda % 8) || (ldw % 8)) return false;
    // 32 rows per workgroup for one image (no register or bandwidth spent on clamped rows), 64 for a batch (32 at every batch size measured slower in round 6: 7.52-7.63 vs
    // 7.38-7.42 ms at four images, 13.1 vs 12.7-12.9 at eight); the result of a row does not depend on the choice
    if (M <= 32) launch_skinny_mt<2>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);
    else launch_skinny_mt<4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);
    return true;
}
void launch_gemm_f16(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                     float *out, __half *out_h, int ldo, hipStream_t s) {
    static int cus = 0;
    if (!cus) { hipDeviceProp_t prop; cus = hipGetDeviceProperties(&prop, 0) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
    if (g_gemm_arm > 0 && launch_gemm_f16_arm(g_gemm_arm, A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s)) return;
    // Tile shape by workgroup count (round 3, tools/timeline_gemm.py): a launch lasts as long as ONE workgroup's chain (1.5 us to the first request, ~0.5 us per
    // 128 k, the epilogue) as long as every workgroup has a CU to itself -- a CU pulls only 50-75 GB/s through its load path whatever the kernel does, so the second
    // workgroup of a CU roughly doubles that CU's time (330 tiles of 64x64 on 256 CUs: 16 us; 198 tiles of 128x64: 12.8 us).  Hence: the shape with the MOST
    // workgroups that still fit one per CU (counted per XCD: the tiles are dealt to the 8 XCDs in equal shares); nothing fits -> from 512 rows on the 128x128 kernel
    // with two workgroups per CU (k_gemm_f16_big), below that the shape with the fewest workgroups.  Every shape accumulates an output element in the same order
    // (tests: ..._tile_shapes_are_bit_identical), so the choice never changes a result.
    struct Shape { int bm, bn, arm; };
    static const Shape shapes[5] = {{64, 32, 11}, {64, 64, 0}, {128, 64, 31}, {64, 128, 5}, {128, 128, 32}};   // arms of launch_gemm_f16_arm; 0 = the 64x64 launch below
    int pick = -1, pick_n = 0;
    for (int i = 0; i < 5 && g_gemm_arm == 0; i++) {
        const int n = (((M + shapes[i].bm - 1) / shapes[i].bm) * ((N + shapes[i].bn - 1) / shapes[i].bn) + 7) / 8 * 8;
        if (pick < 0 || (n <= cus ? (pick_n > cus || n > pick_n) : (pick_n > cus && n < pick_n))) { pick = i; pick_n = n; }
    }
    if ((pick < 0 || pick_n > cus) && launch_gemm_big(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s)) return;   // M >= 512 (MINIGPT4_GEMM_BIG_M)
    if (pick >= 0 && shapes[pick].arm) {
        if (launch_gemm_f16_arm(shapes[pick].arm, A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s)) return;
        if (shapes[pick].arm == 31 && launch_gemm_f16_arm(4, A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s)) return;   // K % 64: the register-staged 128x64
        if (shapes[pick].arm == 32 && launch_gemm_big(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s)) return;
    }
    launch_gemm_t<128, 64, 64>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);
}
// The same GEMM with its fp32 output stored head-major (struct HeadMajor above): N = 3 x D columns -> out[3][D / hd][M][hd].  Every tile shape the dispatcher may pick maps its
// stores the same way, so the choice still never changes a value or where it lands.
void launch_gemm_f16_head_major(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const Tables &tb, float *out, int D, int hd, hipStream_t s) {
    if (D <= 0 || hd <= 0 || D % hd || N % D) throw HipError{hipErrorInvalidValue, "head-major GEMM output: N must be a multiple of D and D of hd", __FILE__, __LINE__};
    struct Scope { Scope(int rows, int D_, int hd_) { t_hm = HeadMajor{rows, D_, hd_}; } ~Scope() { t_hm = HeadMajor{0, 0, 0}; } } scope(M, D, hd);
    launch_gemm_f16(A, lda, W, ldw, M, N, K, bias, nullptr, false, tb, out, nullptr, N, s);
}

// tile-shape arms of k_gemm_f16 for the micro-benchmark (test library only calls this): 3 = BK 64, 4 = 128x64 tiles, 5 = 64x128 tiles, 6 = BK 256
bool launch_gemm_f16_arm(int arm, const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, const float *bias, const float *residual, bool gelu, const Tables &tb,
                         float *out, __half *out_h, int ldo, hipStream_t s) {
    switch (arm) {
    case 3: launch_gemm_t<64, 64, 64>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;
    case 4: launch_gemm_t<128, 128, 64>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;          // 8 waves, one tile each
    case 5: launch_gemm_t<128, 64, 128>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;
    case 7: launch_gemm_t<128, 128, 64, 2, 1>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;    // 4 waves of 64x32
    case 8: launch_gemm_t<64, 128, 64, 2, 1>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;
    case 9: launch_gemm_t<128, 64, 128, 1, 2>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;    // 4 waves of 32x64
    case 10: launch_gemm_t<64, 64, 128, 1, 2>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;
    case 11: launch_gemm_t<128, 64, 32>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;          // 2 waves
    case 12: launch_gemm_t<64, 64, 32>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;
    case 13: launch_gemm_t<64, 128, 128, 2, 2>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;   // 4 waves of 64x64
    case 14: launch_gemm_t<128, 32, 64>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;          // 2 waves, 32 rows
    case 20: return launch_gemm_dma_t<64, 64, 1, 1, 1, 4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);     // 64 KB: two workgroups per CU
    case 21: return launch_gemm_dma_t<64, 64, 1, 1, 2, 3>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);     // 96 KB
    case 22: return launch_gemm_dma_t<64, 64, 1, 1, 2, 4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);     // 128 KB
    case 23: return launch_gemm_dma_t<128, 64, 2, 1, 1, 4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);    // 96 KB, 4 waves of 64x32
    case 24: return launch_gemm_dma_t<128, 64, 2, 1, 2, 3>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);    // 144 KB
    case 25: return launch_gemm_dma_t<64, 128, 1, 2, 1, 4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);    // 4 waves of 32x64
    case 26: return launch_gemm_dma_t<64, 128, 1, 2, 2, 3>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);
    case 27: return launch_gemm_dma_t<64, 32, 1, 1, 1, 4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);     // 2 waves, 48 KB
    case 28: return launch_gemm_dma_t<64, 32, 1, 1, 2, 4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);     // 2 waves, 96 KB
    case 29: return launch_gemm_dma_t<32, 64, 1, 1, 2, 4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);     // 2 waves, 32 rows
    case 30: return launch_gemm_dma_t<128, 128, 2, 2, 1, 4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);   // 4 waves of 64x64, 128 KB
    case 31: return launch_gemm_dma_t<128, 64, 1, 1, 1, 4>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);    // 8 waves of 32x32
    case 32: return launch_gemm_dma_t<128, 128, 2, 2, 1, 3>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);   // the large-M tile with 3 / 2 stages
    case 33: return launch_gemm_dma_t<128, 128, 2, 2, 1, 2>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);
    case 34: return launch_gemm_dma_t<256, 128, 2, 2, 1, 3>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);   // 8 waves of 64x64, 144 KB
    case 35: return launch_gemm_dma_t<128, 256, 2, 2, 1, 3>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);
    case 36: return launch_gemm_dma_t<256, 128, 2, 2, 1, 2>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);   // 96 KB
    case 38: return launch_gemm_dma_t<256, 256, 2, 2, 1, 2>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);   // 16 waves of 64x64, 128 KB
    case 39: return launch_gemm_dma_t<256, 256, 4, 2, 1, 2>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s);   // 8 waves of 128x64
    case 37: launch_gemm_t<128, 64, 64>(A, lda, W, ldw, M, N, K, bias, residual, gelu, tb, out, out_h, ldo, s); return true;          // the 64x64 launch
    default: return false;
    }
}

// =====================================================================================================================
// LayerNorm: one workgroup per row
// =====================================================================================================================
__device__ __forceinline__ double block_sum_d(double v, double *red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// Split-K form for the GEMMs whose N gives fewer 64x64 tiles than CUs (ViT attn.proj / mlp.fc2, the Q-Former's 768-wide layers): `slices` workgroups
// share an output tile, each multiplies a contiguous K range and writes raw fp32 partial sums into its own slab [M][ldo]; k_splitk_reduce_ln adds
// the slabs in a fixed order (deterministic) together with bias / residual and the LayerNorm that follows in the graph.
// (The slice count must not depend on the number of rows: image b of a batch has to equal the same image encoded alone bit for bit, and the slices fix the order in
// which an output element's k ranges are added.  4 images' fc2 would prefer 3 slices of 256x128 tiles -- 61 -> 48 us, profiles/r03b_gemm_batched_encode.log -- not taken.)
int gemm_split_slices(int K, int want) { const int nk = (K + 127) / 128; int s = std::max(1, std::min(want, nk)); const int per = (nk + s - 1) / s; return (nk + per - 1) / per; }
void launch_gemm_f16_splitk(const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, int slices, float *slabs, size_t slab_stride, int ldo, hipStream_t s) {
    Tables tb{};
    if (g_gemm_sk_arm && launch_gemm_f16_splitk_arm(g_gemm_sk_arm, A, lda, W, ldw, M, N, K, slices, slabs, slab_stride, ldo, s)) return;
    // Several images per pass (round 6): the LDS-DMA ring tiles of the language model's prompt GEMMs -- 256x128 from three images on, 128x128 at two -- instead of 64x64
    // tiles: the ViT's fc2 at four images 43 -> 30 us (profiles/r06_gemm_batched_tile_sweep.log: with the reduce 55.5 -> 42.4 us at M = 1028, 46.2 -> 40.2 at 771,
    // 36.4 -> 30.2 at 514, 109.7 -> 75.4 at 2056).  Only when both forms cut K at the same columns (the 64x64 form counts slices in 128-wide k tiles, the ring in 64-wide
    // ones): the slice boundaries fix the order in which an output's k ranges are added, and image b of a batch must equal the same image encoded alone bit for bit.
    const int kps128 = slices > 1 ? ((K + 127) / 128 + slices - 1) / slices * 128 : 0, kps64 = slices > 1 && K % 64 == 0 ? (K / 64 + slices - 1) / slices * 64 : -1;
    if (g_gemm_arm == 0 && kps128 == kps64) {
        if (M >= 640 && launch_gemm_dma_t<256, 128, 2, 2, 1, 3>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride)) return;
        if (M >= 384 && launch_gemm_dma_t<128, 128, 2, 2, 1, 3>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride)) return;
    }
    launch_gemm_t<128, 64, 64>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride);
}
bool launch_gemm_f16_splitk_arm(int arm, const __half *A, int lda, const __half *W, int ldw, int M, int N, int K, int slices, float *slabs, size_t slab_stride, int ldo, hipStream_t s) {
    Tables tb{};
    switch (arm) {   // micro-benchmark arms (slices counted in whole k tiles of the arm's BK by launch_gemm_t)
    case 0: launch_gemm_t<128, 64, 64>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride); return true;
    case 3: launch_gemm_t<64, 64, 64>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride); return true;
    case 7: launch_gemm_t<128, 128, 64, 2, 1>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride); return true;
    case 8: launch_gemm_t<64, 128, 64, 2, 1>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride); return true;
    case 11: launch_gemm_t<128, 64, 32>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride); return true;
    case 12: launch_gemm_t<64, 64, 32>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride); return true;
    case 13: launch_gemm_t<64, 128, 128, 2, 2>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride); return true;
    case 20: return launch_gemm_dma_t<64, 64, 1, 1, 1, 4>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride);
    case 21: return launch_gemm_dma_t<64, 64, 1, 1, 2, 3>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride);
    case 23: return launch_gemm_dma_t<128, 64, 2, 1, 1, 4>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride);
    case 27: return launch_gemm_dma_t<64, 32, 1, 1, 1, 4>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride);
    case 28: return launch_gemm_dma_t<64, 32, 1, 1, 2, 4>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride);
    case 32: return launch_gemm_dma_t<128, 128, 2, 2, 1, 3>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride);
    case 34: return launch_gemm_dma_t<256, 128, 2, 2, 1, 3>(A, lda, W, ldw, M, N, K, nullptr, nullptr, false, tb, slabs, nullptr, ldo, s, slices, slab_stride);
    default: return false;
    }
}
// x = residual + (bias + sum_z slab_z)   [x_out, fp32];   then, when ln_w is given, ggml_norm(x) * ln_w + ln_b -> ln_out (fp32) / ln_out_h (fp16).
// One workgroup per row, n <= 2048.
// MAXZ >= n_slabs, MAXE >= ceil(n / 256) (round 6: instantiated per size class -- the one-size form issued 12 x 8 + 40 loads per thread whatever the row needed, 96 of
// the 136 redundant for the ViT's 4 slabs of 1408, and the address path of four workgroups per CU made the 4-image launch 12.9 us; same additions in the same order)
template <int MAXZ, int MAXE>
__global__ __launch_bounds__(256) void k_splitk_reduce_ln(const float *__restrict__ slabs, int n_slabs, size_t slab_stride, const float *__restrict__ bias, const float *residual, int n,
                                                          float *x_out, const float *__restrict__ ln_w, const float *__restrict__ ln_b, float *__restrict__ ln_out,
                                                          __half *__restrict__ ln_out_h) {
    __shared__ double red[4];
    const size_t row = blockIdx.x;
    float xv[MAXE];
    double s = 0.0;
    // every slab value of the row is requested before the first one is used (clamped indices, no load under a branch: the round-1 form walked the slabs with one
    // dependent round trip each -- 13 us for a 4-slab row of 1408; same additions in the same order)
    float part[MAXZ][MAXE], rv[MAXE], bv[MAXE], lw[MAXE], lb[MAXE];
#pragma unroll
    for (int z = 0; z < MAXZ; z++) {
        const size_t zo = (size_t)min(z, n_slabs - 1) * slab_stride + row * n;
#pragma unroll
        for (int e = 0; e < MAXE; e++) part[z][e] = slabs[zo + min((int)threadIdx.x + 256 * e, n - 1)];
    }
#pragma unroll
    for (int e = 0; e < MAXE; e++) {
        const int ic = min((int)threadIdx.x + 256 * e, n - 1);
        rv[e] = residual ? residual[row * n + ic] : 0.0f;
        bv[e] = bias ? bias[ic] : 0.0f;
        lw[e] = ln_w ? ln_w[ic] : 0.0f; lb[e] = ln_b ? ln_b[ic] : 0.0f;     // requested with the first loads: not a second round trip after the reductions
    }
#pragma unroll
    for (int e = 0; e < MAXE; e++) {
        const int i = threadIdx.x + 256 * e;
        float v = 0.0f;
        if (i < n) {
            float a = part[0][e];
#pragma unroll
            for (int z = 1; z < MAXZ; z++) a = z < n_slabs ? a + part[z][e] : a;
            v = bias ? bv[e] + a : a;
            if (residual) v = rv[e] + v;
            if (x_out) x_out[row * n + i] = v;
        }
        xv[e] = v; s += (double)v;
    }
    if (!ln_w) return;
    const float mean = (float)(block_sum_d(s, red) / (double)n);
    double s2 = 0.0;
#pragma unroll
    for (int e = 0; e < MAXE; e++) { const int i = threadIdx.x + 256 * e; if (i < n) { const float v = xv[e] - mean; s2 += (double)(v * v); } }
    const float variance = (float)(block_sum_d(s2, red) / (double)n);
    const float scale = 1.0f / sqrtf(variance + 1e-5f);
#pragma unroll
    for (int e = 0; e < MAXE; e++) {
        const int i = threadIdx.x + 256 * e;
        if (i < n) {
            float v = (xv[e] - mean) * scale;
            v = lw[e] * v;
            if (ln_b) v = v + lb[e];
            if (ln_out) ln_out[row * n + i] = v;
            if (ln_out_h) ln_out_h[row * n + i] = f2h_rn(v);
        }
    }
}
// (Round 6 measured the same kernel on 16-byte pieces -- a quarter of the load / store instructions, bit-identical rows -- and lost: 3.79 vs 3.75 ms per encode, 7.37 vs 7.27 at
// four images (profiles/r06_attn_query_tiles.log, run 18): 4 slabs x 2 float4 + operands live per thread cost more residency than the address path gained.  Removed.)
void launch_splitk_reduce_ln(const float *slabs, int n_slabs, size_t slab_stride, const float *bias, const float *residual, int rows, int n, float *x_out, const float *ln_w,
                             const float *ln_b, float *ln_out, __half *ln_out_h, hipStream_t s) {
    if (n > 2048 || n_slabs < 1 || n_slabs > 12) throw HipError{hipErrorInvalidValue, "splitk reduce: row longer than 2048 or more than 12 slabs", __FILE__, __LINE__};
#define MG4_RLN(Z_, E_) hipLaunchKernelGGL((k_splitk_reduce_ln<Z_, E_>), dim3((unsigned)rows), dim3(256), 0, s, slabs, n_slabs, slab_stride, bias, residual, n, x_out, ln_w, ln_b, ln_out, ln_out_h)
#define MG4_RLN_E(Z_) do { if (n <= 768) MG4_RLN(Z_, 3); else if (n <= 1536) MG4_RLN(Z_, 6); else MG4_RLN(Z_, 8); } while (0)
    if (n_slabs <= 2) MG4_RLN_E(2); else if (n_slabs <= 4) MG4_RLN_E(4); else if (n_slabs <= 8) MG4_RLN_E(8); else MG4_RLN_E(12);
#undef MG4_RLN_E
#undef MG4_RLN
}

template <int MAXE>                               // >= ceil(n / 256)
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b, int n, float *__restrict__ out,
                                                   __half *__restrict__ out_h, int seq) {
    __shared__ double red[4];
    const size_t row = blockIdx.x;
    const float *xr = x + row * n;
    float xv[MAXE];
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < MAXE; e++) { const int i = threadIdx.x + 256 * e; xv[e] = i < n ? xr[i] : 0.0f; s += (double)xv[e]; }
    // weight and bias requested together with the row (the barriers of the two reductions would otherwise put this memory round trip at the END of the kernel)
    float wv[MAXE], bb[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; e++) { const int ic = min((int)threadIdx.x + 256 * e, n - 1); wv[e] = w[ic]; bb[e] = b ? b[ic] : 0.0f; }
    float mean, variance;
    if (seq) {   // MINIGPT4_PARITY: ggml_norm's two loops literally -- one double accumulator each, element order (oracle/refcpu.c layer_norm)
        if (threadIdx.x == 0) {
            double a = 0.0; for (int i = 0; i < n; i++) a += (double)xr[i];
            const float m = (float)(a / (double)n);
            double a2 = 0.0; for (int i = 0; i < n; i++) { const float v = xr[i] - m; a2 += (double)(v * v); }
            red[0] = (double)m; red[1] = a2;
        }
        __syncthreads();
        mean = (float)red[0]; variance = (float)(red[1] / (double)n);
    } else {
    mean = (float)(block_sum_d(s, red) / (double)n);
    double s2 = 0.0;
#pragma unroll
    for (int e = 0; e < MAXE; e++) { const int i = threadIdx.x + 256 * e; if (i < n) { const float v = xv[e] - mean; s2 += (double)(v * v); } }
    variance = (float)(block_sum_d(s2, red) / (double)n);
    }
    const float scale = 1.0f / sqrtf(variance + 1e-5f);
#pragma unroll
    for (int e = 0; e < MAXE; e++) {
        const int i = threadIdx.x + 256 * e;
        if (i < n) {
            float v = (xv[e] - mean) * scale;
            v = wv[e] * v;
            if (b) v = v + bb[e];
            if (out) out[row * n + i] = v;
            if (out_h) out_h[row * n + i] = f2h_rn(v);
        }
    }
}
void launch_layernorm(const float *x, const float *w, const float *b, int rows, int n, float *out, __half *out_h, hipStream_t s, bool sequential_sums) {
    if (n > 2048) throw HipError{hipErrorInvalidValue, "layernorm: row longer than 2048", __FILE__, __LINE__};
    const int seq = sequential_sums ? 1 : 0;
    if (n <= 768) hipLaunchKernelGGL((k_layernorm<3>), dim3((unsigned)rows), dim3(256), 0, s, x, w, b, n, out, out_h, seq);
    else if (n <= 1536) hipLaunchKernelGGL((k_layernorm<6>), dim3((unsigned)rows), dim3(256), 0, s, x, w, b, n, out, out_h, seq);
    else hipLaunchKernelGGL((k_layernorm<8>), dim3((unsigned)rows), dim3(256), 0, s, x, w, b, n, out, out_h, seq);
}
// MINIGPT4_PARITY attention of the vision tower / Q-Former: oracle/refcpu.c attention_f32 with every fp32 chain in its order -- thread = one key for the scores
// (q * prescale, sequential fma over the head dimension, / score_div), max, fp16-table exp, exact double sum, p = e * (1 / sum) in fp32 (no fp16 rounding here: ggml's
// f32 x f32 mul_mat), thread = one output dimension for P.V (sequential over the keys).  One workgroup per (head, query, image).
__global__ __launch_bounds__(256) void k_attn_vref(const float *__restrict__ q, int ldq, const float *__restrict__ k, const float *__restrict__ v, int ldk, int nq, int nk, int hd,
                                                   float q_prescale, float score_div, const Tables tb, float *__restrict__ out, int ldo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_vr[];
    float *sc = reinterpret_cast<float *>(smem_vr);               // [nk]
    float *qv = sc + ((nk + 3) & ~3);                             // [hd]
    __shared__ float red_f[4]; __shared__ double red_d[4];
    const int h = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    { const size_t z = blockIdx.z; q += z * nq * ldq; k += z * nk * ldk; v += z * nk * ldk; out += z * nq * ldo; }
    for (int i = tid; i < hd; i += 256) { float x = q[(size_t)t * ldq + (size_t)h * hd + i]; if (q_prescale != 0.0f) x *= q_prescale; qv[i] = x; }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < nk; j += 256) {
        const float *kr = k + (size_t)j * ldk + (size_t)h * hd;
        float s = 0.0f;
        for (int i = 0; i < hd; i++) s = fmaf(kr[i], qv[i], s);
        if (score_div != 0.0f) s = s / score_div;
        sc[j] = s; mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red_f[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    double sum = 0.0;
    for (int j = tid; j < nk; j += 256) { const float e = tab_v(tb.exp, sc[j] - mx); sc[j] = e; sum += (double)e; }   // fp16 values: exact in any order
    sum = wave_sum_d(sum);
    if ((tid & 63) == 0) red_d[tid >> 6] = sum;
    __syncthreads();
    const float inv = (float)(1.0 / (((red_d[0] + red_d[1]) + red_d[2]) + red_d[3]));
    for (int j = tid; j < nk; j += 256) sc[j] = sc[j] * inv;
    __syncthreads();
    for (int i = tid; i < hd; i += 256) {
        const float *vr = v + (size_t)h * hd + i;
        float s = 0.0f;
        for (int j = 0; j < nk; j++) s = fmaf(vr[(size_t)j * ldk], sc[j], s);
        out[(size_t)t * ldo + (size_t)h * hd + i] = s;
    }
}
void launch_attn_vref(const float *q, int ldq, const float *k, const float *v, int ldk, int nq, int nk, int heads, int hd, float q_prescale, float score_div, const Tables &tb,
                      float *out, int ldo, hipStream_t s, int batch) {
    const size_t lds = (size_t)((nk + 3) & ~3) * 4 + (size_t)hd * 4 + 64;
    hipLaunchKernelGGL(k_attn_vref, dim3((unsigned)heads, (unsigned)nq, (unsigned)batch), dim3(256), lds, s, q, ldq, k, v, ldk, nq, nk, hd, q_prescale, score_div, tb, out, ldo);
}

typedef float float4_t __attribute__((ext_vector_type(4)));
// the VALUE of ggml's fp16 exp table, computed: table[x] = fp16(expf(fp16(x))) (qtraits.hpp::exp_h: differs from the host libm's entry by one fp16 ulp in about 1 of 10^4 values)
__device__ __forceinline__ float exp_c16(float x) { return __half2float(f2h_rn(__expf(__half2float(f2h_rn(x))))); }
// ---------------------------------------------------------------------------------------------------------------------
// k_attn_vit -- fp32 attention without K / V staging (round 2; replaces the LDS-staged k_attn_mfma for nk <= 64 * TPW keys).
//   * workgroup = (head, 16 queries, image); its 4 waves split the KEYS (wave w owns key tiles w, w + 4, ...), so a ViT layer is 16 x 17 = 272 workgroups
//     (one per CU) whose critical path is a quarter of a head instead of the whole head;
//   * scores are computed TRANSPOSED, S^T = K . Q^T (v_mfma_f32_16x16x4_f32; A = 16 keys, B = 16 queries): the MFMA reduction index may be labelled freely as
//     long as A and B agree, so Q and K fragments come straight from global memory as 16-byte loads (per instruction 64 contiguous bytes of each of the tile's 16
//     rows), no LDS, no transposition;
//   * the C layout of S^T (lane (query, g), register r <-> key 16 kt + 4 g + r) IS the B-operand layout of O^T = V^T . P^T, so the probabilities never leave
//     their registers; V enters as the A operand (lane (dim, g) <- V[key 16 kt + 4 g + r][dim]), again straight from global memory (64-byte segments);
//   * softmax exactly as before (fp32 max, the fp16 exp table, fp64 sum -- exact in any order: <= 2^9 terms of 11-bit values --, p = e * (float)(1 / sum)),
//     but the table's live part (arguments in [-17.4, -0]: tb.exp_neg_n entries from code 0x8000) sits in LDS (LDS-DMA at kernel entry, hidden behind the
//     Q / K loads); anything outside (NaN) falls back to the global table, so the values are the same ones;
//   * the 4 waves' partial O^T tiles are added in wave order through LDS (deterministic) and stored 4 dims at a time.
// ---------------------------------------------------------------------------------------------------------------------
// Round 6: a workgroup can serve `qt` CONSECUTIVE 16-query tiles of its head, one after the other, with the K and V fragments it loaded once (MULTI; they stay in
// registers, the next tile's Q rows are requested before the current tile's softmax).  The arithmetic per query does not depend on qt (same key split over the waves, same
// exchange order), so every qt gives bit-identical rows; what changes is the traffic -- each workgroup pulls its head's K and V (180 KB at the ViT's shapes) through its
// CU's load path, which sustains only 50-70 GB/s, so a layer's 272 workgroups move 50 MB per image for 1.1 MB of distinct K / V -- and the number of workgroup rounds.
// Measured: no gain over qt = 1 at any batch size (launch_attn_f32), which therefore stays the default.
template <int HD, int TPW>
__device__ __forceinline__ void attn_vit_scores(const float (&kf)[TPW][4 * ((HD + 15) / 16)], const float (&qf)[4 * ((HD + 15) / 16)], int wave, int g, int nk, float score_div,
                                                float (&sc)[TPW][4], float &mx) {
    constexpr int KS4 = 4 * ((HD + 15) / 16);
    mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        float4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < KS4; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t][ks], qf[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float sv = acc[r];
            if (score_div != 0.0f) sv = sv / score_div;
            sv = (wave + 4 * t) * 16 + 4 * g + r < nk ? sv : -INFINITY;
            sc[t][r] = sv; mx = fmaxf(mx, sv);
        }
    }
}
template <int HD>
__device__ __forceinline__ void attn_vit_load_q(const float *__restrict__ q, int ldq, int row, size_t hoff, int g, float q_prescale, float (&qf)[4 * ((HD + 15) / 16)]) {
    constexpr int DT = (HD + 15) / 16;
    const float *qp = q + (size_t)row * ldq + hoff;
#pragma unroll
    for (int u = 0; u < DT; u++) {
        const int d0 = 16 * u + 4 * g;
        const float4 t = *reinterpret_cast<const float4 *>(qp + min(d0, HD - 4));
        const bool ok = d0 < HD;
        qf[4 * u] = ok ? t.x : 0.0f; qf[4 * u + 1] = ok ? t.y : 0.0f; qf[4 * u + 2] = ok ? t.z : 0.0f; qf[4 * u + 3] = ok ? t.w : 0.0f;
    }
    if (q_prescale != 0.0f) {
#pragma unroll
        for (int u = 0; u < 4 * DT; u++) qf[u] *= q_prescale;
    }
}
// MULTI = false is the single-tile kernel of rounds 2-5 (the loop folds away: 190 registers, two workgroups per CU); MULTI = true keeps K, V and the next tile's Q rows
// live across the loop (one workgroup per CU).  COMPUTED: the exponentials are computed instead of gathered from the table's LDS copy (fast mode, round 5); as a template
// parameter since round 6 (161 + 24 registers).  (Bounding that form to three waves per SIMD -- 168 VGPRs, 12 B of scratch -- did not pay: 3.82 vs 3.77 ms at one image,
// 7.38 vs 7.25 at four; profiles/r06_attn_query_tiles.log.)
template <int HD, int TPW, bool MULTI, bool COMPUTED>
__global__ __launch_bounds__(256) void k_attn_vit(const float *__restrict__ q, int ldq, const float *__restrict__ k, const float *__restrict__ v, int ldk, int nq, int nk,
                                                  float q_prescale, float score_div, const Tables tb, float *__restrict__ out, __half *__restrict__ out_h, int ldo, int qt, int hsq, int hsk) {
    // hsq / hsk: floats between consecutive heads of q and of k / v (HD when the heads sit side by side in a row; rows x HD in the head-major layout)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int DT = (HD + 15) / 16, KS4 = 4 * DT;
    static_assert(HD % 4 == 0, "16-byte row pieces");
    { const size_t z = blockIdx.z; q += z * nq * ldq; k += z * nk * ldk; v += z * nk * ldk; if (out) out += z * nq * ldo; if (out_h) out_h += z * nq * ldo; }
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
    float *red_m = reinterpret_cast<float *>(smem);                         // [4][16]
    double *red_s = reinterpret_cast<double *>(smem + 256);                 // [4][16]
    float *part = reinterpret_cast<float *>(smem + 768);                    // [4][DT * 4][64]
    __half *etab = reinterpret_cast<__half *>(smem + 768 + 4 * DT * 4 * 64 * 4);
    const unsigned NT = (unsigned)tb.exp_neg_n;                             // multiple of 2048
    // tb.exp == null (fast mode, round 5): the exponentials are computed -- fp16(__expf(fp16 argument)), the table's value up to its last fp16 bit -- and no table travels: the
    // 40 KB LDS-DMA per workgroup was the first thing in every wave's memory queue (~1.6 us of a CU's DMA rate, two workgroups per CU on 16 of the ViT's CUs), in front of the
    // Q / K requests.  Parity mode (k_attn_vref) and the generic path keep the table.
    constexpr bool computed = COMPUTED;                                     // (the launcher instantiates by tb.exp == nullptr)
    MG4_TLA(0);
    if (!computed) for (unsigned c0 = 0; c0 < NT; c0 += 2048)
        __builtin_amdgcn_global_load_lds((g_glb_ptr_t)(tb.exp + 0x8000 + c0 + tid * 8), (g_lds_ptr_t)(reinterpret_cast<unsigned char *>(etab) + (c0 + wave * 512) * 2), 16, 0, 0);
    // (Round 3 tried 16 x 16 = 256 workgroups for the ViT's 257 queries -- the last tile's workgroup serving the left-over query in a second softmax + P.V pass over the
    // K / V fragments it holds: 23.9 -> 20.9 us back to back, but 18.8 -> 19.9 us inside the encoder, where the 16 extra workgroups of the 272 overlap the next launch's
    // ramp.  Not kept; profiles/r03_experiments_not_adopted.log.)
    if (!MULTI) qt = 1;
    int q0 = (int)blockIdx.y * qt * 16;
    const int q_end = min(nq, q0 + qt * 16);
    // Q / K fragments: instruction u of a 16-row tile reads, per row, the 64 contiguous bytes of dims 16 u .. 16 u + 15 (lane (row, g): 16 bytes = dims 16 u + 4 g + e),
    // i.e. 16 whole sectors per wave instruction.  (First version: lane (row, g) read its own 88-byte quarter row 8 bytes at a time -- 64 different sectors per
    // instruction, and the address path, not the 90 KB of K, set the kernel's time.)  MFMA step (u, e) therefore reduces over dims {16 u + 4 g + e : g}; dims past HD
    // (HD = 88: u = 5, g >= 2) contribute zeros.
    float qf[KS4];
    const size_t hoq = (size_t)h * hsq, hok = (size_t)h * hsk;
    attn_vit_load_q<HD>(q, ldq, min(q0 + j, nq - 1), hoq, g, q_prescale, qf);
    float kf[TPW][KS4];
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const float *kp = k + (size_t)min((wave + 4 * t) * 16 + j, nk - 1) * ldk + hok;
#pragma unroll
        for (int u = 0; u < DT; u++) {
            const float4 x = *reinterpret_cast<const float4 *>(kp + min(16 * u + 4 * g, HD - 4));
            kf[t][4 * u] = x.x; kf[t][4 * u + 1] = x.y; kf[t][4 * u + 2] = x.z; kf[t][4 * u + 3] = x.w;     // past HD: finite duplicates, multiplied by the zeros in qf
        }
    }
    MG4_TLA(1);
    float sc[TPW][4];
    float mx;
    attn_vit_scores<HD, TPW>(kf, qf, wave, g, nk, score_div, sc, mx);
    MG4_TLA(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the table DMA (issued first, returned first) has landed
    MG4_TLA(3);
    // V fragments of this wave's keys: requested now, consumed after the softmax (and kept for the workgroup's later query tiles)
    float vf[TPW][4][DT];
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float *vp = v + (size_t)min((wave + 4 * t) * 16 + 4 * g + r, nk - 1) * ldk + hok;
#pragma unroll
            for (int dt = 0; dt < DT; dt++) vf[t][r][dt] = vp[min(dt * 16 + j, HD - 1)];
        }
    MG4_TLA(4);
#pragma clang loop unroll(disable)
    for (;;) {
        const bool more = MULTI && q0 + 16 < q_end;                         // workgroup-uniform
        float qn[KS4];
        if (more) attn_vit_load_q<HD>(q, ldq, min(q0 + 16 + j, nq - 1), hoq, g, q_prescale, qn);   // the next tile's Q rows travel during this tile's softmax
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (g == 0) red_m[wave * 16 + j] = mx;
        __syncthreads();
        MG4_TLA(5);
        mx = fmaxf(fmaxf(red_m[j], red_m[16 + j]), fmaxf(red_m[32 + j], red_m[48 + j]));
        double sum = 0.0;
        // all LDS gathers first (clamped index, no branch: the branchy form made 20 serial LDS round trips), then the three special cases by selects; a code outside
        // all of them (NaN, or a positive difference -- cannot happen for finite scores) takes the global table in a wave-uniform slow path
        unsigned code[TPW][4]; float el[TPW][4];
        bool odd = false;
        if constexpr (computed) {
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) { el[t][r] = exp_c16(sc[t][r] - mx); code[t][r] = 0u; }     // (-inf -> 0, 0 -> 1)
        } else {
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                code[t][r] = f2h_bits_v(sc[t][r] - mx);
                el[t][r] = __half2float(etab[min(code[t][r] ^ 0x8000u, NT - 1u)]);
            }
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned c = code[t][r], idx = c ^ 0x8000u;
                float e = idx < NT ? el[t][r] : (c == 0u ? 1.0f : 0.0f);    // +0: the row maximum itself, table[+0] = fp16(expf(0)) = 1; 0xFC00 = -inf: table[-inf] = 0
                odd = odd || (idx >= NT && c != 0u && c != 0xFC00u);
                el[t][r] = e;
            }
        }
        if (!computed && __builtin_amdgcn_ballot_w64(odd) != 0ull) {
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned c = code[t][r], idx = c ^ 0x8000u;
                    if (idx >= NT && c != 0u && c != 0xFC00u) el[t][r] = __half2float(tb.exp[c]);
                }
        }
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) { sc[t][r] = el[t][r]; sum += (double)el[t][r]; }   // (a masked key's score is -inf: code 0xFC00 above, e = 0; the maximum is finite)
        sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
        if (g == 0) red_s[wave * 16 + j] = sum;
        MG4_TLA(6);
        __syncthreads();
        MG4_TLA(7);
        sum = ((red_s[j] + red_s[16 + j]) + red_s[32 + j]) + red_s[48 + j];
        const float inv = (float)(1.0 / sum);
        float4_t o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; dt++) { o[dt][0] = 0.0f; o[dt][1] = 0.0f; o[dt][2] = 0.0f; o[dt][3] = 0.0f; }
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float p = sc[t][r] * inv;
#pragma unroll
                for (int dt = 0; dt < DT; dt++) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[t][r][dt], p, o[dt], 0, 0, 0);
            }
        MG4_TLA(8);
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int r = 0; r < 4; r++) part[(wave * DT * 4 + dt * 4 + r) * 64 + lane] = o[dt][r];
        __syncthreads();
        MG4_TLA(9);
        for (int e = tid; e < DT * 64; e += 256) {
            const int dt = e >> 6, ln = e & 63, dim0 = dt * 16 + 4 * (ln >> 4), qrow = q0 + (ln & 15);
            float s4[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o1 = (dt * 4 + r) * 64 + ln;
                s4[r] = ((part[o1] + part[DT * 4 * 64 + o1]) + part[2 * DT * 4 * 64 + o1]) + part[3 * DT * 4 * 64 + o1];
            }
            if (qrow < nq && dim0 < HD) {
                const size_t oo = (size_t)qrow * ldo + h * HD + dim0;
                if (out) *reinterpret_cast<float4 *>(out + oo) = make_float4(s4[0], s4[1], s4[2], s4[3]);
                if (out_h) { __half2 a = __halves2half2(f2h_rn(s4[0]), f2h_rn(s4[1])), b = __halves2half2(f2h_rn(s4[2]), f2h_rn(s4[3])); uint2 w; w.x = *reinterpret_cast<unsigned *>(&a); w.y = *reinterpret_cast<unsigned *>(&b); *reinterpret_cast<uint2 *>(out_h + oo) = w; }
            }
        }
        if (!more) break;
        // next query tile of this workgroup: its scores against the K fragments already in registers.  (The exchange buffers are safe to reuse: every wave passes the
        // barriers above in order, and the reads of `part` precede the next tile's first barrier.)
        q0 += 16;
#pragma unroll
        for (int u = 0; u < KS4; u++) qf[u] = qn[u];
        attn_vit_scores<HD, TPW>(kf, qf, wave, g, nk, score_div, sc, mx);
    }
#ifdef MG4_TIMELINE
    MG4_TLA(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG4_TLA(11);
#endif
}

// ViT / BERT attention (fp32 scores and outputs on the exact-f32 matrix cores, k_attn_vit).  nk <= 320 keys (the reference's graphs have 257 or 32).
static int g_attn_qt = 0;                          // forced query tiles per workgroup (0 = choose); experiments / A-B only (set_attn_vit_qt)
void set_attn_vit_qt(int qt) { g_attn_qt = qt < 0 ? 0 : qt; }
void launch_attn_f32(const float *q, int ldq, const float *k, const float *v, int ldk, int nq, int nk, int heads, int hd, float q_prescale, float score_div,
                     const Tables &tb, float *out, __half *out_h, int ldo, hipStream_t s, int batch, int hsq, int hsk) {
    if (hsq <= 0) hsq = hd;
    if (hsk <= 0) hsk = hd;
    static bool attr_set = false;
    if (!attr_set) {   // > 64 KiB of dynamic LDS for the table forms (gfx950 has 160 KiB per CU)
        HIP_IGNORE(lds_optin_max(&k_attn_vit<88, 5, false, false>)); HIP_IGNORE(lds_optin_max(&k_attn_vit<88, 5, true, false>));
        HIP_IGNORE(lds_optin_max(&k_attn_vit<64, 5, false, false>)); HIP_IGNORE(lds_optin_max(&k_attn_vit<64, 5, true, false>));
        HIP_IGNORE(lds_optin_max(&k_attn_vit<64, 1, false, false>));
        attr_set = true;
    }
    if (hd != 88 && hd != 64) throw HipError{hipErrorInvalidValue, "attn_f32: head size must be 88 or 64", __FILE__, __LINE__};
    if (nk > 320 || (tb.exp && (tb.exp_neg_n <= 0 || tb.exp_neg_n % 2048))) throw HipError{hipErrorInvalidValue, "attn_f32: more than 320 keys, or the exp table's LDS part is not set up", __FILE__, __LINE__};
    const size_t lds_v = 768 + (size_t)4 * ((hd + 15) / 16) * 4 * 64 * 4 + (tb.exp ? (size_t)tb.exp_neg_n * 2 : 0);
    // Query tiles per workgroup: 1.  The looping form (qt > 1: k_attn_vit's header) was built to put several images' workgroups into one round (4 images: 1088 -> 256) and
    // measured against the single-tile kernel inside the encoder (profiles/r06_attn_query_tiles.log): 2 images 5.51 vs 5.41 ms, 4 images 7.65 vs 7.70, 8 images 13.05 vs
    // 12.81, one image 4.03 (qt 2) vs 3.91 -- no gain: the K / V / next-Q registers it keeps live (256 + 84 AGPRs, one workgroup per CU) cost what the saved rounds bring.
    // What DID pay in that experiment: the exponential mode as a template parameter (190 -> 161 registers on the single-tile kernel: 4.00 -> 3.91 ms, 8.47 -> 8.05 with
    // the fc2 ring tiles).  MINIGPT4_ATTN_QT forces a tile count (A/B, bit-identical: tests/test_gpu_parity.py).
    const int tiles = (nq + 15) / 16;
    int qt = 1;
    if (g_attn_qt > 0) qt = std::min(g_attn_qt, tiles);
    if (hd == 64 && nk <= 64) qt = 1;                                       // the Q-Former's self-attention (32 x 32): the short-key instantiation has no looping form
    const bool comp = tb.exp == nullptr;
    dim3 grid((unsigned)heads, (unsigned)((tiles + qt - 1) / qt), (unsigned)batch);
#define MG4_ATTN_LAUNCH(HD_, TPW_, MULTI_, COMP_) hipLaunchKernelGGL((k_attn_vit<HD_, TPW_, MULTI_, COMP_>), grid, dim3(256), lds_v, s, q, ldq, k, v, ldk, nq, nk, q_prescale, score_div, tb, out, out_h, ldo, qt, hsq, hsk)
    if (hd == 88) { if (qt > 1) { if (comp) MG4_ATTN_LAUNCH(88, 5, true, true); else MG4_ATTN_LAUNCH(88, 5, true, false); } else { if (comp) MG4_ATTN_LAUNCH(88, 5, false, true); else MG4_ATTN_LAUNCH(88, 5, false, false); } }
    else if (nk <= 64) { if (comp) MG4_ATTN_LAUNCH(64, 1, false, true); else MG4_ATTN_LAUNCH(64, 1, false, false); }
    else { if (qt > 1) { if (comp) MG4_ATTN_LAUNCH(64, 5, true, true); else MG4_ATTN_LAUNCH(64, 5, true, false); } else { if (comp) MG4_ATTN_LAUNCH(64, 5, false, true); else MG4_ATTN_LAUNCH(64, 5, false, false); } }
#undef MG4_ATTN_LAUNCH
}

// =====================================================================================================================
// generic Linear epilogue (vision files whose Linear weights are not F16: the product W.x comes from the LLM mat-mul kernels as plain fp32):
// v = bias + y ; optional fp16-table GELU ; optional residual + v  -- the same order as the fused GEMM epilogue above (NNLinear: repeat(bias) + mul_mat).
// =====================================================================================================================
__global__ __launch_bounds__(256) void k_lin_epilogue(const float *__restrict__ y, const float *__restrict__ bias, const float *residual, const Tables tb, int gelu, size_t total, int n,
                                                      float *out, __half *__restrict__ out_h) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float v = y[i];
    if (bias) v = bias[i % (size_t)n] + v;
    if (gelu) v = tab_v(tb.gelu, v);
    if (residual) v = residual[i] + v;
    if (out) out[i] = v;
    if (out_h) out_h[i] = f2h_rn(v);
}
void launch_lin_epilogue(const float *y, const float *bias, const float *residual, bool gelu, const Tables &tb, int rows, int n, float *out, __half *out_h, hipStream_t s) {
    const size_t total = (size_t)rows * n;
    hipLaunchKernelGGL(k_lin_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, y, bias, residual, tb, gelu ? 1 : 0, total, n, out, out_h);
}

// =====================================================================================================================
// small data-movement kernels
// =====================================================================================================================
__global__ void k_im2col(const float *__restrict__ img, __half *__restrict__ patches, int ldp) {
    const int p = blockIdx.x, oh = p >> 4, ow = p & 15;
    img += (size_t)blockIdx.y * 3 * 224 * 224; patches += (size_t)blockIdx.y * 256 * ldp;   // image y of a batch
    for (int kk = threadIdx.x; kk < ldp; kk += blockDim.x) {
        float v = 0.0f;
        if (kk < 588) { const int c = kk / 196, r = kk - c * 196, kh = r / 14, kw = r - kh * 14; v = img[(size_t)c * 224 * 224 + (size_t)(oh * 14 + kh) * 224 + ow * 14 + kw]; }
        patches[(size_t)p * ldp + kk] = f2h_rn(v);
    }
}
void launch_im2col(const float *image, __half *patches, int ldp, hipStream_t s, int batch) { hipLaunchKernelGGL(k_im2col, dim3(256, (unsigned)batch), dim3(256), 0, s, image, patches, ldp); }

__global__ void k_assemble(const float *__restrict__ cls, const float *__restrict__ pe, const float *__restrict__ pos, int D, float *__restrict__ x) {
    const int r = blockIdx.x;
    pe += (size_t)blockIdx.y * 256 * D; x += (size_t)blockIdx.y * 257 * D;   // image y of a batch
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float base = r == 0 ? (0.0f + cls[i]) : (0.0f + pe[(size_t)(r - 1) * D + i]);
        x[(size_t)r * D + i] = base + pos[(size_t)r * D + i];
    }
}
void launch_assemble_embeddings(const float *cls, const float *pe, const float *pos, int D, float *x, hipStream_t s, int batch) {
    hipLaunchKernelGGL(k_assemble, dim3(257, (unsigned)batch), dim3(256), 0, s, cls, pe, pos, D, x);
}
__global__ void k_f32_to_f16(const float *__restrict__ x, __half *__restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = f2h_rn(x[i]);
}
void launch_f32_to_f16(const float *x, __half *y, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_f32_to_f16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n); }

}  // namespace mg4
