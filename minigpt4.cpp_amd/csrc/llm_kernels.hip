// gfx950 kernels for the Vicuna (LLaMA) path: fused-dequant integer-dot mat-vec / mat-mul over repacked weight planes,
// RMSNorm + activation quantisation, RoPE + KV append, causal attention over the fp16 KV cache, argmax, embedding gather.
//
// Arithmetic follows the reference's CPU path (ggml @ llama.cpp master-31cfbb1 behind llama_eval, reference
// minigpt4.cpp:2373/2412): activations are quantised to the weight type's vec_dot_type (Q8_0 / Q8_1 / Q8_K), block dots are
// exact int32 (v_dot4_i32_i8), block results are scaled and accumulated in fp32.  See DESIGN.md "Numerics".
#include "kernels.hpp"
#include "devutil.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <tuple>
#include <vector>
#include <hip/hip_ext.h>

namespace mg4 {

// =====================================================================================================================
// helpers
// =====================================================================================================================
// ---- measurement: kernel symbols and per-dispatch timestamps of the launches (Engine::profile_sites) ----------------------------------------------------------------
// While tracing is on, every launcher of this file notes the symbol it is about to launch, and the launch itself goes through hipExtLaunchKernel with a start / stop
// event pair: the runtime stamps those with the dispatch's own begin / end times -- the interval rocprofv3 --kernel-trace reports -- instead of bracketing the launch
// with marker events (measured +2.0...2.9 us of packet processing per pair, profiles/r02_bench_n1.json vs r02_decode_kernel_stats.csv).
static bool g_kname_on = false;
static char g_kname[192] = "";
struct LaunchProbe { hipEvent_t start, stop; };
static std::vector<LaunchProbe> g_probes;      // one per noted launch since tracing was switched on
static size_t g_probe_next = 0;                // first probe not yet attached to a launch
void kernel_name_tracing(bool on) {
    g_kname_on = on; g_kname[0] = 0;
    if (on) { for (auto &p : g_probes) { HIP_IGNORE(hipEventDestroy(p.start)); HIP_IGNORE(hipEventDestroy(p.stop)); } g_probes.clear(); g_probe_next = 0; }
}
void reset_kernel_name() { g_kname[0] = 0; }
const char *last_kernel_name() { return g_kname; }
size_t launch_probe_count() { return g_probes.size(); }
float launch_probe_us(size_t first, size_t last) {   // sum of the dispatch durations of probes [first, last); call after the stream has been synchronised
    float us = 0.0f;
    for (size_t i = first; i < last && i < g_probes.size(); i++) { float ms = 0.0f; if (hipEventElapsedTime(&ms, g_probes[i].start, g_probes[i].stop) != hipSuccess) return -1.0f; us += ms * 1e3f; }
    return us;
}
static void note_kernel(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static void note_kernel(const char *fmt, ...) {
    if (!g_kname_on) return;
    char one[128]; va_list ap; va_start(ap, fmt); vsnprintf(one, sizeof(one), fmt, ap); va_end(ap);
    const size_t have = strlen(g_kname);
    if (have && have + 3 < sizeof(g_kname)) strcat(g_kname, " + ");
    strncat(g_kname, one, sizeof(g_kname) - strlen(g_kname) - 1);
    LaunchProbe p{};
    if (hipEventCreate(&p.start) == hipSuccess && hipEventCreate(&p.stop) == hipSuccess) g_probes.push_back(p);
}
// every kernel launch of this file: plain <<< >>> unless a noted launch is waiting for its probe
template <typename... KA, typename... Args>
static inline void launch_k(void (*k)(KA...), dim3 g, dim3 b, size_t lds, hipStream_t s, Args... args) {
    static_assert(sizeof...(KA) == sizeof...(Args), "kernel argument count");
    if (g_kname_on && g_probe_next < g_probes.size()) {
        LaunchProbe &p = g_probes[g_probe_next++];
        std::tuple<KA...> tup{static_cast<KA>(args)...};
        std::apply([&](auto &...e) { void *ptrs[] = {static_cast<void *>(&e)...}; HIP_IGNORE(hipExtLaunchKernel(reinterpret_cast<const void *>(k), g, b, ptrs, lds, s, p.start, p.stop, 0)); }, tup);
        return;
    }
    k<<<g, b, lds, s>>>(args...);
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(k, g, b, l, s, ...) launch_k(k, g, b, l, s, __VA_ARGS__)

}  // namespace mg4
#include "qtraits.hpp"
namespace mg4 {

// =====================================================================================================================
// load-time repack: ggml array-of-blocks -> planes.  One thread per unit (16 bytes of the main plane).
// =====================================================================================================================
size_t plan_qweight(int type, int rows, int cols, QWeight &w, uint8_t *base) {
    w = QWeight{};
    w.type = type; w.rows = rows; w.cols = cols;
    const size_t n = (size_t)rows * cols;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { uint8_t *p = base ? base + off : nullptr; off += al(bytes); return p; };
    switch (type) {
    case GT_F32: w.qs = take(n * 4); break;
    case GT_F16: w.qs = take(n * 2); break;
    case GT_Q4_0: w.qs = take(n / 32 * 16); w.sc = take(n / 32 * 2); break;
    case GT_Q4_1: w.qs = take(n / 32 * 16); w.sc = take(n / 32 * 4); break;
    case GT_Q5_0: w.qs = take(n / 32 * 16); w.qh = take(n / 32 * 4); w.sc = take(n / 32 * 2); break;
    case GT_Q5_1: w.qs = take(n / 32 * 16); w.qh = take(n / 32 * 4); w.sc = take(n / 32 * 4); break;
    case GT_Q8_0: w.qs = take(n / 32 * 32); w.sc = take(n / 32 * 2); break;
    case GT_Q2_K: w.qs = take(n / 32 * 8); w.sc = take(n / 32 * 2); w.d = take(n / 256 * 4); break;   // unit = 32 consecutive weights = two 16-weight sub-blocks: 2-bit planes, {scale | min << 4} x 2, {d, dmin} per super-block
    case GT_Q4_K: w.qs = take(n / 256 * 128); w.sc = take(n / 256 * 16); break;
    case GT_Q5_K: w.qs = take(n / 256 * 128); w.qh = take(n / 256 * 32); w.sc = take(n / 256 * 16); break;
    case GT_Q6_K: w.qs = take(n / 256 * 128); w.qh = take(n / 256 * 64); w.sc = take(n / 256 * 16); w.d = take(n / 256 * 2); break;
    default: return 0;
    }
    w.bytes = gt_nbytes(type, n);
    return off;
}
bool qweight_supported(int type) {
    switch (type) { case GT_F32: case GT_F16: case GT_Q4_0: case GT_Q4_1: case GT_Q5_0: case GT_Q5_1: case GT_Q8_0: case GT_Q2_K: case GT_Q4_K: case GT_Q5_K: case GT_Q6_K: return true; default: return false; }
}

// high-bit transposition for 5-bit types: element e of 32 (lo: e<16 -> dword k=e/4, byte i=e%4, slot w=k; hi: e>=16 -> w=4+k)
__device__ __forceinline__ unsigned pack_hb1(const unsigned char hb_lo[16], const unsigned char hb_hi[16]) {
    unsigned P = 0;
#pragma unroll
    for (int e = 0; e < 16; e++) { const int k = e >> 2, i = e & 3; P |= (unsigned)(hb_lo[e] & 1) << (8 * i + k); P |= (unsigned)(hb_hi[e] & 1) << (8 * i + 4 + k); }
    return P;
}

__global__ void k_repack(const uint8_t *__restrict__ raw, QWeight w, size_t n_units) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_units) return;
    uint8_t *qs = const_cast<uint8_t *>(w.qs), *qh = const_cast<uint8_t *>(w.qh), *sc = const_cast<uint8_t *>(w.sc), *dd = const_cast<uint8_t *>(w.d);
    switch (w.type) {
    case GT_Q4_0: { const uint8_t *b = raw + g * 18; sc[g * 2] = b[0]; sc[g * 2 + 1] = b[1]; for (int i = 0; i < 16; i++) qs[g * 16 + i] = b[2 + i]; break; }
    case GT_Q4_1: { const uint8_t *b = raw + g * 20; for (int i = 0; i < 4; i++) sc[g * 4 + i] = b[i]; for (int i = 0; i < 16; i++) qs[g * 16 + i] = b[4 + i]; break; }
    case GT_Q5_0: case GT_Q5_1: {
        const int hdr = w.type == GT_Q5_0 ? 2 : 4; const uint8_t *b = raw + g * (size_t)(hdr + 20);
        for (int i = 0; i < hdr; i++) sc[g * hdr + i] = b[i];
        const unsigned q = (unsigned)b[hdr] | ((unsigned)b[hdr + 1] << 8) | ((unsigned)b[hdr + 2] << 16) | ((unsigned)b[hdr + 3] << 24);
        unsigned char lo[16], hi[16];
        for (int e = 0; e < 16; e++) { lo[e] = (q >> e) & 1; hi[e] = (q >> (16 + e)) & 1; }
        *reinterpret_cast<unsigned *>(qh + g * 4) = pack_hb1(lo, hi);
        for (int i = 0; i < 16; i++) qs[g * 16 + i] = b[hdr + 4 + i];
        break; }
    case GT_Q8_0: { const size_t blk = g >> 1; const int half = (int)(g & 1); const uint8_t *b = raw + blk * 34;
        if (!half) { sc[blk * 2] = b[0]; sc[blk * 2 + 1] = b[1]; }
        for (int i = 0; i < 16; i++) qs[g * 16 + i] = b[2 + half * 16 + i];
        break; }
    case GT_Q2_K: {   // block_q2_K = scales[16] | qs[64] | d | dmin; element 128 n + 32 j + l = (qs[32 n + l] >> 2 j) & 3; unit i of a super-block = elements 32 i .. 32 i + 31
        const size_t sb = g >> 3; const int i = (int)(g & 7), n = i >> 2, j = i & 3; const uint8_t *b = raw + sb * 84;
        if (i == 0) for (int k = 0; k < 4; k++) dd[sb * 4 + k] = b[80 + k];
        unsigned P0 = 0, P1 = 0;   // weight e (0..15) of a half at bits 8 (e & 3) + 2 (e >> 2): (P >> 2 k) & 0x03030303 = the four weights of dword k
        for (int e = 0; e < 16; e++) { const int k = e >> 2, c = e & 3;
            P0 |= (((unsigned)b[16 + 32 * n + e] >> (2 * j)) & 3u) << (8 * c + 2 * k); P1 |= (((unsigned)b[16 + 32 * n + 16 + e] >> (2 * j)) & 3u) << (8 * c + 2 * k); }
        reinterpret_cast<unsigned *>(qs + g * 8)[0] = P0; reinterpret_cast<unsigned *>(qs + g * 8)[1] = P1;
        sc[g * 2] = b[2 * i]; sc[g * 2 + 1] = b[2 * i + 1];
        break; }
    case GT_Q4_K: { const size_t sb = g >> 3; const int u = (int)(g & 7); const uint8_t *b = raw + sb * 144;
        if (u == 0) for (int i = 0; i < 16; i++) sc[sb * 16 + i] = b[i];
        for (int i = 0; i < 16; i++) qs[g * 16 + i] = b[16 + u * 16 + i];
        break; }
    case GT_Q5_K: { const size_t sb = g >> 3; const int u = (int)(g & 7), j = u >> 1, h = u & 1; const uint8_t *b = raw + sb * 176;
        if (u == 0) for (int i = 0; i < 16; i++) sc[sb * 16 + i] = b[i];
        unsigned char lo[16], hi[16];
        for (int e = 0; e < 16; e++) { const unsigned char v = b[16 + 16 * h + e]; lo[e] = (v >> (2 * j)) & 1; hi[e] = (v >> (2 * j + 1)) & 1; }
        *reinterpret_cast<unsigned *>(qh + g * 4) = pack_hb1(lo, hi);
        for (int i = 0; i < 16; i++) qs[g * 16 + i] = b[48 + u * 16 + i];
        break; }
    case GT_Q6_K: { const size_t sb = g >> 3; const int u = (int)(g & 7), n = u >> 2, c = (u >> 1) & 1, h = u & 1; const uint8_t *b = raw + sb * 210;
        if (u == 0) { dd[sb * 2] = b[208]; dd[sb * 2 + 1] = b[209]; }
        unsigned Plo = 0, Phi = 0;
        for (int e = 0; e < 16; e++) { const unsigned v = b[128 + 32 * n + 16 * h + e]; const int k = e >> 2, i = e & 3;
            Plo |= ((v >> (2 * c)) & 3u) << (8 * i + 2 * k); Phi |= ((v >> (4 + 2 * c)) & 3u) << (8 * i + 2 * k); }
        reinterpret_cast<unsigned *>(qh + g * 8)[0] = Plo; reinterpret_cast<unsigned *>(qh + g * 8)[1] = Phi;
        sc[g * 2] = b[192 + 8 * n + 2 * c + h]; sc[g * 2 + 1] = b[192 + 8 * n + 4 + 2 * c + h];
        for (int i = 0; i < 16; i++) qs[g * 16 + i] = b[u * 16 + i];
        break; }
    case GT_F16: case GT_F32: { for (int i = 0; i < 16; i++) qs[g * 16 + i] = raw[g * 16 + i]; break; }
    default: break;
    }
}
void launch_repack(const uint8_t *raw, const QWeight &w, hipStream_t s) {
    const size_t n = (size_t)w.rows * w.cols;
    size_t units;
    switch (w.type) { case GT_F32: units = n / 4; break; case GT_F16: units = n / 8; break; case GT_Q8_0: units = n / 16; break; default: units = n / 32; }
    const int bs = 256;
    hipLaunchKernelGGL(k_repack, dim3((unsigned)((units + bs - 1) / bs)), dim3(bs), 0, s, raw, w, units);
}


// =====================================================================================================================
// y[t][r] = W[r] . act[t] (+ residual).  One wave owns R consecutive rows; its 64 lanes stride over the row's units
// (16-byte coalesced loads, 1 KiB per wave-instruction); each lane keeps the activation unit in registers and reuses it
// for the R rows and TN tokens.  Wave-level xor-shuffle reduction at the end; no LDS.
// =====================================================================================================================
template <int T, int R, int TN>
__global__ __launch_bounds__(256) void k_mul_mat(const QWeight W, const ActQ A, const int N, float *y, const int ldy, const float *residual) {
    using X = Tr<T>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * R;
    const int t0 = blockIdx.y * TN;
    if (row0 >= W.rows) return;          // wave-uniform
    const int K = W.cols, U = K / X::EPU;
    float acc[R][TN];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int t = 0; t < TN; t++) acc[r][t] = 0.0f;
    int rows_i[R];
#pragma unroll
    for (int r = 0; r < R; r++) rows_i[r] = min(row0 + r, W.rows - 1);
    for (int u0 = 0; u0 < U; u0 += 64) {
        const int u = u0 + lane;
        const bool ok = u < U;
        const int uc = ok ? u : 0;       // out-of-range lanes fetch unit 0 (valid address), their result is discarded
        typename X::AU a[TN];
#pragma unroll
        for (int t = 0; t < TN; t++) X::loada(A, min(t0 + t, N - 1), K, uc, a[t]);
        typename X::WU w[R];
#pragma unroll
        for (int r = 0; r < R; r++) X::loadw(W, (size_t)rows_i[r] * U, uc, w[r]);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int t = 0; t < TN; t++) { float c = acc[r][t]; X::dot(w[r], a[t], c); acc[r][t] = ok ? c : acc[r][t]; }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int t = 0; t < TN; t++) {
            const float v = wave_sum(acc[r][t]);
            if (lane == 0 && row0 + r < W.rows && t0 + t < N) {
                const size_t o = (size_t)(t0 + t) * ldy + row0 + r;
                y[o] = residual ? v + residual[o] : v;
            }
        }
}

template <int T>
static void launch_mul_mat_t(const QWeight &W, const ActQ &A, int N, float *y, int ldy, const float *residual, hipStream_t s) {
    note_kernel("k_mul_mat<%d, 2, %d>", T, N == 1 ? 1 : 4);
    if (N == 1) {
        constexpr int R = 2;
        dim3 grid((unsigned)((W.rows + 4 * R - 1) / (4 * R)), 1);
        hipLaunchKernelGGL((k_mul_mat<T, R, 1>), grid, dim3(256), 0, s, W, A, N, y, ldy, residual);
    } else {
        constexpr int R = 2, TN = 4;
        dim3 grid((unsigned)((W.rows + 4 * R - 1) / (4 * R)), (unsigned)((N + TN - 1) / TN));
        hipLaunchKernelGGL((k_mul_mat<T, R, TN>), grid, dim3(256), 0, s, W, A, N, y, ldy, residual);
    }
}
// =====================================================================================================================
// MINIGPT4_PARITY=1 -- the oracle's fp32 ORDER (oracle/refcpu.c vec_dot_*: one sequential chain of fma's per output, block after block).  One wave per
// (row, token): the lanes split the row's units exactly like k_mul_mat and use the same unit traits (same loads, same integer dot products), but the integer parts are
// first combined per ggml block (8 units per k-quant super-block, 2 per Q8_0 block: exact), and the per-block fp32 terms are then added by every lane in block order.
// The fast kernels differ from this one only in the order of those additions; this one is bit-identical to the CPU oracle.  F16 / F32: one fma per ELEMENT in
// element order (ggml_vec_dot_f16 restated as a scalar loop), unit after unit.
// =====================================================================================================================
template <int T> __device__ __forceinline__ float ref_chunk(const typename Tr<T>::WU &w, const typename Tr<T>::AU &a, const bool ok, const int n_here, float acc);   // below
template <int T>
__global__ __launch_bounds__(256) void k_mul_mat_ref(const QWeight W, const ActQ A, const int N, float *y, const int ldy, const float *residual) {
    using X = Tr<T>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave, t = blockIdx.y;
    if (row >= W.rows || t >= N) return;          // wave-uniform
    const int K = W.cols, U = K / X::EPU;
    float acc = 0.0f;
    for (int u0 = 0; u0 < U; u0 += 64) {
        const int u = u0 + lane;
        const bool ok = u < U;
        const int uc = ok ? u : 0;
        const int n_here = min(64, U - u0);
        typename X::AU a; X::loada(A, t, K, uc, a);
        typename X::WU w; X::loadw(W, (size_t)row * U, uc, w);
        if constexpr (T == GT_F16 || T == GT_F32) {
            for (int g = 0; g < n_here; g++) { float c = acc; X::dot(w, a, c); acc = __shfl(c, g); }   // unit g continues the chain where unit g - 1 left it
        } else {
            acc = ref_chunk<T>(w, a, ok, n_here, acc);
        }
    }
    if (lane == 0) { const size_t o = (size_t)t * ldy + row; y[o] = residual ? acc + residual[o] : acc; }
}
// One chunk of <= 64 units (n_here of them; lane = unit) of one output in the oracle's order: the running value `acc` (wave-uniform) comes in, the value after the chunk's
// last block goes out.
template <int T>
__device__ __forceinline__ float ref_chunk(const typename Tr<T>::WU &w, const typename Tr<T>::AU &a, const bool ok, const int n_here, float acc) {
    using X = Tr<T>;
    int i0, i1; X::ints(w, a, i0, i1);
    if (!ok) { i0 = 0; i1 = 0; }
    // the block's integer parts over its GROUP lanes on the DPP crossbar (exact integer sums: any combining order gives the same value; __shfl_xor is an LDS permute)
    if (X::GROUP >= 2) { i0 += dpp_i<0xB1>(i0); i1 += dpp_i<0xB1>(i1); }
    if (X::GROUP >= 4) { i0 += dpp_i<0x4E>(i0); i1 += dpp_i<0x4E>(i1); }
    if (X::GROUP >= 8) { i0 += dpp_i<0x141>(i0); i1 += dpp_i<0x141>(i1); }
    static_assert(X::GROUP <= 8, "one DPP row half per ggml block");
    float f0, v0, f1, v1; X::terms(w, a, i0, i1, f0, v0, f1, v1);
    if constexpr (X::GROUP == 8) {
        // k-quants: the chain walks the 8-lane groups of this register in place.  At step g every lane takes the running value of the lane 8 below it (row_shr:8:
        // the lower half of its 16-lane row) or, where a row begins, of the previous row's last lane (row_bcast:15), and adds ITS block's terms: after step g the
        // lanes of group g hold the oracle's running sum, the other lanes hold values nobody reads.  3 instructions per block instead of 4 v_readlane + 2 fma.
        const int ng = n_here / 8;
        float run = acc;
#pragma unroll
        for (int g = 0; g < 8; g++) {
            if (g < ng) {
                float t = g == 0 ? acc : ((g & 1) ? dpp_f<0x118>(run) : dpp_f<0x142>(run));
                t = fmaf(f0, v0, t);
                if (X::TERMS == 2) t = fmaf(f1, v1, t);
                run = t;
            }
        }
        acc = readlane_f(run, 8 * (ng - 1));
    } else {
        for (int g = 0; g < n_here; g += X::GROUP) {   // g is wave-uniform: v_readlane (the generic __shfl is an LDS permute, ~100 dependent cycles per block term)
            acc = fmaf(readlane_f(f0, g), readlane_f(v0, g), acc);
            if (X::TERMS == 2) acc = fmaf(readlane_f(f1, g), readlane_f(v1, g), acc);
        }
    }
    return acc;
}
// One output from the lanes' NU registers of units (u = lane + 64 i): chunk after chunk.  Returns the wave-uniform result.
template <int T, int NU>
__device__ __forceinline__ float ref_chain(const typename Tr<T>::WU (&w)[NU], const typename Tr<T>::AU (&a)[NU], const bool (&ok)[NU], const int U) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < NU; i++) {
        const int n_here = min(64, U - 64 * i);
        if (n_here <= 0) break;
        acc = ref_chunk<T>(w[i], a[i], ok[i], n_here, acc);
    }
    return acc;
}
// The same chain for ONE activation row (decode) over up to three matrices of one type and K in one launch: every lane's NU units (u = lane + 64 i, as in k_matvec_v2) are
// requested before the first block is added, so a wave pays one memory round trip instead of one per 64 units (the generic kernel above: 3 ... 7 dependent round trips per
// row, 5.5 ms per 13B token in parity mode).  The per-block terms are added in the same order: chunk after chunk, block after block -- bit-identical to k_mul_mat_ref.
struct RefSet { QWeight w[3]; float *y[3]; const float *res[3]; int n; };
template <int T, int NU>
__global__ __launch_bounds__(256) void k_mul_mat_ref_row(const RefSet S, const ActQ A) {
    using X = Tr<T>;
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.y;
    const QWeight &W = S.w[m];
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))), n_waves = (int)gridDim.x * 4;
    const int K = W.cols, U = K / X::EPU, rows = W.rows;
    if (wave >= rows) return;                     // wave-uniform
    typename X::AU a[NU]; bool ok[NU]; int uc[NU];
#pragma unroll
    for (int i = 0; i < NU; i++) { const int u = lane + 64 * i; ok[i] = u < U; uc[i] = ok[i] ? u : 0; X::loada(A, 0, K, uc[i], a[i]); }
    struct Row { typename X::WU w[NU]; };
    auto fetch = [&](int row, Row &Rw) {           // clamped instead of branching: every load of the pipeline is unconditional (counted waits)
        const int r = min(row, rows - 1);
#pragma unroll
        for (int i = 0; i < NU; i++) X::loadw(W, (size_t)r * U, uc[i], Rw.w[i]);
    };
    auto chain = [&](int row, const Row &Rw) {     // the oracle's order: chunk after chunk, block after block
        const float acc = ref_chain<T, NU>(Rw.w, a, ok, U);
        if (lane == 0) { const float *r = S.res[m]; S.y[m][row] = r ? acc + r[row] : acc; }
    };
    Row cur, nxt;                                  // two statically named stages, as in matvec_run: the next row is in flight while this one is chained
    fetch(wave, cur);
    for (int row = wave; row < rows;) {
        fetch(row + n_waves, nxt);
        __builtin_amdgcn_sched_barrier(0);
        chain(row, cur);
        __builtin_amdgcn_sched_barrier(0);
        row += n_waves;
        if (row >= rows) break;
        fetch(row + n_waves, cur);
        __builtin_amdgcn_sched_barrier(0);
        chain(row, nxt);
        __builtin_amdgcn_sched_barrier(0);
        row += n_waves;
    }
}
static int g_ref_cus = 256;
template <int T> static bool launch_mul_mat_ref_row_t(const RefSet &S, const ActQ &A, hipStream_t s) {
    const int U = S.w[0].cols / Tr<T>::EPU, nu = (U + 63) / 64;
    // persistent waves: ~16 per CU over the whole set, each walking rows wave, wave + n_waves, ...
    const int blocks = std::max(1, std::min((S.w[0].rows + 3) / 4, g_ref_cus * 4 / S.n));
    const dim3 grid((unsigned)blocks, (unsigned)S.n);
    note_kernel("k_mul_mat_ref_row<%d, %d>", T, nu);
    switch (nu) {
    case 1: hipLaunchKernelGGL((k_mul_mat_ref_row<T, 1>), grid, dim3(256), 0, s, S, A); return true;
    case 2: hipLaunchKernelGGL((k_mul_mat_ref_row<T, 2>), grid, dim3(256), 0, s, S, A); return true;
    case 3: hipLaunchKernelGGL((k_mul_mat_ref_row<T, 3>), grid, dim3(256), 0, s, S, A); return true;
    case 4: hipLaunchKernelGGL((k_mul_mat_ref_row<T, 4>), grid, dim3(256), 0, s, S, A); return true;
    case 6: hipLaunchKernelGGL((k_mul_mat_ref_row<T, 6>), grid, dim3(256), 0, s, S, A); return true;
    case 7: hipLaunchKernelGGL((k_mul_mat_ref_row<T, 7>), grid, dim3(256), 0, s, S, A); return true;
    default: return false;
    }
}
// One activation row against 1..3 equally shaped matrices of one type (parity mode's decode step); false -> shape outside this kernel, use launch_mul_mat_ref per matrix
bool launch_mul_mat_ref_set(const QWeight *const *W, float *const *y, const float *const *res, int n, const ActQ &A, hipStream_t s) {
    if (n < 1 || n > 3) return false;
    RefSet S{}; S.n = n;
    for (int i = 0; i < n; i++) { if (W[i]->type != W[0]->type || W[i]->rows != W[0]->rows || W[i]->cols != W[0]->cols) return false; S.w[i] = *W[i]; S.y[i] = y[i]; S.res[i] = res ? res[i] : nullptr; }
    switch (W[0]->type) {
    case GT_Q4_0: return launch_mul_mat_ref_row_t<GT_Q4_0>(S, A, s);
    case GT_Q4_1: return launch_mul_mat_ref_row_t<GT_Q4_1>(S, A, s);
    case GT_Q5_0: return launch_mul_mat_ref_row_t<GT_Q5_0>(S, A, s);
    case GT_Q5_1: return launch_mul_mat_ref_row_t<GT_Q5_1>(S, A, s);
    case GT_Q4_K: return launch_mul_mat_ref_row_t<GT_Q4_K>(S, A, s);
    case GT_Q5_K: return launch_mul_mat_ref_row_t<GT_Q5_K>(S, A, s);
    case GT_Q6_K: return launch_mul_mat_ref_row_t<GT_Q6_K>(S, A, s);
    default: return false;                       // (Q8_0 / Q2_K / F16 / F32 rows have other unit widths: the generic kernel serves them)
    }
}
// Prompt rows in parity mode: one wave per WEIGHT row (persistent over rows wave, wave + n_waves, ...), the row's units in registers for all N activation rows -- the
// generic kernel above re-reads the weights once per (output, activation row).  The chain per (row, activation row) is ref_chain's: bit-identical to k_mul_mat_ref.
template <int T, int NU>
__global__ __launch_bounds__(256) void k_mul_mat_ref_rows(const QWeight W, const ActQ A, const int N, float *y, const int ldy, const float *residual) {
    using X = Tr<T>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))), n_waves = (int)gridDim.x * 4;
    const int K = W.cols, U = K / X::EPU, rows = W.rows;
    bool ok[NU]; int uc[NU];
#pragma unroll
    for (int i = 0; i < NU; i++) { const int u = lane + 64 * i; ok[i] = u < U; uc[i] = ok[i] ? u : 0; }
    for (int row = wave; row < rows; row += n_waves) {
        typename X::WU w[NU];
#pragma unroll
        for (int i = 0; i < NU; i++) X::loadw(W, (size_t)row * U, uc[i], w[i]);
        typename X::AU a0[NU], a1[NU];              // two statically named stages: the next activation row is in flight while this one is chained
#pragma unroll
        for (int i = 0; i < NU; i++) X::loada(A, 0, K, uc[i], a0[i]);
        for (int t = 0; t < N;) {
#pragma unroll
            for (int i = 0; i < NU; i++) X::loada(A, min(t + 1, N - 1), K, uc[i], a1[i]);
            __builtin_amdgcn_sched_barrier(0);
            { const float acc = ref_chain<T, NU>(w, a0, ok, U); if (lane == 0) { const size_t o = (size_t)t * ldy + row; y[o] = residual ? acc + residual[o] : acc; } }
            __builtin_amdgcn_sched_barrier(0);
            if (++t >= N) break;
#pragma unroll
            for (int i = 0; i < NU; i++) X::loada(A, min(t + 1, N - 1), K, uc[i], a0[i]);
            __builtin_amdgcn_sched_barrier(0);
            { const float acc = ref_chain<T, NU>(w, a1, ok, U); if (lane == 0) { const size_t o = (size_t)t * ldy + row; y[o] = residual ? acc + residual[o] : acc; } }
            __builtin_amdgcn_sched_barrier(0);
            ++t;
        }
    }
}
template <int T> static bool launch_mul_mat_ref_rows_t(const QWeight &W, const ActQ &A, int N, float *y, int ldy, const float *residual, hipStream_t s) {
    const int U = W.cols / Tr<T>::EPU, nu = (U + 63) / 64;
    const dim3 grid((unsigned)std::max(1, std::min((W.rows + 3) / 4, g_ref_cus * 4)));
    note_kernel("k_mul_mat_ref_rows<%d, %d>", T, nu);
    switch (nu) {
    case 1: hipLaunchKernelGGL((k_mul_mat_ref_rows<T, 1>), grid, dim3(256), 0, s, W, A, N, y, ldy, residual); return true;
    case 2: hipLaunchKernelGGL((k_mul_mat_ref_rows<T, 2>), grid, dim3(256), 0, s, W, A, N, y, ldy, residual); return true;
    case 3: hipLaunchKernelGGL((k_mul_mat_ref_rows<T, 3>), grid, dim3(256), 0, s, W, A, N, y, ldy, residual); return true;
    case 4: hipLaunchKernelGGL((k_mul_mat_ref_rows<T, 4>), grid, dim3(256), 0, s, W, A, N, y, ldy, residual); return true;
    case 6: hipLaunchKernelGGL((k_mul_mat_ref_rows<T, 6>), grid, dim3(256), 0, s, W, A, N, y, ldy, residual); return true;
    case 7: hipLaunchKernelGGL((k_mul_mat_ref_rows<T, 7>), grid, dim3(256), 0, s, W, A, N, y, ldy, residual); return true;
    default: return false;
    }
}
template <int T> static void launch_mul_mat_ref_t(const QWeight &W, const ActQ &A, int N, float *y, int ldy, const float *residual, hipStream_t s) {
    note_kernel("k_mul_mat_ref<%d>", T);
    hipLaunchKernelGGL((k_mul_mat_ref<T>), dim3((unsigned)((W.rows + 3) / 4), (unsigned)N), dim3(256), 0, s, W, A, N, y, ldy, residual);
}
void launch_mul_mat_ref(const QWeight &W, const ActQ &A, int N, float *y, int ldy, const float *residual, hipStream_t s) {
    if (N > 1) {   // prompt rows: the weight row stays in registers for all N activation rows (k-quants; the other types and unit counts take the generic kernel)
        bool done = false;
        switch (W.type) {
        case GT_Q4_K: done = launch_mul_mat_ref_rows_t<GT_Q4_K>(W, A, N, y, ldy, residual, s); break;
        case GT_Q5_K: done = launch_mul_mat_ref_rows_t<GT_Q5_K>(W, A, N, y, ldy, residual, s); break;
        case GT_Q6_K: done = launch_mul_mat_ref_rows_t<GT_Q6_K>(W, A, N, y, ldy, residual, s); break;
        default: break;
        }
        if (done) return;
    }
    switch (W.type) {
    case GT_Q4_0: launch_mul_mat_ref_t<GT_Q4_0>(W, A, N, y, ldy, residual, s); break;
    case GT_Q4_1: launch_mul_mat_ref_t<GT_Q4_1>(W, A, N, y, ldy, residual, s); break;
    case GT_Q5_0: launch_mul_mat_ref_t<GT_Q5_0>(W, A, N, y, ldy, residual, s); break;
    case GT_Q5_1: launch_mul_mat_ref_t<GT_Q5_1>(W, A, N, y, ldy, residual, s); break;
    case GT_Q8_0: launch_mul_mat_ref_t<GT_Q8_0>(W, A, N, y, ldy, residual, s); break;
    case GT_Q2_K: launch_mul_mat_ref_t<GT_Q2_K>(W, A, N, y, ldy, residual, s); break;
    case GT_Q4_K: launch_mul_mat_ref_t<GT_Q4_K>(W, A, N, y, ldy, residual, s); break;
    case GT_Q5_K: launch_mul_mat_ref_t<GT_Q5_K>(W, A, N, y, ldy, residual, s); break;
    case GT_Q6_K: launch_mul_mat_ref_t<GT_Q6_K>(W, A, N, y, ldy, residual, s); break;
    case GT_F16: launch_mul_mat_ref_t<GT_F16>(W, A, N, y, ldy, residual, s); break;
    case GT_F32: launch_mul_mat_ref_t<GT_F32>(W, A, N, y, ldy, residual, s); break;
    default: throw HipError{hipErrorInvalidValue, "unsupported weight type", __FILE__, __LINE__};
    }
}

// =====================================================================================================================
// Decode mat-vec, v2: persistent waves.  The launch covers up to 3 matrices of one type and one K (wq|wk|wv, w1|w3) as one
// concatenated row space.  A wave walks row groups g = wave, wave + n_waves, ...; each lane owns NU fixed units of a row
// (u = lane + 64 i), keeps their activation units in registers for the whole kernel, and software-pipelines the weight
// stream: the R x NU x (2-3) loads of the next row group are issued before the current group's dot products (plain global
// loads straight to VGPRs, counted vmcnt waits by the compiler; no LDS -- the weights are read once and not shared).
// =====================================================================================================================
// Activation preparation fused into the mat-vec prologue (decode): every workgroup redundantly prepares the (tiny, L2-resident) activation
// row in LDS -- rms_norm * w | identity | silu(a) * b, then ggml's Q8_K / Q8_0 quantisation -- while its first weight loads are in flight.
enum ProKind : int { PRO_NONE = 0, PRO_RMS = 1, PRO_PLAIN = 2, PRO_SILU = 3 };
struct ProArgs { const float *x; const float *w; Tables tb; };   // RMS: x, norm weight; PLAIN: x; SILU: a = x, b = w
__device__ __forceinline__ void quant_emit4(const float v[4], const bool in_range, const int idx, const size_t row, const int K, const ActQ &A, const int mask);

struct MatSet {
    QWeight w0;            // matrix 0; matrix m has every plane shifted by m * dmat bytes
    long long dmat;        // byte distance between consecutive matrices (identical for all planes)
    float *y0; long long dy;            // outputs: y_m = y0 + m * dy
    const float *res0; long long dres;  // optional residual inputs
    int n;                 // matrices in the set
    int rows_each;         // rows of each matrix
};

// Workgroup size of the prologue variants: one fat workgroup per CU (as many waves as the register budget of NU admits), so that the redundant
// row preparation is done once per CU instead of once per 4 waves.
// Prologue variants run as ONE fat workgroup per CU of 8..12 waves (512..768 threads): the row preparation is repeated per workgroup, so fewer, fatter
// workgroups repeat it less.  The kernel is compiled for the largest size its register budget admits and launched with the size whose
// rows-per-wave division is the most even (see pick_fat_threads).
constexpr int MV_FAT_MIN = 512;
template <int NU> constexpr int mv_fat_max_threads() { return NU <= 4 ? 768 : 512; }

// EPI_SILU_PAIR (w1|w3 of the feed-forward block; ms.n == 2, R == 2): group g = the row pair (w1[g], w3[g]); the wave writes
// h[g] = silu_table(w1[g] . x) * (w3[g] . x) instead of the two dot products.  The table lookup of a group is consumed one group later, so
// that it never stalls the weight stream.
enum EpiKind : int { EPI_STORE = 0, EPI_SILU_PAIR = 1, EPI_REF = 2 };   // EPI_REF (MINIGPT4_PARITY): store, every fp32 accumulation in the CPU oracle's order (ref_chain; the rms sum = the oracle's value)
// In-kernel timeline (diagnostic builds only: make EXTRA=-DMG4_TIMELINE OUT=../libminigpt4_tl.so OBJ=build_tl).  Thread 0 of every workgroup of a decode mat-vec
// stamps the 100 MHz constant clock at: 0 entry, 1 first weight tiles requested, 2 activation row ready (prologue done), 3 first row group finished, 4 last row
// group finished, 5 results stored.  The last launch wins; read with minigpt4_amd_timeline after a single-launch micro-benchmark (tools/timeline.py).
#ifdef MG4_TIMELINE
__device__ unsigned long long g_tl[1024 * 8];
#define MG4_TL(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_tl[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// slots 6 / 7: the LATEST end / entry over all waves of the workgroup (the clock only grows, so atomicMax needs no reset between launches)
#define MG4_TL_ALL(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) atomicMax(&g_tl[blockIdx.x * 8 + (i)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
// the prompt attention's own slots (workgroup = blockIdx.x + gridDim.x * blockIdx.y): 0 entry, 1 scores done, 2 softmax done, 3 P.V done, 4 stored
__device__ unsigned long long g_tla[2048 * 8];
#define MG4_TLP(i) do { const unsigned lb_ = blockIdx.x + gridDim.x * blockIdx.y; if (threadIdx.x == 0 && lb_ < 2048) g_tla[lb_ * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MG4_TLP(i) do {} while (0)
#define MG4_TL(i) do {} while (0)
#define MG4_TL_ALL(i) do {} while (0)
#endif
// (Round 5 measured "pair packing" -- two consecutive K = 5120 rows as ONE run of 320 units over 5 lane-walks instead of 2 x 3 with 17 % of the lanes masked -- and lost:
// 356 vs 374 tok/s on the 13B file, profiles/r05_experiments_not_adopted.md; removed again.)
template <int T, int NU, int R, int PRO, int EPI>
__device__ __forceinline__ void matvec_run(const MatSet &ms, const ActQ &A, const ProArgs &pa, const int n_groups, const int n_waves, const int wave) {
    static_assert(EPI != EPI_SILU_PAIR || R == 2, "the SiLU pair epilogue works on row pairs");
    // groups of this wave: g = g_first, g_first + g_step, ... < g_last
    const int g_first = wave, g_last = n_groups, g_step = n_waves;
    MG4_TL(0); MG4_TL_ALL(7);
    using X = TrMV<T>;
    const int lane = threadIdx.x & 63;
    const int K = ms.w0.cols, U = K / X::EPU, rows_each = ms.rows_each, total_rows = ms.n * rows_each;
    int uc[NU]; bool ok[NU];
#pragma unroll
    for (int i = 0; i < NU; i++) { const int u = lane + 64 * i; ok[i] = u < U; uc[i] = ok[i] ? u : 0; }
    // NOTE: every load of the pipeline is unconditional (indices are clamped instead of branching): a load inside an exec-masked branch
    // makes hipcc's counted s_waitcnt fall back to (near) vmcnt(0), which drains the prefetch -- see DESIGN.md "mat-vec pipeline".
    struct Grp { typename X::WU w[R][NU]; float res[R]; };
    const bool has_res = ms.res0 != nullptr;
    const float *res_base = has_res ? ms.res0 : ms.y0;                     // always a valid address; the value is dropped when there is no residual
    const long long res_stride = has_res ? ms.dres : ms.dy;
    auto fetch = [&](int g, Grp &G) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            // wave-uniform (scalar) row, clamped instead of branching; the matrix of the row is selected on the operands themselves (a
            // bool -> int conversion would be done on the vector ALU)
            const int row = EPI == EPI_SILU_PAIR ? min(g, rows_each - 1) + r * rows_each : min(g * R + r, total_rows - 1);
            const bool m1 = row >= rows_each, m2 = row >= 2 * rows_each;
            const int lr = row - (m2 ? 2 * rows_each : (m1 ? rows_each : 0));
            const long long d = m2 ? 2 * ms.dmat : (m1 ? ms.dmat : 0ll);
            QWeight W = ms.w0;
            W.qs += d; W.qh += d; W.sc += d; W.d += d;
            WBuf B;
            X::mkb(W, (size_t)lr * U, B);
#pragma unroll
            for (int i = 0; i < NU; i++) X::loadb(B, uc[i], G.w[r][i]);
            G.res[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mkbuf(reinterpret_cast<const uint8_t *>(res_base + (m2 ? 2 * res_stride : (m1 ? res_stride : 0ll)) + lr)), 0, 0, 0));
        }
    };
    Grp cur, nxt;
    typename X::AU a[NU];
    if (PRO == PRO_NONE) {
        fetch(g_first, cur);
        MG4_TL(1);
#pragma unroll
        for (int i = 0; i < NU; i++) X::loada(A, 0, K, uc[i], a[i]);
        MG4_TL(2);
    } else {
        // Row preparation in the prologue.  Order matters (vector-memory results return in issue order): the row (and, for SiLU, the table
        // gathers) is requested BEFORE the first weight tiles, so the preparation runs while those tiles are in flight.
        extern __shared__ __attribute__((aligned(16))) unsigned char smem_mv[];
        ActQ L;                                     // LDS image of the quantised row (only the planes this weight type reads are written)
        double *red = reinterpret_cast<double *>(smem_mv);               // [waves] partial sums (dynamic LDS: shared by both halves of k_matvec_mix)
        unsigned char *p = smem_mv + 128;
        L.q8k = reinterpret_cast<int8_t *>(p); p += (size_t)K;
        L.q80 = reinterpret_cast<int8_t *>(p); p += (size_t)K;
        L.dk = reinterpret_cast<float *>(p); p += (size_t)(K / 256 + 1) * 4;
        L.d0 = reinterpret_cast<float *>(p); p += (size_t)(K / 32) * 4;
        L.d1 = reinterpret_cast<float *>(p); p += (size_t)(K / 32) * 4;
        L.s1 = reinterpret_cast<float *>(p); p += (size_t)(K / 32) * 4;
        L.sum0 = reinterpret_cast<int *>(p); p += (size_t)(K / 32) * 4;
        L.bsk = reinterpret_cast<int16_t *>(p); p += (size_t)(K / 16) * 2;
        L.xh = reinterpret_cast<__half *>(smem_mv + 128);                // F16 weights: the fp16 row takes the place of the two int8 images (2 K bytes)
        L.xf = nullptr;
        constexpr int RND = (NU * 64 * X::EPU / 4 + MV_FAT_MIN - 1) / MV_FAT_MIN;    // K <= NU * 64 * EPU elements, 4 per thread per round, >= MV_FAT_MIN threads
        const int nthr = (int)blockDim.x;
        constexpr int mask = T == GT_F16 ? ACT_F16 : (T == GT_Q4_K || T == GT_Q5_K || T == GT_Q6_K) ? ACT_Q8K : ACT_Q80;
        float4 xv[RND], yv[RND];
        bool in[RND];
#pragma unroll
        for (int r = 0; r < RND; r++) {
            const int i = (r * nthr + (int)threadIdx.x) * 4;
            in[r] = i < K;
            const int ic = in[r] ? i : 0;
            xv[r] = *reinterpret_cast<const float4 *>(pa.x + ic);
            if (PRO != PRO_PLAIN) yv[r] = *reinterpret_cast<const float4 *>(pa.w + ic);
        }
        if (PRO == PRO_SILU) {
#pragma unroll
            for (int r = 0; r < RND; r++) { xv[r].x = tab(pa.tb.silu, xv[r].x); xv[r].y = tab(pa.tb.silu, xv[r].y); xv[r].z = tab(pa.tb.silu, xv[r].z); xv[r].w = tab(pa.tb.silu, xv[r].w); }
        }
        fetch(g_first, cur);
        MG4_TL(1);
        float scale = 1.0f;
        if (PRO == PRO_RMS) {
            double sum = 0.0;
#pragma unroll
            for (int r = 0; r < RND; r++) {
                const float4 v = xv[r];
                double q = 0.0;
                q += (double)(v.x * v.x); q += (double)(v.y * v.y); q += (double)(v.z * v.z); q += (double)(v.w * v.w);
                sum += in[r] ? q : 0.0;
            }
            sum = wave_sum_d(sum);
            if (lane == 0) red[threadIdx.x >> 6] = sum;
            __syncthreads();
            double tot = 0.0;
            for (int w = 0; w < nthr / 64; w++) tot += red[w];
            float mean;
            if (EPI == EPI_REF) {   // the oracle's mean (one double accumulator in element order) from a parallel sum: k_rms_quant's interval argument
                const double delta = (double)K * 4e-16;
                const float lo = (float)(tot * (1.0 - delta) / (double)K), hi = (float)(tot * (1.0 + delta) / (double)K);
                mean = lo;
                if (lo != hi) {     // workgroup-uniform (~1e-4 of the rows): the literal loop
                    __syncthreads();
                    if (threadIdx.x == 0) { double sq = 0.0; for (int i = 0; i < K; i++) sq += (double)(pa.x[i] * pa.x[i]); red[0] = sq; }
                    __syncthreads();
                    mean = (float)(red[0] / (double)K);
                }
            } else mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + 1e-6f);
        }
#pragma unroll
        for (int r = 0; r < RND; r++) {
            const int i = (r * nthr + (int)threadIdx.x) * 4;
            float v[4];
            if (PRO == PRO_RMS) { v[0] = (xv[r].x * scale) * yv[r].x; v[1] = (xv[r].y * scale) * yv[r].y; v[2] = (xv[r].z * scale) * yv[r].z; v[3] = (xv[r].w * scale) * yv[r].w; }
            else if (PRO == PRO_SILU) { v[0] = xv[r].x * yv[r].x; v[1] = xv[r].y * yv[r].y; v[2] = xv[r].z * yv[r].z; v[3] = xv[r].w * yv[r].w; }
            else { v[0] = xv[r].x; v[1] = xv[r].y; v[2] = xv[r].z; v[3] = xv[r].w; }
            if (!in[r]) { v[0] = 0.0f; v[1] = 0.0f; v[2] = 0.0f; v[3] = 0.0f; }
            quant_emit4(v, in[r], i, 0, K, L, mask);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NU; i++) X::loada(L, 0, K, uc[i], a[i]);
        MG4_TL(2);
    }
    unsigned short pend_t = 0; float pend_b = 0.0f; int pend_row = -1;          // EPI_SILU_PAIR: the previous group's table lookup, not yet consumed
    auto flush_pending = [&]() { if (pend_row >= 0 && lane == 0) ms.y0[pend_row] = h2f_bits(pend_t) * pend_b; };
    auto consume = [&](int g, const Grp &G) {
        float out[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if constexpr (EPI == EPI_REF) out[r] = ref_chain<T, NU>(G.w[r], a, ok, U);
            else {
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < NU; i++) { float c = acc; X::dot(G.w[r][i], a[i], c); acc = ok[i] ? c : acc; }
                out[r] = wave_sum(acc);
            }
        }
        if (EPI == EPI_SILU_PAIR) {
            flush_pending();
            pend_t = reinterpret_cast<const unsigned short *>(pa.tb.silu)[f2h_bits(out[0])]; pend_b = out[R - 1]; pend_row = g;
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int row = g * R + r;
                if (lane == 0 && row < total_rows) {
                    const int m = row >= 2 * rows_each ? 2 : (row >= rows_each ? 1 : 0), lr = row - m * rows_each;
                    const float val = has_res ? out[r] + G.res[r] : out[r];
                    ms.y0[(long long)m * ms.dy + lr] = val;
                }
            }
        }
    };
    // two statically named stages (a `cur = nxt` copy would have to wait for the loads in flight; the requested unroll is refused by the compiler)
    // The sched_barriers keep the next group's loads ahead of the current group's dot products (the scheduler otherwise hoists the arithmetic, and
    // with it the wait for the current tiles, above the loads: one tile in flight instead of two).
    for (int g = g_first; g < g_last;) {
        fetch(g + g_step, nxt);
        __builtin_amdgcn_sched_barrier(0);
        consume(g, cur);
        __builtin_amdgcn_sched_barrier(0);
#ifdef MG4_TIMELINE
        if (g == g_first) MG4_TL(3);
#endif
        g += g_step;
        if (g >= g_last) break;
        fetch(g + g_step, cur);
        __builtin_amdgcn_sched_barrier(0);
        consume(g, nxt);
        __builtin_amdgcn_sched_barrier(0);
        g += g_step;
    }
    MG4_TL(4);
    if (EPI == EPI_SILU_PAIR) flush_pending();
#ifdef MG4_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG4_TL(5); MG4_TL_ALL(6);
#endif
}
template <int T, int NU, int R, int PRO, int EPI>
__global__ __launch_bounds__(PRO == PRO_NONE ? 256 : mv_fat_max_threads<NU>()) void k_matvec_v2(const MatSet ms, const ActQ A, const ProArgs pa, const int n_groups, const int n_waves) {
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));   // wave-uniform: scalar loop control
    matvec_run<T, NU, R, PRO, EPI>(ms, A, pa, n_groups, n_waves, wave);
}
// Two weight types in one launch (llama.cpp's k-quant mixes give wv more bits than wq|wk): waves [0, n_waves1) stream set 1, the rest set 2.  Both
// sets share K and the prepared activation row (both types read the Q8_K image); every wave passes the same number of workgroup barriers.
template <int T1, int T2, int NU, int PRO, int EPI = EPI_STORE>
__global__ __launch_bounds__(PRO == PRO_NONE ? 256 : mv_fat_max_threads<NU>()) void k_matvec_mix(const MatSet ms1, const MatSet ms2, const ActQ A, const ProArgs pa, const int n_groups1,
                                                                                                 const int n_waves1, const int n_groups2, const int n_waves2) {
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (wave < n_waves1) matvec_run<T1, NU, 1, PRO, EPI>(ms1, A, pa, n_groups1, n_waves1, wave);
    else matvec_run<T2, NU, 1, PRO, EPI>(ms2, A, pa, n_groups2, n_waves2, wave - n_waves1);
}
static int g_mv_cus = 256;
static int g_mv_force_waves = 0;    // MINIGPT4_MV_WAVES: waves per CU of the prologue-free launches (0 = choose)
static int g_mv_force_fat = 0;      // MINIGPT4_FAT_LB: threads of the fat workgroups (0 = choose)
// Launch geometry (measured on the 13B decode, profiles/r01g_ab_geometry.log): 8 waves per CU everywhere.  More waves lose even where they divide
// the rows more evenly (5120 rows: 2048 waves = 3 | 2 rows per wave, 2560 waves = 2 each, yet 640-thread workgroups are 6 % slower end to
// end than 512-thread ones; 1024-thread ones 5 % slower; 5 / 7 waves per CU for the K = 13824 mat-vec 4 % / 1 % slower than 8).
static int pick_waves_per_cu(int /*groups*/, int max_wpc) { return std::min(g_mv_force_waves ? g_mv_force_waves : 8, max_wpc); }
static int pick_fat_threads(int /*groups*/, int max_threads) { return g_mv_force_fat ? std::max(MV_FAT_MIN, std::min(g_mv_force_fat / 64 * 64, max_threads)) : MV_FAT_MIN; }
static size_t mv_prologue_lds(int K) { return 128 + (size_t)2 * K + (size_t)(K / 256 + 1) * 4 + (size_t)4 * (K / 32) * 4 + (size_t)(K / 16) * 2 + 64; }
template <int NU> constexpr int mv_max_wpc() { return NU <= 3 ? 16 : NU == 4 ? 12 : 8; }      // register budget of the two-stage pipeline
template <int T, int NU, int R, int EPI>
static void launch_v2_t(const MatSet &ms, const ActQ &A, int pro, const ProArgs &pa, hipStream_t s) {
    const int total_rows = ms.n * ms.rows_each;
    const int n_groups = EPI == EPI_SILU_PAIR ? ms.rows_each : (total_rows + R - 1) / R;
    note_kernel("k_matvec_v2<%d, %d, %d, %d, %d>", T, NU, R, EPI == EPI_SILU_PAIR && pro != PRO_NONE ? (int)PRO_RMS : pro, (int)EPI);
    if (pro == PRO_NONE) {
        // EPI_REF: the block chain is a dependent sequence per wave (DPP move + fma per block): as many waves per SIMD as the registers admit, not the stream-optimal 8 per CU
        int n_waves = std::min(n_groups, g_mv_cus * (EPI == EPI_REF ? mv_max_wpc<NU>() : pick_waves_per_cu(n_groups, mv_max_wpc<NU>())));
        n_waves = (n_waves + 3) & ~3;
        hipLaunchKernelGGL((k_matvec_v2<T, NU, R, PRO_NONE, EPI>), dim3((unsigned)(n_waves / 4)), dim3(256), 0, s, ms, A, pa, n_groups, n_waves);
        return;
    }
    const int LB = EPI == EPI_REF ? mv_fat_max_threads<NU>() : pick_fat_threads(n_groups, mv_fat_max_threads<NU>()), WPB = LB / 64;
    const int n_blocks = std::min((n_groups + WPB - 1) / WPB, g_mv_cus);
    const int n_waves = n_blocks * WPB;
    const dim3 grid((unsigned)n_blocks), block((unsigned)LB);
    const int K = ms.w0.cols;
    const size_t lds = mv_prologue_lds(K);
    if constexpr (EPI == EPI_SILU_PAIR) {   // w1|w3 follow the ffn norm: only that prologue is instantiated for the pair epilogue
        hipLaunchKernelGGL((k_matvec_v2<T, NU, R, PRO_RMS, EPI>), grid, block, lds, s, ms, A, pa, n_groups, n_waves);
    } else switch (pro) {
    case PRO_RMS: hipLaunchKernelGGL((k_matvec_v2<T, NU, R, PRO_RMS, EPI>), grid, block, lds, s, ms, A, pa, n_groups, n_waves); break;
    case PRO_PLAIN: hipLaunchKernelGGL((k_matvec_v2<T, NU, R, PRO_PLAIN, EPI>), grid, block, lds, s, ms, A, pa, n_groups, n_waves); break;
    default: if constexpr (EPI == EPI_STORE) hipLaunchKernelGGL((k_matvec_v2<T, NU, R, PRO_SILU, EPI>), grid, block, lds, s, ms, A, pa, n_groups, n_waves); break;
    }
}
#ifndef MG4_R_NU3
#define MG4_R_NU3 1
#endif
#ifndef MG4_R_NU2
#define MG4_R_NU2 1
#endif
// Q8_0 / F16 (16-byte units of 16 / 8 weights): a row needs more units per lane than the 32-weight types, and only a few unit counts are instantiated -- a row takes the
// smallest one that covers it (the surplus units are masked like any ragged tail).  Q8_0: K <= 1024 | 4096 | 5120 | 11264; F16: K <= 512 | 1024 | 4096 | 5120.
template <int T>
static bool launch_v2_narrow(const MatSet &ms, const ActQ &A, int pro, const ProArgs &pa, int epi, hipStream_t s) {
    if (epi != EPI_STORE) return false;
    const int nu = (ms.w0.cols / TrMV<T>::EPU + 63) / 64;
    if constexpr (T == GT_Q8_0) {
        if (nu <= 1) launch_v2_t<T, 1, 2, EPI_STORE>(ms, A, pro, pa, s);
        else if (nu <= 4) launch_v2_t<T, 4, 1, EPI_STORE>(ms, A, pro, pa, s);
        else if (nu <= 5) launch_v2_t<T, 5, 1, EPI_STORE>(ms, A, pro, pa, s);
        else if (nu <= 11) launch_v2_t<T, 11, 1, EPI_STORE>(ms, A, pro, pa, s);
        else return false;
    } else {
        if (nu <= 1) launch_v2_t<T, 1, 2, EPI_STORE>(ms, A, pro, pa, s);
        else if (nu <= 2) launch_v2_t<T, 2, 1, EPI_STORE>(ms, A, pro, pa, s);
        else if (nu <= 8) launch_v2_t<T, 8, 1, EPI_STORE>(ms, A, pro, pa, s);
        else if (nu <= 10) launch_v2_t<T, 10, 1, EPI_STORE>(ms, A, pro, pa, s);
        else return false;
    }
    return true;
}
template <int T>
static bool launch_v2_type(const MatSet &ms, const ActQ &A, int pro, const ProArgs &pa, int epi, hipStream_t s) {
    const int U = ms.w0.cols / Tr<T>::EPU;
    const int nu = (U + 63) / 64;
    bool ok = true;
    if (epi == EPI_SILU_PAIR) {
        if (ms.n != 2 || (pro != PRO_NONE && pro != PRO_RMS)) ok = false;
        else switch (nu) {
        case 1: launch_v2_t<T, 1, 2, EPI_SILU_PAIR>(ms, A, pro, pa, s); break;
        case 2: launch_v2_t<T, 2, 2, EPI_SILU_PAIR>(ms, A, pro, pa, s); break;
        case 3: launch_v2_t<T, 3, 2, EPI_SILU_PAIR>(ms, A, pro, pa, s); break;
        default: ok = false;
        }
    } else if (epi == EPI_REF) {   // parity mode: the k-quants of the headline files (other types: k_mul_mat_ref_row); no SiLU prologue (the row comes from k_silu_mul_quant)
        if constexpr (T == GT_Q4_K || T == GT_Q5_K || T == GT_Q6_K) {
            if (pro == PRO_SILU) return false;
            switch (nu) {
            case 1: launch_v2_t<T, 1, 1, EPI_REF>(ms, A, pro, pa, s); break;
            case 2: launch_v2_t<T, 2, 1, EPI_REF>(ms, A, pro, pa, s); break;
            case 3: launch_v2_t<T, 3, 1, EPI_REF>(ms, A, pro, pa, s); break;
            case 4: launch_v2_t<T, 4, 1, EPI_REF>(ms, A, pro, pa, s); break;
            case 5: launch_v2_t<T, 5, 1, EPI_REF>(ms, A, pro, pa, s); break;
            case 6: launch_v2_t<T, 6, 1, EPI_REF>(ms, A, pro, pa, s); break;
            case 7: launch_v2_t<T, 7, 1, EPI_REF>(ms, A, pro, pa, s); break;
            default: ok = false;
            }
        } else ok = false;
    } else switch (nu) {
    case 1: launch_v2_t<T, 1, 2, EPI_STORE>(ms, A, pro, pa, s); break;
    case 2: launch_v2_t<T, 2, MG4_R_NU2, EPI_STORE>(ms, A, pro, pa, s); break;
    case 3: launch_v2_t<T, 3, MG4_R_NU3, EPI_STORE>(ms, A, pro, pa, s); break;
    case 4: launch_v2_t<T, 4, 1, EPI_STORE>(ms, A, pro, pa, s); break;
    case 5: launch_v2_t<T, 5, 1, EPI_STORE>(ms, A, pro, pa, s); break;
    case 6: launch_v2_t<T, 6, 1, EPI_STORE>(ms, A, pro, pa, s); break;
    case 7: launch_v2_t<T, 7, 1, EPI_STORE>(ms, A, pro, pa, s); break;
    default: ok = false;
    }
    return ok;
}
// copies the timeline stamps of the last decode mat-vec launch (8 x u64 per workgroup); 0 when the library was built without MG4_TIMELINE
int read_attn_timeline(unsigned long long *out, int max_workgroups) {
#ifdef MG4_TIMELINE
    const int n = std::min(max_workgroups, 2048);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tla), (size_t)n * 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return n;
#else
    (void)out; (void)max_workgroups; return 0;
#endif
}
int read_matvec_timeline(unsigned long long *out, int max_workgroups) {
#ifdef MG4_TIMELINE
    const int n = std::min(max_workgroups, 1024);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), (size_t)n * 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return n;
#else
    (void)out; (void)max_workgroups; return 0;
#endif
}
bool matvec_silu_pair_supported(int type, int cols) { return type != GT_Q8_0 && type != GT_F16 && matvec_prologue_supported(type, cols) && cols / 32 <= 3 * 64; }

static bool fill_matset(MatSet &ms, const QWeight *const *W, float *const *y, const float *const *residual, int n) {
    ms = MatSet{};
    ms.n = n; ms.rows_each = W[0]->rows; ms.w0 = *W[0]; ms.y0 = y[0]; ms.res0 = residual ? residual[0] : nullptr;
    if (n > 1) {
        ms.dmat = (long long)(W[1]->qs - W[0]->qs); ms.dy = (long long)(y[1] - y[0]); ms.dres = residual && residual[0] ? (long long)(residual[1] - residual[0]) : 0;
        for (int i = 1; i < n; i++) {
            if (W[i]->type != W[0]->type || W[i]->rows != W[0]->rows || W[i]->cols != W[0]->cols) return false;
            if ((long long)(W[i]->qs - W[0]->qs) != i * ms.dmat || (W[0]->qh && (long long)(W[i]->qh - W[0]->qh) != i * ms.dmat) ||
                (W[0]->sc && (long long)(W[i]->sc - W[0]->sc) != i * ms.dmat) || (W[0]->d && (long long)(W[i]->d - W[0]->d) != i * ms.dmat)) return false;
            if ((long long)(y[i] - y[0]) != i * ms.dy) return false;
            if (residual && residual[0] && (long long)(residual[i] - residual[0]) != i * ms.dres) return false;
        }
    }
    return true;
}
template <int T1, int T2, int NU, int EPI = EPI_STORE>
static void launch_mix_t(const MatSet &m1, const MatSet &m2, double bytes1, double bytes2, const ActQ &A, int pro, const ProArgs &pa, hipStream_t s) {
    const int ng1 = m1.n * m1.rows_each, ng2 = m2.n * m2.rows_each;
    // split the workgroups so that the busiest wave of either set finishes earliest: cost = rows per wave x bytes per row
    const double bpr1 = bytes1 / ng1, bpr2 = bytes2 / ng2;
    int best_t = 0, best_b1 = 0, best_total = 0; double best_cost = 1e300;
    const int t_fix = pro == PRO_NONE ? 256 : (EPI == EPI_REF ? mv_fat_max_threads<NU>() : pick_fat_threads(ng1, mv_fat_max_threads<NU>()));
    for (int t = t_fix; t <= t_fix; t += 64) {
        const int wpb = t / 64;
        const int total = pro == PRO_NONE ? g_mv_cus * (EPI == EPI_REF ? mv_max_wpc<NU>() : pick_waves_per_cu(ng1 + ng2, mv_max_wpc<NU>())) / 4 : g_mv_cus;
        for (int b1 = 1; b1 < total; b1++) {
            const int w1 = b1 * wpb, w2 = (total - b1) * wpb;
            const double c = std::max((double)((ng1 + w1 - 1) / w1) * bpr1, (double)((ng2 + w2 - 1) / w2) * bpr2);
            if (c < best_cost) { best_cost = c; best_t = t; best_b1 = b1; best_total = total; }
        }
    }
    note_kernel("k_matvec_mix<%d, %d, %d, %d, %d>", T1, T2, NU, pro == PRO_NONE ? 0 : 1, (int)EPI);
    const int wpb = best_t / 64, nw1 = best_b1 * wpb, nw2 = (best_total - best_b1) * wpb;
    const dim3 grid((unsigned)best_total), block((unsigned)best_t);
    if (pro == PRO_NONE) hipLaunchKernelGGL((k_matvec_mix<T1, T2, NU, PRO_NONE, EPI>), grid, block, 0, s, m1, m2, A, pa, ng1, nw1, ng2, nw2);
    else hipLaunchKernelGGL((k_matvec_mix<T1, T2, NU, PRO_RMS, EPI>), grid, block, mv_prologue_lds(m1.w0.cols), s, m1, m2, A, pa, ng1, nw1, ng2, nw2);
}
template <int T1, int T2, int EPI = EPI_STORE>
static bool launch_mix_nu(const MatSet &m1, const MatSet &m2, double b1, double b2, const ActQ &A, int pro, const ProArgs &pa, hipStream_t s) {
    const int nu = (m1.w0.cols / 32 + 63) / 64;
    switch (nu) {
    case 1: launch_mix_t<T1, T2, 1, EPI>(m1, m2, b1, b2, A, pro, pa, s); return true;
    case 2: launch_mix_t<T1, T2, 2, EPI>(m1, m2, b1, b2, A, pro, pa, s); return true;
    case 3: launch_mix_t<T1, T2, 3, EPI>(m1, m2, b1, b2, A, pro, pa, s); return true;
    case 4: launch_mix_t<T1, T2, 4, EPI>(m1, m2, b1, b2, A, pro, pa, s); return true;
    default: return false;
    }
}
// Decode mat-vec over two sets of different k-quant types with the same K (wq|wk + wv of a "more bits" layer) in ONE launch.  pro: PRO_NONE or PRO_RMS.
bool launch_matvec_mixed(const QWeight *const *W1, float *const *y1, int n1, const QWeight *const *W2, float *const *y2, int n2, const ActQ &A, hipStream_t s, int pro,
                         const float *px, const float *pw, int epi) {
    if (pro != PRO_NONE && pro != PRO_RMS) return false;
    if (W1[0]->cols != W2[0]->cols || W1[0]->cols % 256) return false;
    MatSet m1, m2;
    if (!fill_matset(m1, W1, y1, nullptr, n1) || !fill_matset(m2, W2, y2, nullptr, n2)) return false;
    ProArgs pa{}; pa.x = px; pa.w = pw;
    double b1 = 0, b2 = 0;
    for (int i = 0; i < n1; i++) b1 += (double)W1[i]->bytes;
    for (int i = 0; i < n2; i++) b2 += (double)W2[i]->bytes;
    const int t1 = W1[0]->type, t2 = W2[0]->type;
    if (epi == EPI_REF) {
        if (t1 == GT_Q5_K && t2 == GT_Q6_K) return launch_mix_nu<GT_Q5_K, GT_Q6_K, EPI_REF>(m1, m2, b1, b2, A, pro, pa, s);
        if (t1 == GT_Q4_K && t2 == GT_Q6_K) return launch_mix_nu<GT_Q4_K, GT_Q6_K, EPI_REF>(m1, m2, b1, b2, A, pro, pa, s);
        return false;
    }
    if (t1 == GT_Q5_K && t2 == GT_Q6_K) return launch_mix_nu<GT_Q5_K, GT_Q6_K>(m1, m2, b1, b2, A, pro, pa, s);
    if (t1 == GT_Q4_K && t2 == GT_Q6_K) return launch_mix_nu<GT_Q4_K, GT_Q6_K>(m1, m2, b1, b2, A, pro, pa, s);
    return false;
}
bool matvec_prologue_supported(int type, int cols) {
    if (type == GT_Q8_0) return cols % 32 == 0 && cols <= 11 * 64 * 16;      // launch_v2_narrow's instantiations
    if (type == GT_F16) return cols % 8 == 0 && cols <= 10 * 64 * 8;
    switch (type) { case GT_Q4_0: case GT_Q4_1: case GT_Q5_0: case GT_Q5_1: case GT_Q4_K: case GT_Q5_K: case GT_Q6_K: break; default: return false; }
    return cols % 32 == 0 && cols / 32 <= 7 * 64 && ((type != GT_Q4_K && type != GT_Q5_K && type != GT_Q6_K) || cols % 256 == 0);
}
void set_matvec_tuning(int waves_per_cu, int fat_threads, int cus) { g_mv_force_waves = std::max(0, waves_per_cu); g_mv_force_fat = std::max(0, fat_threads); if (cus > 0) { g_mv_cus = cus; g_ref_cus = cus; } }
// Decode (N = 1) mat-vec over 1..3 same-type, same-shape, equally spaced matrices.  Returns false when the set is outside the v2 kernel's range.
bool launch_matvec_set(const QWeight *const *W, float *const *y, const float *const *residual, int n, const ActQ &A, hipStream_t s, int pro, const float *px, const float *pw,
                       const Tables *tb, int epi) {
    ProArgs pa{}; pa.x = px; pa.w = pw; if (tb) pa.tb = *tb;
    MatSet ms;
    if (!fill_matset(ms, W, y, residual, n)) return false;
    switch (W[0]->type) {
    case GT_Q4_0: return launch_v2_type<GT_Q4_0>(ms, A, pro, pa, epi, s);
    case GT_Q4_1: return launch_v2_type<GT_Q4_1>(ms, A, pro, pa, epi, s);
    case GT_Q5_0: return launch_v2_type<GT_Q5_0>(ms, A, pro, pa, epi, s);
    case GT_Q5_1: return launch_v2_type<GT_Q5_1>(ms, A, pro, pa, epi, s);
    case GT_Q4_K: return launch_v2_type<GT_Q4_K>(ms, A, pro, pa, epi, s);
    case GT_Q5_K: return launch_v2_type<GT_Q5_K>(ms, A, pro, pa, epi, s);
    case GT_Q6_K: return launch_v2_type<GT_Q6_K>(ms, A, pro, pa, epi, s);
    case GT_Q8_0: return launch_v2_narrow<GT_Q8_0>(ms, A, pro, pa, epi, s);
    case GT_F16: return launch_v2_narrow<GT_F16>(ms, A, pro, pa, epi, s);
    default: return false;   // F32 rows, and Q8_0 / F16 rows wider than launch_v2_narrow's instantiations: served by k_mul_mat
    }
}


// =====================================================================================================================
// Batched decode mat-vec (B conversations, one row each): the same persistent-wave two-stage weight pipeline as k_matvec_v2, for up to TN
// activation rows.  The weights are streamed ONCE; every weight unit is multiplied against the matching activation unit of each row.  The
// quantised activation rows (prepared by k_rms_quant / k_silu_mul_quant) are copied into LDS once per workgroup -- one fat workgroup per CU --
// and read from there inside the loop (registers could hold only K <= 6144 for four rows; LDS traffic is ~1/4 of its bandwidth at HBM speed).
// Per row the arithmetic is exactly the single-row kernel's: the same unit dot products, the same per-lane fma order, the same wave reduction.
// =====================================================================================================================
template <int T> constexpr bool tr_is_kquant() { return T == GT_Q4_K || T == GT_Q5_K || T == GT_Q6_K; }
static size_t mv_tn_lds(int type, int K, int TN) {
    const bool kq = type == GT_Q4_K || type == GT_Q5_K || type == GT_Q6_K;
    const size_t per_row = kq ? (size_t)K + (size_t)(K / 256) * 4 + (size_t)(K / 16) * 2 : (size_t)K + (size_t)(K / 32) * 16;
    return (size_t)TN * per_row + 64;
}
// Are the TN rows' activation fragments register-resident?  512-thread workgroups (two waves per SIMD): 256 registers -> K <= 3 x 64 units at 4 rows, <= 5 x 64 at
// 2 rows; the widest K (NU = 7, K = 13824: w2) runs 256-thread workgroups, one wave per SIMD, 512 registers -> resident at 2 and at 4 rows (4 x 7 x 10 = 280 registers).
template <int NU, int TN> constexpr bool mv_tn_reg() { return NU <= 3 || (TN == 2 && NU <= 5) || NU >= 7; }
template <int NU> constexpr int mv_tn_threads() { return NU >= 7 ? 256 : 512; }   // widest K: one wave per SIMD (up to 512 VGPRs) -- two would spill Q5_K's weight stages; 4 waves x 7 units keep as many bytes in flight as 8 x 3
// PRO = 1 (K <= 3 x 64 units only): the rows are prepared inside the launch -- rms_norm(x_t) * w and the quantisation of k_rms_quant, row by row with the single-row
// prologue's arithmetic (matvec_run), into one LDS image per row -- so a batched decode step needs no standalone preparation launch in front of wq|wk|wv and w1|w3.
static size_t mv_tn_image_bytes(int K) { return ((size_t)2 * K + (size_t)(K / 256 + 1) * 4 + (size_t)4 * (K / 32) * 4 + (size_t)(K / 16) * 2 + 64 + 15) & ~(size_t)15; }
extern __shared__ __attribute__((aligned(16))) unsigned char smem_tn[];
// `wave` = this wave's index among the n_waves that share the set's rows (k_matvec_tn: all waves of the launch; k_matvec_tn_mix: the waves of one of its two sets)
template <int T, int NU, int TN, int PRO>
__device__ __forceinline__ void matvec_tn_run(const MatSet &ms, const ActQ &A, const int N, const int ldy, const int n_groups, const int n_waves, const int wave, const ProArgs &pa,
                                              const int ldx, const int image_bytes) {
    static_assert(PRO == 0 || NU <= 3, "the prologue variant: 512-thread workgroups, K <= 3 x 64 units");
    using X = Tr<T>;
    const int lane = threadIdx.x & 63;
    const int K = ms.w0.cols, U = K / X::EPU, rows_each = ms.rows_each, total_rows = ms.n * rows_each;
    int uc[NU]; bool ok[NU];
#pragma unroll
    for (int i = 0; i < NU; i++) { const int u = lane + 64 * i; ok[i] = u < U; uc[i] = ok[i] ? u : 0; }
    struct Grp { typename X::WU w[NU]; float res[TN]; };
    const bool has_res = ms.res0 != nullptr;
    const float *res_base = has_res ? ms.res0 : ms.y0;
    const long long res_stride = has_res ? ms.dres : ms.dy;
    auto fetch = [&](int g, Grp &G) {           // every load unconditional (clamped indices): see matvec_run
        const int row = min(g, total_rows - 1);
        const bool m1 = row >= rows_each, m2 = row >= 2 * rows_each;
        const int lr = row - (m2 ? 2 * rows_each : (m1 ? rows_each : 0));
        const long long d = m2 ? 2 * ms.dmat : (m1 ? ms.dmat : 0ll);
        QWeight W = ms.w0;
        W.qs += d; W.qh += d; W.sc += d; W.d += d;
        WBuf B;
        X::mkb(W, (size_t)lr * U, B);
#pragma unroll
        for (int i = 0; i < NU; i++) X::loadb(B, uc[i], G.w[i]);
        const float *rb = res_base + (m2 ? 2 * res_stride : (m1 ? res_stride : 0ll)) + lr;
#pragma unroll
        for (int t = 0; t < TN; t++) G.res[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mkbuf(reinterpret_cast<const uint8_t *>(rb + (size_t)min(t, N - 1) * ldy)), 0, 0, 0));
    };
    Grp cur, nxt;
    MG4_TL(0); MG4_TL_ALL(7);
    constexpr int PRND = PRO ? NU : 1;                                   // 512 threads x 4 elements per round: K <= NU * 2048
    float4 pxv[PRO ? TN : 1][PRND], pyv[PRND];
    bool pin[PRND];
    if (PRO) {   // the rows are requested before the first weight tiles (vector-memory results return in issue order)
#pragma unroll
        for (int r = 0; r < PRND; r++) {
            const int i = (r * 512 + (int)threadIdx.x) * 4;
            pin[r] = i < K;
            const int ic = pin[r] ? i : 0;
            pyv[r] = PRO == 1 ? *reinterpret_cast<const float4 *>(pa.w + ic) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
#pragma unroll
            for (int t = 0; t < (PRO ? TN : 1); t++) pxv[t][r] = *reinterpret_cast<const float4 *>(pa.x + (size_t)min(t, N - 1) * ldx + ic);
        }
    }
    fetch(wave, cur);
    MG4_TL(1);
    // K <= 3 x 64 units: every lane works on the SAME units of every weight row, so the TN activation fragments live in registers for the whole launch (no LDS
    // image, no barrier, no LDS re-read per weight row -- the LDS variant below re-reads TN x 5 KB per 3.5 KB weight row and is LDS-bandwidth bound).
    constexpr bool REG = mv_tn_reg<NU, TN>();
    typename X::AU ar[REG ? TN : 1][REG ? NU : 1];
    if (REG && PRO) {
        constexpr int mask = (T == GT_Q4_K || T == GT_Q5_K || T == GT_Q6_K) ? ACT_Q8K : ACT_Q80;
        double *red = reinterpret_cast<double *>(smem_tn);               // [TN][8 waves]
        if (PRO == 1) {
#pragma unroll
            for (int t = 0; t < (PRO ? TN : 1); t++) {
                double sum = 0.0;
#pragma unroll
                for (int r = 0; r < PRND; r++) {
                    const float4 v = pxv[t][r];
                    double q = 0.0;
                    q += (double)(v.x * v.x); q += (double)(v.y * v.y); q += (double)(v.z * v.z); q += (double)(v.w * v.w);
                    sum += pin[r] ? q : 0.0;
                }
                sum = wave_sum_d(sum);
                if (lane == 0) red[t * 8 + (threadIdx.x >> 6)] = sum;
            }
            __syncthreads();
        }
        ActQ Li[PRO ? TN : 1];
#pragma unroll
        for (int t = 0; t < (PRO ? TN : 1); t++) {
            float scale = 1.0f;
            if (PRO == 1) {
                double tot = 0.0;
                for (int w = 0; w < 8; w++) tot += red[t * 8 + w];
                const float mean = (float)(tot / (double)K);
                scale = 1.0f / sqrtf(mean + 1e-6f);
            }
            unsigned char *p = smem_tn + 512 + (size_t)t * image_bytes;
            ActQ L{};
            L.q8k = reinterpret_cast<int8_t *>(p); p += (size_t)K;
            L.q80 = reinterpret_cast<int8_t *>(p); p += (size_t)K;
            L.dk = reinterpret_cast<float *>(p); p += (size_t)(K / 256 + 1) * 4;
            L.d0 = reinterpret_cast<float *>(p); p += (size_t)(K / 32) * 4;
            L.d1 = reinterpret_cast<float *>(p); p += (size_t)(K / 32) * 4;
            L.s1 = reinterpret_cast<float *>(p); p += (size_t)(K / 32) * 4;
            L.sum0 = reinterpret_cast<int *>(p); p += (size_t)(K / 32) * 4;
            L.bsk = reinterpret_cast<int16_t *>(p);
            L.bsq = nullptr; L.xh = nullptr; L.xf = nullptr;
#pragma unroll
            for (int r = 0; r < PRND; r++) {
                const int i = (r * 512 + (int)threadIdx.x) * 4;
                float v[4] = {pxv[t][r].x, pxv[t][r].y, pxv[t][r].z, pxv[t][r].w};                                     // PRO == 2: the rows as they are (attention output -> wo)
                if (PRO == 1) { v[0] = (v[0] * scale) * pyv[r].x; v[1] = (v[1] * scale) * pyv[r].y; v[2] = (v[2] * scale) * pyv[r].z; v[3] = (v[3] * scale) * pyv[r].w; }
                if (!pin[r]) { v[0] = 0.0f; v[1] = 0.0f; v[2] = 0.0f; v[3] = 0.0f; }
                quant_emit4(v, pin[r], i, 0, K, L, mask);
            }
            Li[t] = L;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < (PRO ? TN : 1); t++)
#pragma unroll
            for (int i = 0; i < NU; i++) X::loada(Li[t], 0, K, uc[i], ar[REG ? t : 0][REG ? i : 0]);
    } else if (REG) {
#pragma unroll
        for (int t = 0; t < TN; t++)
#pragma unroll
            for (int i = 0; i < NU; i++) X::loada(A, min(t, N - 1), K, uc[i], ar[REG ? t : 0][REG ? i : 0]);
    }
    // otherwise: LDS image of the N activation rows, laid out like the global ActQ planes (so Tr<T>::loada indexes it unchanged)
    ActQ L;
    if (!REG) {
        unsigned char *p = smem_tn;
        const int nthr = (int)blockDim.x, tid = (int)threadIdx.x;
        // LDS-DMA copy (global_load_lds, 16 bytes per lane, destination lane-linear = a plain contiguous copy): all pieces of all planes are in flight together; the
        // round-1 form (a load -> ds_write loop, one dependent round trip per 4-8 KB) cost 14 serial round trips for a K = 13824 image -- most of the kernel's time.
        const int nwv = nthr >> 6, wv = tid >> 6;
        auto copy16 = [&](void *dst, const void *src, size_t bytes) {      // bytes is a multiple of 16 for every plane x row count used here, except the tails handled below
            const size_t n1k = (bytes + 1023) / 1024;
            for (size_t c = (size_t)wv; c < n1k; c += (size_t)nwv) {
                const size_t off = c * 1024 + (size_t)lane * 16;
                if (off + 16 <= bytes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(src) + off),
                                                     (__attribute__((address_space(3))) void *)(reinterpret_cast<unsigned char *>(dst) + c * 1024), 16, 0, 0);
            }
            const size_t n16 = bytes / 16;
            for (size_t i = n16 * 16 + (size_t)tid; i < bytes; i += (size_t)nthr) reinterpret_cast<unsigned char *>(dst)[i] = reinterpret_cast<const unsigned char *>(src)[i];
        };
        if (tr_is_kquant<T>()) {
            L.q8k = reinterpret_cast<int8_t *>(p); p += (size_t)TN * K;
            L.dk = reinterpret_cast<float *>(p); p += (size_t)TN * (K / 256) * 4;
            L.bsk = reinterpret_cast<int16_t *>(p);
            copy16(L.q8k, A.q8k, (size_t)N * K); copy16(L.dk, A.dk, (size_t)N * (K / 256) * 4); copy16(L.bsk, A.bsk, (size_t)N * (K / 16) * 2);
        } else {
            L.q80 = reinterpret_cast<int8_t *>(p); p += (size_t)TN * K;
            L.d0 = reinterpret_cast<float *>(p); p += (size_t)TN * (K / 32) * 4;
            L.d1 = reinterpret_cast<float *>(p); p += (size_t)TN * (K / 32) * 4;
            L.s1 = reinterpret_cast<float *>(p); p += (size_t)TN * (K / 32) * 4;
            L.sum0 = reinterpret_cast<int *>(p);
            copy16(L.q80, A.q80, (size_t)N * K); copy16(L.d0, A.d0, (size_t)N * (K / 32) * 4); copy16(L.d1, A.d1, (size_t)N * (K / 32) * 4);
            copy16(L.s1, A.s1, (size_t)N * (K / 32) * 4); copy16(L.sum0, A.sum0, (size_t)N * (K / 32) * 4);
        }
    }
    if (!REG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    MG4_TL(2);
    auto consume = [&](int g, const Grp &G) {
        float out[TN];
#pragma unroll
        for (int t = 0; t < TN; t++) out[t] = 0.0f;
        // unit-major: the decoded weight unit (nibble / high-bit unpacking, scale extraction) is shared by the TN rows; per row the units are still
        // accumulated in index order, like the single-row kernel
#pragma unroll
        for (int i = 0; i < NU; i++) {
#pragma unroll
            for (int t = 0; t < TN; t++) {
                float c = out[t];
                if (REG) X::dot(G.w[i], ar[REG ? t : 0][REG ? i : 0], c);
                else { typename X::AU a; X::loada(L, min(t, N - 1), K, uc[i], a); X::dot(G.w[i], a, c); }
                out[t] = ok[i] ? c : out[t];
            }
        }
#pragma unroll
        for (int t = 0; t < TN; t++) out[t] = wave_sum(out[t]);
        if (lane == 0 && g < total_rows) {
            const int m = g >= 2 * rows_each ? 2 : (g >= rows_each ? 1 : 0), lr = g - m * rows_each;
            float *yo = ms.y0 + (long long)m * ms.dy + lr;
#pragma unroll
            for (int t = 0; t < TN; t++) if (t < N) yo[(size_t)t * ldy] = has_res ? out[t] + G.res[t] : out[t];
        }
    };
    for (int g = wave; g < n_groups;) {
        fetch(g + n_waves, nxt);
        __builtin_amdgcn_sched_barrier(0);
        consume(g, cur);
        __builtin_amdgcn_sched_barrier(0);
#ifdef MG4_TIMELINE
        if (g == wave) MG4_TL(3);
#endif
        g += n_waves;
        if (g >= n_groups) break;
        fetch(g + n_waves, cur);
        __builtin_amdgcn_sched_barrier(0);
        consume(g, nxt);
        __builtin_amdgcn_sched_barrier(0);
        g += n_waves;
    }
    MG4_TL(4);
#ifdef MG4_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG4_TL(5); MG4_TL_ALL(6);
#endif
}
template <int T, int NU, int TN, int PRO = 0>
__global__ __launch_bounds__(mv_tn_threads<NU>()) void k_matvec_tn(const MatSet ms, const ActQ A, const int N, const int ldy, const int n_groups, const int n_waves, const ProArgs pa,
                                                                   const int ldx, const int image_bytes) {
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    matvec_tn_run<T, NU, TN, PRO>(ms, A, N, ldy, n_groups, n_waves, wave, pa, ldx, image_bytes);
}
// Two sets of different k-quant types with the same K (wq|wk + wv of a "more bits" layer) against the same TN rows in ONE launch: the workgroups are split between
// the sets by bytes, like k_matvec_mix does for the single-row step.  Both types quantise their rows to Q8_K, so the in-launch preparation (PRO = 1) is the same code
// with the same barriers on both sides of the split (a workgroup is never divided: the split is at workgroup granularity).
template <int T1, int T2, int NU, int TN, int PRO>
__global__ __launch_bounds__(mv_tn_threads<NU>()) void k_matvec_tn_mix(const MatSet ms1, const MatSet ms2, const ActQ A, const int N, const int ldy, const int n_groups1, const int n_waves1,
                                                                       const int n_groups2, const int n_waves2, const ProArgs pa, const int ldx, const int image_bytes) {
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (wave < n_waves1) matvec_tn_run<T1, NU, TN, PRO>(ms1, A, N, ldy, n_groups1, n_waves1, wave, pa, ldx, image_bytes);
    else matvec_tn_run<T2, NU, TN, PRO>(ms2, A, N, ldy, n_groups2, n_waves2, wave - n_waves1, pa, ldx, image_bytes);
}
template <int T, int NU, int TN>
static void launch_tn_n(const MatSet &ms, const ActQ &A, int N, int ldy, hipStream_t s, const float *px, const float *pw, int ldx) {
    const int n_groups = ms.n * ms.rows_each;
    constexpr int WPB = mv_tn_threads<NU>() / 64;
    const int n_blocks = std::min((n_groups + WPB - 1) / WPB, g_mv_cus), n_waves = n_blocks * WPB;
    ProArgs pa{}; pa.x = px; pa.w = pw;
    constexpr bool PRO_OK = NU <= 3 && (T == GT_Q4_0 || T == GT_Q4_K || T == GT_Q5_K || T == GT_Q6_K);
    if constexpr (PRO_OK) {
        if (px) {
            const int img = (int)mv_tn_image_bytes(ms.w0.cols);
            static bool attr1 = false;
            if (!attr1) { HIP_IGNORE(lds_optin_max(&k_matvec_tn<T, NU, TN, 1>));
                          HIP_IGNORE(lds_optin_max(&k_matvec_tn<T, NU, TN, 2>)); attr1 = true; }
            note_kernel("k_matvec_tn<%d, %d, %d, %d>", T, NU, TN, pw ? 1 : 2);
            if (pw) hipLaunchKernelGGL((k_matvec_tn<T, NU, TN, 1>), dim3((unsigned)n_blocks), dim3((unsigned)mv_tn_threads<NU>()), (size_t)512 + (size_t)TN * img, s, ms, A, N, ldy, n_groups, n_waves, pa, ldx, img);
            else hipLaunchKernelGGL((k_matvec_tn<T, NU, TN, 2>), dim3((unsigned)n_blocks), dim3((unsigned)mv_tn_threads<NU>()), (size_t)512 + (size_t)TN * img, s, ms, A, N, ldy, n_groups, n_waves, pa, ldx, img);
            return;
        }
    }
    note_kernel("k_matvec_tn<%d, %d, %d, 0>", T, NU, TN);
    const size_t lds = mv_tn_reg<NU, TN>() ? 0 : mv_tn_lds(T, ms.w0.cols, TN);
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_matvec_tn<T, NU, TN, 0>)); attr = true; }
    hipLaunchKernelGGL((k_matvec_tn<T, NU, TN, 0>), dim3((unsigned)n_blocks), dim3((unsigned)mv_tn_threads<NU>()), lds, s, ms, A, N, ldy, n_groups, n_waves, pa, 0, 0);
}
template <int T, int NU>
static void launch_tn_t(const MatSet &ms, const ActQ &A, int N, int ldy, hipStream_t s, const float *px, const float *pw, int ldx) {   // 2-row instantiation: half the dot products (and LDS reads) of the 4-row one
    if (N <= 2) launch_tn_n<T, NU, 2>(ms, A, N, ldy, s, px, pw, ldx);
    else if (N == 3) launch_tn_n<T, NU, 3>(ms, A, N, ldy, s, px, pw, ldx);      // the kernels are vector-issue bound: a 4th, unused row costs a quarter more
    else launch_tn_n<T, NU, 4>(ms, A, N, ldy, s, px, pw, ldx);
}
template <int T>
static bool launch_tn_type(const MatSet &ms, const ActQ &A, int N, int ldy, hipStream_t s, const float *px, const float *pw, int ldx) {
    const int nu = (ms.w0.cols / Tr<T>::EPU + 63) / 64;
    switch (nu) {
    case 1: launch_tn_t<T, 1>(ms, A, N, ldy, s, px, pw, ldx); return true;
    case 2: launch_tn_t<T, 2>(ms, A, N, ldy, s, px, pw, ldx); return true;
    case 3: launch_tn_t<T, 3>(ms, A, N, ldy, s, px, pw, ldx); return true;
    case 4: launch_tn_t<T, 4>(ms, A, N, ldy, s, px, pw, ldx); return true;
    case 5: launch_tn_t<T, 5>(ms, A, N, ldy, s, px, pw, ldx); return true;
    case 6: launch_tn_t<T, 6>(ms, A, N, ldy, s, px, pw, ldx); return true;
    case 7: launch_tn_t<T, 7>(ms, A, N, ldy, s, px, pw, ldx); return true;
    default: return false;
    }
}
// 2..4 activation rows against 1..3 same-type, same-shape, equally spaced matrices in ONE weight pass: y[m][t * ldy + r] = W_m[r] . act[t] (+ residual[m][t * ldy + r]).
// false -> shape / type outside the kernel's range (the caller falls back to launch_mul_mat per matrix).
bool matvec_rows_prologue_ok(int type, int K) {
    if (type != GT_Q4_0 && type != GT_Q4_K && type != GT_Q5_K && type != GT_Q6_K) return false;
    const int epu = type == GT_Q4_0 ? Tr<GT_Q4_0>::EPU : 32;
    return matvec_prologue_supported(type, K) && K / epu <= 192 && K % 4 == 0;
}
bool launch_matvec_rows(const QWeight *const *W, float *const *y, const float *const *residual, int n, const ActQ &A, int N, int ldy, hipStream_t s, const float *px, const float *pw, int ldx) {
    if (px && !matvec_rows_prologue_ok(W[0]->type, W[0]->cols)) return false;
    if (N < 1 || N > 4 || !matvec_prologue_supported(W[0]->type, W[0]->cols)) return false;
    if (mv_tn_lds(W[0]->type, W[0]->cols, 4) > 150 * 1024) return false;
    MatSet ms;
    if (!fill_matset(ms, W, y, residual, n)) return false;
    switch (W[0]->type) {
    case GT_Q4_0: return launch_tn_type<GT_Q4_0>(ms, A, N, ldy, s, px, pw, ldx);
    case GT_Q4_1: return launch_tn_type<GT_Q4_1>(ms, A, N, ldy, s, px, pw, ldx);
    case GT_Q5_0: return launch_tn_type<GT_Q5_0>(ms, A, N, ldy, s, px, pw, ldx);
    case GT_Q5_1: return launch_tn_type<GT_Q5_1>(ms, A, N, ldy, s, px, pw, ldx);
    case GT_Q4_K: return launch_tn_type<GT_Q4_K>(ms, A, N, ldy, s, px, pw, ldx);
    case GT_Q5_K: return launch_tn_type<GT_Q5_K>(ms, A, N, ldy, s, px, pw, ldx);
    case GT_Q6_K: return launch_tn_type<GT_Q6_K>(ms, A, N, ldy, s, px, pw, ldx);
    default: return false;
    }
}

template <int T1, int T2, int NU, int TN>
static void launch_tn_mix_n(const MatSet &m1, const MatSet &m2, double bytes1, double bytes2, const ActQ &A, int N, int ldy, hipStream_t s, const float *px, const float *pw, int ldx) {
    static_assert(NU <= 3, "register-resident rows, the prologue's range");
    const int ng1 = m1.n * m1.rows_each, ng2 = m2.n * m2.rows_each;
    constexpr int WPB = mv_tn_threads<NU>() / 64;
    const int total = g_mv_cus;
    // split the workgroups so that the busiest wave of either set finishes earliest: cost = rows per wave x bytes per row (launch_mix_t)
    const double bpr1 = bytes1 / ng1, bpr2 = bytes2 / ng2;
    int best_b1 = 1; double best_cost = 1e300;
    for (int b1 = 1; b1 < total; b1++) {
        const int w1 = b1 * WPB, w2 = (total - b1) * WPB;
        const double c = std::max((double)((ng1 + w1 - 1) / w1) * bpr1, (double)((ng2 + w2 - 1) / w2) * bpr2);
        if (c < best_cost) { best_cost = c; best_b1 = b1; }
    }
    const int nw1 = best_b1 * WPB, nw2 = (total - best_b1) * WPB;
    ProArgs pa{}; pa.x = px; pa.w = pw;
    const dim3 grid((unsigned)total), block((unsigned)mv_tn_threads<NU>());
    if (px) {
        const int img = (int)mv_tn_image_bytes(m1.w0.cols);
        static bool attr1 = false;
        if (!attr1) { HIP_IGNORE(lds_optin_max(&k_matvec_tn_mix<T1, T2, NU, TN, 1>)); attr1 = true; }
        note_kernel("k_matvec_tn_mix<%d, %d, %d, %d, 1>", T1, T2, NU, TN);
        hipLaunchKernelGGL((k_matvec_tn_mix<T1, T2, NU, TN, 1>), grid, block, (size_t)512 + (size_t)TN * img, s, m1, m2, A, N, ldy, ng1, nw1, ng2, nw2, pa, ldx, img);
        return;
    }
    note_kernel("k_matvec_tn_mix<%d, %d, %d, %d, 0>", T1, T2, NU, TN);
    hipLaunchKernelGGL((k_matvec_tn_mix<T1, T2, NU, TN, 0>), grid, block, 0, s, m1, m2, A, N, ldy, ng1, nw1, ng2, nw2, pa, 0, 0);
}
template <int T1, int T2>
static bool launch_tn_mix_type(const MatSet &m1, const MatSet &m2, double b1, double b2, const ActQ &A, int N, int ldy, hipStream_t s, const float *px, const float *pw, int ldx) {
    const int nu = (m1.w0.cols / 32 + 63) / 64;
    switch (nu * 3 + (N <= 2 ? 0 : N == 3 ? 1 : 2)) {
    case 3: launch_tn_mix_n<T1, T2, 1, 2>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx); return true;
    case 4: launch_tn_mix_n<T1, T2, 1, 3>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx); return true;
    case 5: launch_tn_mix_n<T1, T2, 1, 4>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx); return true;
    case 6: launch_tn_mix_n<T1, T2, 2, 2>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx); return true;
    case 7: launch_tn_mix_n<T1, T2, 2, 3>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx); return true;
    case 8: launch_tn_mix_n<T1, T2, 2, 4>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx); return true;
    case 9: launch_tn_mix_n<T1, T2, 3, 2>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx); return true;
    case 10: launch_tn_mix_n<T1, T2, 3, 3>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx); return true;
    case 11: launch_tn_mix_n<T1, T2, 3, 4>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx); return true;
    default: return false;
    }
}
// Batched decode, N = 1..4 rows: two sets of different k-quant types with the same K (wq|wk + wv of a "more bits" layer) in ONE launch.  px != null: the rows are
// prepared inside the launch (rms_norm(px_t) * pw, quantised), as in launch_matvec_rows.  false: outside the kernel's range, nothing was launched.
bool launch_matvec_rows_mixed(const QWeight *const *W1, float *const *y1, int n1, const QWeight *const *W2, float *const *y2, int n2, const ActQ &A, int N, int ldy, hipStream_t s,
                              const float *px, const float *pw, int ldx) {
    if (N < 1 || N > 4 || W1[0]->cols != W2[0]->cols || W1[0]->cols % 256 || W1[0]->cols / 32 > 3 * 64) return false;
    if (px && (!pw || !matvec_rows_prologue_ok(W1[0]->type, W1[0]->cols) || !matvec_rows_prologue_ok(W2[0]->type, W2[0]->cols))) return false;
    MatSet m1, m2;
    if (!fill_matset(m1, W1, y1, nullptr, n1) || !fill_matset(m2, W2, y2, nullptr, n2)) return false;
    double b1 = 0, b2 = 0;
    for (int i = 0; i < n1; i++) b1 += (double)W1[i]->bytes;
    for (int i = 0; i < n2; i++) b2 += (double)W2[i]->bytes;
    const int t1 = W1[0]->type, t2 = W2[0]->type;
    if (t1 == GT_Q5_K && t2 == GT_Q6_K) return launch_tn_mix_type<GT_Q5_K, GT_Q6_K>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx);
    if (t1 == GT_Q4_K && t2 == GT_Q6_K) return launch_tn_mix_type<GT_Q4_K, GT_Q6_K>(m1, m2, b1, b2, A, N, ldy, s, px, pw, ldx);
    return false;
}

static int g_mmq_enabled = 2;   // 0: v_dot4 tiles only (tests / A-B), 2: the LDS-staged int8-MFMA kernels of mmq2_kernels.hip for N >= 5 rows
void set_mmq_enabled(int v) { g_mmq_enabled = v; }
int mmq_enabled() { return g_mmq_enabled; }
// Unquantised (F16) weights, prefill: ggml converts the activation rows to fp16 and accumulates exact fp16 products in fp32 (ggml_vec_dot_f16) -- which is
// precisely the vision tower's MFMA GEMM (v_mfma_f32_32x32x16_f16, fp32 accumulators), so rows >= 16 go there: BASELINE.json configs[4]
// (13B f16, 512-token prefill) is MFMA-bound instead of re-streaming 25 GB of weights once per 4 tokens.  MINIGPT4_F16_GEMM=0 keeps the v_dot path.
static int g_f16_gemm = 1;
void set_f16_gemm(int v) { g_f16_gemm = v != 0; }
void launch_mul_mat(const QWeight &W, const ActQ &A, int N, float *y, int ldy, const float *residual, hipStream_t s) {
    if (g_mmq_enabled >= 2 && N >= 5 && (A.bsq || W.type == GT_Q4_0) && mmq2_supported(W.type, W.rows, W.cols)) {
        const QWeight *Wp[1] = {&W}; float *Yp[1] = {y}; const float *Rp[1] = {residual};
        if (launch_mmq2_set(Wp, Yp, residual ? Rp : nullptr, 1, A, N, ldy, s)) return;
    }
    if (W.type == GT_F16 && N >= 16 && W.cols % 8 == 0) {
        if (g_f16_gemm) { launch_gemm_f16(A.xh, W.cols, reinterpret_cast<const __half *>(W.qs), W.cols, N, W.rows, W.cols, nullptr, residual, false, Tables{}, y, nullptr, ldy, s); return; }
    }
    switch (W.type) {
    case GT_Q4_0: launch_mul_mat_t<GT_Q4_0>(W, A, N, y, ldy, residual, s); break;
    case GT_Q4_1: launch_mul_mat_t<GT_Q4_1>(W, A, N, y, ldy, residual, s); break;
    case GT_Q5_0: launch_mul_mat_t<GT_Q5_0>(W, A, N, y, ldy, residual, s); break;
    case GT_Q5_1: launch_mul_mat_t<GT_Q5_1>(W, A, N, y, ldy, residual, s); break;
    case GT_Q8_0: launch_mul_mat_t<GT_Q8_0>(W, A, N, y, ldy, residual, s); break;
    case GT_Q2_K: launch_mul_mat_t<GT_Q2_K>(W, A, N, y, ldy, residual, s); break;
    case GT_Q4_K: launch_mul_mat_t<GT_Q4_K>(W, A, N, y, ldy, residual, s); break;
    case GT_Q5_K: launch_mul_mat_t<GT_Q5_K>(W, A, N, y, ldy, residual, s); break;
    case GT_Q6_K: launch_mul_mat_t<GT_Q6_K>(W, A, N, y, ldy, residual, s); break;
    case GT_F16: launch_mul_mat_t<GT_F16>(W, A, N, y, ldy, residual, s); break;
    case GT_F32: launch_mul_mat_t<GT_F32>(W, A, N, y, ldy, residual, s); break;
    default: throw HipError{hipErrorInvalidValue, "unsupported weight type", __FILE__, __LINE__};
    }
}


constexpr int RQ_THREADS = 1024;
__global__ __launch_bounds__(RQ_THREADS) void k_rms_quant(const float *__restrict__ x, const float *__restrict__ w, const int K, const ActQ A, const int mask, const int seq) {
    const size_t row = blockIdx.x;
    const float *xr = x + row * K;
    __shared__ double red[RQ_THREADS / 64];
    float scale = 1.0f;
    if (w && seq) {   // MINIGPT4_PARITY: the value of ggml_compute_forward_rms_norm's loop -- ONE double accumulator, element order -- without running 5120 dependent double
        // additions on one lane (20 us per norm, 1.6 ms of a 13B token).  Every addend is >= 0, so the sequential sum S and the parallel sum T below both lie within
        // K * 2^-53 (relative) of the exact sum; mean = (float)(S / K) is monotone in S, so if the two ends of T (1 +- 4e-16 K) convert to the SAME float, that float is
        // the oracle's mean, whatever S is.  Otherwise (a sum that close to a float rounding boundary: ~1e-4 of the rows) thread 0 runs the literal loop.
        double sum = 0.0;
        for (int i = threadIdx.x * 4; i < K; i += RQ_THREADS * 4) { const float4 v = *reinterpret_cast<const float4 *>(xr + i);
            sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
        sum = wave_sum_d(sum);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < RQ_THREADS / 64; i++) tot += red[i];
        const double delta = (double)K * 4e-16;          // two sums of K non-negative terms, each within K * 2^-53 of the exact one, and margin for the multiplications below
        const float lo = (float)(tot * (1.0 - delta) / (double)K), hi = (float)(tot * (1.0 + delta) / (double)K);
        float mean = lo;
        if (lo != hi) {                                  // block-uniform: every thread computed the same tot
            __syncthreads();
            if (threadIdx.x == 0) { double sq = 0.0; for (int i = 0; i < K; i++) sq += (double)(xr[i] * xr[i]); red[0] = sq; }
            __syncthreads();
            mean = (float)(red[0] / (double)K);
        }
        scale = 1.0f / sqrtf(mean + 1e-6f);
    } else if (w) {
        double sum = 0.0;
        for (int i = threadIdx.x * 4; i < K; i += RQ_THREADS * 4) { const float4 v = *reinterpret_cast<const float4 *>(xr + i);
            sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
        sum = wave_sum_d(sum);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < RQ_THREADS / 64; i++) tot += red[i];
        const float mean = (float)(tot / (double)K);
        scale = 1.0f / sqrtf(mean + 1e-6f);
    }
    for (int i0 = 0; i0 < K; i0 += RQ_THREADS * 4) {
        const int i = i0 + threadIdx.x * 4;
        const bool in = i < K;
        float v[4] = {0, 0, 0, 0};
        if (in) { const float4 xv = *reinterpret_cast<const float4 *>(xr + i);
            if (w) { const float4 wv = *reinterpret_cast<const float4 *>(w + i); v[0] = (xv.x * scale) * wv.x; v[1] = (xv.y * scale) * wv.y; v[2] = (xv.z * scale) * wv.z; v[3] = (xv.w * scale) * wv.w; }
            else { v[0] = xv.x; v[1] = xv.y; v[2] = xv.z; v[3] = xv.w; } }
        quant_emit4(v, in, i, row, K, A, mask);
    }
}
// Consumers of a split-K mat-mul whose combine step was deferred (round 3; SlabSrc in kernels.hpp): the value of element i of matrix m is
//     (slab_0[i] + slab_1[i] + ... + slab_{ks-1}[i]) (+ residual[i])          -- exactly k_mmq2_reduce_set's additions, in its order --
// formed where it is consumed instead of by a launch of its own (a prompt pass had 3.5 such launches per layer: 0.73 of the 142-row pass's 11.3 ms).
__device__ __forceinline__ float4 slab_sum4(const float *__restrict__ base, int ks, long long stride, const float *__restrict__ res, size_t i) {
    float4 s = *reinterpret_cast<const float4 *>(base + i);
    for (int z = 1; z < ks; z++) { const float4 t = *reinterpret_cast<const float4 *>(base + (size_t)z * stride + i); s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    if (res) { const float4 t = *reinterpret_cast<const float4 *>(res + i); s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    return s;
}
// x_out = residual + sum of slabs (the residual stream, materialised here), then k_rms_quant's norm + quantisation of the row
__global__ __launch_bounds__(RQ_THREADS) void k_rms_quant_slabs(const float *__restrict__ slabs, const int ks, const long long stride, const float *res, float *x_out,
                                                                const float *__restrict__ w, const int K, const ActQ A, const int mask) {
    const size_t row = blockIdx.x;
    float *xr = x_out + row * K;
    __shared__ double red[RQ_THREADS / 64];
    double sum = 0.0;
    for (int i = threadIdx.x * 4; i < K; i += RQ_THREADS * 4) {
        const float4 v = slab_sum4(slabs, ks, stride, res, row * K + i);
        *reinterpret_cast<float4 *>(xr + i) = v;                   // re-read below by the same thread
        sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w);
    }
    sum = wave_sum_d(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < RQ_THREADS / 64; i++) tot += red[i];
    const float mean = (float)(tot / (double)K);
    const float scale = 1.0f / sqrtf(mean + 1e-6f);
    for (int i0 = 0; i0 < K; i0 += RQ_THREADS * 4) {
        const int i = i0 + threadIdx.x * 4;
        const bool in = i < K;
        float v[4] = {0, 0, 0, 0};
        if (in) { const float4 xv = *reinterpret_cast<const float4 *>(xr + i); const float4 wv = *reinterpret_cast<const float4 *>(w + i);
            v[0] = (xv.x * scale) * wv.x; v[1] = (xv.y * scale) * wv.y; v[2] = (xv.z * scale) * wv.z; v[3] = (xv.w * scale) * wv.w; }
        quant_emit4(v, in, i, row, K, A, mask);
    }
}
void launch_rms_quant_slabs(const SlabSrc &src, const float *w, int N, int K, const ActQ &A, int mask, hipStream_t s) {
    note_kernel("k_rms_quant_slabs");
    hipLaunchKernelGGL(k_rms_quant_slabs, dim3((unsigned)N), dim3(RQ_THREADS), 0, s, src.ws, src.ks, src.stride, src.res[0], src.y[0], w, K, A, mask);
}
void launch_rms_quant(const float *x, const float *w, int N, int K, const ActQ &A, int mask, hipStream_t s, bool sequential_sum) {
    note_kernel("k_rms_quant");
    hipLaunchKernelGGL(k_rms_quant, dim3((unsigned)N), dim3(RQ_THREADS), 0, s, x, w, K, A, mask, sequential_sum ? 1 : 0);
}

__global__ __launch_bounds__(256) void k_silu_mul_quant(const float *__restrict__ a, const float *__restrict__ b, const int K, const ActQ A, const int mask, const Tables tb) {
    const size_t row = blockIdx.y;
    const int i = blockIdx.x * 1024 + threadIdx.x * 4;
    const bool in = i < K;
    float v[4] = {0, 0, 0, 0};
    if (in) { const float4 av = *reinterpret_cast<const float4 *>(a + row * K + i);
        if (b) { const float4 bv = *reinterpret_cast<const float4 *>(b + row * K + i);
            v[0] = silu_h(tb.silu, av.x) * bv.x; v[1] = silu_h(tb.silu, av.y) * bv.y; v[2] = silu_h(tb.silu, av.z) * bv.z; v[3] = silu_h(tb.silu, av.w) * bv.w; }
        else { v[0] = av.x; v[1] = av.y; v[2] = av.z; v[3] = av.w; } }
    quant_emit4(v, in, i, row, K, A, mask);
}
// a = sum of matrix 0's slabs, b = sum of matrix 1's (w1 | w3 of a prompt pass): neither product is written out
__global__ __launch_bounds__(256) void k_silu_mul_quant_slabs(const float *__restrict__ slabs, const int ks, const long long stride, const int K, const ActQ A, const int mask, const Tables tb) {
    const size_t row = blockIdx.y;
    const int i = blockIdx.x * 1024 + threadIdx.x * 4;
    const bool in = i < K;
    float v[4] = {0, 0, 0, 0};
    if (in) {
        const float4 av = slab_sum4(slabs, ks, stride, nullptr, row * K + i), bv = slab_sum4(slabs + (size_t)ks * stride, ks, stride, nullptr, row * K + i);
        v[0] = tab(tb.silu, av.x) * bv.x; v[1] = tab(tb.silu, av.y) * bv.y; v[2] = tab(tb.silu, av.z) * bv.z; v[3] = tab(tb.silu, av.w) * bv.w;
    }
    quant_emit4(v, in, i, row, K, A, mask);
}
void launch_silu_mul_quant_slabs(const SlabSrc &src, int N, int K, const ActQ &A, int mask, const Tables &tb, hipStream_t s) {
    note_kernel("k_silu_mul_quant_slabs");
    hipLaunchKernelGGL(k_silu_mul_quant_slabs, dim3((unsigned)((K + 1023) / 1024), (unsigned)N), dim3(256), 0, s, src.ws, src.ks, src.stride, K, A, mask, tb);
}
void launch_silu_mul_quant(const float *a, const float *b, int N, int K, const ActQ &A, int mask, const Tables &tb, hipStream_t s) {
    note_kernel("k_silu_mul_quant");
    hipLaunchKernelGGL(k_silu_mul_quant, dim3((unsigned)((K + 1023) / 1024), (unsigned)N), dim3(256), 0, s, a, b, K, A, mask, tb);
}

// =====================================================================================================================
// embedding gather: dequantise rows of the raw (un-repacked) ggml table
// =====================================================================================================================
__device__ __forceinline__ void sm_k4(int j, const uint8_t *q, int &d, int &m) {
    if (j < 4) { d = q[j] & 63; m = q[j + 4] & 63; } else { d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}
__device__ float dequant_elem(int type, const uint8_t *row, int e) {
    switch (type) {
    case GT_F32: return reinterpret_cast<const float *>(row)[e];
    case GT_F16: return h2f_bits(reinterpret_cast<const unsigned short *>(row)[e]);
    case GT_Q4_0: { const uint8_t *b = row + (e >> 5) * 18; const int j = e & 31; const float d = h2f_bits(b[0] | (b[1] << 8)); const int q = j < 16 ? (b[2 + j] & 15) : (b[2 + j - 16] >> 4); return (float)(q - 8) * d; }
    case GT_Q4_1: { const uint8_t *b = row + (e >> 5) * 20; const int j = e & 31; const float d = h2f_bits(b[0] | (b[1] << 8)), m = h2f_bits(b[2] | (b[3] << 8)); const int q = j < 16 ? (b[4 + j] & 15) : (b[4 + j - 16] >> 4); return (float)q * d + m; }
    case GT_Q5_0: { const uint8_t *b = row + (e >> 5) * 22; const int j = e & 31; const float d = h2f_bits(b[0] | (b[1] << 8)); const unsigned qh = b[2] | (b[3] << 8) | (b[4] << 16) | ((unsigned)b[5] << 24);
        const int q = (j < 16 ? (b[6 + j] & 15) : (b[6 + j - 16] >> 4)) | (((qh >> j) & 1) << 4); return (float)(q - 16) * d; }
    case GT_Q5_1: { const uint8_t *b = row + (e >> 5) * 24; const int j = e & 31; const float d = h2f_bits(b[0] | (b[1] << 8)), m = h2f_bits(b[2] | (b[3] << 8)); const unsigned qh = b[4] | (b[5] << 8) | (b[6] << 16) | ((unsigned)b[7] << 24);
        const int q = (j < 16 ? (b[8 + j] & 15) : (b[8 + j - 16] >> 4)) | (((qh >> j) & 1) << 4); return (float)q * d + m; }
    case GT_Q8_0: { const uint8_t *b = row + (e >> 5) * 34; const float d = h2f_bits(b[0] | (b[1] << 8)); return (float)(signed char)b[2 + (e & 31)] * d; }
    case GT_Q2_K: { const uint8_t *b = row + (e >> 8) * 84; const int i = e & 255, n = i >> 7, j = (i >> 5) & 3, l = i & 31; const uint8_t sc = b[i >> 4];
        const float d = h2f_bits(b[80] | (b[81] << 8)), dm = h2f_bits(b[82] | (b[83] << 8)); const float dl = d * (float)(sc & 0xF), ml = dm * (float)(sc >> 4);
        return dl * (float)((b[16 + 32 * n + l] >> (2 * j)) & 3) - ml; }
    case GT_Q4_K: { const uint8_t *b = row + (e >> 8) * 144; const int i = e & 255, s = i >> 5, l = i & 31; const float d = h2f_bits(b[0] | (b[1] << 8)), dm = h2f_bits(b[2] | (b[3] << 8)); int sc, m; sm_k4(s, b + 4, sc, m);
        const uint8_t qb = b[16 + (s >> 1) * 32 + l]; const int q = (s & 1) ? (qb >> 4) : (qb & 15); return (d * sc) * (float)q - dm * m; }
    case GT_Q5_K: { const uint8_t *b = row + (e >> 8) * 176; const int i = e & 255, s = i >> 5, l = i & 31; const float d = h2f_bits(b[0] | (b[1] << 8)), dm = h2f_bits(b[2] | (b[3] << 8)); int sc, m; sm_k4(s, b + 4, sc, m);
        const uint8_t qb = b[48 + (s >> 1) * 32 + l]; const int q = ((s & 1) ? (qb >> 4) : (qb & 15)) + (((b[16 + l] >> s) & 1) ? 16 : 0); return (d * sc) * (float)q - dm * m; }
    case GT_Q6_K: { const uint8_t *b = row + (e >> 8) * 210; const int i = e & 255, n = i >> 7, a = (i >> 5) & 3, l = i & 31; const float d = h2f_bits(b[208] | (b[209] << 8));
        const uint8_t qlb = b[64 * n + 32 * (a & 1) + l]; const int lo = (a & 2) ? (qlb >> 4) : (qlb & 15); const int hi = (b[128 + 32 * n + l] >> (2 * a)) & 3;
        const int q = (lo | (hi << 4)) - 32; const int sc = (signed char)b[192 + 8 * n + 2 * a + (l >> 4)]; return d * (float)sc * (float)q; }
    default: return 0.0f;
    }
}
__global__ void k_get_rows(int type, const uint8_t *__restrict__ table, int K, size_t row_bytes, const int *__restrict__ tokens, float *__restrict__ out) {
    const int t = blockIdx.x;
    if (tokens[t] < 0) return;                     // embedding row: already written by the host copy
    const uint8_t *row = table + (size_t)tokens[t] * row_bytes;
    const int e = blockIdx.y * blockDim.x + threadIdx.x;
    if (e < K) out[(size_t)t * K + e] = dequant_elem(type, row, e);
}
void launch_get_rows(int type, const uint8_t *raw_table, int K, const int *tokens, int N, float *out, hipStream_t s) {
    note_kernel("k_get_rows");
    hipLaunchKernelGGL(k_get_rows, dim3((unsigned)N, (unsigned)((K + 255) / 256)), dim3(256), 0, s, type, raw_table, K, gt_nbytes(type, (size_t)K), tokens, out);
}

// =====================================================================================================================
// RoPE (ggml mode 0: interleaved pairs; cos/sin table built on the host with ggml's iterative fp32 theta) + KV append
// =====================================================================================================================
__global__ void k_rope_kv(float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v, int E, int hd, const int *__restrict__ n_past,
                          const float *__restrict__ cos_tab, const float *__restrict__ sin_tab, __half *__restrict__ kc, __half *__restrict__ vc) {
    const int t = blockIdx.x, h = blockIdx.y, i = threadIdx.x;   // i < hd/2
    const int pos = *n_past + t;
    const float c = cos_tab[(size_t)pos * (hd / 2) + i], s = sin_tab[(size_t)pos * (hd / 2) + i];
    const size_t o = (size_t)t * E + (size_t)h * hd + 2 * i;
    const float q0 = q[o], q1 = q[o + 1];
    q[o] = q0 * c - q1 * s; q[o + 1] = q0 * s + q1 * c;
    const float k0 = k[o], k1 = k[o + 1];
    const size_t co = (size_t)pos * E + (size_t)h * hd + 2 * i;
    *reinterpret_cast<__half2 *>(kc + co) = __floats2half2_rn(k0 * c - k1 * s, k0 * s + k1 * c);
    *reinterpret_cast<__half2 *>(vc + co) = __floats2half2_rn(v[o], v[o + 1]);
}
// q | k | v = the sums of the three matrices' slabs (wq | wk | wv of a prompt pass); the rotated q is written to `q` as before, k and v only to the caches
struct RopeSlabs { const float *base[3]; int ks[3]; };
__global__ void k_rope_kv_slabs(const RopeSlabs rs, const long long stride, float *__restrict__ q, int E, int hd, const int *__restrict__ n_past,
                                const float *__restrict__ cos_tab, const float *__restrict__ sin_tab, __half *__restrict__ kc, __half *__restrict__ vc) {
    const int t = blockIdx.x, h = blockIdx.y, i = threadIdx.x;   // i < hd/2
    const int pos = *n_past + t;
    const float c = cos_tab[(size_t)pos * (hd / 2) + i], s = sin_tab[(size_t)pos * (hd / 2) + i];
    const size_t o = (size_t)t * E + (size_t)h * hd + 2 * i;
    float2 m[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float *b = rs.base[a] + o;
        float2 acc = *reinterpret_cast<const float2 *>(b);
        for (int z = 1; z < rs.ks[a]; z++) { const float2 u = *reinterpret_cast<const float2 *>(b + (size_t)z * stride); acc.x += u.x; acc.y += u.y; }
        m[a] = acc;
    }
    q[o] = m[0].x * c - m[0].y * s; q[o + 1] = m[0].x * s + m[0].y * c;
    const size_t co = (size_t)pos * E + (size_t)h * hd + 2 * i;
    *reinterpret_cast<__half2 *>(kc + co) = __floats2half2_rn(m[1].x * c - m[1].y * s, m[1].x * s + m[1].y * c);
    *reinterpret_cast<__half2 *>(vc + co) = __floats2half2_rn(m[2].x, m[2].y);
}
void launch_rope_kv_slabs(const SlabSrc &src, int N, int n_head, int hd, const int *n_past, const float *cos_tab, const float *sin_tab, __half *kcache, __half *vcache, hipStream_t s) {
    RopeSlabs rs;
    for (int a = 0; a < 3; a++) { rs.base[a] = src.mbase[a]; rs.ks[a] = src.mks[a]; }
    hipLaunchKernelGGL(k_rope_kv_slabs, dim3((unsigned)N, (unsigned)n_head), dim3((unsigned)(hd / 2)), 0, s, rs, src.stride, src.y[0], n_head * hd, hd, n_past, cos_tab, sin_tab, kcache, vcache);
}
void launch_rope_kv(float *q, const float *k, const float *v, int N, int n_head, int hd, const int *n_past, const float *cos_tab, const float *sin_tab,
                    __half *kcache, __half *vcache, hipStream_t s) {
    hipLaunchKernelGGL(k_rope_kv, dim3((unsigned)N, (unsigned)n_head), dim3((unsigned)(hd / 2)), 0, s, q, k, v, n_head * hd, hd, n_past, cos_tab, sin_tab, kcache, vcache);
}

// =====================================================================================================================
// causal attention over the fp16 KV cache.  One 512-thread workgroup per (head, query token).
//   FUSED (decode, N = 1): RoPE of q/k and the KV append happen in the prologue; the new key/value are also kept in LDS.
//   scores : one lane per key, the whole head row in registers (HD/8 independent 16-byte loads), sequential fp32 fma over the
//            head dim; q is rounded to fp16 first, like ggml's f16 x f32 mul_mat
//   softmax: max, exp through the fp16 table, exact double sum, probabilities rounded to fp16
//   PV     : thread = (key partition, 8-dim chunk): 16-byte V loads, 4 in flight; partitions reduced through LDS
// =====================================================================================================================
constexpr int AT_THREADS = 512;
//   BATCHED (decode of several conversations in one pass; implies FUSED): row t belongs to conversation row_slot[t], whose position is
//            n_past[row_slot[t]] and whose caches start seq_stride * row_slot[t] elements behind kc / vc.
template <int HD, bool FUSED, bool BATCHED = false>
__global__ __launch_bounds__(AT_THREADS) void k_attn_llm(float *__restrict__ q, const float *__restrict__ kin, const float *__restrict__ vin, __half *__restrict__ kc,
                                                         __half *__restrict__ vc, int E, const int *__restrict__ n_past, const float *__restrict__ cos_tab,
                                                         const float *__restrict__ sin_tab, const Tables tb, float *__restrict__ out, const int *__restrict__ row_slot = nullptr,
                                                         size_t seq_stride = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CH = HD / 8, P = AT_THREADS / CH;
    const int h = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    int pos_;
    if (BATCHED) { const int slot = row_slot[t]; pos_ = n_past[slot]; kc += (size_t)slot * seq_stride; vc += (size_t)slot * seq_stride; }
    else pos_ = *n_past + t;
    const int pos = pos_, T = pos + 1;
    const int Tg = FUSED ? pos : T;                               // keys read from the global cache
    const int Tpad = (T + 7) & ~7;
    float *sc = reinterpret_cast<float *>(smem);                  // [Tpad]
    __half *ph = reinterpret_cast<__half *>(sc + Tpad);           // [Tpad]
    __half *qh = ph + Tpad;                                       // [HD]
    __half *knew = qh + HD, *vnew = knew + HD;                    // [HD] each
    float *part = reinterpret_cast<float *>(vnew + HD);           // [P][HD]
    __shared__ float s_red[AT_THREADS / 64];
    __shared__ double s_dred[AT_THREADS / 64];
    const float scale = 1.0f / sqrtf((float)HD);
    const size_t qo = (size_t)t * E + (size_t)h * HD;
    if (FUSED) {
        if (tid < HD / 2) {
            const int i = tid;
            const float c = cos_tab[(size_t)pos * (HD / 2) + i], s = sin_tab[(size_t)pos * (HD / 2) + i];
            const float q0 = q[qo + 2 * i], q1 = q[qo + 2 * i + 1], k0 = kin[qo + 2 * i], k1 = kin[qo + 2 * i + 1];
            const __half2 qr = __floats2half2_rn(q0 * c - q1 * s, q0 * s + q1 * c), kr = __floats2half2_rn(k0 * c - k1 * s, k0 * s + k1 * c);
            const __half2 vr = __floats2half2_rn(vin[qo + 2 * i], vin[qo + 2 * i + 1]);
            *reinterpret_cast<__half2 *>(qh + 2 * i) = qr; *reinterpret_cast<__half2 *>(knew + 2 * i) = kr; *reinterpret_cast<__half2 *>(vnew + 2 * i) = vr;
            const size_t co = (size_t)pos * E + (size_t)h * HD + 2 * i;
            *reinterpret_cast<__half2 *>(kc + co) = kr; *reinterpret_cast<__half2 *>(vc + co) = vr;
        }
    } else {
        for (int i = tid; i < HD; i += AT_THREADS) qh[i] = f2h_rn(q[qo + i]);
    }
    __syncthreads();
    unsigned qreg[HD / 2];
#pragma unroll
    for (int i = 0; i < HD / 8; i++) { const int4 v4 = *reinterpret_cast<const int4 *>(qh + 8 * i); qreg[4 * i] = (unsigned)v4.x; qreg[4 * i + 1] = (unsigned)v4.y; qreg[4 * i + 2] = (unsigned)v4.z; qreg[4 * i + 3] = (unsigned)v4.w; }
    auto dot_row = [&](const __half *kr) {
        int4 kk[HD / 8];
#pragma unroll
        for (int i = 0; i < HD / 8; i++) kk[i] = ld16(kr + 8 * i);
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < HD / 8; i++) {
            const unsigned w[4] = {(unsigned)kk[i].x, (unsigned)kk[i].y, (unsigned)kk[i].z, (unsigned)kk[i].w};
#pragma unroll
            for (int e = 0; e < 4; e++) { s = fmaf(h2f_bits(w[e] & 0xFFFF), h2f_bits(qreg[4 * i + e] & 0xFFFF), s); s = fmaf(h2f_bits(w[e] >> 16), h2f_bits(qreg[4 * i + e] >> 16), s); }
        }
        return s * scale;
    };
    // (Measured and not adopted, profiles/r02m_bench_n1.json vs r02k: requesting these K / V rows at kernel entry, before the RoPE prologue and its barrier, with the exp
    // table's live part in LDS and the KV append moved behind the last barrier -- 11.7 us per launch at a context of 430 against 10.9 us for this form; beside LDS-DMA
    // hipcc waits vmcnt(0) for every ordinary load, so the prologue sat out the whole prefetch.)
    // Scores of the cached keys: 16 consecutive lanes share one key row (lane c holds its dims 8 c .. 8 c + 7 -- one 256-byte row per 16 lanes, four whole rows per
    // wave instruction; the round-1 form, a whole row per lane, asked the address path for 64 different cache lines per instruction and grew by ~0.025 us per key), the
    // 8-dim partial dots are added across the 16 lanes with DPP.  Key and value rows of the same (lane, round) sit at the same offset of the two caches, and neither
    // depends on this step's scores: both are requested here, NPRE rounds deep, so they arrive during the dot products / the softmax.
    const int c = tid % CH, p = tid / CH;
    const __half *kb = kc + (size_t)h * HD + 8 * c, *vb = vc + (size_t)h * HD + 8 * c;
    constexpr int NPRE = 16;                        // x P = 32 key partitions: contexts up to 512 need no second round trip
    int4 kpre[NPRE], vpre[NPRE];
#pragma unroll
    for (int i = 0; i < NPRE; i++) kpre[i] = ld16(kb + (size_t)min(p + i * P, max(Tg - 1, 0)) * E);   // clamped: never branches, never out of the cache
#pragma unroll
    for (int i = 0; i < NPRE; i++) vpre[i] = ld16(vb + (size_t)min(p + i * P, max(Tg - 1, 0)) * E);
    float qd[8];
    {
        const int4 q4 = *reinterpret_cast<const int4 *>(qh + 8 * c);
        const unsigned w[4] = {(unsigned)q4.x, (unsigned)q4.y, (unsigned)q4.z, (unsigned)q4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { qd[2 * e] = h2f_bits(w[e] & 0xFFFF); qd[2 * e + 1] = h2f_bits(w[e] >> 16); }
    }
    auto dot16 = [&](const int4 &kk) {              // all 16 lanes of the row return the key's score
        const unsigned w[4] = {(unsigned)kk.x, (unsigned)kk.y, (unsigned)kk.z, (unsigned)kk.w};
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) { s = fmaf(h2f_bits(w[e] & 0xFFFF), qd[2 * e], s); s = fmaf(h2f_bits(w[e] >> 16), qd[2 * e + 1], s); }
        s += dpp_f<0xB1>(s); s += dpp_f<0x4E>(s);                 // the CH = HD / 8 lanes of the key: 4 (HD 32), 8 (HD 64: half a DPP row) or 16 (HD 128: a DPP row)
        if (CH >= 8) s += dpp_f<0x141>(s);
        if (CH >= 16) s += dpp_f<0x140>(s);
        return s * scale;
    };
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPRE; i++) { const int j = p + i * P; const float s = dot16(kpre[i]); if (j < Tg) { if (c == 0) sc[j] = s; mx = fmaxf(mx, s); } }
    for (int j0 = p + NPRE * P; j0 < Tg; j0 += 8 * P) {      // beyond the prefetch: 8 rows per round trip
        int4 kk[8];
#pragma unroll
        for (int i = 0; i < 8; i++) kk[i] = ld16(kb + (size_t)min(j0 + i * P, Tg - 1) * E);
#pragma unroll
        for (int i = 0; i < 8; i++) { const int j = j0 + i * P; const float s = dot16(kk[i]); if (j < Tg) { if (c == 0) sc[j] = s; mx = fmaxf(mx, s); } }
    }
    if (FUSED && tid == AT_THREADS - 1) { const float s = dot_row(knew); sc[pos] = s; mx = fmaxf(mx, s); }
    mx = wave_max(mx);
    if ((tid & 63) == 0) s_red[tid >> 6] = mx;
    __syncthreads();
    mx = s_red[0];
#pragma unroll
    for (int i = 1; i < AT_THREADS / 64; i++) mx = fmaxf(mx, s_red[i]);
    double sum = 0.0;
    for (int j = tid; j < T; j += AT_THREADS) { const float v = exp_h(tb.exp, sc[j] - mx); sc[j] = v; sum += (double)v; }
    sum = wave_sum_d(sum);
    if ((tid & 63) == 0) s_dred[tid >> 6] = sum;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < AT_THREADS / 64; i++) tot += s_dred[i];
    const float inv = (float)(1.0 / tot);
    for (int j = tid; j < T; j += AT_THREADS) ph[j] = f2h_rn(sc[j] * inv);
    __syncthreads();
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto pv_acc = [&](const int4 &vv, const int j) {
        const float pj = __half2float(ph[j]);
        const unsigned w[4] = {(unsigned)vv.x, (unsigned)vv.y, (unsigned)vv.z, (unsigned)vv.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { o[2 * e] = fmaf(h2f_bits(w[e] & 0xFFFF), pj, o[2 * e]); o[2 * e + 1] = fmaf(h2f_bits(w[e] >> 16), pj, o[2 * e + 1]); }
    };
#pragma unroll
    for (int i = 0; i < NPRE; i++) { const int j = p + i * P; if (j < Tg) pv_acc(vpre[i], j); }
    for (int j0 = p + NPRE * P; j0 < Tg; j0 += 8 * P) {
        int4 vv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) vv[i] = ld16(vb + (size_t)min(j0 + i * P, Tg - 1) * E);
#pragma unroll
        for (int i = 0; i < 8; i++) { const int j = j0 + i * P; if (j < Tg) pv_acc(vv[i], j); }
    }
    if (FUSED && p == P - 1) {
        const float pj = __half2float(ph[pos]);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = fmaf(__half2float(vnew[8 * c + e]), pj, o[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) part[p * HD + 8 * c + e] = o[e];
    __syncthreads();
    for (int i = tid; i < HD; i += AT_THREADS) { float s = 0.0f;
#pragma unroll 8
        for (int pp = 0; pp < P; pp++) s += part[pp * HD + i];
        out[qo + i] = s; }
}

template <int HD>
static void launch_attn_hd(float *q, const float *k, const float *v, __half *kc, __half *vc, int N, int n_head, const int *n_past, int n_ctx, const float *cos_tab,
                           const float *sin_tab, const Tables &tb, float *out, bool fused, hipStream_t s) {
    const int Tpad = (n_ctx + 7) & ~7;
    const size_t lds = (size_t)Tpad * 6 + (size_t)HD * 6 + (size_t)(AT_THREADS / (HD / 8)) * HD * 4 + 64;
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_attn_llm<HD, true>));
                 HIP_IGNORE(lds_optin_max(&k_attn_llm<HD, false>)); attr = true; }
    note_kernel("k_attn_llm<%d, %s, false>", HD, fused ? "true" : "false");
    if (fused) hipLaunchKernelGGL((k_attn_llm<HD, true>), dim3((unsigned)n_head, 1), dim3(AT_THREADS), lds, s, q, k, v, kc, vc, n_head * HD, n_past, cos_tab, sin_tab, tb, out, (const int *)nullptr, (size_t)0);
    else hipLaunchKernelGGL((k_attn_llm<HD, false>), dim3((unsigned)n_head, (unsigned)N), dim3(AT_THREADS), lds, s, q, k, v, kc, vc, n_head * HD, n_past, cos_tab, sin_tab, tb, out, (const int *)nullptr, (size_t)0);
}
template <int HD>
static void launch_attn_batched_hd(float *q, const float *k, const float *v, __half *kc, __half *vc, int B, int n_head, const int *n_past, const int *row_slot, size_t seq_stride,
                                   int n_ctx, const float *cos_tab, const float *sin_tab, const Tables &tb, float *out, hipStream_t s) {
    const int Tpad = (n_ctx + 7) & ~7;
    const size_t lds = (size_t)Tpad * 6 + (size_t)HD * 6 + (size_t)(AT_THREADS / (HD / 8)) * HD * 4 + 64;
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_attn_llm<HD, true, true>)); attr = true; }
    hipLaunchKernelGGL((k_attn_llm<HD, true, true>), dim3((unsigned)n_head, (unsigned)B), dim3(AT_THREADS), lds, s, q, k, v, kc, vc, n_head * HD, n_past, cos_tab, sin_tab, tb, out, row_slot,
                       seq_stride);
}
// Decode attention for B rows of B different conversations (RoPE + KV append fused): row t uses position n_past[row_slot[t]] and the caches of that conversation.
void launch_attn_llm_batched(float *q, const float *k, const float *v, __half *kcache, __half *vcache, int B, int n_head, int hd, const int *n_past, const int *row_slot,
                             size_t seq_stride, int n_ctx, const float *cos_tab, const float *sin_tab, const Tables &tb, float *out, hipStream_t s) {
    switch (hd) {
    case 32: launch_attn_batched_hd<32>(q, k, v, kcache, vcache, B, n_head, n_past, row_slot, seq_stride, n_ctx, cos_tab, sin_tab, tb, out, s); break;
    case 64: launch_attn_batched_hd<64>(q, k, v, kcache, vcache, B, n_head, n_past, row_slot, seq_stride, n_ctx, cos_tab, sin_tab, tb, out, s); break;
    case 128: launch_attn_batched_hd<128>(q, k, v, kcache, vcache, B, n_head, n_past, row_slot, seq_stride, n_ctx, cos_tab, sin_tab, tb, out, s); break;
    default: throw HipError{hipErrorInvalidValue, "unsupported head size", __FILE__, __LINE__};
    }
}
// =====================================================================================================================
// Key-split decode attention (long contexts): k_attn_llm above puts ONE workgroup on a head -- 40 workgroups on a 256-CU chip, 0.015 us per cached key, 35 us per layer
// at 2048 keys.  Here S workgroups share a head's keys (n_head x S >= 240 workgroups) in two launches; nothing spins, so nothing can hang:
//   k_attn_split_scores : (head, split) -> RoPE of q / k (every workgroup; split 0 appends k, v to the cache), scores of its key range into a per-head fp32 row in HBM
//                         (16 lanes per 256-byte key row, DPP reduction -- the arithmetic of k_attn_llm), the new key's score by the last split;
//   k_attn_split_pv     : (head, split) -> every workgroup reads the head's whole score row (T x 4 bytes), max, fp16-table exp, exact sum, 1 / sum (identical in all S
//                         workgroups: the same values in, order-independent exact arithmetic); probabilities rounded to fp16 and P.V over ITS key range; the partial
//                         output goes out as device-scope (sc1) stores, an arrival counter per head names the last workgroup, and that one adds the S partials in
//                         split order (deterministic) and resets the counter for the next launch.
// Softmax semantics are those of k_attn_llm (global max BEFORE the table lookups: the fp16 exp table does not allow the flash-decoding rescale).
// =====================================================================================================================
constexpr int AS_THREADS = 256;
template <int HD>
__global__ __launch_bounds__(AS_THREADS) void k_attn_split_scores(const float *__restrict__ q, const float *__restrict__ kin, const float *__restrict__ vin, __half *__restrict__ kc,
                                                                   __half *__restrict__ vc, int E, const int *__restrict__ n_past, const float *__restrict__ cos_tab,
                                                                   const float *__restrict__ sin_tab, float *__restrict__ scores, int ld_scores, __half *__restrict__ qrot) {
    constexpr int CH = HD / 8, P = AS_THREADS / CH;
    constexpr int NR = 16;                                        // key rows per lane group in flight (16 x P x 256 B = 64 KiB per workgroup at HD 128)
    const int h = blockIdx.x, sp = blockIdx.y, S = gridDim.y, tid = threadIdx.x;
    const int pos = *n_past, Tg = pos;                            // keys 0 .. pos - 1 come from the cache, key pos is this step's
    __shared__ __attribute__((aligned(16))) __half qh[HD];
    __shared__ __attribute__((aligned(16))) __half knew[HD];
    const float scale = 1.0f / sqrtf((float)HD);
    const size_t qo = (size_t)h * HD;
    const int c = tid % CH, p = tid / CH;
    const int per = ((Tg + S - 1) / S + P - 1) / P * P;           // keys per split, a whole number of passes
    const int j_lo = sp * per, j_hi = min(Tg, j_lo + per), j_cl = max(j_hi - 1, 0);
    const __half *kb = kc + (size_t)h * HD + 8 * c;
    // the first NR key rows of this lane group do not depend on q: requested before the RoPE prologue and its barrier
    int4 kk[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) kk[i] = ld16(kb + (size_t)min(j_lo + p + i * P, j_cl) * E);
    if (tid < HD / 2) {
        const int i = tid;
        const float cs = cos_tab[(size_t)pos * (HD / 2) + i], sn = sin_tab[(size_t)pos * (HD / 2) + i];
        const float q0 = q[qo + 2 * i], q1 = q[qo + 2 * i + 1], k0 = kin[qo + 2 * i], k1 = kin[qo + 2 * i + 1];
        const __half2 qr = __halves2half2(f2h_rn(q0 * cs - q1 * sn), f2h_rn(q0 * sn + q1 * cs)), kr = __halves2half2(f2h_rn(k0 * cs - k1 * sn), f2h_rn(k0 * sn + k1 * cs));
        *reinterpret_cast<__half2 *>(qh + 2 * i) = qr; *reinterpret_cast<__half2 *>(knew + 2 * i) = kr;
        if (sp == 0) {
            const __half2 vr = __halves2half2(f2h_rn(vin[qo + 2 * i]), f2h_rn(vin[qo + 2 * i + 1]));
            const size_t co = (size_t)pos * E + (size_t)h * HD + 2 * i;
            *reinterpret_cast<__half2 *>(kc + co) = kr; *reinterpret_cast<__half2 *>(vc + co) = vr;
            *reinterpret_cast<__half2 *>(qrot + qo + 2 * i) = qr;   // (diagnostics / tests)
        }
    }
    __syncthreads();
    float qd[8];
    {
        const int4 q4 = *reinterpret_cast<const int4 *>(qh + 8 * c);
        const unsigned w[4] = {(unsigned)q4.x, (unsigned)q4.y, (unsigned)q4.z, (unsigned)q4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { qd[2 * e] = h2f_bits(w[e] & 0xFFFF); qd[2 * e + 1] = h2f_bits(w[e] >> 16); }
    }
    auto dot16 = [&](const int4 &kv) {
        const unsigned w[4] = {(unsigned)kv.x, (unsigned)kv.y, (unsigned)kv.z, (unsigned)kv.w};
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) { s = fmaf(h2f_bits(w[e] & 0xFFFF), qd[2 * e], s); s = fmaf(h2f_bits(w[e] >> 16), qd[2 * e + 1], s); }
        s += dpp_f<0xB1>(s); s += dpp_f<0x4E>(s);
        if (CH >= 8) s += dpp_f<0x141>(s);
        if (CH >= 16) s += dpp_f<0x140>(s);
        return s * scale;
    };
    float *srow = scores + (size_t)h * ld_scores;
    for (int j0 = j_lo + p; j0 < j_hi; j0 += NR * P) {
        int4 nx[NR];
        const bool more = j0 + NR * P < j_hi;                     // wave-uniform up to the lane group; clamped loads keep it branch-free
#pragma unroll
        for (int i = 0; i < NR; i++) nx[i] = ld16(kb + (size_t)min(j0 + (NR + i) * P, j_cl) * E);
#pragma unroll
        for (int i = 0; i < NR; i++) { const int j = j0 + i * P; const float s = dot16(kk[i]); if (j < j_hi && c == 0) srow[j] = s; }
        (void)more;
#pragma unroll
        for (int i = 0; i < NR; i++) kk[i] = nx[i];
    }
    if (sp == S - 1 && tid == 0) {                                // this step's key: one lane, the whole row in element order (k_attn_llm's dot_row)
        float s = 0.0f;
        for (int i = 0; i < HD; i++) s = fmaf(__half2float(knew[i]), __half2float(qh[i]), s);
        srow[pos] = s * scale;
    }
}
template <int HD>
__global__ __launch_bounds__(AS_THREADS) void k_attn_split_pv(const float *__restrict__ scores, int ld_scores, const __half *__restrict__ vc, int E, const int *__restrict__ n_past,
                                                               const Tables tb, float *__restrict__ partial, unsigned *__restrict__ arrive, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sp[];
    constexpr int CH = HD / 8, P = AS_THREADS / CH;
    const int h = blockIdx.x, sp = blockIdx.y, S = gridDim.y, tid = threadIdx.x;
    const int pos = *n_past, T = pos + 1;
    float *sc = reinterpret_cast<float *>(smem_sp);               // [T] scores, then exp values
    float *part = sc + ((T + 3) & ~3);                            // [P][HD]
    __shared__ float s_red[AS_THREADS / 64];
    __shared__ double s_dred[AS_THREADS / 64];
    __shared__ int s_last;
    constexpr int NR = 16;
    // this split's key range of the T keys (key pos included: it was appended by the score launch); its first NR value rows do not depend on the softmax: requested first
    const int per = ((T + S - 1) / S + P - 1) / P * P;
    const int j_lo = sp * per, j_hi = min(T, j_lo + per), j_cl = max(j_hi - 1, 0);
    const int c = tid % CH, p = tid / CH;
    const __half *vb = vc + (size_t)h * HD + 8 * c;
    int4 vv[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) vv[i] = ld16(vb + (size_t)min(j_lo + p + i * P, j_cl) * E);
    const float *srow = scores + (size_t)h * ld_scores;
    float mx = -INFINITY;
    for (int j = tid; j < T; j += AS_THREADS) { const float s = srow[j]; sc[j] = s; mx = fmaxf(mx, s); }
    mx = wave_max(mx);
    if ((tid & 63) == 0) s_red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    double sum = 0.0;
    for (int j0 = tid; j0 < T; j0 += 4 * AS_THREADS) {            // four independent table gathers in flight per thread
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = j0 + u * AS_THREADS; v[u] = j < T ? exp_h(tb.exp, sc[j] - mx) : 0.0f; }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = j0 + u * AS_THREADS; if (j < T) { sc[j] = v[u]; sum += (double)v[u]; } }   // fp16 values: the double sum is exact in any order
    }
    sum = wave_sum_d(sum);
    if ((tid & 63) == 0) s_dred[tid >> 6] = sum;
    __syncthreads();
    const float inv = (float)(1.0 / (((s_dred[0] + s_dred[1]) + s_dred[2]) + s_dred[3]));
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j0 = j_lo + p; j0 < j_hi; j0 += NR * P) {
        int4 nx[NR];
#pragma unroll
        for (int i = 0; i < NR; i++) nx[i] = ld16(vb + (size_t)min(j0 + (NR + i) * P, j_cl) * E);
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const int j = j0 + i * P;
            if (j < j_hi) {
                const float pj = __half2float(f2h_rn(sc[j] * inv));
                const unsigned w[4] = {(unsigned)vv[i].x, (unsigned)vv[i].y, (unsigned)vv[i].z, (unsigned)vv[i].w};
#pragma unroll
                for (int e = 0; e < 4; e++) { o[2 * e] = fmaf(h2f_bits(w[e] & 0xFFFF), pj, o[2 * e]); o[2 * e + 1] = fmaf(h2f_bits(w[e] >> 16), pj, o[2 * e + 1]); }
            }
        }
#pragma unroll
        for (int i = 0; i < NR; i++) vv[i] = nx[i];
    }
#pragma unroll
    for (int e = 0; e < 8; e++) part[p * HD + 8 * c + e] = o[e];
    __syncthreads();
    float *mine = partial + ((size_t)h * S + sp) * HD;
    for (int i = tid; i < HD; i += AS_THREADS) {
        float s = 0.0f;
#pragma unroll 8
        for (int pp = 0; pp < P; pp++) s += part[pp * HD + i];
        __hip_atomic_store(mine + i, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // device-scope (sc1) store: visible to a reader on another XCD without a cache flush
    }
    // Hand-off form (cdna_hip_programming.md Guideline 16, R1; MI355X_MICROARCH.md "Valid forms"): payload by write-through (sc1) stores, EVERY storing thread drains
    // vmcnt(0), workgroup barrier, ONE lane's agent-scope arrival; the reader (below) takes the payload with sc1 loads, which the guide lists as replacing the acquire when
    // the producer stored sc1.  A release on the arrival would add buffer_wbl2 (~1.7 us) to every one of the n_head x S workgroups of every layer.  Exercised across XCDs
    // at the real head count by tests/test_gpu_parity.py::test_key_split_attention_at_the_13b_head_count_is_bit_identical.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this thread's partial stores have completed ...
    __syncthreads();                                              // ... and so have everybody's, before the arrival is counted
    if (tid == 0) s_last = __hip_atomic_fetch_add(arrive + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(S - 1);
    __syncthreads();
    if (!s_last) return;
    for (int i = tid; i < HD; i += AS_THREADS) {                  // the last workgroup of the head: the S partials in split order
        float s = 0.0f;
        for (int k = 0; k < S; k++) s += __hip_atomic_load(partial + ((size_t)h * S + k) * HD + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out[(size_t)h * HD + i] = s;
    }
    if (tid == 0) __hip_atomic_store(arrive + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every other workgroup of this head has arrived: ready for the next launch
}
size_t attn_split_workspace_bytes(int n_head, int hd, int n_ctx, int splits) {
    return (size_t)n_head * (size_t)((n_ctx + 3) & ~3) * 4 + (size_t)n_head * splits * hd * 4 + (size_t)n_head * hd * 2 + (size_t)n_head * 4 + 1024;
}
int attn_split_count(int n_head, int cus) { return std::max(2, std::min(16, (cus - cus / 16) / std::max(1, n_head))); }   // ~240 of 256 CUs: 6 splits for 40 heads, 7 for 32
template <int HD>
static void launch_attn_split_hd(float *q, const float *k, const float *v, __half *kc, __half *vc, int n_head, const int *n_past, int n_ctx, const float *cos_tab, const float *sin_tab,
                                 const Tables &tb, float *out, void *ws, int splits, hipStream_t s) {
    const int ld = (n_ctx + 3) & ~3;
    float *scores = reinterpret_cast<float *>(ws);
    float *partial = scores + (size_t)n_head * ld;
    __half *qrot = reinterpret_cast<__half *>(partial + (size_t)n_head * splits * HD);
    unsigned *arrive = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(qrot) + (((size_t)n_head * HD * 2 + 255) & ~(size_t)255));
    note_kernel("k_attn_split_scores<%d>", HD);
    hipLaunchKernelGGL((k_attn_split_scores<HD>), dim3((unsigned)n_head, (unsigned)splits), dim3(AS_THREADS), 0, s, q, k, v, kc, vc, n_head * HD, n_past, cos_tab, sin_tab, scores, ld, qrot);
    const size_t lds = (size_t)ld * 4 + (size_t)(AS_THREADS / (HD / 8)) * HD * 4 + 64;
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_attn_split_pv<HD>)); attr = true; }
    note_kernel("k_attn_split_pv<%d>", HD);
    hipLaunchKernelGGL((k_attn_split_pv<HD>), dim3((unsigned)n_head, (unsigned)splits), dim3(AS_THREADS), lds, s, scores, ld, vc, n_head * HD, n_past, tb, partial, arrive, out);
}
// decode (one row): RoPE + KV append + attention with the keys of every head shared by `splits` workgroups; `ws` >= attn_split_workspace_bytes(...), its last 1 KiB zeroed once
void launch_attn_llm_split(float *q, const float *k, const float *v, __half *kcache, __half *vcache, int n_head, int hd, const int *n_past, int n_ctx, const float *cos_tab,
                           const float *sin_tab, const Tables &tb, float *out, void *ws, int splits, hipStream_t s) {
    switch (hd) {
    case 32: launch_attn_split_hd<32>(q, k, v, kcache, vcache, n_head, n_past, n_ctx, cos_tab, sin_tab, tb, out, ws, splits, s); break;
    case 64: launch_attn_split_hd<64>(q, k, v, kcache, vcache, n_head, n_past, n_ctx, cos_tab, sin_tab, tb, out, ws, splits, s); break;
    case 128: launch_attn_split_hd<128>(q, k, v, kcache, vcache, n_head, n_past, n_ctx, cos_tab, sin_tab, tb, out, ws, splits, s); break;
    default: throw HipError{hipErrorInvalidValue, "unsupported head size", __FILE__, __LINE__};
    }
}

// =====================================================================================================================
// Prefill attention (N > 1 query rows of one conversation) on the exact-f32 matrix cores.  The per-token kernel above re-reads a head's whole K and V once per
// query (142 x 40 workgroups per layer for the image-turn prompt: 82 us per layer); here a workgroup owns (head, 16 consecutive queries) and streams the keys the
// LAST of its queries may see through LDS in tiles of 64 -- K once for the scores, V once for the output.
//   scores : v_mfma_f32_16x16x4_f32 over the head dim; operands are the fp16 values (q rounded to fp16, cached k) widened to fp32, so every product is exact and
//            the accumulator is the k-ordered fp32 fma chain of k_attn_llm's dot_row -- bit-identical scores; then * 1/sqrt(hd)
//   softmax: per query over its own causal range: max, fp16-table exp, exact double sum, probabilities rounded to fp16 (as k_attn_llm)
//   output : P V on the same MFMA, keys in index order (k_attn_llm adds key partitions in another order: fp32 rounding only)
// =====================================================================================================================
typedef float pf4_t __attribute__((ext_vector_type(4)));
constexpr int AP_QT = 16, AP_KT = 64;
template <int HD>
__device__ __forceinline__ void ap_stage(float *kv, const __half *__restrict__ cache, int E, int h, int key0, int T, int tid) {
    constexpr int C8 = HD / 8, LDV = HD + 1, PER = AP_KT * C8 / 256;     // 16-byte pieces per thread
    int4 x[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) { const int e = tid + 256 * u, j = e / C8, c = e - j * C8; x[u] = ld16(cache + (size_t)min(key0 + j, T - 1) * E + (size_t)h * HD + 8 * c); }
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int e = tid + 256 * u, j = e / C8, c = e - j * C8;
        const bool live = key0 + j < T;
        const unsigned w[4] = {(unsigned)x[u].x, (unsigned)x[u].y, (unsigned)x[u].z, (unsigned)x[u].w};
        float *d = kv + j * LDV + 8 * c;
#pragma unroll
        for (int i = 0; i < 4; i++) { d[2 * i] = live ? h2f_bits(w[i] & 0xFFFF) : 0.0f; d[2 * i + 1] = live ? h2f_bits(w[i] >> 16) : 0.0f; }
    }
}
// QS = query sub-tiles of 16 per workgroup: 2 from 256 prompt rows on -- the staged key / value tile (the fp16 -> fp32 conversion and its LDS writes are most of the
// kernel's time) then serves 32 queries instead of 16.
template <int HD, int QS>
__global__ __launch_bounds__(256) void k_attn_prefill(const float *__restrict__ q, const __half *__restrict__ kc, const __half *__restrict__ vc, int E, int N, const int *__restrict__ n_past,
                                                      const Tables tb, float *__restrict__ out, int LS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ap[];
    constexpr int LDV = HD + 1, KS = HD / 4, DT = HD / 16, QT = AP_QT * QS;
    static_assert(DT % 4 == 0 || DT == 2, "dim tiles are dealt to the 4 waves");
    float *S = reinterpret_cast<float *>(smem_ap);                // [QT][LS]
    float *kv = S + (size_t)QT * LS;                              // [64][LDV]
    const int h = blockIdx.x, q0 = blockIdx.y * QT, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int np = *n_past;
    const int T = np + min(q0 + QT - 1, N - 1) + 1;               // keys the last query of this tile sees
    const int nkt = (T + AP_KT - 1) / AP_KT;
    const float scale = 1.0f / sqrtf((float)HD);
    // Q fragments: A[i = lane & 15 (query)][kk = lane >> 4] per k-step, rounded to fp16 like ggml's f16 x f32 mul_mat
    float qf[QS][KS];
#pragma unroll
    for (int qs = 0; qs < QS; qs++) {
        const float *qp = q + (size_t)min(q0 + 16 * qs + (lane & 15), N - 1) * E + (size_t)h * HD + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[qs][ks] = f16r(qp[4 * ks]);
    }
    for (int kt = 0; kt < nkt; kt++) {
        __syncthreads();
        ap_stage<HD>(kv, kc, E, h, kt * AP_KT, T, tid);
        __syncthreads();
        const int key0 = kt * AP_KT + 16 * wave;
        if (key0 < T) {
            const float *kb = kv + (size_t)(16 * wave + (lane & 15)) * LDV + (lane >> 4);
            pf4_t acc[QS];
#pragma unroll
            for (int qs = 0; qs < QS; qs++) acc[qs] = pf4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const float kval = kb[4 * ks];
#pragma unroll
                for (int qs = 0; qs < QS; qs++) acc[qs] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[qs][ks], kval, acc[qs], 0, 0, 0);
            }
#pragma unroll
            for (int qs = 0; qs < QS; qs++)
#pragma unroll
                for (int r = 0; r < 4; r++) S[(size_t)(16 * qs + (lane >> 4) * 4 + r) * LS + key0 + (lane & 15)] = acc[qs][r] * scale;
        }
    }
    __syncthreads();
#pragma unroll
    for (int qs = 0; qs < QS; qs++) {   // softmax: 16 lanes per query row; row r sees keys 0 .. np + q0 + r
        const int row = 16 * qs + (tid >> 4), sub = tid & 15;
        const int Tq = min(np + q0 + row + 1, T);
        float *sr = S + (size_t)row * LS;
        float mx = -INFINITY;
        for (int j = sub; j < Tq; j += 16) mx = fmaxf(mx, sr[j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4)); mx = fmaxf(mx, __shfl_xor(mx, 8));
        double sum = 0.0;
        for (int j0 = sub; j0 < Tq; j0 += 64) {                  // table gathers in batches of 4
            float e[4];
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = tab(tb.exp, sr[min(j0 + 16 * u, Tq - 1)] - mx);
#pragma unroll
            for (int u = 0; u < 4; u++) { const int j = j0 + 16 * u; if (j < Tq) { sr[j] = e[u]; sum += (double)e[u]; } }
        }
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
        const float inv = (float)(1.0 / sum);
        for (int j = sub; j < nkt * AP_KT; j += 16) sr[j] = j < Tq ? f16r(sr[j] * inv) : 0.0f;
    }
    // O = P V: wave w owns dim tiles w, w + 4, ..; accumulators persist across the key tiles
    constexpr int DPW = (DT + 3) / 4;
    pf4_t oacc[QS][DPW];
#pragma unroll
    for (int qs = 0; qs < QS; qs++)
#pragma unroll
        for (int i = 0; i < DPW; i++) oacc[qs][i] = pf4_t{0.0f, 0.0f, 0.0f, 0.0f};
    for (int kt = 0; kt < nkt; kt++) {
        __syncthreads();
        ap_stage<HD>(kv, vc, E, h, kt * AP_KT, T, tid);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DPW; i++) {
            const int dt = wave + 4 * i;
            if (dt < DT) {
                const float *vb = kv + (size_t)(lane >> 4) * LDV + dt * 16 + (lane & 15);
#pragma unroll
                for (int ks = 0; ks < AP_KT / 4; ks++) {
                    const float vval = vb[(size_t)4 * ks * LDV];
#pragma unroll
                    for (int qs = 0; qs < QS; qs++)
                        oacc[qs][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(S[(size_t)(16 * qs + (lane & 15)) * LS + kt * AP_KT + (lane >> 4) + 4 * ks], vval, oacc[qs][i], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int qs = 0; qs < QS; qs++)
#pragma unroll
        for (int i = 0; i < DPW; i++) {
            const int dt = wave + 4 * i;
            if (dt < DT) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int qrow = q0 + 16 * qs + (lane >> 4) * 4 + r;
                    if (qrow < N) out[(size_t)qrow * E + (size_t)h * HD + dt * 16 + (lane & 15)] = oacc[qs][i][r];
                }
            }
        }
}
// The same on the fp16 matrix cores (round 3): every operand of both products IS an fp16 value in the reference (K / V cache rows, q rounded to fp16 by ggml's f16 x f32
// mul_mat, probabilities rounded to fp16 before P.V), so v_mfma_f32_16x16x32_f16 multiplies them exactly and accumulates in fp32 -- 16x the rate of the f32 MFMA above and
// no fp16 -> fp32 staging pass.  K tiles go to LDS as they are (16-byte chunks XOR-swizzled by the key index: conflict-free fragment reads); V tiles are written
// TRANSPOSED ([dim][key], 70-half rows) so that a lane's 8 consecutive keys of one dim are contiguous.  Softmax as above.  fp32 accumulation order inside the MFMA differs
// from the sequential chain of k_attn_llm / the oracle (fp32 rounding only; MINIGPT4_PARITY uses k_attn_ref).
typedef _Float16 aph8_t __attribute__((ext_vector_type(8)));
constexpr int APH_LDT = 70;
template <int HD, int QS>
__global__ __launch_bounds__(256) void k_attn_prefill_h(const float *__restrict__ q, const __half *__restrict__ kc, const __half *__restrict__ vc, int E, int N, const int *__restrict__ n_past,
                                                        const Tables tb, float *__restrict__ out, int LS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_aph[];
    constexpr int C8 = HD / 8, QT = AP_QT * QS, KSQ = HD / 32, DT = HD / 16, PER = AP_KT * C8 / 256;
    static_assert(PER >= 1 && (DT % 4 == 0 || DT == 2), "tile geometry");
    float *S = reinterpret_cast<float *>(smem_aph);               // [QT][LS], LS = 4 mod 64
    __half *Kt = reinterpret_cast<__half *>(S + (size_t)QT * LS); // [64][HD], chunk c of row j at slot c ^ (j & (C8 - 1))
    __half *Vt = Kt + AP_KT * HD;                                 // [HD][APH_LDT]
    // causal: the LAST query tile sees the most keys -- it is dispatched first (longest-first), so the short tiles fill the tail of the launch
    const int h = blockIdx.x, q0 = (int)(gridDim.y - 1 - blockIdx.y) * QT, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    MG4_TLP(0);
    const int np = *n_past;
    const int T = np + min(q0 + QT - 1, N - 1) + 1;               // keys the last query of this tile sees
    const int nkt = (T + AP_KT - 1) / AP_KT;
    const float scale = 1.0f / sqrtf((float)HD);
    const int l15 = lane & 15, l4 = lane >> 4;
    aph8_t qf[QS][KSQ];                                           // A fragments: query l15, dims 32 ks + 8 l4 .. + 7, rounded to fp16
#pragma unroll
    for (int qs = 0; qs < QS; qs++) {
        const float *qp = q + (size_t)min(q0 + 16 * qs + l15, N - 1) * E + (size_t)h * HD + 8 * l4;
#pragma unroll
        for (int ks = 0; ks < KSQ; ks++) {
            const float4 a = *reinterpret_cast<const float4 *>(qp + 32 * ks), b = *reinterpret_cast<const float4 *>(qp + 32 * ks + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; e++) qf[qs][ks][e] = (_Float16)__half2float(f2h_rn(v[e]));
        }
    }
    // the global loads of tile kt + 1 are issued before the MFMAs of tile kt (register double buffer): one exposed memory round trip per phase instead of one per tile
    int4 xk[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) { const int e = tid + 256 * u, j = e / C8, c = e - j * C8; xk[u] = ld16(kc + (size_t)min(j, T - 1) * E + (size_t)h * HD + 8 * c); }
    for (int kt = 0; kt < nkt; kt++) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++) { const int e = tid + 256 * u, j = e / C8, c = e - j * C8; *reinterpret_cast<int4 *>(Kt + j * HD + ((c ^ (j & (C8 - 1))) << 3)) = xk[u]; }
#pragma unroll
        for (int u = 0; u < PER; u++) { const int e = tid + 256 * u, j = e / C8, c = e - j * C8; xk[u] = ld16(kc + (size_t)min((kt + 1) * AP_KT + j, T - 1) * E + (size_t)h * HD + 8 * c); }
        __syncthreads();
        const int key0 = kt * AP_KT + 16 * wave;
        if (key0 < T) {
            const int row = 16 * wave + l15;
            pf4_t acc[QS];
#pragma unroll
            for (int qs = 0; qs < QS; qs++) acc[qs] = pf4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < KSQ; ks++) {
                const aph8_t kf = *reinterpret_cast<const aph8_t *>(Kt + row * HD + (((4 * ks + l4) ^ (row & (C8 - 1))) << 3));
#pragma unroll
                for (int qs = 0; qs < QS; qs++) acc[qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[qs][ks], kf, acc[qs], 0, 0, 0);
            }
#pragma unroll
            for (int qs = 0; qs < QS; qs++)
#pragma unroll
                for (int r = 0; r < 4; r++) S[(size_t)(16 * qs + l4 * 4 + r) * LS + key0 + l15] = acc[qs][r] * scale;
        }
    }
    __syncthreads();
    MG4_TLP(1);
#pragma unroll
    for (int qs = 0; qs < QS; qs++) {
        // softmax: 16 lanes per query row, lane `sub` owns the keys 64 s + 4 sub .. + 3 (16-byte LDS accesses); row r sees keys 0 .. np + q0 + r.  Same values as
        // k_attn_prefill's loop (max, fp16-table exp, exact double sum -- order-free --, p = fp16(e * (1 / sum))); what changed in round 3 is the shape: the timeline
        // had this phase at 15 of a workgroup's 35 us -- 16 dependent rounds of 4 table gathers per row pass -- now 16 gathers are in flight per round.
        const int row = 16 * qs + (tid >> 4), sub = tid & 15;
        const int Tq = min(np + q0 + row + 1, T), nb = nkt * AP_KT;
        float *sr = S + (size_t)row * LS;
        float mx = -INFINITY;
        for (int j = 4 * sub; j < nb; j += 64) {
            const float4 v = *reinterpret_cast<const float4 *>(sr + j);
            mx = fmaxf(mx, j < Tq ? v.x : -INFINITY); mx = fmaxf(mx, j + 1 < Tq ? v.y : -INFINITY); mx = fmaxf(mx, j + 2 < Tq ? v.z : -INFINITY); mx = fmaxf(mx, j + 3 < Tq ? v.w : -INFINITY);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4)); mx = fmaxf(mx, __shfl_xor(mx, 8));
        double sum = 0.0;
        for (int j0 = 0; j0 < nb; j0 += 256) {
            float x[16]; unsigned short t[16];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int jj = min(j0 + 64 * u, nb - 64) + 4 * sub;          // a block past the end repeats the last one (its values are dropped below)
                const float4 v = *reinterpret_cast<const float4 *>(sr + jj);
                x[4 * u] = v.x; x[4 * u + 1] = v.y; x[4 * u + 2] = v.z; x[4 * u + 3] = v.w;
            }
            // (the table stays in global memory: an LDS copy -- 39 KB by LDS-DMA per workgroup -- left this phase at 8.0 us against 7.7 and cost the 142-row launch its
            //  co-resident workgroups, 11.3 -> 15.1 us: the phase is bound by its ~30 vector instructions per element, not by the gathers)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int jj = j0 + 64 * (e >> 2) + 4 * sub + (e & 3);
                t[e] = reinterpret_cast<const unsigned short *>(tb.exp)[f2h_bits(jj < Tq ? x[e] - mx : 0.0f)];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int jj = j0 + 64 * u + 4 * sub;
                if (j0 + 64 * u < nb) {                                       // uniform over the 16 lanes of a row
                    float4 ev;
                    ev.x = jj < Tq ? h2f_bits(t[4 * u]) : 0.0f; ev.y = jj + 1 < Tq ? h2f_bits(t[4 * u + 1]) : 0.0f; ev.z = jj + 2 < Tq ? h2f_bits(t[4 * u + 2]) : 0.0f; ev.w = jj + 3 < Tq ? h2f_bits(t[4 * u + 3]) : 0.0f;
                    sum += (double)ev.x; sum += (double)ev.y; sum += (double)ev.z; sum += (double)ev.w;
                    *reinterpret_cast<float4 *>(sr + jj) = ev;
                }
            }
        }
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
        const float inv = (float)(1.0 / sum);
        for (int j = 4 * sub; j < nb; j += 64) {                             // keys past the row's limit already hold 0
            float4 v = *reinterpret_cast<const float4 *>(sr + j);
            v.x = f16r(v.x * inv); v.y = f16r(v.y * inv); v.z = f16r(v.z * inv); v.w = f16r(v.w * inv);
            *reinterpret_cast<float4 *>(sr + j) = v;
        }
    }
    MG4_TLP(2);
    constexpr int DPW = (DT + 3) / 4;
    pf4_t oacc[QS][DPW];
#pragma unroll
    for (int qs = 0; qs < QS; qs++)
#pragma unroll
        for (int i = 0; i < DPW; i++) oacc[qs][i] = pf4_t{0.0f, 0.0f, 0.0f, 0.0f};
    int4 xv[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) { const int e = tid + 256 * u, j = e / C8, c = e - j * C8; xv[u] = ld16(vc + (size_t)min(j, T - 1) * E + (size_t)h * HD + 8 * c); }
    for (int kt = 0; kt < nkt; kt++) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = tid + 256 * u, j = e / C8, c = e - j * C8;
            const bool live = kt * AP_KT + j < T;
            const unsigned w[4] = {(unsigned)xv[u].x, (unsigned)xv[u].y, (unsigned)xv[u].z, (unsigned)xv[u].w};
            unsigned short *d = reinterpret_cast<unsigned short *>(Vt) + (8 * c) * APH_LDT + j;
#pragma unroll
            for (int i = 0; i < 4; i++) { d[(2 * i) * APH_LDT] = live ? (unsigned short)(w[i] & 0xFFFF) : (unsigned short)0; d[(2 * i + 1) * APH_LDT] = live ? (unsigned short)(w[i] >> 16) : (unsigned short)0; }
        }
#pragma unroll
        for (int u = 0; u < PER; u++) { const int e = tid + 256 * u, j = e / C8, c = e - j * C8; xv[u] = ld16(vc + (size_t)min((kt + 1) * AP_KT + j, T - 1) * E + (size_t)h * HD + 8 * c); }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DPW; i++) {
            const int dt = wave + 4 * i;
            if (dt < DT) {
#pragma unroll
                for (int k2 = 0; k2 < AP_KT / 32; k2++) {
                    const unsigned *vp = reinterpret_cast<const unsigned *>(Vt + (16 * dt + l15) * APH_LDT + 32 * k2 + 8 * l4);   // 4-byte aligned: 140-byte rows
                    union { unsigned u[4]; aph8_t v; } vb;
#pragma unroll
                    for (int e = 0; e < 4; e++) vb.u[e] = vp[e];
#pragma unroll
                    for (int qs = 0; qs < QS; qs++) {
                        const float *pp = S + (size_t)(16 * qs + l15) * LS + kt * AP_KT + 32 * k2 + 8 * l4;
                        const float4 a = *reinterpret_cast<const float4 *>(pp), b = *reinterpret_cast<const float4 *>(pp + 4);
                        aph8_t pf;
                        pf[0] = (_Float16)a.x; pf[1] = (_Float16)a.y; pf[2] = (_Float16)a.z; pf[3] = (_Float16)a.w; pf[4] = (_Float16)b.x; pf[5] = (_Float16)b.y; pf[6] = (_Float16)b.z; pf[7] = (_Float16)b.w;
                        oacc[qs][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vb.v, oacc[qs][i], 0, 0, 0);
                    }
                }
            }
        }
    }
    MG4_TLP(3);
#pragma unroll
    for (int qs = 0; qs < QS; qs++)
#pragma unroll
        for (int i = 0; i < DPW; i++) {
            const int dt = wave + 4 * i;
            if (dt < DT) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int qrow = q0 + 16 * qs + l4 * 4 + r;
                    if (qrow < N) out[(size_t)qrow * E + (size_t)h * HD + dt * 16 + l15] = oacc[qs][i][r];
                }
            }
        }
#ifdef MG4_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG4_TLP(4);
#endif
}
// ---------------------------------------------------------------------------------------------------------------------
// k_attn_prefill_h8 (round 3): the same arithmetic as k_attn_prefill_h in a 512-thread workgroup whose waves have ROLES.  The timeline of the 4-wave form had a 512-key
// workgroup at 6.2 us of scores + 7.7 us of softmax + 12.7 us of P.V, each key tile paying [barrier, LDS write of the staged tile, barrier, MFMAs] in sequence on the same
// four waves.  Here waves 4..7 are LOADERS (global -> registers two tiles ahead -> LDS tile kt, transposing V) while waves 0..3 multiply tile kt - 1 out of the OTHER LDS
// buffer: one barrier per tile and the staging runs beside the MFMAs; the softmax uses all 512 threads (32 rows x 16 lanes at once).  QS = 2 only (32 queries).
// Every value is formed by the same operations in the same order as in k_attn_prefill_h (scores: one MFMA chain per (query group, key tile) over ks; softmax: order-free
// max / exact sum; P.V: tiles in key order), so the two kernels are bit-identical (MINIGPT4_ATTN_PREFILL_W8=0 selects the 4-wave form, A/B).
// ---------------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(512) void k_attn_prefill_h8(const float *__restrict__ q, const __half *__restrict__ kc, const __half *__restrict__ vc, int E, int N, const int *__restrict__ n_past,
                                                         const Tables tb, float *__restrict__ out, int LS, __half *__restrict__ out_h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_aph8[];
    constexpr int QS = 2, C8 = HD / 8, QT = AP_QT * QS, KSQ = HD / 32, DT = HD / 16, PER = AP_KT * C8 / 256;
    constexpr int KVB = (AP_KT * HD * 2 > HD * APH_LDT * 2 ? AP_KT * HD * 2 : HD * APH_LDT * 2);        // bytes of one K (or transposed V) tile buffer
    static_assert(PER >= 1 && (DT % 4 == 0 || DT == 2), "tile geometry");
    float *S = reinterpret_cast<float *>(smem_aph8);                                                 // [QT][LS]
    unsigned char *kvb = smem_aph8 + (size_t)QT * LS * 4;                                             // two tile buffers (K tiles, later V tiles)
    const int h = blockIdx.x, q0 = (int)(gridDim.y - 1 - blockIdx.y) * QT, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int lt = tid & 255, mw = wave & 3;                                                           // loader thread / MFMA wave index
    MG4_TLP(0);
    const int np = *n_past;
    const int T = np + min(q0 + QT - 1, N - 1) + 1;
    const int nkt = (T + AP_KT - 1) / AP_KT;
    const float scale = 1.0f / sqrtf((float)HD);
    const int l15 = lane & 15, l4 = lane >> 4;
    aph8_t qf[QS][KSQ];
    if (!loader) {
#pragma unroll
        for (int qs = 0; qs < QS; qs++) {
            const float *qp = q + (size_t)min(q0 + 16 * qs + l15, N - 1) * E + (size_t)h * HD + 8 * l4;
#pragma unroll
            for (int ks = 0; ks < KSQ; ks++) {
                const float4 a = *reinterpret_cast<const float4 *>(qp + 32 * ks), b = *reinterpret_cast<const float4 *>(qp + 32 * ks + 4);
                const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int e = 0; e < 8; e++) qf[qs][ks][e] = (_Float16)__half2float(f2h_rn(v[e]));
            }
        }
    }
    // ---- scores: iteration kt stages tile kt (loaders) and multiplies tile kt - 1 (MFMA waves)
    int4 xa[PER], xb[PER];                                                                            // loaders: tiles kt and kt + 1 in flight
#define MG4_KV_LOAD(X, base, tile)                                                                          \
    _Pragma("unroll") for (int u = 0; u < PER; u++) { const int e = lt + 256 * u, j = e / C8, c = e - j * C8;   \
        X[u] = ld16(base + (size_t)min((tile) * AP_KT + j, T - 1) * E + (size_t)h * HD + 8 * c); }
    if (loader) { MG4_KV_LOAD(xa, kc, 0) MG4_KV_LOAD(xb, kc, 1) }
    auto k_write = [&](const int4 (&X)[PER], int buf) {
        __half *Kt = reinterpret_cast<__half *>(kvb + (size_t)buf * KVB);
#pragma unroll
        for (int u = 0; u < PER; u++) { const int e = lt + 256 * u, j = e / C8, c = e - j * C8; *reinterpret_cast<int4 *>(Kt + j * HD + ((c ^ (j & (C8 - 1))) << 3)) = X[u]; }
    };
    auto k_mul = [&](int kt, int buf) {
        const __half *Kt = reinterpret_cast<const __half *>(kvb + (size_t)buf * KVB);
        const int key0 = kt * AP_KT + 16 * mw;
        if (key0 < T) {
            const int row = 16 * mw + l15;
            pf4_t acc[QS];
#pragma unroll
            for (int qs = 0; qs < QS; qs++) acc[qs] = pf4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < KSQ; ks++) {
                const aph8_t kf = *reinterpret_cast<const aph8_t *>(Kt + row * HD + (((4 * ks + l4) ^ (row & (C8 - 1))) << 3));
#pragma unroll
                for (int qs = 0; qs < QS; qs++) acc[qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[qs][ks], kf, acc[qs], 0, 0, 0);
            }
#pragma unroll
            for (int qs = 0; qs < QS; qs++)
#pragma unroll
                for (int r = 0; r < 4; r++) S[(size_t)(16 * qs + l4 * 4 + r) * LS + key0 + l15] = acc[qs][r] * scale;
        }
    };
    for (int kt = 0; kt <= nkt; kt += 2) {
        if (loader) { if (kt < nkt) { k_write(xa, 0); MG4_KV_LOAD(xa, kc, kt + 2) } }
        else if (kt >= 1) k_mul(kt - 1, 1);
        __syncthreads();
        if (kt + 1 > nkt) break;
        if (loader) { if (kt + 1 < nkt) { k_write(xb, 1); MG4_KV_LOAD(xb, kc, kt + 3) } }
        else k_mul(kt, 0);
        __syncthreads();
    }
    MG4_TLP(1);
    // the first two V tiles are requested now: they arrive while the softmax runs
    if (loader) { MG4_KV_LOAD(xa, vc, 0) MG4_KV_LOAD(xb, vc, 1) }
    {   // softmax: 32 rows x 16 lanes at once (k_attn_prefill_h's code with qs = tid >> 8)
        const int row = tid >> 4, sub = tid & 15;
        const int Tq = min(np + q0 + row + 1, T), nb = nkt * AP_KT;
        float *sr = S + (size_t)row * LS;
        float mx = -INFINITY;
        for (int j = 4 * sub; j < nb; j += 64) {
            const float4 v = *reinterpret_cast<const float4 *>(sr + j);
            mx = fmaxf(mx, j < Tq ? v.x : -INFINITY); mx = fmaxf(mx, j + 1 < Tq ? v.y : -INFINITY); mx = fmaxf(mx, j + 2 < Tq ? v.z : -INFINITY); mx = fmaxf(mx, j + 3 < Tq ? v.w : -INFINITY);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4)); mx = fmaxf(mx, __shfl_xor(mx, 8));
        double sum = 0.0;
        for (int j0 = 0; j0 < nb; j0 += 256) {
            float x[16]; unsigned short t[16];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int jj = min(j0 + 64 * u, nb - 64) + 4 * sub;
                const float4 v = *reinterpret_cast<const float4 *>(sr + jj);
                x[4 * u] = v.x; x[4 * u + 1] = v.y; x[4 * u + 2] = v.z; x[4 * u + 3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int jj = j0 + 64 * (e >> 2) + 4 * sub + (e & 3);
                t[e] = reinterpret_cast<const unsigned short *>(tb.exp)[f2h_bits(jj < Tq ? x[e] - mx : 0.0f)];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int jj = j0 + 64 * u + 4 * sub;
                if (j0 + 64 * u < nb) {
                    float4 ev;
                    ev.x = jj < Tq ? h2f_bits(t[4 * u]) : 0.0f; ev.y = jj + 1 < Tq ? h2f_bits(t[4 * u + 1]) : 0.0f; ev.z = jj + 2 < Tq ? h2f_bits(t[4 * u + 2]) : 0.0f; ev.w = jj + 3 < Tq ? h2f_bits(t[4 * u + 3]) : 0.0f;
                    sum += (double)ev.x; sum += (double)ev.y; sum += (double)ev.z; sum += (double)ev.w;
                    *reinterpret_cast<float4 *>(sr + jj) = ev;
                }
            }
        }
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
        const float inv = (float)(1.0 / sum);
        for (int j = 4 * sub; j < nb; j += 64) {
            float4 v = *reinterpret_cast<const float4 *>(sr + j);
            v.x = f16r(v.x * inv); v.y = f16r(v.y * inv); v.z = f16r(v.z * inv); v.w = f16r(v.w * inv);
            *reinterpret_cast<float4 *>(sr + j) = v;
        }
    }
    MG4_TLP(2);
    // ---- P.V: the same two-role loop over the V tiles (written transposed: [dim][key])
    constexpr int DPW = (DT + 3) / 4;
    pf4_t oacc[QS][DPW];
#pragma unroll
    for (int qs = 0; qs < QS; qs++)
#pragma unroll
        for (int i = 0; i < DPW; i++) oacc[qs][i] = pf4_t{0.0f, 0.0f, 0.0f, 0.0f};
    auto v_write = [&](const int4 (&X)[PER], int kt, int buf) {
        __half *Vt = reinterpret_cast<__half *>(kvb + (size_t)buf * KVB);
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = lt + 256 * u, j = e / C8, c = e - j * C8;
            const bool live = kt * AP_KT + j < T;
            const unsigned w[4] = {(unsigned)X[u].x, (unsigned)X[u].y, (unsigned)X[u].z, (unsigned)X[u].w};
            unsigned short *d = reinterpret_cast<unsigned short *>(Vt) + (8 * c) * APH_LDT + j;
#pragma unroll
            for (int i = 0; i < 4; i++) { d[(2 * i) * APH_LDT] = live ? (unsigned short)(w[i] & 0xFFFF) : (unsigned short)0; d[(2 * i + 1) * APH_LDT] = live ? (unsigned short)(w[i] >> 16) : (unsigned short)0; }
        }
    };
    auto v_mul = [&](int kt, int buf) {
        const __half *Vt = reinterpret_cast<const __half *>(kvb + (size_t)buf * KVB);
#pragma unroll
        for (int i = 0; i < DPW; i++) {
            const int dt = mw + 4 * i;
            if (dt < DT) {
#pragma unroll
                for (int k2 = 0; k2 < AP_KT / 32; k2++) {
                    const unsigned *vp = reinterpret_cast<const unsigned *>(Vt + (16 * dt + l15) * APH_LDT + 32 * k2 + 8 * l4);
                    union { unsigned u[4]; aph8_t v; } vb;
#pragma unroll
                    for (int e = 0; e < 4; e++) vb.u[e] = vp[e];
#pragma unroll
                    for (int qs = 0; qs < QS; qs++) {
                        const float *pp = S + (size_t)(16 * qs + l15) * LS + kt * AP_KT + 32 * k2 + 8 * l4;
                        const float4 a = *reinterpret_cast<const float4 *>(pp), b = *reinterpret_cast<const float4 *>(pp + 4);
                        aph8_t pf;
                        pf[0] = (_Float16)a.x; pf[1] = (_Float16)a.y; pf[2] = (_Float16)a.z; pf[3] = (_Float16)a.w; pf[4] = (_Float16)b.x; pf[5] = (_Float16)b.y; pf[6] = (_Float16)b.z; pf[7] = (_Float16)b.w;
                        oacc[qs][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vb.v, oacc[qs][i], 0, 0, 0);
                    }
                }
            }
        }
    };
    __syncthreads();                                                     // normalised probabilities visible; the K buffers are free
    for (int kt = 0; kt <= nkt; kt += 2) {
        if (loader) { if (kt < nkt) { v_write(xa, kt, 0); MG4_KV_LOAD(xa, vc, kt + 2) } }
        else if (kt >= 1) v_mul(kt - 1, 1);
        __syncthreads();
        if (kt + 1 > nkt) break;
        if (loader) { if (kt + 1 < nkt) { v_write(xb, kt + 1, 1); MG4_KV_LOAD(xb, vc, kt + 3) } }
        else v_mul(kt, 0);
        __syncthreads();
    }
#undef MG4_KV_LOAD
    MG4_TLP(3);
    if (!loader) {
#pragma unroll
        for (int qs = 0; qs < QS; qs++)
#pragma unroll
            for (int i = 0; i < DPW; i++) {
                const int dt = mw + 4 * i;
                if (dt < DT) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int qrow = q0 + 16 * qs + l4 * 4 + r;
                        // out_h: the rows as the consumer wants them (an F16 wo multiplies the fp16-rounded attention output: the conversion launch is saved)
                        if (out_h) {   // pairs of dims from neighbouring lanes: 4-byte stores
                            const float o0 = oacc[qs][i][r], o1 = __shfl_xor(o0, 1);
                            if (qrow < N && !(l15 & 1)) *reinterpret_cast<__half2 *>(out_h + (size_t)qrow * E + (size_t)h * HD + dt * 16 + l15) = __halves2half2(f2h_rn(o0), f2h_rn(o1));
                        } else if (qrow < N) out[(size_t)qrow * E + (size_t)h * HD + dt * 16 + l15] = oacc[qs][i][r];
                    }
                }
            }
    }
#ifdef MG4_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG4_TLP(4);
#endif
}
static int g_attn_prefill_w8 = 1;    // 1: the 8-wave loader / MFMA form for 32-query tiles; MINIGPT4_ATTN_PREFILL_W8, read by Engine::init
void set_attn_prefill_w8(int v) { g_attn_prefill_w8 = v != 0; }
template <int HD>
static bool launch_attn_prefill_h8(const float *q, const __half *kc, const __half *vc, int N, int n_head, const int *n_past, int t_max, const Tables &tb, float *out, hipStream_t s, __half *out_h) {
    const int LS = ((t_max + AP_KT - 1) / AP_KT) * AP_KT + 4;
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_attn_prefill_h8<HD>)); attr = true; }
    const size_t kvbytes = std::max((size_t)AP_KT * HD * 2, (size_t)HD * APH_LDT * 2);
    const size_t lds = (size_t)AP_QT * 2 * LS * 4 + 2 * kvbytes;
    if (lds > 160 * 1024 - 512) return false;
    hipLaunchKernelGGL((k_attn_prefill_h8<HD>), dim3((unsigned)n_head, (unsigned)((N + AP_QT * 2 - 1) / (AP_QT * 2))), dim3(512), lds, s, q, kc, vc, n_head * HD, N, n_past, tb, out, LS, out_h);
    return true;
}
static int g_attn_prefill_f16 = 1;   // 1: prompt attention on the fp16 matrix cores (k_attn_prefill_h), 0: the exact-f32 MFMA kernel (k_attn_prefill); test-library setter only
void set_attn_prefill_f16(int v) { g_attn_prefill_f16 = v != 0; }
template <int HD, int QS>
static bool launch_attn_prefill_h_qs(const float *q, const __half *kc, const __half *vc, int N, int n_head, const int *n_past, int t_max, const Tables &tb, float *out, hipStream_t s) {
    const int LS = ((t_max + AP_KT - 1) / AP_KT) * AP_KT + 4;
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_attn_prefill_h<HD, QS>)); attr = true; }
    const size_t lds = (size_t)AP_QT * QS * LS * 4 + (size_t)AP_KT * HD * 2 + (size_t)HD * APH_LDT * 2;
    if (lds > 160 * 1024 - 512) return false;
    hipLaunchKernelGGL((k_attn_prefill_h<HD, QS>), dim3((unsigned)n_head, (unsigned)((N + AP_QT * QS - 1) / (AP_QT * QS))), dim3(256), lds, s, q, kc, vc, n_head * HD, N, n_past, tb, out, LS);
    return true;
}
template <int HD>
static bool launch_attn_prefill_hd(const float *q, const __half *kc, const __half *vc, int N, int n_head, const int *n_past, int t_max, const Tables &tb, float *out, hipStream_t s, __half *out_h, bool *wrote_h) {
    if (g_attn_prefill_f16) {   // 32 queries per staged K / V tile once that still gives every CU a workgroup
        if (g_attn_prefill_w8 && n_head * ((N + 31) / 32) >= 256 && launch_attn_prefill_h8<HD>(q, kc, vc, N, n_head, n_past, t_max, tb, out, s, out_h)) { if (wrote_h) *wrote_h = out_h != nullptr; return true; }
        if (n_head * ((N + 31) / 32) >= 256 && launch_attn_prefill_h_qs<HD, 2>(q, kc, vc, N, n_head, n_past, t_max, tb, out, s)) return true;
        if (launch_attn_prefill_h_qs<HD, 1>(q, kc, vc, N, n_head, n_past, t_max, tb, out, s)) return true;
    }
    // Fallback: k_attn_prefill<HD, 1>, the round-2 kernel with fp32 score / value products.  Taken (a) after set_attn_prefill_f16(0) (test library: the A/B switch of
    // the fp16 kernels), (b) when the fp16 kernels refuse because their LDS image (score rows + fp16 K tile + transposed V tile) exceeds the 160 KiB of a CU.  Both forms
    // hold 16 score rows of the whole context: at head size 128 both end at 1984 keys; beyond, this returns false and Engine::forward runs the rows through the decode
    // attention kernel (k_attn_llm, one query row per workgroup), which has no such limit.
    const int LS = ((t_max + AP_KT - 1) / AP_KT) * AP_KT + 1;
    static bool attr = false;
    if (!attr) { HIP_IGNORE(lds_optin_max(&k_attn_prefill<HD, 1>)); attr = true; }
    // (QS = 2 -- 32 queries per staged key / value tile from 256 prompt rows on -- is bit-identical and was measured 2 % SLOWER at 512 rows: 13B Q5_K_M 31.7 vs 31.0 ms,
    // 13B f16 27.9 vs 27.2 ms, profiles/r02r_prefill_ksplit_fill_sweep.log; half the workgroups, each twice as long: not instantiated.)
    const size_t lds = ((size_t)AP_QT * LS + (size_t)AP_KT * (HD + 1)) * 4;
    if (lds > 160 * 1024 - 512) return false;
    hipLaunchKernelGGL((k_attn_prefill<HD, 1>), dim3((unsigned)n_head, (unsigned)((N + AP_QT - 1) / AP_QT)), dim3(256), lds, s, q, kc, vc, n_head * HD, N, n_past, tb, out, LS);
    return true;
}
// N > 1 query rows at positions *n_past .. *n_past + N - 1 (launch_rope_kv has run); t_max >= *n_past + N (the host's view, sizes the LDS score rows).
// false -> the score rows do not fit LDS (very long contexts): the caller runs launch_attn_llm instead.
bool launch_attn_prefill(const float *q, const __half *kcache, const __half *vcache, int N, int n_head, int hd, const int *n_past, int t_max, const Tables &tb, float *out, hipStream_t s,
                         __half *out_h, bool *wrote_h) {
    if (wrote_h) *wrote_h = false;
    switch (hd) {
    case 32: return launch_attn_prefill_hd<32>(q, kcache, vcache, N, n_head, n_past, t_max, tb, out, s, out_h, wrote_h);
    case 64: return launch_attn_prefill_hd<64>(q, kcache, vcache, N, n_head, n_past, t_max, tb, out, s, out_h, wrote_h);
    case 128: return launch_attn_prefill_hd<128>(q, kcache, vcache, N, n_head, n_past, t_max, tb, out, s, out_h, wrote_h);
    default: return false;
    }
}
bool attn_head_size_supported(int hd) { return hd == 32 || hd == 64 || hd == 128; }
static size_t attn_lds_bytes(int n_ctx, int hd) { const int Tpad = (n_ctx + 7) & ~7; return (size_t)Tpad * 6 + (size_t)hd * 6 + (size_t)(AT_THREADS / (hd / 8)) * hd * 4 + 64; }
int attn_max_ctx(int hd) { int n = 0; while (attn_lds_bytes(n + 8, hd) + 256 /* static reduction arrays */ <= 160 * 1024) n += 8; return n; }
// fused = true (N must be 1): q,k,v are the raw projections; RoPE + KV append happen inside.  fused = false: launch_rope_kv must have run.
void launch_attn_llm(float *q, const float *k, const float *v, __half *kcache, __half *vcache, int N, int n_head, int hd, const int *n_past, int n_ctx,
                     const float *cos_tab, const float *sin_tab, const Tables &tb, float *out, bool fused, hipStream_t s) {
    switch (hd) {
    case 32: launch_attn_hd<32>(q, k, v, kcache, vcache, N, n_head, n_past, n_ctx, cos_tab, sin_tab, tb, out, fused, s); break;
    case 64: launch_attn_hd<64>(q, k, v, kcache, vcache, N, n_head, n_past, n_ctx, cos_tab, sin_tab, tb, out, fused, s); break;
    case 128: launch_attn_hd<128>(q, k, v, kcache, vcache, N, n_head, n_past, n_ctx, cos_tab, sin_tab, tb, out, fused, s); break;
    default: throw HipError{hipErrorInvalidValue, "unsupported head size", __FILE__, __LINE__};
    }
}

// =====================================================================================================================
// MINIGPT4_PARITY=1 attention: the oracle's loops (oracle/refcpu.c orc_llama_eval, = ggml's mul_mat(K, Q) / soft_max / mul_mat(V, P) on f16 operands) with every fp32
// chain in ITS order: thread = one key for the scores (sequential fma over the head dimension), exact double sum for the softmax, thread = one output
// dimension for P.V (sequential fma over the keys).  After launch_rope_kv (q rotated in place, caches appended).  One workgroup per (head, query row).
// =====================================================================================================================
// FUSED (one query row, decode): q | k | v are the raw projections; RoPE and the cache append (k_rope_kv's arithmetic) happen here, the new key / value row is also kept in LDS
// (the cache line written by this workgroup is not read back).  P.V: the value rows travel through LDS in chunks of 128 keys (16-byte coalesced loads, the next chunk in
// registers while this one is chained) -- a thread's chain over the keys is unchanged, it no longer waits for a strided 2-byte load per key.
template <bool FUSED>
__global__ __launch_bounds__(256) void k_attn_ref(const float *__restrict__ q, const float *__restrict__ kraw, const float *__restrict__ vraw, __half *__restrict__ kc, __half *__restrict__ vc,
                                                  int E, int hd, const int *__restrict__ n_past, const float *__restrict__ cos_tab, const float *__restrict__ sin_tab, const Tables tb, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ref[];
    const int h = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const int T = *n_past + t + 1;
    constexpr int CH = 128;                                       // keys per value chunk
    __half *vbuf = reinterpret_cast<__half *>(smem_ref);          // [2][CH][hd]
    float *sc = reinterpret_cast<float *>(vbuf + 2 * CH * hd);    // [T]
    __half *ph = reinterpret_cast<__half *>(sc + ((T + 3) & ~3)); // [T]
    __half *qh = ph + ((T + 7) & ~7);                             // [hd]
    __half *knew = qh + hd, *vnew = knew + hd;                    // [hd] each (FUSED)
    __shared__ float red_f[4]; __shared__ double red_d[4];
    if (FUSED) {
        if (tid < hd / 2) {
            const int i = tid, pos = T - 1;
            const float c = cos_tab[(size_t)pos * (hd / 2) + i], s = sin_tab[(size_t)pos * (hd / 2) + i];
            const size_t o = (size_t)h * hd + 2 * i;
            const float q0 = q[o], q1 = q[o + 1];
            const float r0 = q0 * c - q1 * s, r1 = q0 * s + q1 * c;
            qh[2 * i] = f2h_rn(r0); qh[2 * i + 1] = f2h_rn(r1);
            const float k0 = kraw[o], k1 = kraw[o + 1];
            const size_t co = (size_t)pos * E + (size_t)h * hd + 2 * i;
            const __half2 kk = __floats2half2_rn(k0 * c - k1 * s, k0 * s + k1 * c), vv = __floats2half2_rn(vraw[o], vraw[o + 1]);
            *reinterpret_cast<__half2 *>(kc + co) = kk; *reinterpret_cast<__half2 *>(vc + co) = vv;
            *reinterpret_cast<__half2 *>(knew + 2 * i) = kk; *reinterpret_cast<__half2 *>(vnew + 2 * i) = vv;
        }
    } else {
        for (int i = tid; i < hd; i += 256) qh[i] = __float2half_rn(q[(size_t)t * E + (size_t)h * hd + i]);
    }
    __syncthreads();
    const float kq_scale = 1.0f / sqrtf((float)hd);
    float mx = -INFINITY;
    for (int j = tid; j < T; j += 256) {                          // the row in 16-byte pieces (hd % 8 == 0), the fma chain in element order as before
        const __half *kr = (FUSED && j == T - 1) ? knew : kc + (size_t)j * E + (size_t)h * hd;
        float s = 0.0f;
        for (int i0 = 0; i0 < hd; i0 += 32) {
            int4 kk[4];
#pragma unroll
            for (int c = 0; c < 4; c++) kk[c] = i0 + 8 * c < hd ? ld16(kr + i0 + 8 * c) : make_int4(0, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (i0 + 8 * c >= hd) break;
                const unsigned wds[4] = {(unsigned)kk[c].x, (unsigned)kk[c].y, (unsigned)kk[c].z, (unsigned)kk[c].w};
#pragma unroll
                for (int e = 0; e < 4; e++) { const int i = i0 + 8 * c + 2 * e; s = fmaf(h2f_bits(wds[e] & 0xFFFF), __half2float(qh[i]), s); s = fmaf(h2f_bits(wds[e] >> 16), __half2float(qh[i + 1]), s); }
            }
        }
        s *= kq_scale; sc[j] = s; mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red_f[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    double sum = 0.0;
    for (int j = tid; j < T; j += 256) { const float v = tab(tb.exp, sc[j] - mx); sc[j] = v; sum += (double)v; }   // <= 2048 fp16 values: the double sum is exact in any order
    sum = wave_sum_d(sum);
    if ((tid & 63) == 0) red_d[tid >> 6] = sum;
    __syncthreads();
    const float inv = (float)(1.0 / (((red_d[0] + red_d[1]) + red_d[2]) + red_d[3]));
    for (int j = tid; j < T; j += 256) ph[j] = f2h_rn(sc[j] * inv);
    // ---- P.V: chunk c = keys c CH .. of this head's value rows -> LDS; 16 threads x 16 bytes cover a row's hd (= 128) halves, 16 keys per pass of the workgroup
    const int per_row = hd / 8, keys_per_pass = 256 / per_row, passes = CH / keys_per_pass;   // hd = 128: 16, 16, 8; hd = 64: 8, 32, 4; hd = 32: 4, 64, 2
    const int kr_ = tid / per_row, part = tid - kr_ * per_row;
    const int n_chunks = (T + CH - 1) / CH;
    int4 stage[8];
    auto request = [&](int c) {
#pragma unroll
        for (int p = 0; p < 8; p++) if (p < passes) {
            const int j = min(c * CH + p * keys_per_pass + kr_, T - 1);                     // clamped: rows past T are never chained
            const __half *src = (FUSED && j == T - 1) ? vnew + part * 8 : vc + (size_t)j * E + (size_t)h * hd + part * 8;
            stage[p] = ld16(src);
        }
    };
    auto deposit = [&](int b) {
#pragma unroll
        for (int p = 0; p < 8; p++) if (p < passes) *reinterpret_cast<int4 *>(vbuf + ((size_t)b * CH + p * keys_per_pass + kr_) * hd + part * 8) = stage[p];
    };
    request(0);
    __syncthreads();                                              // ph complete (and knew / vnew visible)
    deposit(0);
    float s = 0.0f;
    for (int c = 0; c < n_chunks; c++) {
        __syncthreads();                                          // chunk c deposited; everybody has left the buffer chunk c + 1 goes to
        if (c + 1 < n_chunks) request(c + 1);
        if (tid < hd) {
            const __half *vb = vbuf + (size_t)(c & 1) * CH * hd + tid;
            const int j0 = c * CH, n = min(CH, T - j0);
            for (int u = 0; u < n; u++) s = fmaf(__half2float(vb[(size_t)u * hd]), __half2float(ph[j0 + u]), s);   // key order
        }
        if (c + 1 < n_chunks) deposit((c + 1) & 1);
    }
    if (tid < hd) out[(size_t)t * E + (size_t)h * hd + tid] = s;
}
static size_t attn_ref_lds(int t_max, int hd) { return (size_t)2 * 128 * hd * 2 + (size_t)((t_max + 3) & ~3) * 4 + (size_t)((t_max + 7) & ~7) * 2 + (size_t)hd * 2 * 3 + 64; }
// largest n_ctx whose score / probability rows fit k_attn_ref's LDS next to its value staging (parity mode): smaller than attn_max_ctx(hd), which is the fast kernel's
constexpr size_t ATTN_REF_DYN_LDS = 160 * 1024 - 512;                     // the CU's 160 KiB minus the kernel's static reduction arrays (48 B), rounded
int attn_ref_max_ctx(int hd) { int n = 0; while (attn_ref_lds(n + 8, hd) <= ATTN_REF_DYN_LDS) n += 8; return n; }
// the > 64 KiB dynamic-LDS opt-in of a kernel: once per DEVICE (the attribute belongs to the device's code object), and a refusal is an error here, not a launch failure later
static unsigned long long g_attn_ref_optin_mask = 0;                       // bit d: device d has the attribute for both instantiations (<= 64 devices per process)
void attn_ref_prepare() {   // called by Engine::init / set_parity when parity mode is switched on -- never inside a stream capture (an attribute call there is refused)
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && (g_attn_ref_optin_mask >> dev & 1)) return;
    // (a request for the full 160 KiB is refused -- invalid argument -- because the kernels also have static LDS; rounds 3-4 asked for exactly that behind HIP_IGNORE, so the
    // opt-in never took effect and parity mode silently depended on staying below the 64 KiB default)
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_attn_ref<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_REF_DYN_LDS));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_attn_ref<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_REF_DYN_LDS));
    if (dev < 64) g_attn_ref_optin_mask |= 1ull << dev;
}
template <typename K> static void attn_ref_lds_optin(K kernel) {       // launchers reached without attn_ref_prepare (test hooks): best effort, as before
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
    if (dev < 64 && (g_attn_ref_optin_mask >> dev & 1)) return;
    HIP_IGNORE(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_REF_DYN_LDS));
}
void launch_attn_ref(const float *q, const __half *kcache, const __half *vcache, int N, int n_head, int hd, const int *n_past, int t_max, const Tables &tb, float *out, hipStream_t s) {
    note_kernel("k_attn_ref");
    attn_ref_lds_optin(&k_attn_ref<false>);
    hipLaunchKernelGGL(k_attn_ref<false>, dim3((unsigned)n_head, (unsigned)N), dim3(256), attn_ref_lds(t_max, hd), s, q, nullptr, nullptr, const_cast<__half *>(kcache), const_cast<__half *>(vcache),
                       n_head * hd, hd, n_past, nullptr, nullptr, tb, out);
}
// one query row: RoPE + cache append + attention in one launch (q, k, v: the raw projections; q is left unrotated)
void launch_attn_ref_fused(const float *q, const float *k, const float *v, __half *kcache, __half *vcache, int n_head, int hd, const int *n_past, int t_max, const float *cos_tab, const float *sin_tab,
                           const Tables &tb, float *out, hipStream_t s) {
    note_kernel("k_attn_ref");
    attn_ref_lds_optin(&k_attn_ref<true>);
    hipLaunchKernelGGL(k_attn_ref<true>, dim3((unsigned)n_head, 1), dim3(256), attn_ref_lds(t_max, hd), s, q, k, v, kcache, vcache, n_head * hd, hd, n_past, cos_tab, sin_tab, tb, out);
}

// =====================================================================================================================
// argmax (first maximum wins, like llama_sample_token_greedy) and small elementwise helpers
// =====================================================================================================================
constexpr int AM_BLOCKS = 64;
__device__ __forceinline__ void argmax_combine(float &best, int &bi, float ov, int oi) { if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; } }
__device__ __forceinline__ void argmax_wave(float &best, int &bi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o); argmax_combine(best, bi, ov, oi); }
}
__global__ __launch_bounds__(256) void k_argmax_part(const float *__restrict__ x, int n, float *__restrict__ pv, int *__restrict__ pi) {
    float best = -INFINITY; int bi = 0x7FFFFFFF;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += AM_BLOCKS * 256) { const float v = x[i]; if (v > best) { best = v; bi = i; } }
    __shared__ float sv[4]; __shared__ int si[4];
    argmax_wave(best, bi);
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < 4; w++) argmax_combine(best, bi, sv[w], si[w]); pv[blockIdx.x] = best; pi[blockIdx.x] = bi; }
}
__global__ __launch_bounds__(64) void k_argmax_final(const float *__restrict__ pv, const int *__restrict__ pi, int *__restrict__ out) {
    float best = pv[threadIdx.x]; int bi = pi[threadIdx.x];
    argmax_wave(best, bi);
    if (threadIdx.x == 0) *out = bi == 0x7FFFFFFF ? 0 : bi;
}
// (Round 5 measured argmax + advance as ONE launch -- last-arriving workgroup reduces the partial maxima and advances the position: 374.1 vs 374.4 tok/s, no difference;
// removed again, profiles/r05_experiments_not_adopted.md.)
// first maximum wins, like llama_sample_token_greedy.  `scratch` holds AM_BLOCKS floats + AM_BLOCKS ints.
void launch_argmax(const float *logits, int n, int *out, void *scratch, hipStream_t s) {
    note_kernel("k_argmax_part"); note_kernel("k_argmax_final");
    float *pv = static_cast<float *>(scratch); int *pi = reinterpret_cast<int *>(pv + AM_BLOCKS);
    hipLaunchKernelGGL(k_argmax_part, dim3(AM_BLOCKS), dim3(256), 0, s, logits, n, pv, pi);
    hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(64), 0, s, pv, pi, out);
}

__global__ void k_add_inplace(float *__restrict__ x, const float *__restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] + y[i];
}
void launch_add_inplace(float *x, const float *y, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n); }

__global__ __launch_bounds__(256) void k_checksum(const unsigned *__restrict__ p, size_t n_words, unsigned long long *out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);                      // integer sum: order does not matter
}
uint64_t device_checksum(const void *p, size_t bytes, hipStream_t s) {
    struct Word { unsigned long long *d = nullptr; ~Word() { if (d) HIP_IGNORE(hipFree(d)); } } w;   // freed on every path, also when a check below throws
    unsigned long long h = 0;
    HIP_CHECK(hipMalloc((void **)&w.d, 8));
    HIP_CHECK(hipMemsetAsync(w.d, 0, 8, s));
    hipLaunchKernelGGL(k_checksum, dim3(2048), dim3(256), 0, s, static_cast<const unsigned *>(p), bytes / 4, w.d);
    HIP_CHECK(hipMemcpyAsync(&h, w.d, 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return (uint64_t)h;
}
// Fault injection for the native broadcast's bounded waits (MINIGPT4_DIST_TEST_STALL_MS, tests/test_gpu_serve.py): one lane keeps the stream busy for `ms` milliseconds
// of the 100 MHz constant clock -- what a collective whose peer died looks like to the host -- and then ENDS (at most 5 s: it can never hang the device).
__global__ void k_stall(unsigned ms) {
    const unsigned long long t0 = wall_clock64(), ticks = (unsigned long long)(ms > 5000u ? 5000u : ms) * 100000ull;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
void launch_stall(unsigned ms, hipStream_t s) { hipLaunchKernelGGL(k_stall, dim3(1), dim3(1), 0, s, ms); }
__global__ void k_fill_u16(unsigned short *p, size_t n, unsigned short v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_u16(void *p, size_t n, unsigned short v, hipStream_t s) { hipLaunchKernelGGL(k_fill_u16, dim3(1024), dim3(256), 0, s, (unsigned short *)p, n, v); }
__global__ void k_set_int(int *p, int v) { *p = v; }
void launch_set_int(int *p, int v, hipStream_t s) { hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, s, p, v); }
// Batched decode epilogue, one workgroup per row: greedy argmax of the row's logits (first maximum wins), stored with the logits' owner slot; the
// conversation's position advances by one and the greedy token becomes its next input.
__global__ __launch_bounds__(1024) void k_batch_finish(const float *__restrict__ logits, int n_vocab, const int *__restrict__ row_slot, int *__restrict__ n_past, int *__restrict__ argmax,
                                                      int *__restrict__ feed, float *__restrict__ slot_logits) {
    const int r = blockIdx.x, slot = row_slot[r];
    const float *x = logits + (size_t)r * n_vocab;
    float *keep = slot_logits + (size_t)slot * n_vocab;                   // the conversation's own copy (sampling with temp > 0, minigpt4_amd_get_logits)
    float best = -INFINITY; int bi = 0x7FFFFFFF;
    // 16 waves, 16-byte loads (a 256-thread scalar loop took 35 us for 32 000 logits: 125 dependent trips); every thread visits its indices in ascending order and
    // argmax_combine prefers the lower index, so ties resolve to the first maximum as before
    const int n4 = (n_vocab & 3) == 0 ? n_vocab >> 2 : 0;
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i]; reinterpret_cast<float4 *>(keep)[i] = v;
        if (v.x > best) { best = v.x; bi = 4 * i; } if (v.y > best) { best = v.y; bi = 4 * i + 1; } if (v.z > best) { best = v.z; bi = 4 * i + 2; } if (v.w > best) { best = v.w; bi = 4 * i + 3; }
    }
    for (int i = 4 * n4 + threadIdx.x; i < n_vocab; i += 1024) { const float v = x[i]; keep[i] = v; if (v > best) { best = v; bi = i; } }
    __shared__ float sv[16]; __shared__ int si[16];
    argmax_wave(best, bi);
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) argmax_combine(best, bi, sv[w], si[w]);
        const int id = bi == 0x7FFFFFFF ? 0 : bi;
        argmax[slot] = id; feed[slot] = id; n_past[slot] += 1;
    }
}
void launch_batch_finish(const float *logits, int n_vocab, int B, const int *row_slot, int *n_past, int *argmax, int *feed, float *slot_logits, hipStream_t s) {
    hipLaunchKernelGGL(k_batch_finish, dim3((unsigned)B), dim3(1024), 0, s, logits, n_vocab, row_slot, n_past, argmax, feed, slot_logits);
}
// batched decode prologue: the host's view of each row's position (a conversation may have been reset) -> n_past[slot]
__global__ void k_batch_begin(int *__restrict__ n_past, const int *__restrict__ row_slot, const int *__restrict__ row_pos, int B) {
    const int r = threadIdx.x;
    if (r < B) n_past[row_slot[r]] = row_pos[r];
}
void launch_batch_begin(int *n_past, const int *row_slot, const int *row_pos, int B, hipStream_t s) { hipLaunchKernelGGL(k_batch_begin, dim3(1), dim3(64), 0, s, n_past, row_slot, row_pos, B); }
// end of a decode step: the KV position advances and the greedy token becomes the next input (the host may overwrite it)
__global__ void k_advance(int *n_past, int n, int *tok0, const int *argmax) { *n_past += n; if (tok0) *tok0 = *argmax; }
void launch_advance(int *n_past, int n, int *tok0, const int *argmax, hipStream_t s) { note_kernel("k_advance"); hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, s, n_past, n, tok0, argmax); }
// Profiling gate (Engine::profile_sites): one wave that keeps the stream busy for `us` microseconds (s_memrealtime: the constant 100 MHz clock) while the host
// queues the step's launches behind it, so the per-site event pairs time the GPU and not the host's launch rate.  Bounded: at most 2^20 polls even if the clock stood still.
__global__ void k_delay(int us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), ticks = (unsigned long long)us * 100ull;
    for (int i = 0; i < (1 << 20) && __builtin_amdgcn_s_memrealtime() - t0 < ticks; i++) __builtin_amdgcn_s_sleep(32);
}
void launch_delay(int us, hipStream_t s) { hipLaunchKernelGGL(k_delay, dim3(1), dim3(1), 0, s, us); }

}  // namespace mg4
