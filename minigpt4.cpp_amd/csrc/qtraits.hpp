// Device-side unit traits of the quantised weight planes (how one lane fetches a 16-byte weight unit and the matching activation unit, and how they dot) and the
// activation quantiser, shared by the kernel files that stream weights (llm_kernels.hip: launch-per-op kernels; decode_engine.hip: the persistent decode engine).
#pragma once
#include "common.hpp"
#include "devutil.hpp"

namespace mg4 {

__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
__device__ __forceinline__ float h2f_bits(unsigned short h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ unsigned short f2h_bits(float f) { return __half_as_ushort(f2h_rn(f)); }
__device__ __forceinline__ float f16r(float f) { return __half2float(f2h_rn(f)); }
__device__ __forceinline__ float tab(const __half *t, float x) { return __half2float(t[f2h_bits(x)]); }
// Fast mode (round 5, SURVEY.md 9.5: "in fast mode evaluate in fp32"): the VALUE of ggml's fp16 tables computed instead of gathered -- table[x] = fp16(f(fp16(x))) with f
// evaluated in fp32 by the host libm; here f comes from the GPU's exp (v_exp_f32, ~1 ulp), so a result differs from the table's by one fp16 ulp only when f lands within
// ~1e-7 of an fp16 rounding boundary (about one value in 10^4).  A null table pointer selects this form: the DEFAULT of fast mode for the decode step's attention / SiLU launches (Engine::tabs_dec_) and for the ViT / Q-Former
// attention (Engine::tabs_vis_.exp; MINIGPT4_COMPUTED_TABLES=0 passes the tables again, kept under test) -- a stated deviation from ggml's table VALUES by at most one fp16 ulp,
// inside the fast-mode bound.  Parity mode, every prompt pass and the GELU of the vision tower always read the tables.
__device__ __forceinline__ float exp_h(const __half *t, float x) { if (t) return tab(t, x); return f16r(__expf(f16r(x))); }
__device__ __forceinline__ float silu_h(const __half *t, float x) { if (t) return tab(t, x); const float xh = f16r(x); return f16r(xh / (1.0f + __expf(-xh))); }

__device__ __forceinline__ int4 ld16(const void *p) { return *reinterpret_cast<const int4 *>(p); }
// weight-plane loads: streamed once per token by exactly one wave -> non-temporal (MG4_NT_WEIGHTS=0 builds the default-policy variant for A/B runs)
#ifndef MG4_NT_WEIGHTS
#define MG4_NT_WEIGHTS 1
#endif
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <typename N> __device__ __forceinline__ N ldw_raw(const void *p) {
#if MG4_NT_WEIGHTS
    return __builtin_nontemporal_load(reinterpret_cast<const N *>(p));
#else
    return *reinterpret_cast<const N *>(p);
#endif
}
template <typename V> __device__ __forceinline__ V ldw(const void *p) { return ldw_raw<V>(p); }
template <> __device__ __forceinline__ int4 ldw<int4>(const void *p) { const v4i_t v = ldw_raw<v4i_t>(p); return make_int4(v.x, v.y, v.z, v.w); }
template <> __device__ __forceinline__ uint2 ldw<uint2>(const void *p) { const v2u_t v = ldw_raw<v2u_t>(p); return make_uint2(v.x, v.y); }
template <> __device__ __forceinline__ float4 ldw<float4>(const void *p) { const v4f_t v = ldw_raw<v4f_t>(p); return make_float4(v.x, v.y, v.z, v.w); }

// Buffer-descriptor form of the weight-plane loads (decode mat-vec): the descriptor is built from wave-uniform values (the row's plane
// addresses, SGPRs), the lane's byte offset within the row is a loop-invariant VGPR -> no per-load vector address arithmetic, so no VALU
// temporary can alias (and therefore wait for) the destination of a load that is still in flight.
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2ub_t __attribute__((ext_vector_type(2)));
constexpr int WAUX = MG4_NT_WEIGHTS ? 2 : 0;
struct WBuf { __amdgpu_buffer_rsrc_t qs, qh, sc, d; };
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkbuf(const uint8_t *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, 0x7FFFFFFF, 0x00020000); }
__device__ __forceinline__ int4 bld16(__amdgpu_buffer_rsrc_t r, int off) { const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, WAUX); return make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w); }
__device__ __forceinline__ uint2 bld8(__amdgpu_buffer_rsrc_t r, int off) { const v2ub_t v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, WAUX); return make_uint2(v.x, v.y); }
__device__ __forceinline__ unsigned bld4(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, WAUX); }
__device__ __forceinline__ unsigned short bld2(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, WAUX); }

// =====================================================================================================================
// per-type unit traits: how one lane fetches a weight unit / the matching activation unit, and how they dot.
// =====================================================================================================================
template <int T> struct Tr;


template <> struct Tr<GT_Q4_0> {
    static constexpr int EPU = 32;
    struct WU { int4 q; unsigned short dh; };
    struct AU { int4 a0, a1; float d; int sum; };
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) { w.q = ldw<int4>(W.qs + g0 * 16 + (unsigned)(u * 16)); w.dh = ldw<unsigned short>(W.sc + g0 * 2 + (unsigned)(u * 2)); }
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 16); B.sc = mkbuf(W.sc + g0 * 2); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) { w.q = bld16(B.qs, u * 16); w.dh = bld2(B.sc, u * 2); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) {
        const int8_t *p = A.q80 + (size_t)t * K + (size_t)u * 32; a.a0 = ld16(p); a.a1 = ld16(p + 16);
        const size_t b = (size_t)t * (K / 32) + u; a.d = A.d0[b]; a.sum = A.sum0[b]; }
    static constexpr int GROUP = 1, TERMS = 1;   // units per ggml block, fp32 terms per block (k_mul_mat_ref)
    static __device__ __forceinline__ void ints(const WU &w, const AU &a, int &i0, int &i1) {
        int s = 0;
        s = dot4(w.q.x & 0x0F0F0F0F, a.a0.x, s); s = dot4(w.q.y & 0x0F0F0F0F, a.a0.y, s); s = dot4(w.q.z & 0x0F0F0F0F, a.a0.z, s); s = dot4(w.q.w & 0x0F0F0F0F, a.a0.w, s);
        s = dot4((w.q.x >> 4) & 0x0F0F0F0F, a.a1.x, s); s = dot4((w.q.y >> 4) & 0x0F0F0F0F, a.a1.y, s); s = dot4((w.q.z >> 4) & 0x0F0F0F0F, a.a1.z, s); s = dot4((w.q.w >> 4) & 0x0F0F0F0F, a.a1.w, s);
        i0 = s - 8 * a.sum; i1 = 0;
    }
    static __device__ __forceinline__ void terms(const WU &w, const AU &a, int i0, int, float &f0, float &v0, float &f1, float &v1) { f0 = h2f_bits(w.dh) * a.d; v0 = (float)i0; f1 = v1 = 0.0f; }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        int s = 0;
        s = dot4(w.q.x & 0x0F0F0F0F, a.a0.x, s); s = dot4(w.q.y & 0x0F0F0F0F, a.a0.y, s); s = dot4(w.q.z & 0x0F0F0F0F, a.a0.z, s); s = dot4(w.q.w & 0x0F0F0F0F, a.a0.w, s);
        s = dot4((w.q.x >> 4) & 0x0F0F0F0F, a.a1.x, s); s = dot4((w.q.y >> 4) & 0x0F0F0F0F, a.a1.y, s); s = dot4((w.q.z >> 4) & 0x0F0F0F0F, a.a1.z, s); s = dot4((w.q.w >> 4) & 0x0F0F0F0F, a.a1.w, s);
        s -= 8 * a.sum;
        acc = fmaf(h2f_bits(w.dh) * a.d, (float)s, acc);
    }
};
template <> struct Tr<GT_Q4_1> {
    static constexpr int EPU = 32;
    struct WU { int4 q; unsigned dm; };
    struct AU { int4 a0, a1; float d, s; };
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) { w.q = ldw<int4>(W.qs + g0 * 16 + (unsigned)(u * 16)); w.dm = ldw<unsigned>(W.sc + g0 * 4 + (unsigned)(u * 4)); }
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 16); B.sc = mkbuf(W.sc + g0 * 4); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) { w.q = bld16(B.qs, u * 16); w.dm = bld4(B.sc, u * 4); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) {
        const int8_t *p = A.q80 + (size_t)t * K + (size_t)u * 32; a.a0 = ld16(p); a.a1 = ld16(p + 16);
        const size_t b = (size_t)t * (K / 32) + u; a.d = A.d1[b]; a.s = A.s1[b]; }
    static constexpr int GROUP = 1, TERMS = 2;
    static __device__ __forceinline__ void ints(const WU &w, const AU &a, int &i0, int &i1) {
        int s = 0;
        s = dot4(w.q.x & 0x0F0F0F0F, a.a0.x, s); s = dot4(w.q.y & 0x0F0F0F0F, a.a0.y, s); s = dot4(w.q.z & 0x0F0F0F0F, a.a0.z, s); s = dot4(w.q.w & 0x0F0F0F0F, a.a0.w, s);
        s = dot4((w.q.x >> 4) & 0x0F0F0F0F, a.a1.x, s); s = dot4((w.q.y >> 4) & 0x0F0F0F0F, a.a1.y, s); s = dot4((w.q.z >> 4) & 0x0F0F0F0F, a.a1.z, s); s = dot4((w.q.w >> 4) & 0x0F0F0F0F, a.a1.w, s);
        i0 = s; i1 = 0;
    }
    static __device__ __forceinline__ void terms(const WU &w, const AU &a, int i0, int, float &f0, float &v0, float &f1, float &v1) { f0 = h2f_bits(w.dm & 0xFFFF) * a.d; v0 = (float)i0; f1 = h2f_bits(w.dm >> 16); v1 = a.s; }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        int s = 0;
        s = dot4(w.q.x & 0x0F0F0F0F, a.a0.x, s); s = dot4(w.q.y & 0x0F0F0F0F, a.a0.y, s); s = dot4(w.q.z & 0x0F0F0F0F, a.a0.z, s); s = dot4(w.q.w & 0x0F0F0F0F, a.a0.w, s);
        s = dot4((w.q.x >> 4) & 0x0F0F0F0F, a.a1.x, s); s = dot4((w.q.y >> 4) & 0x0F0F0F0F, a.a1.y, s); s = dot4((w.q.z >> 4) & 0x0F0F0F0F, a.a1.z, s); s = dot4((w.q.w >> 4) & 0x0F0F0F0F, a.a1.w, s);
        acc = fmaf(h2f_bits(w.dm & 0xFFFF) * a.d, (float)s, acc);
        acc = fmaf(h2f_bits(w.dm >> 16), a.s, acc);
    }
};
// 5-bit: P holds the high bits pre-transposed (see pack_hb1): lo dword k -> (P << (4-k)) & 0x10101010, hi dword k -> (P >> k) & 0x10101010
#define MG4_Q5_DOT8(q, P, a0, a1, s)                                                                                   \
    s = dot4((q.x & 0x0F0F0F0F) | ((P << 4) & 0x10101010), a0.x, s); s = dot4((q.y & 0x0F0F0F0F) | ((P << 3) & 0x10101010), a0.y, s); \
    s = dot4((q.z & 0x0F0F0F0F) | ((P << 2) & 0x10101010), a0.z, s); s = dot4((q.w & 0x0F0F0F0F) | ((P << 1) & 0x10101010), a0.w, s); \
    s = dot4(((q.x >> 4) & 0x0F0F0F0F) | (P & 0x10101010), a1.x, s); s = dot4(((q.y >> 4) & 0x0F0F0F0F) | ((P >> 1) & 0x10101010), a1.y, s); \
    s = dot4(((q.z >> 4) & 0x0F0F0F0F) | ((P >> 2) & 0x10101010), a1.z, s); s = dot4(((q.w >> 4) & 0x0F0F0F0F) | ((P >> 3) & 0x10101010), a1.w, s);
template <> struct Tr<GT_Q5_0> {
    static constexpr int EPU = 32;
    struct WU { int4 q; unsigned P; unsigned short dh; };
    using AU = Tr<GT_Q4_0>::AU;
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) { w.q = ldw<int4>(W.qs + g0 * 16 + (unsigned)(u * 16)); w.P = ldw<unsigned>(W.qh + g0 * 4 + (unsigned)(u * 4)); w.dh = ldw<unsigned short>(W.sc + g0 * 2 + (unsigned)(u * 2)); }
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 16); B.qh = mkbuf(W.qh + g0 * 4); B.sc = mkbuf(W.sc + g0 * 2); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) { w.q = bld16(B.qs, u * 16); w.P = bld4(B.qh, u * 4); w.dh = bld2(B.sc, u * 2); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) { Tr<GT_Q4_0>::loada(A, t, K, u, a); }
    static constexpr int GROUP = 1, TERMS = 1;
    static __device__ __forceinline__ void ints(const WU &w, const AU &a, int &i0, int &i1) {
        int s = 0; const unsigned P = w.P;
        MG4_Q5_DOT8(w.q, P, a.a0, a.a1, s)
        i0 = s - 16 * a.sum; i1 = 0;
    }
    static __device__ __forceinline__ void terms(const WU &w, const AU &a, int i0, int, float &f0, float &v0, float &f1, float &v1) { f0 = h2f_bits(w.dh) * a.d; v0 = (float)i0; f1 = v1 = 0.0f; }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        int s = 0; const unsigned P = w.P;
        MG4_Q5_DOT8(w.q, P, a.a0, a.a1, s)
        s -= 16 * a.sum;
        acc = fmaf(h2f_bits(w.dh) * a.d, (float)s, acc);
    }
};
template <> struct Tr<GT_Q5_1> {
    static constexpr int EPU = 32;
    struct WU { int4 q; unsigned P; unsigned dm; };
    using AU = Tr<GT_Q4_1>::AU;
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) { w.q = ldw<int4>(W.qs + g0 * 16 + (unsigned)(u * 16)); w.P = ldw<unsigned>(W.qh + g0 * 4 + (unsigned)(u * 4));
        w.dm = ldw<unsigned>(W.sc + g0 * 4 + (unsigned)(u * 4)); }
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 16); B.qh = mkbuf(W.qh + g0 * 4); B.sc = mkbuf(W.sc + g0 * 4); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) { w.q = bld16(B.qs, u * 16); w.P = bld4(B.qh, u * 4); w.dm = bld4(B.sc, u * 4); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) { Tr<GT_Q4_1>::loada(A, t, K, u, a); }
    static constexpr int GROUP = 1, TERMS = 2;
    static __device__ __forceinline__ void ints(const WU &w, const AU &a, int &i0, int &i1) {
        int s = 0; const unsigned P = w.P;
        MG4_Q5_DOT8(w.q, P, a.a0, a.a1, s)
        i0 = s; i1 = 0;
    }
    static __device__ __forceinline__ void terms(const WU &w, const AU &a, int i0, int, float &f0, float &v0, float &f1, float &v1) { f0 = h2f_bits(w.dm & 0xFFFF) * a.d; v0 = (float)i0; f1 = h2f_bits(w.dm >> 16); v1 = a.s; }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        int s = 0; const unsigned P = w.P;
        MG4_Q5_DOT8(w.q, P, a.a0, a.a1, s)
        acc = fmaf(h2f_bits(w.dm & 0xFFFF) * a.d, (float)s, acc);
        acc = fmaf(h2f_bits(w.dm >> 16), a.s, acc);
    }
};
template <> struct Tr<GT_Q8_0> {   // unit = half a block (16 int8); the two halves are combined across the lane pair before scaling
    static constexpr int EPU = 16;
    struct WU { int4 q; unsigned short dh; };
    struct AU { int4 a; float d; };
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) { w.q = ldw<int4>(W.qs + g0 * 16 + (unsigned)(u * 16)); w.dh = ldw<unsigned short>(W.sc + (g0 >> 1) * 2 + (unsigned)((u >> 1) * 2)); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) { a.a = ld16(A.q80 + (size_t)t * K + (size_t)u * 16); a.d = A.d0[(size_t)t * (K / 32) + (u >> 1)]; }
    static constexpr int GROUP = 2, TERMS = 1;
    static __device__ __forceinline__ void ints(const WU &w, const AU &a, int &i0, int &i1) {
        int s = 0;
        s = dot4(w.q.x, a.a.x, s); s = dot4(w.q.y, a.a.y, s); s = dot4(w.q.z, a.a.z, s); s = dot4(w.q.w, a.a.w, s);
        i0 = s; i1 = 0;
    }
    static __device__ __forceinline__ void terms(const WU &w, const AU &a, int i0, int, float &f0, float &v0, float &f1, float &v1) { f0 = h2f_bits(w.dh) * a.d; v0 = (float)i0; f1 = v1 = 0.0f; }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        int s = 0;
        s = dot4(w.q.x, a.a.x, s); s = dot4(w.q.y, a.a.y, s); s = dot4(w.q.z, a.a.z, s); s = dot4(w.q.w, a.a.w, s);
        s += __shfl_xor(s, 1);
        if (!(threadIdx.x & 1)) acc = fmaf(h2f_bits(w.dh) * a.d, (float)s, acc);
    }
};
// k-quants ---------------------------------------------------------------------------------------------------------------
// sub-block scale x integer block sum: |scale| <= 127 and |sum| <= 32 x 63 x 127 + 32 x 16 x 127 < 2^19, so the 24-bit multiplier is exact -- and full rate, where
// the 32-bit v_mul_lo_u32 / v_mad_u64_u32 the compiler emits for int * int issue at a quarter of it (two of them per 32 weights and activation row)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
struct AK { int4 lo, hi; float d; int bs_lo, bs_hi; };
__device__ __forceinline__ void scale_min_pair(const int4 &h, int j, int &sc0, int &sc1, int &m0, int &m1) {
    // h.y,h.z,h.w = the 12 packed 6-bit (scale,min) bytes of a Q4_K/Q5_K super-block; pair j -> sub-blocks 2j, 2j+1
    const unsigned s0 = (unsigned)h.y, s1 = (unsigned)h.z, s2 = (unsigned)h.w;
    const unsigned sc_lo = s0 & 0x3f3f3f3fu, m_lo = s1 & 0x3f3f3f3fu;
    const unsigned sc_hi = (s2 & 0x0f0f0f0fu) | (((s0 >> 6) & 0x03030303u) << 4);
    const unsigned m_hi = ((s2 >> 4) & 0x0f0f0f0fu) | (((s1 >> 6) & 0x03030303u) << 4);
    const unsigned scw = (j & 2) ? sc_hi : sc_lo, mw = (j & 2) ? m_hi : m_lo;
    const int sh = (j & 1) * 16;
    sc0 = (scw >> sh) & 0xFF; sc1 = (scw >> (sh + 8)) & 0xFF; m0 = (mw >> sh) & 0xFF; m1 = (mw >> (sh + 8)) & 0xFF;
}
template <> struct Tr<GT_Q2_K> {   // supported, not tuned: only the generic tile kernel k_mul_mat streams this type (no persistent-wave / MFMA path)
    static constexpr int EPU = 32;
    struct WU { uint2 p; unsigned short sc; unsigned dm; };
    using AU = AK;
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) {
        w.p = ldw<uint2>(W.qs + g0 * 8 + (unsigned)(u * 8)); w.sc = ldw<unsigned short>(W.sc + g0 * 2 + (unsigned)(u * 2)); w.dm = ldw<unsigned>(W.d + (g0 >> 3) * 4 + (unsigned)((u >> 3) * 4)); }
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 8); B.sc = mkbuf(W.sc + g0 * 2); B.d = mkbuf(W.d + (g0 >> 3) * 4); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) { w.p = bld8(B.qs, u * 8); w.sc = bld2(B.sc, u * 2); w.dm = bld4(B.d, (u >> 3) * 4); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) {
        const int8_t *p = A.q8k + (size_t)t * K + (size_t)u * 32; a.lo = ld16(p); a.hi = ld16(p + 16);
        a.d = A.dk[(size_t)t * (K / 256) + (u >> 3)]; const int16_t *bs = A.bsk + (size_t)t * (K / 16) + 2 * u; a.bs_lo = bs[0]; a.bs_hi = bs[1]; }
    static constexpr int GROUP = 8, TERMS = 2;
    static __device__ __forceinline__ void ints(const WU &w, const AU &a, int &i0, int &i1) {
        int s0 = 0, s1 = 0; const unsigned L = w.p.x, H = w.p.y;
        s0 = dot4(L & 0x03030303, a.lo.x, s0); s0 = dot4((L >> 2) & 0x03030303, a.lo.y, s0); s0 = dot4((L >> 4) & 0x03030303, a.lo.z, s0); s0 = dot4((L >> 6) & 0x03030303, a.lo.w, s0);
        s1 = dot4(H & 0x03030303, a.hi.x, s1); s1 = dot4((H >> 2) & 0x03030303, a.hi.y, s1); s1 = dot4((H >> 4) & 0x03030303, a.hi.z, s1); s1 = dot4((H >> 6) & 0x03030303, a.hi.w, s1);
        const int sc0 = w.sc & 0xF, m0 = (w.sc >> 4) & 0xF, sc1 = (w.sc >> 8) & 0xF, m1 = w.sc >> 12;
        i0 = mul24(sc0, s0) + mul24(sc1, s1); i1 = m0 * a.bs_lo + m1 * a.bs_hi;
    }
    static __device__ __forceinline__ void terms(const WU &w, const AU &a, int i0, int i1, float &f0, float &v0, float &f1, float &v1) {
        const float d = h2f_bits(w.dm & 0xFFFF), dmin = h2f_bits(w.dm >> 16); f0 = d * a.d; v0 = (float)i0; f1 = -(dmin * a.d); v1 = (float)i1; }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        int s0 = 0, s1 = 0; const unsigned L = w.p.x, H = w.p.y;
        s0 = dot4(L & 0x03030303, a.lo.x, s0); s0 = dot4((L >> 2) & 0x03030303, a.lo.y, s0); s0 = dot4((L >> 4) & 0x03030303, a.lo.z, s0); s0 = dot4((L >> 6) & 0x03030303, a.lo.w, s0);
        s1 = dot4(H & 0x03030303, a.hi.x, s1); s1 = dot4((H >> 2) & 0x03030303, a.hi.y, s1); s1 = dot4((H >> 4) & 0x03030303, a.hi.z, s1); s1 = dot4((H >> 6) & 0x03030303, a.hi.w, s1);
        const int sc0 = w.sc & 0xF, m0 = (w.sc >> 4) & 0xF, sc1 = (w.sc >> 8) & 0xF, m1 = w.sc >> 12;
        const float d = h2f_bits(w.dm & 0xFFFF), dmin = h2f_bits(w.dm >> 16);
        acc = fmaf(d * a.d, (float)(mul24(sc0, s0) + mul24(sc1, s1)), acc);
        acc = fmaf(-(dmin * a.d), (float)(m0 * a.bs_lo + m1 * a.bs_hi), acc);
    }
};
template <> struct Tr<GT_Q4_K> {
    static constexpr int EPU = 32;
    struct WU { int4 q; int4 h; };
    using AU = AK;
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) { w.q = ldw<int4>(W.qs + g0 * 16 + (unsigned)(u * 16)); w.h = ldw<int4>(W.sc + (g0 >> 3) * 16 + (unsigned)((u >> 3) * 16)); }
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 16); B.sc = mkbuf(W.sc + (g0 >> 3) * 16); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) { w.q = bld16(B.qs, u * 16); w.h = bld16(B.sc, (u >> 3) * 16); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) {
        const int sb = u >> 3, i = u & 7, j = i >> 1, h = i & 1;
        const int8_t *p = A.q8k + (size_t)t * K + (size_t)sb * 256 + 64 * j + 16 * h; a.lo = ld16(p); a.hi = ld16(p + 32);
        a.d = A.dk[(size_t)t * (K / 256) + sb]; const int16_t *bs = A.bsk + (size_t)t * (K / 16) + sb * 16 + 4 * j + h; a.bs_lo = bs[0]; a.bs_hi = bs[2]; }
    static constexpr int GROUP = 8, TERMS = 2;
    static __device__ __forceinline__ void ints(const WU &w, const AU &a, int &i0, int &i1) {
        const int j = (threadIdx.x & 7) >> 1;
        int sc0, sc1, m0, m1; scale_min_pair(w.h, j, sc0, sc1, m0, m1);
        int s0 = 0, s1 = 0;
        s0 = dot4(w.q.x & 0x0F0F0F0F, a.lo.x, s0); s0 = dot4(w.q.y & 0x0F0F0F0F, a.lo.y, s0); s0 = dot4(w.q.z & 0x0F0F0F0F, a.lo.z, s0); s0 = dot4(w.q.w & 0x0F0F0F0F, a.lo.w, s0);
        s1 = dot4((w.q.x >> 4) & 0x0F0F0F0F, a.hi.x, s1); s1 = dot4((w.q.y >> 4) & 0x0F0F0F0F, a.hi.y, s1); s1 = dot4((w.q.z >> 4) & 0x0F0F0F0F, a.hi.z, s1); s1 = dot4((w.q.w >> 4) & 0x0F0F0F0F, a.hi.w, s1);
        i0 = mul24(sc0, s0) + mul24(sc1, s1); i1 = m0 * a.bs_lo + m1 * a.bs_hi;
    }
    static __device__ __forceinline__ void terms(const WU &w, const AU &a, int i0, int i1, float &f0, float &v0, float &f1, float &v1) {
        const float d = h2f_bits((unsigned)w.h.x & 0xFFFF), dmin = h2f_bits((unsigned)w.h.x >> 16); f0 = d * a.d; v0 = (float)i0; f1 = -(dmin * a.d); v1 = (float)i1; }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        const int j = (threadIdx.x & 7) >> 1;
        int sc0, sc1, m0, m1; scale_min_pair(w.h, j, sc0, sc1, m0, m1);
        int s0 = 0, s1 = 0;
        s0 = dot4(w.q.x & 0x0F0F0F0F, a.lo.x, s0); s0 = dot4(w.q.y & 0x0F0F0F0F, a.lo.y, s0); s0 = dot4(w.q.z & 0x0F0F0F0F, a.lo.z, s0); s0 = dot4(w.q.w & 0x0F0F0F0F, a.lo.w, s0);
        s1 = dot4((w.q.x >> 4) & 0x0F0F0F0F, a.hi.x, s1); s1 = dot4((w.q.y >> 4) & 0x0F0F0F0F, a.hi.y, s1); s1 = dot4((w.q.z >> 4) & 0x0F0F0F0F, a.hi.z, s1); s1 = dot4((w.q.w >> 4) & 0x0F0F0F0F, a.hi.w, s1);
        const float d = h2f_bits((unsigned)w.h.x & 0xFFFF), dmin = h2f_bits((unsigned)w.h.x >> 16);
        acc = fmaf(d * a.d, (float)(mul24(sc0, s0) + mul24(sc1, s1)), acc);
        acc = fmaf(-(dmin * a.d), (float)(m0 * a.bs_lo + m1 * a.bs_hi), acc);
    }
};
template <> struct Tr<GT_Q5_K> {
    static constexpr int EPU = 32;
    struct WU { int4 q; int4 h; unsigned P; };
    using AU = AK;
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) { w.q = ldw<int4>(W.qs + g0 * 16 + (unsigned)(u * 16)); w.h = ldw<int4>(W.sc + (g0 >> 3) * 16 + (unsigned)((u >> 3) * 16)); w.P = ldw<unsigned>(W.qh + g0 * 4 + (unsigned)(u * 4)); }
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 16); B.sc = mkbuf(W.sc + (g0 >> 3) * 16); B.qh = mkbuf(W.qh + g0 * 4); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) { w.q = bld16(B.qs, u * 16); w.h = bld16(B.sc, (u >> 3) * 16); w.P = bld4(B.qh, u * 4); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) { Tr<GT_Q4_K>::loada(A, t, K, u, a); }
    static constexpr int GROUP = 8, TERMS = 2;
    static __device__ __forceinline__ void ints(const WU &w, const AU &a, int &i0, int &i1) {
        const int j = (threadIdx.x & 7) >> 1;
        int sc0, sc1, m0, m1; scale_min_pair(w.h, j, sc0, sc1, m0, m1);
        const unsigned P = w.P; int s0 = 0, s1 = 0;
        s0 = dot4((w.q.x & 0x0F0F0F0F) | ((P << 4) & 0x10101010), a.lo.x, s0); s0 = dot4((w.q.y & 0x0F0F0F0F) | ((P << 3) & 0x10101010), a.lo.y, s0);
        s0 = dot4((w.q.z & 0x0F0F0F0F) | ((P << 2) & 0x10101010), a.lo.z, s0); s0 = dot4((w.q.w & 0x0F0F0F0F) | ((P << 1) & 0x10101010), a.lo.w, s0);
        s1 = dot4(((w.q.x >> 4) & 0x0F0F0F0F) | (P & 0x10101010), a.hi.x, s1); s1 = dot4(((w.q.y >> 4) & 0x0F0F0F0F) | ((P >> 1) & 0x10101010), a.hi.y, s1);
        s1 = dot4(((w.q.z >> 4) & 0x0F0F0F0F) | ((P >> 2) & 0x10101010), a.hi.z, s1); s1 = dot4(((w.q.w >> 4) & 0x0F0F0F0F) | ((P >> 3) & 0x10101010), a.hi.w, s1);
        i0 = mul24(sc0, s0) + mul24(sc1, s1); i1 = m0 * a.bs_lo + m1 * a.bs_hi;
    }
    static __device__ __forceinline__ void terms(const WU &w, const AU &a, int i0, int i1, float &f0, float &v0, float &f1, float &v1) {
        const float d = h2f_bits((unsigned)w.h.x & 0xFFFF), dmin = h2f_bits((unsigned)w.h.x >> 16); f0 = d * a.d; v0 = (float)i0; f1 = -(dmin * a.d); v1 = (float)i1; }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        const int j = (threadIdx.x & 7) >> 1;
        int sc0, sc1, m0, m1; scale_min_pair(w.h, j, sc0, sc1, m0, m1);
        const unsigned P = w.P; int s0 = 0, s1 = 0;
        s0 = dot4((w.q.x & 0x0F0F0F0F) | ((P << 4) & 0x10101010), a.lo.x, s0); s0 = dot4((w.q.y & 0x0F0F0F0F) | ((P << 3) & 0x10101010), a.lo.y, s0);
        s0 = dot4((w.q.z & 0x0F0F0F0F) | ((P << 2) & 0x10101010), a.lo.z, s0); s0 = dot4((w.q.w & 0x0F0F0F0F) | ((P << 1) & 0x10101010), a.lo.w, s0);
        s1 = dot4(((w.q.x >> 4) & 0x0F0F0F0F) | (P & 0x10101010), a.hi.x, s1); s1 = dot4(((w.q.y >> 4) & 0x0F0F0F0F) | ((P >> 1) & 0x10101010), a.hi.y, s1);
        s1 = dot4(((w.q.z >> 4) & 0x0F0F0F0F) | ((P >> 2) & 0x10101010), a.hi.z, s1); s1 = dot4(((w.q.w >> 4) & 0x0F0F0F0F) | ((P >> 3) & 0x10101010), a.hi.w, s1);
        const float d = h2f_bits((unsigned)w.h.x & 0xFFFF), dmin = h2f_bits((unsigned)w.h.x >> 16);
        acc = fmaf(d * a.d, (float)(mul24(sc0, s0) + mul24(sc1, s1)), acc);
        acc = fmaf(-(dmin * a.d), (float)(m0 * a.bs_lo + m1 * a.bs_hi), acc);
    }
};
template <> struct Tr<GT_Q6_K> {
    static constexpr int EPU = 32;
    struct WU { int4 q; unsigned Plo, Phi; unsigned short sc, dh; };
    using AU = AK;
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) {
        w.q = ldw<int4>(W.qs + g0 * 16 + (unsigned)(u * 16)); const uint2 p = ldw<uint2>(W.qh + g0 * 8 + (unsigned)(u * 8)); w.Plo = p.x; w.Phi = p.y;
        w.sc = ldw<unsigned short>(W.sc + g0 * 2 + (unsigned)(u * 2));
        w.dh = ldw<unsigned short>(W.d + (g0 >> 3) * 2 + (unsigned)((u >> 3) * 2)); }
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 16); B.qh = mkbuf(W.qh + g0 * 8); B.sc = mkbuf(W.sc + g0 * 2); B.d = mkbuf(W.d + (g0 >> 3) * 2); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) {
        w.q = bld16(B.qs, u * 16); const uint2 p = bld8(B.qh, u * 8); w.Plo = p.x; w.Phi = p.y;
        w.sc = bld2(B.sc, u * 2);
        w.dh = bld2(B.d, (u >> 3) * 2); }
static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) {
        const int sb = u >> 3, i = u & 7, n = i >> 2, c = (i >> 1) & 1, h = i & 1;
        const int8_t *p = A.q8k + (size_t)t * K + (size_t)sb * 256 + 128 * n + 32 * c + 16 * h; a.lo = ld16(p); a.hi = ld16(p + 64);
        a.d = A.dk[(size_t)t * (K / 256) + sb]; const int16_t *bs = A.bsk + (size_t)t * (K / 16) + sb * 16 + 8 * n + 2 * c + h; a.bs_lo = bs[0]; a.bs_hi = bs[4]; }
    static constexpr int GROUP = 8, TERMS = 1;
    static __device__ __forceinline__ void ints(const WU &w, const AU &a, int &i0, int &i1) {
        int s0 = 0, s1 = 0; const unsigned L = w.Plo, H = w.Phi;
        s0 = dot4((w.q.x & 0x0F0F0F0F) | ((L << 4) & 0x30303030), a.lo.x, s0); s0 = dot4((w.q.y & 0x0F0F0F0F) | ((L << 2) & 0x30303030), a.lo.y, s0);
        s0 = dot4((w.q.z & 0x0F0F0F0F) | (L & 0x30303030), a.lo.z, s0); s0 = dot4((w.q.w & 0x0F0F0F0F) | ((L >> 2) & 0x30303030), a.lo.w, s0);
        s1 = dot4(((w.q.x >> 4) & 0x0F0F0F0F) | ((H << 4) & 0x30303030), a.hi.x, s1); s1 = dot4(((w.q.y >> 4) & 0x0F0F0F0F) | ((H << 2) & 0x30303030), a.hi.y, s1);
        s1 = dot4(((w.q.z >> 4) & 0x0F0F0F0F) | (H & 0x30303030), a.hi.z, s1); s1 = dot4(((w.q.w >> 4) & 0x0F0F0F0F) | ((H >> 2) & 0x30303030), a.hi.w, s1);
        s0 -= 32 * a.bs_lo; s1 -= 32 * a.bs_hi;
        i0 = mul24((int)(signed char)(w.sc & 0xFF), s0) + mul24((int)(signed char)(w.sc >> 8), s1); i1 = 0;
    }
    static __device__ __forceinline__ void terms(const WU &w, const AU &a, int i0, int, float &f0, float &v0, float &f1, float &v1) { f0 = h2f_bits(w.dh) * a.d; v0 = (float)i0; f1 = v1 = 0.0f; }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        int s0 = 0, s1 = 0; const unsigned L = w.Plo, H = w.Phi;
        s0 = dot4((w.q.x & 0x0F0F0F0F) | ((L << 4) & 0x30303030), a.lo.x, s0); s0 = dot4((w.q.y & 0x0F0F0F0F) | ((L << 2) & 0x30303030), a.lo.y, s0);
        s0 = dot4((w.q.z & 0x0F0F0F0F) | (L & 0x30303030), a.lo.z, s0); s0 = dot4((w.q.w & 0x0F0F0F0F) | ((L >> 2) & 0x30303030), a.lo.w, s0);
        s1 = dot4(((w.q.x >> 4) & 0x0F0F0F0F) | ((H << 4) & 0x30303030), a.hi.x, s1); s1 = dot4(((w.q.y >> 4) & 0x0F0F0F0F) | ((H << 2) & 0x30303030), a.hi.y, s1);
        s1 = dot4(((w.q.z >> 4) & 0x0F0F0F0F) | (H & 0x30303030), a.hi.z, s1); s1 = dot4(((w.q.w >> 4) & 0x0F0F0F0F) | ((H >> 2) & 0x30303030), a.hi.w, s1);
        s0 -= 32 * a.bs_lo; s1 -= 32 * a.bs_hi;
        acc = fmaf(h2f_bits(w.dh) * a.d, (float)(mul24((int)(signed char)(w.sc & 0xFF), s0) + mul24((int)(signed char)(w.sc >> 8), s1)), acc);
    }
};
template <> struct Tr<GT_F16> {
    static constexpr int EPU = 8;
    struct WU { int4 q; };
    struct AU { int4 a; };
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) { w.q = ldw<int4>(W.qs + g0 * 16 + (unsigned)(u * 16)); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) { a.a = ld16(reinterpret_cast<const uint8_t *>(A.xh) + ((size_t)t * K + (size_t)u * 8) * 2); }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        const unsigned wq[4] = {(unsigned)w.q.x, (unsigned)w.q.y, (unsigned)w.q.z, (unsigned)w.q.w}, aq[4] = {(unsigned)a.a.x, (unsigned)a.a.y, (unsigned)a.a.z, (unsigned)a.a.w};
#pragma unroll
        for (int i = 0; i < 4; i++) { acc = fmaf(h2f_bits(wq[i] & 0xFFFF), h2f_bits(aq[i] & 0xFFFF), acc); acc = fmaf(h2f_bits(wq[i] >> 16), h2f_bits(aq[i] >> 16), acc); }
    }
};
template <> struct Tr<GT_F32> {
    static constexpr int EPU = 4;
    struct WU { float4 q; };
    struct AU { float4 a; };
    static __device__ __forceinline__ void loadw(const QWeight &W, size_t g0, int u, WU &w) { w.q = ldw<float4>(W.qs + g0 * 16 + (unsigned)(u * 16)); }
    static __device__ __forceinline__ void loada(const ActQ &A, int t, int K, int u, AU &a) { a.a = *reinterpret_cast<const float4 *>(A.xf + (size_t)t * K + (size_t)u * 4); }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) { acc = fmaf(w.q.x, a.a.x, acc); acc = fmaf(w.q.y, a.a.y, acc); acc = fmaf(w.q.z, a.a.z, acc); acc = fmaf(w.q.w, a.a.w, acc); }
};

// =====================================================================================================================
// Decode mat-vec traits (k_matvec_v2): the unit traits above, plus buffer-descriptor loads for the two types whose 16-byte unit covers fewer than 32 weights -- Q8_0
// (16 weights: half a block) and F16 (8 weights).  A lane's units stay lane-contiguous 16-byte pieces (u = lane + 64 i: every wave load instruction reads 1 KiB of whole
// cache lines), so a row needs more units per lane than the 32-weight types (K = 5120: F16 10, Q8_0 5; K = 11008: Q8_0 11).  Q8_0: each HALF block is scaled by the
// block's d on its own lane (d * (s_lo + s_hi) = d * s_lo + d * s_hi up to fp32 rounding; the oracle-order kernels keep the lane-pair sum of Tr<GT_Q8_0>).
// The real Vicuna-v0 file has n_vocab = 32001, for which llama.cpp's k-quant mixes fall back to an F16 output matrix (327.7 MB streamed per token).
// =====================================================================================================================
template <int T> struct TrMV : Tr<T> {};
template <> struct TrMV<GT_Q8_0> : Tr<GT_Q8_0> {
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 16); B.sc = mkbuf(W.sc + (g0 >> 1) * 2); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) { w.q = bld16(B.qs, u * 16); w.dh = bld2(B.sc, (u >> 1) * 2); }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {
        int s = 0;
        s = dot4(w.q.x, a.a.x, s); s = dot4(w.q.y, a.a.y, s); s = dot4(w.q.z, a.a.z, s); s = dot4(w.q.w, a.a.w, s);
        acc = fmaf(h2f_bits(w.dh) * a.d, (float)s, acc);
    }
};
typedef _Float16 v2h_t __attribute__((ext_vector_type(2)));
template <> struct TrMV<GT_F16> : Tr<GT_F16> {
    static __device__ __forceinline__ void mkb(const QWeight &W, size_t g0, WBuf &B) { B.qs = mkbuf(W.qs + g0 * 16); }
    static __device__ __forceinline__ void loadb(const WBuf &B, int u, WU &w) { w.q = bld16(B.qs, u * 16); }
    static __device__ __forceinline__ void dot(const WU &w, const AU &a, float &acc) {   // v_dot2_f32_f16: exact fp16 products, fp32 accumulate
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h_t, w.q.x), __builtin_bit_cast(v2h_t, a.a.x), acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h_t, w.q.y), __builtin_bit_cast(v2h_t, a.a.y), acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h_t, w.q.z), __builtin_bit_cast(v2h_t, a.a.z), acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h_t, w.q.w), __builtin_bit_cast(v2h_t, a.a.w), acc, false);
    }
};

// =====================================================================================================================
// activation preparation: (rms_norm * w | silu(a)*b | identity) -> {Q8_K, Q8_0/Q8_1, f16, f32}
// thread t of a 256-thread group owns 4 consecutive values; a wave = one 256-wide Q8_K block, 8 lanes = one 32-wide block.
// =====================================================================================================================
__device__ __forceinline__ void quant_emit4(const float v[4], const bool in_range, const int idx /*first element index in the row*/, const size_t row, const int K,
                                            const ActQ &A, const int mask) {
    const int lane = threadIdx.x & 63;
    if (mask & ACT_F32) { if (in_range) *reinterpret_cast<float4 *>(A.xf + row * K + idx) = make_float4(v[0], v[1], v[2], v[3]); }
    if (mask & ACT_F16) { if (in_range) { __half2 h0 = __halves2half2(f2h_rn(v[0]), f2h_rn(v[1])), h1 = __halves2half2(f2h_rn(v[2]), f2h_rn(v[3]));
            uint2 o; o.x = *reinterpret_cast<unsigned *>(&h0); o.y = *reinterpret_cast<unsigned *>(&h1); *reinterpret_cast<uint2 *>(A.xh + row * K + idx) = o; } }
    if (mask & ACT_Q8K) {
        // signed value of the FIRST element with the largest magnitude in the 256-block (ggml quantize_row_q8_K): the wave = one block,
        // lanes are in element order, so it is the first maximum of the lowest lane whose local maximum equals the wave maximum.
        float best = 0.0f, bav = -1.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) { const float av = fabsf(v[e]); if (av > bav) { bav = av; best = v[e]; } }
        const float amax = wave_max(bav);
        const unsigned long long m = __ballot(bav == amax);
        const float maxv = readlane_f(best, __ffsll((long long)m) - 1);
        int q[4] = {0, 0, 0, 0}; float d = 0.0f;
        if (amax != 0.0f) {
            const float iscale = -128.f / maxv;
#pragma unroll
            for (int e = 0; e < 4; e++) { const int r = (int)rintf(iscale * v[e]); q[e] = r > 127 ? 127 : r; }
            d = 1.0f / iscale;
        }
        int s = q[0] + q[1] + q[2] + q[3];
        s += dpp_i<0xB1>(s); s += dpp_i<0x4E>(s);
        if (in_range) {
            const unsigned pk = (unsigned)(q[0] & 0xFF) | ((unsigned)(q[1] & 0xFF) << 8) | ((unsigned)(q[2] & 0xFF) << 16) | ((unsigned)(q[3] & 0xFF) << 24);
            *reinterpret_cast<unsigned *>(A.q8k + row * K + idx) = pk;
            if ((lane & 3) == 0) A.bsk[row * (K / 16) + idx / 16] = (int16_t)s;
            if (lane == 0) A.dk[row * (K / 256) + idx / 256] = d;
        }
        if (A.bsq) {   // prefill (csrc/mmq2_kernels.hip): the per-32 sums as two int8 digits, so that sum_j m_j * bsum_j runs on the int8 matrix cores
            const int s32 = s + dpp_i<0x141>(s);                              // row_half_mirror: the other 16-group of this 32-block
            if (in_range && (lane & 7) == 0) { int8_t *o = A.bsq + (row * (K / 256) + idx / 256) * 16 + ((idx & 255) >> 5); o[0] = (int8_t)(s32 & 127); o[8] = (int8_t)(s32 >> 7);
                if (A.bs16) { __half *oh = A.bs16 + (row * (K / 256) + idx / 256) * 16 + ((idx & 255) >> 5); oh[0] = __int2half_rn(s32 & 127); oh[8] = __int2half_rn(s32 >> 7); } }
        }
        if (A.q16 && in_range) {   // the same int8 values as fp16 in k_mmqh_q45k's fragment order (common.hpp): this thread's 4 elements are one half of one 16-byte chunk
            const int i = idx & 255, c = (i >> 6) * 8 + ((i >> 4) & 1) * 4 + ((i >> 2) & 3), x = (i >> 5) & 1;
            const __half2 h0 = __halves2half2(__int2half_rn(q[0]), __int2half_rn(q[2])), h1 = __halves2half2(__int2half_rn(q[1]), __int2half_rn(q[3]));
            uint2 o; o.x = *reinterpret_cast<const unsigned *>(&h0); o.y = *reinterpret_cast<const unsigned *>(&h1);
            *reinterpret_cast<uint2 *>(A.q16 + row * K + (size_t)(idx & ~255) + c * 8 + x * 4) = o;
        }
    }
    if (mask & ACT_Q80) {
        float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        amax = fmaxf(amax, dpp_f<0xB1>(amax)); amax = fmaxf(amax, dpp_f<0x4E>(amax)); amax = fmaxf(amax, dpp_f<0x141>(amax));   // 8 lanes = one 32-block
        const float d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
        int q[4];
#pragma unroll
        for (int e = 0; e < 4; e++) q[e] = (int)rintf(v[e] * id);
        int s = q[0] + q[1] + q[2] + q[3];
        s += dpp_i<0xB1>(s); s += dpp_i<0x4E>(s); s += dpp_i<0x141>(s);
        if (in_range) {
            const unsigned pk = (unsigned)(q[0] & 0xFF) | ((unsigned)(q[1] & 0xFF) << 8) | ((unsigned)(q[2] & 0xFF) << 16) | ((unsigned)(q[3] & 0xFF) << 24);
            *reinterpret_cast<unsigned *>(A.q80 + row * K + idx) = pk;
            if ((lane & 7) == 0) { const size_t b = row * (K / 32) + idx / 32; A.d0[b] = f16r(d); A.d1[b] = d; A.s1[b] = d * (float)s; A.sum0[b] = s; }
        }
    }
}

}  // namespace mg4
