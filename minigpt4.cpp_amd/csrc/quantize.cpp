// minigpt4_quantize_model (reference minigpt4.cpp:2817-2982) -- host-only offline tool: re-quantises the eligible Linear weights of a vision file
// ("ggml" v1 container) with ggml's REFERENCE block quantisers and writes the result in the same container (MiniGPT4ModelLoader::dump, :1632-1717).
//
// The quantisers below restate ggml @ llama.cpp master-31cfbb1 (`quantize_row_q4_0_reference` ... in ggml.c, `quantize_row_q4_K_reference` ... and the
// `make_qkx1_quants` / `make_qx_quants` searches in k_quants.c) from their published form [UPSTREAM-RECALL: the sources are not on this machine; the same
// algorithms are restated independently in oracle/refquant.py and the two must agree byte for byte, tests/test_cpu_quantize.py].  `ggml_quantize_chunk`
// (reference call site :2932) runs these over the FLAT element array of a tensor.
#include "quantize.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "common.hpp"
#include "formats.hpp"

namespace mg4 {
namespace {

inline uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }
inline float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
inline void put16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)(v & 0xFF); p[1] = (uint8_t)(v >> 8); }
inline int nearest_int(float fval) { float val = fval + 12582912.f; int i; memcpy(&i, &val, sizeof(int)); return (i & 0x007fffff) - 0x00400000; }

// ---- 32-element blocks (ggml.c) ----------------------------------------------------------------------------------------------
void q4_0_block(const float *x, uint8_t *y) {
    float amax = 0.0f, max = 0.0f;
    for (int j = 0; j < 32; j++) { const float v = x[j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
    const float d = max / -8; const float id = d ? 1.0f / d : 0.0f;
    put16(y, f2h(d));
    for (int j = 0; j < 16; j++) {
        const float x0 = x[j] * id, x1 = x[16 + j] * id;
        const uint8_t xi0 = (uint8_t)std::min(15, (int)(int8_t)(x0 + 8.5f)), xi1 = (uint8_t)std::min(15, (int)(int8_t)(x1 + 8.5f));
        y[2 + j] = (uint8_t)(xi0 | (xi1 << 4));
    }
}
void q4_1_block(const float *x, uint8_t *y) {
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int j = 0; j < 32; j++) { const float v = x[j]; if (v < mn) mn = v; if (v > mx) mx = v; }
    const float d = (mx - mn) / ((1 << 4) - 1); const float id = d ? 1.0f / d : 0.0f;
    put16(y, f2h(d)); put16(y + 2, f2h(mn));
    for (int j = 0; j < 16; j++) {
        const float x0 = (x[j] - mn) * id, x1 = (x[16 + j] - mn) * id;
        const uint8_t xi0 = (uint8_t)std::min(15, (int)(int8_t)(x0 + 0.5f)), xi1 = (uint8_t)std::min(15, (int)(int8_t)(x1 + 0.5f));
        y[4 + j] = (uint8_t)(xi0 | (xi1 << 4));
    }
}
void q5_0_block(const float *x, uint8_t *y) {
    float amax = 0.0f, max = 0.0f;
    for (int j = 0; j < 32; j++) { const float v = x[j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
    const float d = max / -16; const float id = d ? 1.0f / d : 0.0f;
    put16(y, f2h(d));
    uint32_t qh = 0;
    for (int j = 0; j < 16; j++) {
        const float x0 = x[j] * id, x1 = x[16 + j] * id;
        const uint8_t xi0 = (uint8_t)std::min(31, (int)(int8_t)(x0 + 16.5f)), xi1 = (uint8_t)std::min(31, (int)(int8_t)(x1 + 16.5f));
        y[6 + j] = (uint8_t)((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
        qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
        qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
    }
    memcpy(y + 2, &qh, 4);
}
void q5_1_block(const float *x, uint8_t *y) {
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int j = 0; j < 32; j++) { const float v = x[j]; if (v < mn) mn = v; if (v > mx) mx = v; }
    const float d = (mx - mn) / ((1 << 5) - 1); const float id = d ? 1.0f / d : 0.0f;
    put16(y, f2h(d)); put16(y + 2, f2h(mn));
    uint32_t qh = 0;
    for (int j = 0; j < 16; j++) {
        const float x0 = (x[j] - mn) * id, x1 = (x[16 + j] - mn) * id;
        const uint8_t xi0 = (uint8_t)(x0 + 0.5f), xi1 = (uint8_t)(x1 + 0.5f);
        y[8 + j] = (uint8_t)((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
        qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
        qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
    }
    memcpy(y + 4, &qh, 4);
}
void q8_0_block(const float *x, uint8_t *y) {
    float amax = 0.0f;
    for (int j = 0; j < 32; j++) amax = std::max(amax, fabsf(x[j]));
    const float d = amax / ((1 << 7) - 1); const float id = d ? 1.0f / d : 0.0f;
    put16(y, f2h(d));
    for (int j = 0; j < 32; j++) y[2 + j] = (uint8_t)(int8_t)roundf(x[j] * id);
}

// ---- k-quants (k_quants.c) ------------------------------------------------------------------------------------------------------
float make_qkx1_quants(int n, int nmax, const float *x, uint8_t *L, float *the_min, int ntry) {
    float mn = x[0], mx = x[0];
    for (int i = 1; i < n; i++) { if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
    if (mx == mn) { for (int i = 0; i < n; i++) L[i] = 0; *the_min = 0; return 0.f; }
    if (mn > 0) mn = 0;
    float iscale = nmax / (mx - mn);
    float scale = 1 / iscale;
    for (int itry = 0; itry < ntry; itry++) {
        float sumlx = 0; int suml2 = 0; bool did_change = false;
        for (int i = 0; i < n; i++) {
            int l = nearest_int(iscale * (x[i] - mn));
            l = std::max(0, std::min(nmax, l));
            if (l != L[i]) { L[i] = (uint8_t)l; did_change = true; }
            sumlx += (x[i] - mn) * l;
            suml2 += l * l;
        }
        scale = sumlx / suml2;
        float sum = 0;
        for (int i = 0; i < n; i++) sum += x[i] - scale * L[i];
        mn = sum / n;
        if (mn > 0) mn = 0;
        iscale = 1 / scale;
        if (!did_change) break;
    }
    *the_min = -mn;
    return scale;
}
// rmse_type 1: weights x^2, as quantize_row_q6_K_reference calls it
float make_qx_quants(int n, int nmax, const float *x, int8_t *L, int rmse_type) {
    float max = 0, amax = 0;
    for (int i = 0; i < n; i++) { const float ax = fabsf(x[i]); if (ax > amax) { amax = ax; max = x[i]; } }
    if (!amax) { for (int i = 0; i < n; i++) L[i] = 0; return 0.f; }
    float iscale = -nmax / max;
    if (rmse_type == 0) {
        for (int i = 0; i < n; i++) { const int l = nearest_int(iscale * x[i]); L[i] = (int8_t)(nmax + std::max(-nmax, std::min(nmax - 1, l))); }
        return 1 / iscale;
    }
    const int weight_type = rmse_type % 2;
    float sumlx = 0, suml2 = 0;
    for (int i = 0; i < n; i++) {
        int l = nearest_int(iscale * x[i]);
        l = std::max(-nmax, std::min(nmax - 1, l));
        L[i] = (int8_t)(l + nmax);
        const float w = weight_type == 1 ? x[i] * x[i] : 1;
        sumlx += w * x[i] * l;
        suml2 += w * l * l;
    }
    float scale = sumlx / suml2;
    float best = scale * sumlx;
    for (int itry = 0; itry < 3; itry++) {
        iscale = 1 / scale;
        float slx = 0, sl2 = 0; bool changed = false;
        for (int i = 0; i < n; i++) {
            int l = nearest_int(iscale * x[i]);
            l = std::max(-nmax, std::min(nmax - 1, l));
            if (l + nmax != L[i]) changed = true;
            const float w = weight_type == 1 ? x[i] * x[i] : 1.f;
            slx += w * x[i] * l;
            sl2 += w * l * l;
        }
        if (!changed || sl2 == 0 || slx * slx <= best * sl2) break;
        for (int i = 0; i < n; i++) { const int l = nearest_int(iscale * x[i]); L[i] = (int8_t)(nmax + std::max(-nmax, std::min(nmax - 1, l))); }
        sumlx = slx; suml2 = sl2;
        scale = sumlx / suml2;
        best = scale * sumlx;
    }
    for (int itry = 0; itry < 5; itry++) {
        int n_changed = 0;
        for (int i = 0; i < n; i++) {
            const float w = weight_type == 1 ? x[i] * x[i] : 1;
            const int l = L[i] - nmax;
            float slx = sumlx - w * x[i] * l;
            if (slx > 0) {
                float sl2 = suml2 - w * l * l;
                int new_l = nearest_int(x[i] * sl2 / slx);
                new_l = std::max(-nmax, std::min(nmax - 1, new_l));
                if (new_l != l) {
                    slx += w * x[i] * new_l;
                    sl2 += w * new_l * new_l;
                    if (sl2 > 0 && slx * slx * suml2 > sumlx * sumlx * sl2) {
                        L[i] = (int8_t)(nmax + new_l); sumlx = slx; suml2 = sl2;
                        scale = sumlx / suml2; best = scale * sumlx;
                        ++n_changed;
                    }
                }
            }
        }
        if (!n_changed) break;
    }
    return scale;
}
inline void get_scale_min_k4(int j, const uint8_t *q, uint8_t *d, uint8_t *m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (uint8_t)((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4)); *m = (uint8_t)((q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4)); }
}
// shared front half of Q4_K / Q5_K: per-32 (scale, min) search, 6-bit packing, re-quantisation against the packed scales.  Returns L[256].
void q45_K_levels(const float *x, int nmax, uint8_t *y_d, uint8_t *y_scales, uint8_t *L) {
    float mins[8], scales[8];
    float max_scale = 0, max_min = 0;
    for (int j = 0; j < 8; j++) {
        scales[j] = make_qkx1_quants(32, nmax, x + 32 * j, L + 32 * j, &mins[j], 5);
        if (scales[j] > max_scale) max_scale = scales[j];
        if (mins[j] > max_min) max_min = mins[j];
    }
    const float inv_scale = max_scale > 0 ? 63.f / max_scale : 0.f, inv_min = max_min > 0 ? 63.f / max_min : 0.f;
    memset(y_scales, 0, 12);
    for (int j = 0; j < 8; j++) {
        uint8_t ls = (uint8_t)std::min(63, nearest_int(inv_scale * scales[j])), lm = (uint8_t)std::min(63, nearest_int(inv_min * mins[j]));
        if (j < 4) { y_scales[j] = ls; y_scales[j + 4] = lm; }
        else { y_scales[j + 4] = (uint8_t)((ls & 0xF) | ((lm & 0xF) << 4)); y_scales[j - 4] |= (uint8_t)((ls >> 4) << 6); y_scales[j - 0] |= (uint8_t)((lm >> 4) << 6); }
    }
    put16(y_d, f2h(max_scale / 63.f)); put16(y_d + 2, f2h(max_min / 63.f));
    const float dd = h2f((uint16_t)(y_d[0] | (y_d[1] << 8))), dmin = h2f((uint16_t)(y_d[2] | (y_d[3] << 8)));
    for (int j = 0; j < 8; j++) {
        uint8_t sc, m; get_scale_min_k4(j, y_scales, &sc, &m);
        const float d = dd * sc; if (!d) continue;
        const float dm = dmin * m;
        for (int ii = 0; ii < 32; ii++) { int l = nearest_int((x[32 * j + ii] + dm) / d); l = std::max(0, std::min(nmax, l)); L[32 * j + ii] = (uint8_t)l; }
    }
}
// quantize_row_q2_K_reference [UPSTREAM-RECALL]: per 16 weights make_qkx1_quants(16, 3, ..., 5 tries) -> (scale, min); both against the largest of the super-block on
// 4 bits; levels re-derived against the packed (fp16-rounded) scales; four 2-bit planes per 128 weights.  block_q2_K = {scales[16], qs[64], d, dmin}
void q2_K_block(const float *x, uint8_t *y) {
    uint8_t L[256]; float mins[16], scales[16];
    float max_scale = 0, max_min = 0;
    for (int j = 0; j < 16; j++) {
        scales[j] = make_qkx1_quants(16, 3, x + 16 * j, L + 16 * j, &mins[j], 5);
        if (scales[j] > max_scale) max_scale = scales[j];
        if (mins[j] > max_min) max_min = mins[j];
    }
    uint8_t *ysc = y, *qs = y + 16, *yd = y + 80, *ydm = y + 82;
    if (max_scale > 0) { const float iscale = 15.f / max_scale; for (int j = 0; j < 16; j++) ysc[j] = (uint8_t)nearest_int(iscale * scales[j]); put16(yd, f2h(max_scale / 15.f)); }
    else { memset(ysc, 0, 16); put16(yd, f2h(0.f)); }
    if (max_min > 0) { const float iscale = 15.f / max_min; for (int j = 0; j < 16; j++) ysc[j] |= (uint8_t)(nearest_int(iscale * mins[j]) << 4); put16(ydm, f2h(max_min / 15.f)); }
    else put16(ydm, f2h(0.f));
    const float dd = h2f((uint16_t)(yd[0] | (yd[1] << 8))), dmin = h2f((uint16_t)(ydm[0] | (ydm[1] << 8)));
    for (int j = 0; j < 16; j++) {
        const float d = dd * (ysc[j] & 0xF); if (!d) continue;
        const float dm = dmin * (ysc[j] >> 4);
        for (int ii = 0; ii < 16; ii++) { int l = nearest_int((x[16 * j + ii] + dm) / d); l = std::max(0, std::min(3, l)); L[16 * j + ii] = (uint8_t)l; }
    }
    for (int j = 0; j < 256; j += 128) for (int l = 0; l < 32; l++) qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
}
// make_q3_quants(n, nmax, x, L, do_rmse = true) [UPSTREAM-RECALL]: signed levels in [-nmax, nmax - 1] against the signed maximum, then up to five sweeps of single-level
// moves that raise (sum w x l)^2 / (sum w l^2) with w = x^2; returns sum w x l / sum w l^2 and stores the levels offset by nmax
float make_q3_quants(int n, int nmax, const float *x, int8_t *L) {
    float max = 0, amax = 0;
    for (int i = 0; i < n; i++) { const float ax = fabsf(x[i]); if (ax > amax) { amax = ax; max = x[i]; } }
    if (!amax) { for (int i = 0; i < n; i++) L[i] = 0; return 0.f; }
    const float iscale = -nmax / max;
    float sumlx = 0, suml2 = 0;
    for (int i = 0; i < n; i++) {
        int l = nearest_int(iscale * x[i]); l = std::max(-nmax, std::min(nmax - 1, l));
        L[i] = (int8_t)l;
        const float w = x[i] * x[i];
        sumlx += w * x[i] * l; suml2 += w * l * l;
    }
    for (int itry = 0; itry < 5; itry++) {
        int n_changed = 0;
        for (int i = 0; i < n; i++) {
            const float w = x[i] * x[i];
            float slx = sumlx - w * x[i] * L[i];
            if (slx > 0) {
                float sl2 = suml2 - w * L[i] * L[i];
                int new_l = nearest_int(x[i] * sl2 / slx); new_l = std::max(-nmax, std::min(nmax - 1, new_l));
                if (new_l != L[i]) {
                    slx += w * x[i] * new_l; sl2 += w * new_l * new_l;
                    if (sl2 > 0 && slx * slx * suml2 > sumlx * sumlx * sl2) { L[i] = (int8_t)new_l; sumlx = slx; suml2 = sl2; ++n_changed; }
                }
            }
        }
        if (!n_changed) break;
    }
    for (int i = 0; i < n; i++) L[i] = (int8_t)(L[i] + nmax);
    return sumlx / suml2;
}
// quantize_row_q3_K_reference [UPSTREAM-RECALL]: per 16 weights make_q3_quants(16, 4) -> signed scale; the sixteen scales against the one of largest magnitude on 6 bits
// (offset 32: low nibbles in scales[0..7], the two high bits in scales[8..11]); levels re-derived against the packed scales; bit 2 of every level in hmask (weight j -> byte
// j % 32, bit j / 32), the low two bits as four 2-bit planes per 128 weights.  block_q3_K = {hmask[32], qs[64], scales[12], d}
void q3_K_block(const float *x, uint8_t *y) {
    int8_t L[256]; float scales[16];
    float max_scale = 0, amax = 0;
    for (int j = 0; j < 16; j++) {
        scales[j] = make_q3_quants(16, 4, x + 16 * j, L + 16 * j);
        const float a = fabsf(scales[j]); if (a > amax) { amax = a; max_scale = scales[j]; }
    }
    uint8_t *hmask = y, *qs = y + 32, *ysc = y + 96, *yd = y + 108;
    memset(ysc, 0, 12);
    if (max_scale) {
        const float iscale = -32.f / max_scale;
        for (int j = 0; j < 16; j++) {
            int l = (int8_t)nearest_int(iscale * scales[j]); l = std::max(-32, std::min(31, l)) + 32;
            if (j < 8) ysc[j] = (uint8_t)(l & 0xF); else ysc[j - 8] |= (uint8_t)((l & 0xF) << 4);
            ysc[j % 4 + 8] |= (uint8_t)((l >> 4) << (2 * (j / 4)));
        }
        put16(yd, f2h(1 / iscale));
    } else put16(yd, f2h(0.f));
    const float dd = h2f((uint16_t)(yd[0] | (yd[1] << 8)));
    for (int j = 0; j < 16; j++) {
        int sc = j < 8 ? ysc[j] & 0xF : ysc[j - 8] >> 4;
        sc = (sc | (((ysc[8 + j % 4] >> (2 * (j / 4))) & 3) << 4)) - 32;
        const float d = dd * sc; if (!d) continue;
        for (int ii = 0; ii < 16; ii++) { int l = nearest_int(x[16 * j + ii] / d); l = std::max(-4, std::min(3, l)); L[16 * j + ii] = (int8_t)(l + 4); }
    }
    memset(hmask, 0, 32);
    for (int j = 0; j < 256; j++) if (L[j] > 3) { hmask[j % 32] |= (uint8_t)(1u << (j / 32)); L[j] = (int8_t)(L[j] - 4); }
    for (int j = 0; j < 256; j += 128) for (int l = 0; l < 32; l++) qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
}
void q4_K_block(const float *x, uint8_t *y) {   // {d, dmin, scales[12], qs[128]}
    uint8_t L[256];
    q45_K_levels(x, 15, y, y + 4, L);
    uint8_t *q = y + 16;
    for (int j = 0; j < 256; j += 64) { for (int l = 0; l < 32; l++) q[l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 4)); q += 32; }
}
void q5_K_block(const float *x, uint8_t *y) {   // {d, dmin, scales[12], qh[32], qs[128]}
    uint8_t L[256];
    q45_K_levels(x, 31, y, y + 4, L);
    uint8_t *qh = y + 16, *ql = y + 48;
    memset(qh, 0, 32);
    uint8_t m1 = 1, m2 = 2;
    for (int n = 0; n < 256; n += 64) {
        for (int j = 0; j < 32; j++) {
            int l1 = L[n + j]; if (l1 > 15) { l1 -= 16; qh[j] |= m1; }
            int l2 = L[n + j + 32]; if (l2 > 15) { l2 -= 16; qh[j] |= m2; }
            ql[j] = (uint8_t)(l1 | (l2 << 4));
        }
        m1 = (uint8_t)(m1 << 2); m2 = (uint8_t)(m2 << 2); ql += 32;
    }
}
void q6_K_block(const float *x, uint8_t *y) {   // {ql[128], qh[64], scales[16] (int8), d}
    int8_t L[256]; float scales[16];
    float max_scale = 0, max_abs_scale = 0;
    for (int ib = 0; ib < 16; ib++) {
        const float scale = make_qx_quants(16, 32, x + 16 * ib, L + 16 * ib, 1);
        scales[ib] = scale;
        const float a = fabsf(scale);
        if (a > max_abs_scale) { max_abs_scale = a; max_scale = scale; }
    }
    int8_t *sc = reinterpret_cast<int8_t *>(y + 192);
    const float iscale = -128.f / max_scale;
    put16(y + 208, f2h(1 / iscale));
    for (int ib = 0; ib < 16; ib++) sc[ib] = (int8_t)std::min(127, nearest_int(iscale * scales[ib]));
    const float dd = h2f((uint16_t)(y[208] | (y[209] << 8)));
    for (int j = 0; j < 16; j++) {
        const float d = dd * sc[j]; if (!d) continue;
        for (int ii = 0; ii < 16; ii++) { int l = nearest_int(x[16 * j + ii] / d); l = std::max(-32, std::min(31, l)); L[16 * j + ii] = (int8_t)(l + 32); }
    }
    uint8_t *ql = y, *qh = y + 128;
    for (int j = 0; j < 256; j += 128) {
        for (int l = 0; l < 32; l++) {
            const uint8_t q1 = L[j + l] & 0xF, q2 = L[j + l + 32] & 0xF, q3 = L[j + l + 64] & 0xF, q4 = L[j + l + 96] & 0xF;
            ql[l] = (uint8_t)(q1 | (q3 << 4)); ql[l + 32] = (uint8_t)(q2 | (q4 << 4));
            qh[l] = (uint8_t)((L[j + l] >> 4) | ((L[j + l + 32] >> 4) << 2) | ((L[j + l + 64] >> 4) << 4) | ((L[j + l + 96] >> 4) << 6));
        }
        ql += 64; qh += 32;
    }
}
}  // namespace

bool quantize_supported(int ggml_type) {
    switch (ggml_type) { case GT_Q4_0: case GT_Q4_1: case GT_Q5_0: case GT_Q5_1: case GT_Q8_0: case GT_Q2_K: case GT_Q3_K: case GT_Q4_K: case GT_Q5_K: case GT_Q6_K: return true; default: return false; }
}
size_t quantize_chunk(int ggml_type, const float *x, uint8_t *dst, size_t n) {
    const size_t blk = (size_t)gt_block(ggml_type), bytes = (size_t)gt_bytes(ggml_type);
    if (!quantize_supported(ggml_type) || n % blk) return 0;
    void (*fn)(const float *, uint8_t *) = nullptr;
    switch (ggml_type) {
    case GT_Q4_0: fn = q4_0_block; break; case GT_Q4_1: fn = q4_1_block; break; case GT_Q5_0: fn = q5_0_block; break; case GT_Q5_1: fn = q5_1_block; break;
    case GT_Q8_0: fn = q8_0_block; break; case GT_Q2_K: fn = q2_K_block; break; case GT_Q3_K: fn = q3_K_block; break; case GT_Q4_K: fn = q4_K_block; break; case GT_Q5_K: fn = q5_K_block; break; default: fn = q6_K_block; break;
    }
    const size_t nb = n / blk;
    // blocks are independent: plain std::thread fan-out (an offline tool; the 13B vision file has ~4 M super-blocks)
    const size_t nthr = nb < 4096 ? 1 : std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), 64));
    auto work = [&](size_t b0, size_t b1) { for (size_t b = b0; b < b1; b++) fn(x + b * blk, dst + b * bytes); };
    if (nthr == 1) work(0, nb);
    else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nthr; t++) th.emplace_back(work, nb * t / nthr, nb * (t + 1) / nthr);
        for (auto &t : th) t.join();
    }
    return nb * bytes;
}

static int ggml_to_mg4(int t) {   // MiniGPT4DataType numbering (reference minigpt4.h:30-48)
    switch (t) { case GT_F16: return 0; case GT_F32: return 1; case GT_I32: return 2; case GT_I64: return 3; case GT_Q4_0: return 4; case GT_Q4_1: return 5; case GT_Q5_0: return 6; case GT_Q5_1: return 7;
        case GT_Q8_0: return 8; case GT_Q8_1: return 9; case GT_Q2_K: return 10; case GT_Q3_K: return 11; case GT_Q4_K: return 12; case GT_Q5_K: return 13; case GT_Q6_K: return 14; case GT_Q8_K: return 15; default: return -1; }
}
static int mg4_to_ggml_q(int t) {
    switch (t) { case 0: return GT_F16; case 1: return GT_F32; case 4: return GT_Q4_0; case 5: return GT_Q4_1; case 6: return GT_Q5_0; case 7: return GT_Q5_1; case 8: return GT_Q8_0; case 9: return GT_Q8_1;
        case 10: return GT_Q2_K; case 11: return GT_Q3_K; case 12: return GT_Q4_K; case 13: return GT_Q5_K; case 14: return GT_Q6_K; case 15: return GT_Q8_K; default: return -1; }
}

// which tensors the reference re-quantises (minigpt4.cpp:2893-2920)
static bool eligible(const std::string &model, const TensorMeta &t) {
    auto ends_with = [](const std::string &s, const char *suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; };
    return (t.type == GT_F16 || t.type == GT_F32) && ends_with(t.name, "weight") && t.ne.size() >= 2 && t.name.find("norm") == std::string::npos &&
           t.name.find("Norm") == std::string::npos && model != "ln_vision" && model != "query_tokens" && model != "llama_proj" && t.name != "patch_embed.proj.weight";
}

int quantize_vision_file(const char *in_path, const char *out_path, int mg4_data_type) {
    struct stat st;
    if (!in_path || stat(in_path, &st) != 0) return E_PathDoesNotExist;
    VisionFile vf;
    if (int e = vf.load(in_path)) return e;
    const int out_type = mg4_to_ggml_q(mg4_data_type);
    if (!quantize_supported(out_type)) { set_last_error("minigpt4_quantize_model: target type must be one of Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q2_K Q3_K Q4_K Q5_K Q6_K"); return E_LoadModelMiniGPT4DataType; }
    if (!out_path) return E_DumpModelFileOpen;
    {   // the input stays mmap'd while the output is written: truncating it through a second name (same path or a hard link) would SIGBUS the next tensor read
        struct stat so;
        if (stat(out_path, &so) == 0 && so.st_dev == st.st_dev && so.st_ino == st.st_ino) { set_last_error("quantize: the output path is the input file"); return E_DumpModelFileOpen; }
    }
    FILE *f = fopen(out_path, "wb");
    if (!f) { set_last_error(std::string("cannot open ") + out_path); return E_DumpModelFileOpen; }
    auto w32 = [&](int32_t v) { fwrite(&v, 4, 1, f); };
    auto wstr = [&](const std::string &s) { w32((int32_t)s.size()); fwrite(s.data(), 1, s.size(), f); };
    fwrite("ggml", 1, 4, f);
    w32(vf.version);
    w32(ggml_to_mg4(out_type));                                     // set_file_data_type(out_type), :2970
    wstr(vf.config_json);                                           // the reference re-serialises the parsed JSON (:1666); the text is kept as read here
    size_t orig_total = 0, new_total = 0;
    std::vector<float> f32; std::vector<uint8_t> q;
    for (const std::string &mname : vf.model_order) {
        const auto &model = vf.models.at(mname);
        // file order of the tensors = their data offsets
        std::vector<const TensorMeta *> order;
        for (auto &kv : model) order.push_back(&kv.second);
        std::sort(order.begin(), order.end(), [](const TensorMeta *a, const TensorMeta *b) { return a->offset < b->offset; });
        std::vector<int> new_type(order.size());
        for (size_t i = 0; i < order.size(); i++) {
            const TensorMeta &t = *order[i];
            // ggml_quantize_chunk works on the flat array; a row length that is not a whole number of blocks would give a file no ggml build can load
            // (ggml_new_tensor asserts ne[0] % blck_size == 0), so such a tensor keeps its type here
            new_type[i] = eligible(mname, t) && !t.ne.empty() && t.ne[0] % gt_block(out_type) == 0 ? out_type : t.type;
        }
        wstr(mname);
        w32((int32_t)order.size());
        for (size_t i = 0; i < order.size(); i++) {
            const TensorMeta &t = *order[i];
            wstr(t.name);
            w32((int32_t)t.ne.size());
            for (int64_t d : t.ne) w32((int32_t)d);
            w32(ggml_to_mg4(new_type[i]));
        }
        for (size_t i = 0; i < order.size(); i++) {
            const TensorMeta &t = *order[i];
            long pos = ftell(f);
            if (pos & 4095) { pos = (pos + 4096) & ~4095L; fseek(f, pos, SEEK_SET); }   // align_to_next_page, :1693-1707
            const uint8_t *src = vf.mf.data + t.offset;
            orig_total += t.nbytes;
            if (new_type[i] == t.type) { fwrite(src, 1, t.nbytes, f); new_total += t.nbytes; continue; }
            const size_t n = (size_t)t.nelements();
            f32.resize(n);
            if (t.type == GT_F16) { const uint16_t *h = reinterpret_cast<const uint16_t *>(src); for (size_t k = 0; k < n; k++) f32[k] = h2f(h[k]); }   // ggml_fp16_to_fp32_row, :2925
            else memcpy(f32.data(), src, n * 4);
            q.resize(gt_nbytes(out_type, n));
            const size_t wrote = quantize_chunk(out_type, f32.data(), q.data(), n);
            fwrite(q.data(), 1, wrote, f);
            new_total += wrote;
            MG4_INFO("%s.%s | Original %10.2f MB -> New %10.2f MB", mname.c_str(), t.name.c_str(), t.nbytes / 1048576.0, wrote / 1048576.0);
        }
    }
    const bool ok = !ferror(f);
    fclose(f);
    if (!ok) { set_last_error("write error"); return E_DumpModelFileOpen; }
    MG4_INFO("Original size %10.2f MB", orig_total / 1048576.0);
    MG4_INFO("Quantized size %10.2f MB", new_total / 1048576.0);
    return E_None;
}


// ---- Q3_K -> Q6_K (load time; see quantize.hpp).  block_q3_K = hmask[32] | qs[64] | scales[12] (sixteen 6-bit values) | d; element 128 n + 32 j + l of a super-block is
// ((qs[32 n + l] >> 2 j) & 3) - (hmask[l] bit (4 n + j) ? 0 : 4) with scale index 8 n + 2 j + l / 16 (k_quants.c dequantize_row_q3_K); block_q6_K = ql[128] | qh[64] |
// scales[16] int8 | d with element 128 n + 32 a + l in ql[64 n + 32 (a & 1) + l] (nibble a >> 1) and bits 2 a of qh[32 n + l], the same scale index.
static void q3k_to_q6k_range(const uint8_t *src, uint8_t *dst, size_t b0, size_t b1) {
    for (size_t b = b0; b < b1; b++) {
        const uint8_t *x = src + b * 110, *hm = x, *qs = x + 32, *sb = x + 96;
        uint8_t *y = dst + b * 210, *ql = y, *qh = y + 128;
        memset(y, 0, 192);
        for (int n = 0; n < 2; n++) for (int a = 0; a < 4; a++) for (int l = 0; l < 32; l++) {
            const int q3 = ((qs[32 * n + l] >> (2 * a)) & 3) - (((hm[l] >> (4 * n + a)) & 1) ? 0 : 4);
            const unsigned v = (unsigned)(q3 + 32);                                            // 6-bit field of Q6_K: value + 32
            ql[64 * n + 32 * (a & 1) + l] |= (uint8_t)((v & 15) << ((a >> 1) * 4));
            qh[32 * n + l] |= (uint8_t)((v >> 4) << (2 * a));
        }
        int8_t *sc = reinterpret_cast<int8_t *>(y + 192);
        for (int j = 0; j < 4; j++) {
            const int hi = sb[8 + j];
            sc[j] = (int8_t)(((sb[j] & 15) | (((hi >> 0) & 3) << 4)) - 32); sc[4 + j] = (int8_t)(((sb[4 + j] & 15) | (((hi >> 2) & 3) << 4)) - 32);
            sc[8 + j] = (int8_t)(((sb[j] >> 4) | (((hi >> 4) & 3) << 4)) - 32); sc[12 + j] = (int8_t)(((sb[4 + j] >> 4) | (((hi >> 6) & 3) << 4)) - 32);
        }
        y[208] = x[108]; y[209] = x[109];
    }
}
void q3k_to_q6k(const uint8_t *src, uint8_t *dst, size_t n_blocks) {
    const size_t nthr = n_blocks < 4096 ? 1 : std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), 32));
    if (nthr == 1) { q3k_to_q6k_range(src, dst, 0, n_blocks); return; }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nthr; t++) th.emplace_back(q3k_to_q6k_range, src, dst, n_blocks * t / nthr, n_blocks * (t + 1) / nthr);
    for (auto &t : th) t.join();
}

}  // namespace mg4
