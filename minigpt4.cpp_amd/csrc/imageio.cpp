// Image file decoding on the host + Pillow's bicubic coefficient tables.  See imageio.hpp.
#include "imageio.hpp"

#include <sys/mman.h>
#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <thread>

#include "common.hpp"

namespace mg4 {

void *bigbuf_alloc(size_t bytes) {
    constexpr size_t HP = (size_t)2 << 20;
    void *p = nullptr;
    if (bytes >= 2 * HP) {
        const size_t r = (bytes + HP - 1) & ~(HP - 1);
        p = aligned_alloc(HP, r);
        if (p) (void)madvise(p, r, MADV_HUGEPAGE);               // advisory: without THP the pages are ordinary ones
    } else p = malloc(bytes);
    if (!p) throw std::bad_alloc();
    return p;
}
namespace {
constexpr uint64_t MAX_PIXELS = 1ull << 27;   // 134 Mpixel: larger headers are treated as corrupt

// =====================================================================================================================
// inflate (RFC 1951) behind a zlib wrapper (RFC 1950)
// =====================================================================================================================
struct BitsLSB {
    const uint8_t *p; size_t n, pos = 0; uint64_t buf = 0; int cnt = 0; bool overrun = false;
    BitsLSB(const uint8_t *d, size_t len) : p(d), n(len) {}
    inline void fill(int need) {
        while (cnt < need) { uint64_t b = 0; if (pos < n) b = p[pos]; else overrun = pos >= n + 8 ? true : overrun; pos++; buf |= b << cnt; cnt += 8; }
    }
    inline unsigned peek(int k) { fill(k); return (unsigned)(buf & ((1ull << k) - 1)); }
    inline void drop(int k) { buf >>= k; cnt -= k; }
    inline unsigned get(int k) { if (!k) return 0; const unsigned v = peek(k); drop(k); return v; }
    inline void align_byte() { drop(cnt & 7); }
    bool past_end() const { return pos > n + 8; }
};

struct HuffLSB {   // canonical Huffman code, LSB-first bit stream: 10-bit direct table + bit-serial fallback
    static constexpr int FAST = 10;
    uint16_t fast[1 << FAST];      // (symbol << 4) | length, 0 = not in the table
    uint16_t count[16], symbol[320];
    bool build(const uint8_t *lens, int n) {
        memset(count, 0, sizeof(count)); memset(fast, 0, sizeof(fast));
        for (int i = 0; i < n; i++) count[lens[i]]++;
        if (count[0] == n) return true;          // no codes: legal for an unused distance tree
        int left = 1;
        for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; i++) if (lens[i]) symbol[offs[lens[i]]++] = (uint16_t)i;
        // direct table
        unsigned code = 0; int idx = 0;
        for (int l = 1; l <= FAST; l++) {
            for (int k = 0; k < count[l]; k++, code++, idx++) {
                unsigned rev = 0; for (int b = 0; b < l; b++) rev |= ((code >> b) & 1u) << (l - 1 - b);
                for (unsigned f = rev; f < (1u << FAST); f += 1u << l) fast[f] = (uint16_t)((symbol[idx] << 4) | l);
            }
            code <<= 1;
        }
        return true;
    }
    inline int decode(BitsLSB &br) const {
        const unsigned e = fast[br.peek(FAST)];
        if (e) { br.drop(e & 15); return e >> 4; }
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)br.get(1);
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

bool inflate_raw(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t expect, std::string &err) {
    static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    BitsLSB br(src, n);
    out.clear(); out.reserve(expect);
    HuffLSB lit, dist;
    for (;;) {
        const unsigned last = br.get(1), type = br.get(2);
        if (type == 0) {
            br.align_byte();
            const unsigned len = br.get(16), nlen = br.get(16);
            if ((len ^ 0xFFFF) != nlen) { err = "inflate: stored block length mismatch"; return false; }
            for (unsigned i = 0; i < len && out.size() < expect; i++) out.push_back((uint8_t)br.get(8));
            if (out.size() >= expect) return true;
        } else if (type == 1 || type == 2) {
            uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 144; i++) lens[i] = 8; for (int i = 144; i < 256; i++) lens[i] = 9; for (int i = 256; i < 280; i++) lens[i] = 7; for (int i = 280; i < 288; i++) lens[i] = 8;
                lit.build(lens, 288);
                for (int i = 0; i < 30; i++) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                const int hlit = (int)br.get(5) + 257, hdist = (int)br.get(5) + 1, hclen = (int)br.get(4) + 4;
                if (hlit > 286 || hdist > 30) { err = "inflate: bad code counts"; return false; }
                uint8_t cl[19] = {0};
                for (int i = 0; i < hclen; i++) cl[ORDER[i]] = (uint8_t)br.get(3);
                HuffLSB clh;
                if (!clh.build(cl, 19)) { err = "inflate: bad code-length code"; return false; }
                int i = 0;
                while (i < hlit + hdist) {
                    const int sym = clh.decode(br);
                    if (sym < 0) { err = "inflate: bad code-length symbol"; return false; }
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int rep; uint8_t v = 0;
                        if (sym == 16) { if (!i) { err = "inflate: repeat without a previous length"; return false; } v = lens[i - 1]; rep = 3 + (int)br.get(2); }
                        else if (sym == 17) rep = 3 + (int)br.get(3);
                        else rep = 11 + (int)br.get(7);
                        if (i + rep > hlit + hdist) { err = "inflate: length repeat overflows"; return false; }
                        while (rep--) lens[i++] = v;
                    }
                }
                if (!lens[256]) { err = "inflate: no end-of-block code"; return false; }
                if (!lit.build(lens, hlit) || !dist.build(lens + hlit, hdist)) { err = "inflate: over-subscribed code"; return false; }
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0) { err = "inflate: bad literal/length symbol"; return false; }
                if (sym < 256) { out.push_back((uint8_t)sym); if (out.size() >= expect) return true; if (br.past_end()) { err = "inflate: truncated stream"; return false; } continue; }
                if (sym == 256) break;
                sym -= 257;
                if (sym >= 29) { err = "inflate: bad length symbol"; return false; }
                const int len = LBASE[sym] + (int)br.get(LEXT[sym]);
                const int ds = dist.decode(br);
                if (ds < 0 || ds >= 30) { err = "inflate: bad distance symbol"; return false; }
                const size_t d = (size_t)DBASE[ds] + br.get(DEXT[ds]);
                if (d > out.size()) { err = "inflate: distance beyond the start of the output"; return false; }
                const size_t from = out.size() - d;
                for (int k = 0; k < len; k++) out.push_back(out[from + (size_t)k]);
                if (out.size() >= expect) return true;       // the image is complete; anything further is ignored (and a corrupt stream cannot grow without bound)
                if (br.past_end()) { err = "inflate: truncated stream"; return false; }
            }
        } else { err = "inflate: reserved block type"; return false; }
        if (br.past_end()) { err = "inflate: truncated stream"; return false; }
        if (last) break;
    }
    return true;
}

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

// =====================================================================================================================
// PNG -> RGB8 with cv::imread(IMREAD_COLOR) semantics: alpha dropped (not composited), 16 bit -> high byte, palette expanded,
// 1/2/4-bit grey scaled to 0..255, grey replicated to three channels.
// =====================================================================================================================
inline int paeth(int a, int b, int c) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

bool png_unfilter(uint8_t *d, size_t rows, size_t rowbytes, int bpp, std::string &err) {   // d: rows x (1 + rowbytes), in place
    std::vector<uint8_t> zero(rowbytes, 0);
    const uint8_t *prev = zero.data();
    for (size_t y = 0; y < rows; y++) {
        uint8_t *row = d + y * (rowbytes + 1) + 1;
        const int ft = row[-1];
        switch (ft) {
        case 0: break;
        case 1: for (size_t i = (size_t)bpp; i < rowbytes; i++) row[i] = (uint8_t)(row[i] + row[i - (size_t)bpp]); break;
        case 2: for (size_t i = 0; i < rowbytes; i++) row[i] = (uint8_t)(row[i] + prev[i]); break;
        case 3: for (size_t i = 0; i < rowbytes; i++) { const int a = i >= (size_t)bpp ? row[i - (size_t)bpp] : 0; row[i] = (uint8_t)(row[i] + ((a + prev[i]) >> 1)); } break;
        case 4: for (size_t i = 0; i < rowbytes; i++) { const int a = i >= (size_t)bpp ? row[i - (size_t)bpp] : 0, c = i >= (size_t)bpp ? prev[i - (size_t)bpp] : 0; row[i] = (uint8_t)(row[i] + paeth(a, prev[i], c)); } break;
        default: err = "png: bad filter type"; return false;
        }
        prev = row;
    }
    return true;
}
}  // namespace

bool decode_png(const uint8_t *data, size_t n, ImageRGB8 &out, std::string &err) {
    static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || memcmp(data, SIG, 8)) { err = "png: bad signature"; return false; }
    size_t pos = 8;
    uint32_t W = 0, H = 0; int bd = 0, ct = -1, interlace = 0; bool have_hdr = false, end = false;
    std::vector<uint8_t> z; uint8_t pal[256 * 3]; int npal = 0;
    memset(pal, 0, sizeof(pal));
    while (!end && pos + 12 <= n) {
        const uint32_t len = be32(data + pos); const uint8_t *ty = data + pos + 4, *cd = data + pos + 8;
        if ((size_t)len > n - pos - 12) { err = "png: truncated chunk"; return false; }
        if (!memcmp(ty, "IHDR", 4)) {
            if (len != 13) { err = "png: bad IHDR"; return false; }
            W = be32(cd); H = be32(cd + 4); bd = cd[8]; ct = cd[9]; interlace = cd[12];
            if (cd[10] || cd[11] || interlace > 1) { err = "png: unsupported compression / filter / interlace method"; return false; }
            have_hdr = true;
        } else if (!memcmp(ty, "PLTE", 4)) { npal = (int)std::min<uint32_t>(len / 3, 256); memcpy(pal, cd, (size_t)npal * 3); }
        else if (!memcmp(ty, "IDAT", 4)) z.insert(z.end(), cd, cd + len);
        else if (!memcmp(ty, "IEND", 4)) end = true;
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || !W || !H || W > (1u << 24) || H > (1u << 24) || (uint64_t)W * H > MAX_PIXELS) { err = "png: missing IHDR or absurd size"; return false; }
    int ch;
    switch (ct) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: err = "png: bad colour type"; return false; }
    const bool bd_ok = (ct == 0 && (bd == 1 || bd == 2 || bd == 4 || bd == 8 || bd == 16)) || (ct == 3 && (bd == 1 || bd == 2 || bd == 4 || bd == 8)) || ((ct == 2 || ct == 4 || ct == 6) && (bd == 8 || bd == 16));
    if (!bd_ok) { err = "png: bad bit depth"; return false; }
    if (z.size() < 2 || (z[0] & 15) != 8 || ((z[0] << 8) | z[1]) % 31 || (z[1] & 0x20)) { err = "png: bad zlib header"; return false; }
    const int bits_pp = ch * bd, bpp = std::max(1, bits_pp / 8);
    auto rowbytes = [&](size_t w) { return (w * (size_t)bits_pp + 7) / 8; };
    static const int XS[7] = {0, 4, 0, 2, 0, 1, 0}, YS[7] = {0, 0, 4, 0, 2, 0, 1}, XD[7] = {8, 8, 4, 4, 2, 2, 1}, YD[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t expect = 0;
    if (!interlace) expect = (size_t)H * (1 + rowbytes(W));
    else for (int p = 0; p < 7; p++) { const size_t pw = (W + XD[p] - 1 - XS[p]) / XD[p], ph = (H + YD[p] - 1 - YS[p]) / YD[p]; if (pw && ph) expect += ph * (1 + rowbytes(pw)); }
    std::vector<uint8_t> raw;
    if (!inflate_raw(z.data() + 2, z.size() - 2, raw, expect, err)) return false;
    if (raw.size() < expect) { err = "png: image data too short"; return false; }
    out.w = (int)W; out.h = (int)H; out.px.assign((size_t)W * H * 3, 0);
    // one decoded (unfiltered) row of `pw` pixels -> RGB pixels at (x0 + i * dx, y)
    auto emit_row = [&](const uint8_t *row, size_t pw, size_t y, size_t x0, size_t dx) {
        uint8_t *dst = out.px.data() + y * W * 3;
        for (size_t i = 0; i < pw; i++) {
            uint8_t r, g, b;
            if (ct == 0 || ct == 3) {
                unsigned v;
                if (bd == 8) v = row[i]; else if (bd == 16) v = row[2 * i];
                else { const unsigned per = 8u / (unsigned)bd, byte = row[i / per], sh = (per - 1 - (unsigned)(i % per)) * (unsigned)bd; v = (byte >> sh) & ((1u << bd) - 1); }
                if (ct == 3) { if ((int)v >= npal) v = 0; r = pal[v * 3]; g = pal[v * 3 + 1]; b = pal[v * 3 + 2]; }
                else { if (bd < 8) v = v * (255u / ((1u << bd) - 1)); r = g = b = (uint8_t)v; }
            } else if (ct == 4) { r = g = b = row[i * (size_t)(bd == 16 ? 4 : 2)]; }
            else { const size_t st = (size_t)(bd == 16 ? 2 : 1), px = i * (size_t)ch * st; r = row[px]; g = row[px + st]; b = row[px + 2 * st]; }
            uint8_t *q = dst + (x0 + i * dx) * 3; q[0] = r; q[1] = g; q[2] = b;
        }
    };
    if (!interlace) {
        const size_t rb = rowbytes(W);
        if (!png_unfilter(raw.data(), H, rb, bpp, err)) return false;
        for (size_t y = 0; y < H; y++) emit_row(raw.data() + y * (rb + 1) + 1, W, y, 0, 1);
    } else {
        size_t off = 0;
        for (int p = 0; p < 7; p++) {
            const size_t pw = (W + XD[p] - 1 - XS[p]) / XD[p], ph = (H + YD[p] - 1 - YS[p]) / YD[p];
            if (!pw || !ph) continue;
            const size_t rb = rowbytes(pw);
            if (!png_unfilter(raw.data() + off, ph, rb, bpp, err)) return false;
            for (size_t y = 0; y < ph; y++) emit_row(raw.data() + off + y * (rb + 1) + 1, pw, (size_t)YS[p] + y * (size_t)YD[p], (size_t)XS[p], (size_t)XD[p]);
            off += ph * (rb + 1);
        }
    }
    return true;
}

// =====================================================================================================================
// JPEG (ITU T.81 Huffman modes) with libjpeg(-turbo)'s default decompression choices, so the pixels equal cv::imread's:
// JDCT_ISLOW inverse DCT, "fancy" (triangle) chroma upsampling for 2:1 ratios, box replication for other integral ratios, 16-bit
// fixed-point YCbCr -> RGB.
// =====================================================================================================================
namespace {
const uint8_t ZZ[64 + 16] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                             29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct JHuff {
    bool present = false;
    uint8_t bits[17] = {0}, vals[256] = {0};
    int mincode[18], maxcode[18], valptr[18];
    uint16_t fast[512];   // 9-bit lookahead: (len << 8) | symbol, 0 = miss
    // AC tables: when the code AND the magnitude bits that follow it fit the 9-bit window, one lookup yields the finished coefficient:
    // (value << 8) | (run << 4) | (code length + magnitude bits); 0 = take the two-step path.  Most coefficients of a photograph are small: this is the common case
    int16_t fast_ac[512];
    bool build() {
        int code = 0, k = 0;
        memset(fast, 0, sizeof(fast));
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k; mincode[l] = code;
            for (int i = 0; i < bits[l]; i++, k++, code++) {
                if (k >= 256 || code >= (1 << l)) return false;      // over-subscribed code
                if (l <= 9) { const int base = code << (9 - l); for (int f = 0; f < (1 << (9 - l)); f++) fast[base + f] = (uint16_t)((l << 8) | vals[k]); }
            }
            maxcode[l] = bits[l] ? code - 1 : -1;
            if (code > (1 << l)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        for (int i = 0; i < 512; i++) {
            fast_ac[i] = 0;
            const unsigned e = fast[i];
            if (!e) continue;
            const int len = (int)(e >> 8), rs = (int)(e & 255), run = rs >> 4, mag = rs & 15;
            if (!mag || len + mag > 9) continue;
            int v = ((i << len) & 511) >> (9 - mag);                 // the magnitude bits behind the code
            if (v < (1 << (mag - 1))) v = v - (1 << mag) + 1;         // receive_extend
            if (v >= -128 && v <= 127) fast_ac[i] = (int16_t)(v * 256 + run * 16 + len + mag);
        }
        return true;
    }
};

struct JBits {   // MSB-first entropy-coded segment reader: 0xFF00 unstuffing, stops at markers (feeds zeros, as libjpeg does)
    const uint8_t *p; size_t n, pos; uint64_t buf = 0; int cnt = 0; int marker = 0; int fed_zero = 0;
    JBits(const uint8_t *d, size_t len, size_t at) : p(d), n(len), pos(at) {}
    inline void fill() {
        while (cnt <= 48) {
            unsigned b = 0;
            if (!marker && pos < n) {
                b = p[pos];
                if (b == 0xFF) {
                    size_t q = pos + 1;
                    while (q < n && p[q] == 0xFF) q++;          // fill bytes
                    if (q < n && p[q] == 0) { pos = q + 1; }     // stuffed zero
                    else { marker = q < n ? p[q] : 0xD9; pos = q < n ? q + 1 : n; b = 0; fed_zero++; }
                } else pos++;
            } else fed_zero++;
            buf |= (uint64_t)b << (56 - cnt); cnt += 8;
        }
    }
    inline unsigned peek(int k) { if (cnt < k) fill(); return (unsigned)(buf >> (64 - k)); }
    inline void drop(int k) { buf <<= k; cnt -= k; }
    inline unsigned get(int k) { if (!k) return 0; if (cnt < k) fill(); const unsigned v = (unsigned)(buf >> (64 - k)); buf <<= k; cnt -= k; return v; }
    inline int receive_extend(int s) { if (!s) return 0; const int v = (int)get(s); return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
    inline int decode(const JHuff &h) {
        if (cnt < 16) fill();
        const unsigned look = (unsigned)(buf >> 55);
        const unsigned e = h.fast[look];
        if (e) { drop((int)(e >> 8)); return (int)(e & 255); }
        int l = 10; int code = (int)(buf >> 54);
        while (l <= 16 && code > h.maxcode[l]) { l++; code = (int)(buf >> (64 - l)); }
        if (l > 16) { drop(16); return 0; }     // garbage input: libjpeg warns and returns 0
        drop(l);
        return h.vals[(h.valptr[l] + code - h.mincode[l]) & 255];
    }
    // restart interval boundary: discard the padding bits, find and consume the RSTn marker (it may not have been prefetched yet)
    void restart() {
        buf = 0; cnt = 0; fed_zero = 0;
        if (!marker) {
            while (pos + 1 < n && !(p[pos] == 0xFF && p[pos + 1] != 0 && p[pos + 1] != 0xFF)) pos++;
            if (pos + 1 < n) { marker = p[pos + 1]; pos += 2; }
        }
        if (marker >= 0xD0 && marker <= 0xD7) marker = 0;
    }
};

struct JComp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int wb = 0, hb = 0;          // blocks covering the component's own size (non-interleaved scan geometry)
    int wbp = 0, hbp = 0;        // blocks allocated (padded to whole MCUs)
    int dw = 0, dh = 0;          // downsampled_width / downsampled_height in samples
    int pred = 0;
    BigBuf<int16_t> coef;        // [hbp][wbp][64], natural order
    PixelBuf plane;              // [hbp * 8][wbp * 8]
};

inline uint8_t idct_limit(int64_t v) { int x = (int)(((v + 512) & 1023) - 512) + 128; return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }   // libjpeg's range_limit[(v) & RANGE_MASK]

// The stages after the entropy decoder -- inverse DCT, chroma upsampling, colour conversion -- are independent per block row / pixel row: a 12-megapixel photograph spends
// a quarter of its decode time there (the rest is the sequential Huffman stage), so they run on up to 8 host threads (same arithmetic per element: the pixels do not depend on the thread count; small images stay serial).
template <typename F> void parallel_rows(int n, size_t work_per_row, F &&fn) {
    unsigned nt = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if ((size_t)n * work_per_row < ((size_t)1 << 20) || n < 2 * (int)nt) nt = 1;
    if (nt <= 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int per = (n + (int)nt - 1) / (int)nt;
    for (unsigned t = 0; t < nt; t++) { const int a = (int)t * per, b = std::min(n, a + per); if (a < b) th.emplace_back([&fn, a, b] { fn(a, b); }); }
    for (auto &x : th) x.join();
}
// jidctint.c jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2)
void idct_islow(const int16_t *in, const uint16_t *q, uint8_t *out, size_t stride) {
    constexpr int64_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819,
                      F_2_562 = 20995, F_3_072 = 25172;
    int ws[64];
    auto descale = [](int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; };
    for (int c = 0; c < 8; c++) {
        auto D = [&](int r) { return (int64_t)in[r * 8 + c] * (int64_t)q[r * 8 + c]; };
        int64_t z2 = D(2), z3 = D(6);
        int64_t z1 = (z2 + z3) * F_0_541;
        int64_t tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        z2 = D(0); z3 = D(4);
        int64_t tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = D(7); tmp1 = D(5); tmp2 = D(3); tmp3 = D(1);
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int64_t z4 = tmp1 + tmp3; const int64_t z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[0 * 8 + c] = (int)descale(tmp10 + tmp3, 11); ws[7 * 8 + c] = (int)descale(tmp10 - tmp3, 11);
        ws[1 * 8 + c] = (int)descale(tmp11 + tmp2, 11); ws[6 * 8 + c] = (int)descale(tmp11 - tmp2, 11);
        ws[2 * 8 + c] = (int)descale(tmp12 + tmp1, 11); ws[5 * 8 + c] = (int)descale(tmp12 - tmp1, 11);
        ws[3 * 8 + c] = (int)descale(tmp13 + tmp0, 11); ws[4 * 8 + c] = (int)descale(tmp13 - tmp0, 11);
    }
    for (int r = 0; r < 8; r++) {
        const int *w = ws + r * 8; uint8_t *o = out + (size_t)r * stride;
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * F_0_541;
        int64_t tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        int64_t tmp0 = ((int64_t)w[0] + (int64_t)w[4]) * 8192, tmp1 = ((int64_t)w[0] - (int64_t)w[4]) * 8192;
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int64_t z4 = tmp1 + tmp3; const int64_t z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        o[0] = idct_limit(descale(tmp10 + tmp3, 18)); o[7] = idct_limit(descale(tmp10 - tmp3, 18));
        o[1] = idct_limit(descale(tmp11 + tmp2, 18)); o[6] = idct_limit(descale(tmp11 - tmp2, 18));
        o[2] = idct_limit(descale(tmp12 + tmp1, 18)); o[5] = idct_limit(descale(tmp12 - tmp1, 18));
        o[3] = idct_limit(descale(tmp13 + tmp0, 18)); o[4] = idct_limit(descale(tmp13 - tmp0, 18));
    }
}

// EXIF orientation (APP1): 1..8, 1 when absent / malformed
int exif_orientation(const uint8_t *p, size_t n) {
    if (n < 14 || memcmp(p, "Exif\0\0", 6)) return 1;
    const uint8_t *t = p + 6; const size_t tn = n - 6;
    const bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return 1;
    auto r16 = [&](size_t o) -> unsigned { return le ? (unsigned)(t[o] | (t[o + 1] << 8)) : (unsigned)((t[o] << 8) | t[o + 1]); };
    auto r32 = [&](size_t o) -> uint32_t { return le ? ((uint32_t)t[o] | ((uint32_t)t[o + 1] << 8) | ((uint32_t)t[o + 2] << 16) | ((uint32_t)t[o + 3] << 24)) : be32(t + o); };
    if (r16(2) != 42) return 1;
    const size_t ifd = r32(4);
    if (ifd + 2 > tn) return 1;
    const unsigned cnt = r16(ifd);
    for (unsigned i = 0; i < cnt; i++) {
        const size_t e = ifd + 2 + (size_t)i * 12;
        if (e + 12 > tn) return 1;
        if (r16(e) == 0x0112) { const unsigned v = r16(e + 8); return v >= 1 && v <= 8 ? (int)v : 1; }
    }
    return 1;
}

void apply_orientation(ImageRGB8 &im, int o) {   // OpenCV ExifTransform (modules/imgcodecs/src/exif.cpp): 2 flip-h, 3 rot180, 4 flip-v, 5 transpose, 6 rot90cw, 7 transverse, 8 rot270cw
    if (o <= 1 || o > 8) return;
    const int w = im.w, h = im.h;
    const bool swap = o >= 5;
    const int ow = swap ? h : w, oh = swap ? w : h;
    PixelBuf dst; dst.resize((size_t)ow * oh * 3);
    for (int y = 0; y < oh; y++) for (int x = 0; x < ow; x++) {
        int sx, sy;
        switch (o) {
        case 2: sx = w - 1 - x; sy = y; break;
        case 3: sx = w - 1 - x; sy = h - 1 - y; break;
        case 4: sx = x; sy = h - 1 - y; break;
        case 5: sx = y; sy = x; break;
        case 6: sx = y; sy = h - 1 - x; break;
        case 7: sx = w - 1 - y; sy = h - 1 - x; break;
        default: sx = w - 1 - y; sy = x; break;   // 8
        }
        memcpy(&dst[((size_t)y * ow + x) * 3], &im.px[((size_t)sy * w + sx) * 3], 3);
    }
    im.w = ow; im.h = oh; im.px.swap(dst);
}
}  // namespace

bool decode_jpeg(const uint8_t *data, size_t n, ImageRGB8 &out, std::string &err) {
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) { err = "jpeg: missing SOI"; return false; }
    uint16_t qt[4][64]; bool qt_ok[4] = {false, false, false, false};
    JHuff hdc[4], hac[4];
    std::vector<JComp> comps;
    int W = 0, H = 0, hmax = 1, vmax = 1, restart_interval = 0, orientation = 1;
    bool progressive = false, have_sof = false, jfif = false, adobe = false; int adobe_transform = -1;
    int mcux = 0, mcuy = 0;
    bool any_scan = false;
    size_t pos = 2;
    auto fail = [&](const char *m) { err = m; return false; };
    for (;;) {
        // next marker
        while (pos < n && data[pos] != 0xFF) pos++;
        while (pos < n && data[pos] == 0xFF) pos++;
        if (pos >= n) break;                         // premature end: libjpeg inserts EOI
        const int m = data[pos++];
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0) continue;
        if (pos + 2 > n) break;
        const size_t len = be16(data + pos);
        if (len < 2 || pos + len > n) return fail("jpeg: truncated marker segment");
        const uint8_t *s = data + pos + 2; const size_t sl = len - 2;
        if (m == 0xDB) {   // DQT
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15; o++;
                if (tq > 3 || o + (pq ? 128u : 64u) > sl) return fail("jpeg: bad DQT");
                for (int i = 0; i < 64; i++) { qt[tq][ZZ[i]] = pq ? be16(s + o + 2 * (size_t)i) : s[o + (size_t)i]; }
                o += pq ? 128 : 64; qt_ok[tq] = true;
            }
        } else if (m == 0xC4) {   // DHT
            size_t o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15; o++;
                if (tc > 1 || th > 3) return fail("jpeg: bad DHT");
                JHuff &h = tc ? hac[th] : hdc[th];
                int total = 0; h.bits[0] = 0;
                for (int i = 1; i <= 16; i++) { h.bits[i] = s[o + (size_t)i - 1]; total += h.bits[i]; }
                o += 16;
                if (total > 256 || o + (size_t)total > sl) return fail("jpeg: bad DHT counts");
                memset(h.vals, 0, sizeof(h.vals)); memcpy(h.vals, s + o, (size_t)total); o += (size_t)total;
                if (!h.build()) return fail("jpeg: bad Huffman table");
                h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {   // SOF0/1/2
            if (have_sof) return fail("jpeg: multiple SOF");
            if (sl < 6) return fail("jpeg: bad SOF");
            if (s[0] != 8) return fail("jpeg: only 8-bit precision is supported");
            H = be16(s + 1); W = be16(s + 3); const int nc = s[5];
            if (!W || !H) return fail("jpeg: zero dimension (DNL is not supported)");
            if ((uint64_t)W * H > MAX_PIXELS) return fail("jpeg: image too large");
            if ((nc != 1 && nc != 3) || sl < 6 + (size_t)nc * 3) return fail("jpeg: unsupported component count (CMYK / YCCK are not supported)");
            progressive = m == 0xC2;
            comps.resize((size_t)nc);
            for (int i = 0; i < nc; i++) { JComp &c = comps[(size_t)i]; c.id = s[6 + i * 3]; c.h = s[7 + i * 3] >> 4; c.v = s[7 + i * 3] & 15; c.tq = s[8 + i * 3];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("jpeg: bad sampling factors"); hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
            mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (JComp &c : comps) {
                if (hmax % c.h || vmax % c.v) return fail("jpeg: fractional sampling ratios are not supported");
                c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
                c.wb = (c.dw + 7) / 8; c.hb = (c.dh + 7) / 8;
                c.wbp = mcux * c.h; c.hbp = mcuy * c.v;
                c.coef.assign((size_t)c.wbp * c.hbp * 64, 0);
            }
            have_sof = true;
        } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return fail("jpeg: lossless / hierarchical / arithmetic-coded files are not supported");
        } else if (m == 0xDD) { if (sl >= 2) restart_interval = be16(s); }
        else if (m == 0xE0) { if (sl >= 5 && !memcmp(s, "JFIF", 5)) jfif = true; }
        else if (m == 0xE1) { const int o = exif_orientation(s, sl); if (o != 1) orientation = o; }
        else if (m == 0xEE) { if (sl >= 12 && !memcmp(s, "Adobe", 5)) { adobe = true; adobe_transform = s[11]; } }
        else if (m == 0xDA) {   // SOS
            if (!have_sof) return fail("jpeg: SOS before SOF");
            if (sl < 1) return fail("jpeg: bad SOS");
            const int ns = s[0];
            if (ns < 1 || ns > (int)comps.size() || sl < 1 + (size_t)ns * 2 + 3) return fail("jpeg: bad SOS");
            JComp *sc[4];
            for (int i = 0; i < ns; i++) {
                sc[i] = nullptr;
                for (JComp &c : comps) if (c.id == s[1 + i * 2]) sc[i] = &c;
                if (!sc[i]) return fail("jpeg: SOS names an unknown component");
                sc[i]->td = s[2 + i * 2] >> 4; sc[i]->ta = s[2 + i * 2] & 15;
                if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("jpeg: bad table selector");
            }
            int Ss = s[1 + ns * 2], Se = s[2 + ns * 2]; const int Ah = s[3 + ns * 2] >> 4, Al = s[3 + ns * 2] & 15;
            if (!progressive) { Ss = 0; Se = 63; }
            if (Ss > Se || Se > 63 || (progressive && Ss == 0 && Se != 0) || (progressive && Ss > 0 && ns != 1) || Al > 13) return fail("jpeg: bad spectral selection");
            for (int i = 0; i < ns; i++) {
                if ((!progressive || Ss == 0) && !(progressive && Ah) && !hdc[sc[i]->td].present) return fail("jpeg: missing DC Huffman table");
                if ((!progressive || Ss > 0) && !hac[sc[i]->ta].present) return fail("jpeg: missing AC Huffman table");
            }
            JBits br(data, n, pos + len);
            for (JComp &c : comps) c.pred = 0;
            int eobrun = 0;
            // geometry of this scan
            const bool inter = ns > 1;
            const int nx = inter ? mcux : sc[0]->wb, ny = inter ? mcuy : sc[0]->hb;
            int until_restart = restart_interval;
            auto decode_block = [&](JComp &c, int16_t *blk) {
                if (!progressive) {
                    const JHuff &dc = hdc[c.td], &ac = hac[c.ta];
                    const int t = br.decode(dc);
                    c.pred += br.receive_extend(t & 15);
                    blk[0] = (int16_t)c.pred;
                    for (int k = 1; k < 64; k++) {
                        if (br.cnt < 16) br.fill();
                        const int fa = ac.fast_ac[(unsigned)(br.buf >> 55)];
                        if (fa) { k += (fa >> 4) & 15; br.drop(fa & 15); blk[ZZ[k]] = (int16_t)(fa >> 8); continue; }     // code + magnitude in one lookup
                        const int rs = br.decode(ac), r = rs >> 4, sz = rs & 15;
                        if (sz) { k += r; blk[ZZ[k]] = (int16_t)br.receive_extend(sz); }
                        else { if (r != 15) break; k += 15; }
                    }
                } else if (Ss == 0) {
                    if (!Ah) { const int t = br.decode(hdc[c.td]); c.pred += br.receive_extend(t & 15); blk[0] = (int16_t)(c.pred * (1 << Al)); }
                    else if (br.get(1)) blk[0] = (int16_t)(blk[0] | (1 << Al));
                } else if (!Ah) {
                    if (eobrun > 0) { eobrun--; return; }
                    const JHuff &ac = hac[c.ta];
                    for (int k = Ss; k <= Se; k++) {
                        const int rs = br.decode(ac), r = rs >> 4, sz = rs & 15;
                        if (sz) { k += r; blk[ZZ[k]] = (int16_t)(br.receive_extend(sz) * (1 << Al)); }
                        else { if (r != 15) { eobrun = (1 << r) - 1; if (r) eobrun += (int)br.get(r); break; } k += 15; }
                    }
                } else {
                    const int p1 = 1 << Al, m1 = -(1 << Al);
                    const JHuff &ac = hac[c.ta];
                    int k = Ss;
                    auto refine = [&](int16_t &cf) { if (br.get(1) && !(cf & p1)) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1)); };
                    if (eobrun == 0) {
                        for (; k <= Se; k++) {
                            const int rs = br.decode(ac); int r = rs >> 4; const int sz = rs & 15;
                            int value = 0;
                            if (sz) value = br.get(1) ? p1 : m1;
                            else if (r != 15) { eobrun = 1 << r; if (r) eobrun += (int)br.get(r); break; }
                            do {
                                int16_t &cf = blk[ZZ[k]];
                                if (cf != 0) refine(cf);
                                else if (--r < 0) break;
                                k++;
                            } while (k <= Se);
                            if (sz && k <= Se) blk[ZZ[k]] = (int16_t)value;
                        }
                    }
                    if (eobrun > 0) {
                        for (; k <= Se; k++) { int16_t &cf = blk[ZZ[k]]; if (cf != 0) refine(cf); }
                        eobrun--;
                    }
                }
            };
            for (int my = 0; my < ny; my++) for (int mx = 0; mx < nx; mx++) {
                if (restart_interval && until_restart == 0) { br.restart(); for (JComp &c : comps) c.pred = 0; eobrun = 0; until_restart = restart_interval; }
                if (inter) {
                    for (int i = 0; i < ns; i++) { JComp &c = *sc[i];
                        for (int by = 0; by < c.v; by++) for (int bx = 0; bx < c.h; bx++)
                            decode_block(c, &c.coef[((size_t)(my * c.v + by) * c.wbp + (size_t)(mx * c.h + bx)) * 64]); }
                } else decode_block(*sc[0], &sc[0]->coef[((size_t)my * sc[0]->wbp + (size_t)mx) * 64]);
                until_restart--;
                if (br.fed_zero > 4096) return fail("jpeg: entropy-coded data ends prematurely");
            }
            any_scan = true;
            // continue parsing after the entropy-coded segment: at the pending marker if one was seen, else scan forward
            if (br.marker) { pos = br.pos - 2 < n ? br.pos - 2 : n; while (pos < n && data[pos] != 0xFF) pos++; }
            else { pos = br.pos; }
            continue;
        }
        pos += len;
    }
    if (!have_sof || !any_scan) return fail("jpeg: no image data");
    // dequantise + inverse DCT
    for (JComp &c : comps) {
        if (!qt_ok[c.tq]) return fail("jpeg: missing quantisation table");
        const size_t stride = (size_t)c.wbp * 8;
        c.plane.resize(stride * (size_t)c.hbp * 8);                        // every sample is written by its block's inverse DCT
        parallel_rows(c.hbp, (size_t)c.wbp * 64, [&](int by0, int by1) {
            for (int by = by0; by < by1; by++) for (int bx = 0; bx < c.wbp; bx++)
                idct_islow(&c.coef[((size_t)by * c.wbp + (size_t)bx) * 64], qt[c.tq], &c.plane[(size_t)by * 8 * stride + (size_t)bx * 8], stride);
        });
        c.coef.clear();
    }
    // upsample every component to W x H (jdsample.c, do_fancy_upsampling = TRUE)
    std::vector<PixelBuf> full(comps.size());
    for (size_t ci = 0; ci < comps.size(); ci++) {
        const JComp &c = comps[ci];
        const int hx = hmax / c.h, vx = vmax / c.v;
        const size_t st = (size_t)c.wbp * 8;
        PixelBuf &f = full[ci];
        const int ow = c.dw * hx, oh = c.dh * vx;   // >= W, H
        f.resize((size_t)ow * oh);                                         // every row is written by the branch below
        auto rowp = [&](int r) { return &c.plane[(size_t)std::min(std::max(r, 0), c.dh - 1) * st]; };   // context rows replicate the first / last real row
        if (hx == 1 && vx == 1) { parallel_rows(oh, (size_t)ow, [&](int y0, int y1) { for (int y = y0; y < y1; y++) memcpy(&f[(size_t)y * ow], rowp(y), (size_t)ow); }); }
        else if (hx == 2 && vx == 1 && c.dw > 2) {   // h2v1_fancy_upsample (jinit_upsampler: fancy only when downsampled_width > 2)
            parallel_rows(oh, (size_t)ow, [&](int y0, int y1) { for (int y = y0; y < y1; y++) { const uint8_t *in = rowp(y); uint8_t *o = &f[(size_t)y * ow]; const int nin = c.dw;
                int iv = in[0]; o[0] = (uint8_t)iv; o[1] = (uint8_t)((iv * 3 + in[1] + 2) >> 2);
                for (int x = 1; x < nin - 1; x++) { iv = in[x] * 3; o[2 * x] = (uint8_t)((iv + in[x - 1] + 1) >> 2); o[2 * x + 1] = (uint8_t)((iv + in[x + 1] + 2) >> 2); }
                iv = in[nin - 1]; o[2 * nin - 2] = (uint8_t)((iv * 3 + in[nin - 2] + 1) >> 2); o[2 * nin - 1] = (uint8_t)iv; } });
        } else if (hx == 2 && vx == 2 && c.dw > 2) {   // h2v2_fancy_upsample (same width condition)
            parallel_rows(c.dh, (size_t)ow * 2, [&](int y0, int y1) { for (int y = y0; y < y1; y++) for (int v = 0; v < 2; v++) {
                const uint8_t *in0 = rowp(y), *in1 = rowp(v == 0 ? y - 1 : y + 1); uint8_t *o = &f[(size_t)(2 * y + v) * ow]; const int nin = c.dw;
                int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                o[0] = (uint8_t)((thiscol * 4 + 8) >> 4); o[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol; thiscol = nextcol;
                for (int x = 1; x < nin - 1; x++) {
                    nextcol = in0[x + 1] * 3 + in1[x + 1];
                    o[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); o[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol; thiscol = nextcol;
                }
                o[2 * nin - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); o[2 * nin - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
            } });
        } else if (hx == 1 && vx == 2) {   // h1v2_fancy_upsample (libjpeg-turbo)
            for (int y = 0; y < c.dh; y++) for (int v = 0; v < 2; v++) {
                const uint8_t *in0 = rowp(y), *in1 = rowp(v == 0 ? y - 1 : y + 1); uint8_t *o = &f[(size_t)(2 * y + v) * ow]; const int bias = v == 0 ? 1 : 2;
                for (int x = 0; x < c.dw; x++) o[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
            }
        } else {   // int_upsample: box replication
            for (int y = 0; y < oh; y++) { const uint8_t *in = rowp(y / vx); uint8_t *o = &f[(size_t)y * ow]; for (int x = 0; x < ow; x++) o[x] = in[x / hx]; }
        }
    }
    out.w = W; out.h = H; out.px.resize((size_t)W * H * 3);              // every byte is written below
    auto at = [&](size_t ci, int x, int y) { const JComp &c = comps[ci]; return full[ci][(size_t)y * (size_t)(c.dw * (hmax / c.h)) + (size_t)x]; };
    if (comps.size() == 1) {
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const uint8_t v = at(0, x, y); uint8_t *q = &out.px[((size_t)y * W + x) * 3]; q[0] = q[1] = q[2] = v; }
    } else {
        // colour space guess of jdapimin.c default_decompress_parms
        bool ycc = true;
        if (jfif) ycc = true;
        else if (adobe) ycc = adobe_transform != 0;
        else { const int a = comps[0].id, b = comps[1].id, cc = comps[2].id; if (a == 1 && b == 2 && cc == 3) ycc = true; else if (a == 'R' && b == 'G' && cc == 'B') ycc = false; else ycc = true; }
        if (!ycc) {
            for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { uint8_t *q = &out.px[((size_t)y * W + x) * 3]; q[0] = at(0, x, y); q[1] = at(1, x, y); q[2] = at(2, x, y); }
        } else {   // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert (SCALEBITS 16)
            int cr_r[256], cb_b[256]; int64_t cr_g[256], cb_g[256];
            auto FIX = [](double v) { return (int64_t)(v * 65536.0 + 0.5); };
            for (int i = 0; i < 256; i++) { const int64_t x = i - 128;
                cr_r[i] = (int)((FIX(1.40200) * x + 32768) >> 16); cb_b[i] = (int)((FIX(1.77200) * x + 32768) >> 16);
                cr_g[i] = (-FIX(0.71414)) * x; cb_g[i] = (-FIX(0.34414)) * x + 32768; }
            auto cl = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
            parallel_rows(H, (size_t)W * 3, [&](int y0, int y1) { for (int y = y0; y < y1; y++) {
                const uint8_t *py = &full[0][(size_t)y * (size_t)(comps[0].dw * (hmax / comps[0].h))], *pb = &full[1][(size_t)y * (size_t)(comps[1].dw * (hmax / comps[1].h))],
                              *pr = &full[2][(size_t)y * (size_t)(comps[2].dw * (hmax / comps[2].h))];
                uint8_t *q = &out.px[(size_t)y * W * 3];
                for (int x = 0; x < W; x++, q += 3) {
                    const int Y = py[x], cb = pb[x], cr = pr[x];
                    q[0] = cl(Y + cr_r[cr]); q[1] = cl(Y + (int)((cb_g[cb] + cr_g[cr]) >> 16)); q[2] = cl(Y + cb_b[cb]);
                } } });
        }
    }
    apply_orientation(out, orientation);
    return true;
}

// =====================================================================================================================
// BMP (uncompressed 8 / 24 / 32 bit) and binary PNM (P5 / P6, maxval <= 255)
// =====================================================================================================================
namespace {
bool decode_bmp(const uint8_t *d, size_t n, ImageRGB8 &out, std::string &err) {
    auto le32 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); };
    auto le16 = [&](size_t o) { return (unsigned)(d[o] | (d[o + 1] << 8)); };
    if (n < 54) { err = "bmp: truncated header"; return false; }
    const size_t off = le32(10), hsz = le32(14);
    if (hsz < 40) { err = "bmp: unsupported header"; return false; }
    const int w = (int)le32(18); int h = (int)le32(22); const unsigned bpp = le16(28), comp = le32(30);
    const bool flip = h > 0; if (h < 0) h = -h;
    if (w <= 0 || h <= 0 || (comp != 0 && !(comp == 3 && bpp == 32)) || (bpp != 8 && bpp != 24 && bpp != 32) || (uint64_t)w * h > MAX_PIXELS) { err = "bmp: unsupported variant"; return false; }
    const size_t stride = ((size_t)w * bpp + 31) / 32 * 4;
    if (off + stride * (size_t)h > n) { err = "bmp: truncated pixel data"; return false; }
    const uint8_t *pal = d + 14 + hsz;
    unsigned ncol = le32(46); if (!ncol) ncol = 256;
    if (bpp == 8 && 14 + hsz + (size_t)ncol * 4 > n) { err = "bmp: truncated palette"; return false; }
    out.w = w; out.h = h; out.px.assign((size_t)w * h * 3, 0);
    for (int y = 0; y < h; y++) {
        const uint8_t *row = d + off + stride * (size_t)(flip ? h - 1 - y : y); uint8_t *q = &out.px[(size_t)y * w * 3];
        for (int x = 0; x < w; x++, q += 3) {
            if (bpp == 8) { const unsigned i = row[x] < ncol ? row[x] : 0; q[0] = pal[i * 4 + 2]; q[1] = pal[i * 4 + 1]; q[2] = pal[i * 4]; }
            else { const uint8_t *p = row + (size_t)x * (bpp / 8); q[0] = p[2]; q[1] = p[1]; q[2] = p[0]; }
        }
    }
    return true;
}
bool decode_pnm(const uint8_t *d, size_t n, ImageRGB8 &out, std::string &err) {
    const bool color = d[1] == '6';
    size_t p = 2; int vals[3], got = 0;
    while (got < 3 && p < n) {
        if (d[p] == '#') { while (p < n && d[p] != '\n') p++; continue; }
        if (isspace(d[p])) { p++; continue; }
        if (!isdigit(d[p])) { err = "pnm: bad header"; return false; }
        long v = 0; while (p < n && isdigit(d[p])) { v = v * 10 + (d[p] - '0'); if (v > (1 << 24)) { err = "pnm: bad header"; return false; } p++; }
        vals[got++] = (int)v;
    }
    if (got < 3 || p >= n) { err = "pnm: truncated header"; return false; }
    p++;   // single whitespace after maxval
    const int w = vals[0], h = vals[1], mv = vals[2];
    if (w <= 0 || h <= 0 || mv <= 0 || mv > 255 || (uint64_t)w * h > MAX_PIXELS) { err = "pnm: unsupported variant"; return false; }
    const size_t need = (size_t)w * h * (color ? 3 : 1);
    if (p + need > n) { err = "pnm: truncated pixel data"; return false; }
    out.w = w; out.h = h; out.px.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) for (int c = 0; c < 3; c++) { const unsigned v = d[p + (color ? i * 3 + (size_t)c : i)]; out.px[i * 3 + (size_t)c] = (uint8_t)(mv == 255 ? v : (v * 255 + (unsigned)mv / 2) / (unsigned)mv); }
    return true;
}
}  // namespace

bool decode_image(const uint8_t *data, size_t n, ImageRGB8 &out, std::string &err) {
    try {
        if (n >= 8 && data[0] == 0x89 && data[1] == 'P') return decode_png(data, n, out, err);
        if (n >= 4 && data[0] == 0xFF && data[1] == 0xD8) return decode_jpeg(data, n, out, err);
        if (n >= 2 && data[0] == 'B' && data[1] == 'M') return decode_bmp(data, n, out, err);
        if (n >= 3 && data[0] == 'P' && (data[1] == '5' || data[1] == '6')) return decode_pnm(data, n, out, err);
    } catch (const std::exception &e) { err = std::string("image decode failed: ") + e.what(); return false; }   // bad_alloc / length_error on absurd headers
    err = "unrecognised image format (supported: PNG, JPEG, BMP, binary PGM/PPM)";
    return false;
}

int load_image_file(const char *path, ImageRGB8 &out) {
    struct stat st;
    if (!path || stat(path, &st) != 0) { set_last_error(std::string("image file does not exist: ") + (path ? path : "(null)")); return E_PathDoesNotExist; }
    FILE *f = fopen(path, "rb");
    if (!f) { set_last_error(std::string("cannot open ") + path); return E_OpenImage; }
    std::vector<uint8_t> buf((size_t)st.st_size);
    const size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    std::string err;
    if (got != buf.size() || !decode_image(buf.data(), buf.size(), out, err)) { set_last_error(std::string(path) + ": " + (err.empty() ? "short read" : err)); return E_OpenImage; }
    return E_None;
}

// =====================================================================================================================
// Pillow libImaging/Resample.c: bicubic_filter, precompute_coeffs (whole-image box), normalize_coeffs_8bpc
// =====================================================================================================================
void precompute_bicubic_8bpc(int in_size, int out_size, ResampleCoeffs &c) {
    constexpr int PRECISION_BITS = 32 - 8 - 2;
    auto filter = [](double x) -> double {
        const double a = -0.5;
        if (x < 0.0) x = -x;
        if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
        if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
        return 0.0;
    };
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    c.in_size = in_size; c.out_size = out_size; c.ksize = ksize;
    c.first.assign((size_t)out_size, 0); c.count.assign((size_t)out_size, 0); c.kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k((size_t)ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; xx++) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5); if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; x++) { const double w = filter((x + xmin - center + 0.5) * ss); k[(size_t)x] = w; ww += w; }
        for (int x = 0; x < xmax; x++) {
            double v = k[(size_t)x]; if (ww != 0.0) v /= ww;
            c.kk[(size_t)xx * ksize + (size_t)x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        c.first[(size_t)xx] = xmin; c.count[(size_t)xx] = xmax;
    }
}

}  // namespace mg4
