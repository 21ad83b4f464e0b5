#include "formats.hpp"
#include "common.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdarg>
#include <cstring>
#include <queue>

namespace mg4 {

// ---------------------------------------------------------------------------------------------------- logging / errors
int g_verbosity = 1;
static thread_local std::string t_last_error;
void set_last_error(const std::string &s) { t_last_error = s; }
const std::string &last_error() { return t_last_error; }
void log_msg(int level, const char *tag, const char *fmt, ...) {
    if (g_verbosity < level) return;
    char buf[2048];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    FILE *f = level == 1 ? stderr : stdout;   // reference: DEBUG/INFO -> stdout, ERR -> stderr (minigpt4.cpp:197-228)
    fprintf(f, "%s: %s\n", tag, buf);
    fflush(f);
}
const char *gt_name(int t) {
    switch (t) { case GT_F32: return "f32"; case GT_F16: return "f16"; case GT_Q4_0: return "q4_0"; case GT_Q4_1: return "q4_1"; case GT_Q5_0: return "q5_0";
        case GT_Q5_1: return "q5_1"; case GT_Q8_0: return "q8_0"; case GT_Q8_1: return "q8_1"; case GT_Q2_K: return "q2_K"; case GT_Q3_K: return "q3_K";
        case GT_Q4_K: return "q4_K"; case GT_Q5_K: return "q5_K"; case GT_Q6_K: return "q6_K"; case GT_Q8_K: return "q8_K"; case GT_I32: return "i32"; case GT_I64: return "i64"; default: return "?"; }
}

// ---------------------------------------------------------------------------------------------------- mmap
bool MappedFile::open(const std::string &path) {
    close();
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(); return false; }
    size = (size_t)st.st_size;
    void *p = mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) { data = nullptr; close(); return false; }
    data = (const uint8_t *)p;
    madvise(p, size, MADV_WILLNEED);
    return true;
}
void MappedFile::close() {
    if (data) munmap((void *)data, size);
    if (fd >= 0) ::close(fd);
    data = nullptr; fd = -1; size = 0;
}

namespace {
struct Reader {
    const uint8_t *p; size_t size, pos = 0; bool ok = true;
    Reader(const uint8_t *p_, size_t s) : p(p_), size(s) {}
    bool need(size_t n) { if (pos + n > size || pos + n < pos) { ok = false; return false; } return true; }
    int32_t s4() { int32_t v = 0; if (need(4)) { memcpy(&v, p + pos, 4); pos += 4; } return v; }
    uint32_t u4() { return (uint32_t)s4(); }
    float f4() { float v = 0; if (need(4)) { memcpy(&v, p + pos, 4); pos += 4; } return v; }
    std::string str(size_t n) { std::string s; if (need(n)) { s.assign((const char *)p + pos, n); pos += n; } return s; }
};
// MiniGPT4DataType -> ggml type (reference minigpt4.h:30-48; mapping at minigpt4.cpp:555-739)
int mg4_to_ggml(int dt) {
    static const int map[16] = {GT_F16, GT_F32, GT_I32, GT_I64, GT_Q4_0, GT_Q4_1, GT_Q5_0, GT_Q5_1, GT_Q8_0, GT_Q8_1, GT_Q2_K, GT_Q3_K, GT_Q4_K, GT_Q5_K, GT_Q6_K, GT_Q8_K};
    return dt >= 0 && dt < 16 ? map[dt] : -1;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------- format A
int VisionFile::load(const std::string &path) {
    if (!mf.open(path)) { set_last_error("cannot mmap " + path); return E_MmapSupport; }
    Reader r(mf.data, mf.size);
    if (r.str(4) != "ggml") { set_last_error("vision file: bad magic"); return E_LoadModelFileHeader; }
    version = r.s4();
    if (!r.ok || version == 0) { set_last_error("vision file: unknown version"); return E_LoadModelFileVersion; }
    ftype = r.s4();
    if (mg4_to_ggml(ftype) < 0) return E_LoadModelMiniGPT4DataType;
    const int32_t clen = r.s4();
    if (clen < 0) return E_LoadModelFileHeader;
    config_json = r.str((size_t)clen);
    if (!r.ok) return E_LoadModelFileHeader;
    while (r.pos < r.size) {
        const int32_t nlen = r.s4();
        if (!r.ok || nlen < 0 || nlen > 4096) return E_LoadModelFileHeader;
        const std::string mname = r.str((size_t)nlen);
        const int32_t nt = r.s4();
        if (!r.ok || nt < 0 || (size_t)nt > (r.size - r.pos) / 12) return E_LoadModelFileHeader;   // a tensor header is at least 12 bytes: a corrupt count cannot make us allocate
        std::vector<TensorMeta> metas((size_t)nt);
        for (auto &t : metas) {
            const int32_t l = r.s4();
            if (!r.ok || l < 0 || l > 4096) return E_LoadModelFileHeader;
            t.name = r.str((size_t)l);
            const int32_t nd = r.s4();
            if (!r.ok || nd < 0 || nd > 8) return E_LoadModelFileHeader;
            int64_t prod = 1;
            for (int i = 0; i < nd; i++) { const int32_t d = r.s4(); if (d < 0 || (prod *= (int64_t)d) > ((int64_t)1 << 40)) return E_LoadModelFileHeader; t.ne.push_back(d); }
            t.type = mg4_to_ggml(r.s4());
            if (!r.ok) return E_LoadModelFileHeader;
            if (t.type < 0) return E_LoadModelMiniGPT4DataType;
            const int64_t n = t.nelements();
            if (n < 0 || (gt_block(t.type) > 1 && (t.ne.empty() || t.ne[0] % gt_block(t.type)))) return E_LoadModelMiniGPT4DataType;
            t.nbytes = gt_nbytes(t.type, (size_t)n);
        }
        auto &m = models[mname];
        model_order.push_back(mname);
        for (auto &t : metas) {
            if (r.pos & 4095) r.pos = (r.pos + 4096) & ~(size_t)4095;   // seek_to_alignment(PAGE_SIZE), minigpt4.cpp:1567
            t.offset = r.pos;
            if (!r.need(t.nbytes)) { set_last_error("vision file: tensor " + t.name + " runs past EOF"); return E_LoadModelFileHeader; }
            r.pos += t.nbytes;
            m[t.name] = t;
        }
    }
    return E_None;
}
const TensorMeta *VisionFile::find(const std::string &model, const std::string &name) const {
    auto mi = models.find(model);
    if (mi == models.end()) return nullptr;
    auto ti = mi->second.find(name);
    return ti == mi->second.end() ? nullptr : &ti->second;
}
int64_t VisionFile::config_int(const std::string &key, int64_t dflt) const {
    const std::string pat = "\"" + key + "\"";
    size_t p = config_json.find(pat);
    if (p == std::string::npos) return dflt;
    p = config_json.find(':', p + pat.size());
    if (p == std::string::npos) return dflt;
    p++;
    while (p < config_json.size() && (config_json[p] == ' ' || config_json[p] == '\t' || config_json[p] == '\n')) p++;
    char *end = nullptr;
    const long long v = strtoll(config_json.c_str() + p, &end, 10);
    return end == config_json.c_str() + p ? dflt : (int64_t)v;
}

// ---------------------------------------------------------------------------------------------------- format B
int LLMFile::load(const std::string &path, bool vocab_only) {
    if (!mf.open(path)) { set_last_error("cannot mmap " + path); return E_LoadLanguageModel; }
    Reader r(mf.data, mf.size);
    const uint32_t magic = r.u4(), ver = r.u4();
    if (!r.ok || magic != 0x67676a74u || ver != 3) { set_last_error("LLM file: expected GGJT v3 (magic 0x67676a74, version 3)"); return E_LoadLanguageModel; }
    n_vocab = r.u4(); n_embd = r.u4(); n_mult = r.u4(); n_head = r.u4(); n_layer = r.u4(); n_rot = r.u4(); ftype = r.u4();
    if (!r.ok || !n_vocab || !n_embd || !n_mult || !n_head || !n_layer || n_embd % n_head || n_vocab > (1u << 24) || (size_t)n_vocab > (r.size - r.pos) / 8 || n_embd > (1u << 20) ||
        n_layer > (1u << 16) || n_mult > (1u << 20)) { set_last_error("LLM file: bad hparams"); return E_LoadLanguageModel; }   // a vocab entry is at least 8 bytes
    pieces.resize(n_vocab); scores.resize(n_vocab);
    for (uint32_t i = 0; i < n_vocab; i++) {
        const uint32_t len = r.u4();
        if (!r.ok || len > (1u << 16)) { set_last_error("LLM file: bad vocab entry"); return E_LoadLanguageModel; }
        pieces[i] = r.str(len); scores[i] = r.f4();
    }
    if (!r.ok) return E_LoadLanguageModel;
    if (vocab_only) return E_None;
    while (r.pos < r.size) {
        const uint32_t nd = r.u4(), nl = r.u4(), ty = r.u4();
        if (!r.ok || nd < 1 || nd > 2 || nl > 4096) { set_last_error("LLM file: bad tensor header"); return E_LoadLanguageModel; }
        TensorMeta t; t.type = (int)ty;
        for (uint32_t i = 0; i < nd; i++) { const uint32_t d = r.u4(); if (d > (1u << 30)) { set_last_error("LLM file: absurd tensor shape"); return E_LoadLanguageModel; } t.ne.push_back(d); }
        if (t.nelements() > ((int64_t)1 << 40)) { set_last_error("LLM file: absurd tensor shape"); return E_LoadLanguageModel; }   // two dims <= 2^30: the product cannot overflow
        t.name = r.str(nl);
        if (!r.ok || gt_bytes(t.type) == 0 || t.ne[0] % gt_block(t.type)) { set_last_error("LLM file: tensor " + t.name + " has an unusable type/shape"); return E_LoadLanguageModel; }
        r.pos = (r.pos + 31) & ~(size_t)31;
        t.offset = r.pos; t.nbytes = gt_nbytes(t.type, (size_t)t.nelements());
        if (!r.need(t.nbytes)) { set_last_error("LLM file: tensor " + t.name + " runs past EOF"); return E_LoadLanguageModel; }
        r.pos += t.nbytes;
        tensors[t.name] = t;
    }
    return E_None;
}

// ---------------------------------------------------------------------------------------------------- tokenizer
void Tokenizer::init(const LLMFile &f) {
    pieces = &f.pieces; scores = &f.scores;
    token_to_id.clear();
    token_to_id.reserve(f.pieces.size() * 2);
    for (size_t i = 0; i < f.pieces.size(); i++) token_to_id[f.pieces[i]] = (int)i;   // later duplicates win, as in llama.cpp
}
std::vector<int> Tokenizer::tokenize(const std::string &text, bool add_bos) const {
    std::vector<int> out;
    if (add_bos) out.push_back(1);
    if (text.empty()) return out;
    struct Sym { size_t start, n; int prev, next; };
    std::vector<Sym> syms;
    for (size_t i = 0; i < text.size();) {
        const unsigned char b = (unsigned char)text[i];
        size_t len = b < 0x80 ? 1 : (b & 0xE0) == 0xC0 ? 2 : (b & 0xF0) == 0xE0 ? 3 : (b & 0xF8) == 0xF0 ? 4 : 1;
        len = std::min(len, text.size() - i);
        syms.push_back({i, len, (int)syms.size() - 1, (int)syms.size() + 1});
        i += len;
    }
    syms.back().next = -1;
    struct Bigram { int left, right; float score; size_t size; };
    auto cmp = [](const Bigram &l, const Bigram &r) { return (l.score < r.score) || (l.score == r.score && l.left > r.left); };
    std::priority_queue<Bigram, std::vector<Bigram>, decltype(cmp)> work(cmp);
    auto try_add = [&](int l, int r) {
        if (l == -1 || r == -1) return;
        const std::string piece = text.substr(syms[l].start, syms[l].n + syms[r].n);
        auto it = token_to_id.find(piece);
        if (it == token_to_id.end() || (size_t)it->second >= scores->size()) return;
        work.push({l, r, (*scores)[it->second], piece.size()});
    };
    for (size_t i = 1; i < syms.size(); i++) try_add((int)i - 1, (int)i);
    while (!work.empty()) {
        const Bigram bg = work.top(); work.pop();
        Sym &L = syms[bg.left], &R = syms[bg.right];
        if (L.n == 0 || R.n == 0 || L.n + R.n != bg.size) continue;
        L.n += R.n; R.n = 0;
        L.next = R.next;
        if (R.next >= 0) syms[R.next].prev = bg.left;
        try_add(L.prev, bg.left);
        try_add(bg.left, L.next);
    }
    for (int i = 0; i != -1; i = syms[i].next) {
        const Sym &s = syms[i];
        auto it = token_to_id.find(text.substr(s.start, s.n));
        if (it == token_to_id.end()) { for (size_t j = 0; j < s.n; j++) out.push_back((int)(unsigned char)text[s.start + j] + 3); }
        else out.push_back(it->second);
    }
    return out;
}

}  // namespace mg4
