#include "formats.hpp"
#include <functional>
#include "common.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
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
    if (r.ok && magic == 0x46554747u) return load_gguf(vocab_only);   // "GGUF"
    if (!r.ok || magic != 0x67676a74u || ver != 3) { set_last_error("LLM file: expected GGJT v3 (magic 0x67676a74, version 3) or GGUF v2/v3"); return E_LoadLanguageModel; }
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

// ---------------------------------------------------------------------------------------------------- format B': GGUF
// The reference pins llama.cpp at a GGJT-v3-era commit (CMakeLists.txt:318) and cannot read GGUF; current converters only write GGUF.  A GGUF file of a
// LLaMA-1 / Vicuna-v0 style model (no grouped-query attention, RMS eps 1e-6, rope base 10000) holds exactly the tensors of the GGJT file under new names, so it
// is mapped onto the same LLMFile view and everything downstream (repack, kernels, tokenizer of the pinned llama.cpp) is unchanged:
//   token_embd -> tok_embeddings, output_norm -> norm, output -> output, blk.N.{attn_norm, attn_q, attn_k, attn_v, attn_output, ffn_norm, ffn_gate, ffn_down,
//   ffn_up} -> layers.N.{attention_norm, attention.wq, .wk, .wv, .wo, ffn_norm, feed_forward.w1, .w2, .w3};  vocabulary pieces: U+2581 -> ' ' and "<0xXX>" byte
//   tokens -> the raw byte, which is the form the GGJT converter of that era stored.
int LLMFile::load_gguf(bool vocab_only) {
    auto fail = [&](const std::string &m) { set_last_error("GGUF file: " + m); return (int)E_LoadLanguageModel; };
    Reader r(mf.data, mf.size);
    r.u4();
    const uint32_t ver = r.u4();
    if (ver != 2 && ver != 3) return fail("only versions 2 and 3 are supported");
    auto u8 = [&]() -> uint64_t { uint64_t v = 0; if (r.need(8)) { memcpy(&v, r.p + r.pos, 8); r.pos += 8; } return v; };
    auto gstr = [&]() -> std::string { const uint64_t n = u8(); if (!r.ok || n > (1u << 24)) { r.ok = false; return std::string(); } return r.str((size_t)n); };
    const uint64_t n_tensors = u8(), n_kv = u8();
    if (!r.ok || n_tensors > (r.size - r.pos) / 24 || n_kv > (r.size - r.pos) / 12) return fail("absurd counts");
    static const size_t scalar_size[13] = {1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8};
    uint64_t alignment = 32, head_kv = 0, head = 0;
    double eps = 1e-6, rope_base = 10000.0;
    std::string arch, tok_model;
    std::vector<int32_t> token_type;
    bool have_tokens = false;
    for (uint64_t i = 0; i < n_kv && r.ok; i++) {
        const std::string key = gstr();
        const uint32_t vt = r.u4();
        auto scalar_u = [&](uint32_t t) -> uint64_t {          // integer-valued scalar of type t
            uint64_t v = 0;
            if (t > 12 || !scalar_size[t] || !r.need(scalar_size[t])) { r.ok = false; return 0; }
            memcpy(&v, r.p + r.pos, scalar_size[t]); r.pos += scalar_size[t];
            return v;
        };
        if (vt == 8) { const std::string v = gstr(); if (key == "general.architecture") arch = v; else if (key == "tokenizer.ggml.model") tok_model = v; continue; }
        if (vt == 9) {
            const uint32_t et = r.u4(); const uint64_t cnt = u8();
            if (!r.ok || cnt > (r.size - r.pos)) return fail("absurd array length");
            if (key == "tokenizer.ggml.tokens" && et == 8) {
                if (cnt > (1u << 24)) return fail("absurd vocabulary");
                pieces.resize((size_t)cnt);
                for (uint64_t k = 0; k < cnt && r.ok; k++) pieces[(size_t)k] = gstr();
                have_tokens = true;
            } else if (key == "tokenizer.ggml.scores" && et == 6) {
                scores.resize((size_t)cnt);
                if (!r.need((size_t)cnt * 4)) return fail("truncated scores");
                memcpy(scores.data(), r.p + r.pos, (size_t)cnt * 4); r.pos += (size_t)cnt * 4;
            } else if (key == "tokenizer.ggml.token_type" && (et == 5 || et == 4)) {
                token_type.resize((size_t)cnt);
                if (!r.need((size_t)cnt * 4)) return fail("truncated token types");
                memcpy(token_type.data(), r.p + r.pos, (size_t)cnt * 4); r.pos += (size_t)cnt * 4;
            } else {   // an array this loader has no use for: skip it (strings one by one; arrays may nest, bounded depth; fixed-size elements in one step)
                std::function<bool(uint32_t, uint64_t, int)> skip = [&](uint32_t t, uint64_t n, int depth) -> bool {
                    if (t == 8) { for (uint64_t k = 0; k < n && r.ok; k++) gstr(); return r.ok; }
                    if (t == 9) {
                        if (depth >= 4) return false;
                        for (uint64_t k = 0; k < n && r.ok; k++) { const uint32_t it = r.u4(); const uint64_t ic = u8(); if (!r.ok || ic > (r.size - r.pos) || !skip(it, ic, depth + 1)) return false; }
                        return r.ok;
                    }
                    if (t > 12 || !scalar_size[t] || !r.need((size_t)n * scalar_size[t])) return false;
                    r.pos += (size_t)n * scalar_size[t];
                    return true;
                };
                if (!skip(et, cnt, 0)) return fail("bad array");
            }
            continue;
        }
        if (vt == 6 || vt == 12) {                                // f32 / f64
            double v = 0;
            if (vt == 6) v = r.f4(); else { if (r.need(8)) { memcpy(&v, r.p + r.pos, 8); r.pos += 8; } }
            if (key == "llama.attention.layer_norm_rms_epsilon") eps = v; else if (key == "llama.rope.freq_base") rope_base = v;
            continue;
        }
        const uint64_t v = scalar_u(vt);
        if (key == "general.alignment") alignment = v;
        else if (key == "llama.embedding_length") n_embd = (uint32_t)v;
        else if (key == "llama.block_count") n_layer = (uint32_t)v;
        else if (key == "llama.feed_forward_length") n_ff_explicit = (uint32_t)v;
        else if (key == "llama.attention.head_count") head = v;
        else if (key == "llama.attention.head_count_kv") head_kv = v;
        else if (key == "llama.rope.dimension_count") n_rot = (uint32_t)v;
        else if (key == "general.file_type") ftype = (uint32_t)v;
    }
    if (!r.ok) return fail("truncated metadata");
    if (arch != "llama") return fail("general.architecture must be \"llama\"");
    if (!tok_model.empty() && tok_model != "llama") return fail("only the SentencePiece (\"llama\") tokenizer model is supported");
    n_head = (uint32_t)head;
    if (!have_tokens || scores.size() != pieces.size() || pieces.empty()) return fail("missing tokenizer.ggml.tokens / scores");
    if (!n_embd || !n_layer || !n_head || !n_ff_explicit || n_embd % n_head || n_embd > (1u << 20) || n_layer > (1u << 16) || n_ff_explicit > (1u << 22)) return fail("bad hyper-parameters");
    if (head_kv && head_kv != head) return fail("grouped-query attention (head_count_kv != head_count) is not supported: the reference's LLaMA-1 / Vicuna-v0 graph has none");
    if (fabs(eps - 1e-6) > 1e-9) return fail("layer_norm_rms_epsilon must be 1e-6 (the value of the reference's pinned llama.cpp)");
    if (fabs(rope_base - 10000.0) > 1e-3) return fail("rope.freq_base must be 10000");
    if (alignment == 0 || alignment > 4096 || (alignment & (alignment - 1))) return fail("bad general.alignment");
    n_vocab = (uint32_t)pieces.size(); n_mult = 1;
    for (size_t i = 0; i < pieces.size(); i++) {                   // SentencePiece pieces -> the GGJT-era form
        std::string &p = pieces[i];
        const bool is_byte = (i < token_type.size() && token_type[i] == 6) || (p.size() == 6 && p.compare(0, 3, "<0x") == 0 && p[5] == '>');
        if (is_byte && p.size() == 6 && p.compare(0, 3, "<0x") == 0) { p = std::string(1, (char)strtol(p.substr(3, 2).c_str(), nullptr, 16)); continue; }
        std::string q; q.reserve(p.size());
        for (size_t k = 0; k < p.size();) {
            if (k + 2 < p.size() && (unsigned char)p[k] == 0xE2 && (unsigned char)p[k + 1] == 0x96 && (unsigned char)p[k + 2] == 0x81) { q.push_back(' '); k += 3; }   // U+2581
            else q.push_back(p[k++]);
        }
        p.swap(q);
    }
    if (vocab_only) return E_None;
    struct Info { std::string name; TensorMeta t; uint64_t rel; };
    std::vector<Info> infos((size_t)n_tensors);
    for (auto &in : infos) {
        in.name = gstr();
        const uint32_t nd = r.u4();
        if (!r.ok || nd < 1 || nd > 2) return fail("tensor " + in.name + ": only 1-D / 2-D tensors are expected");
        for (uint32_t d = 0; d < nd; d++) { const uint64_t v = u8(); if (v > (1u << 30)) return fail("absurd tensor shape"); in.t.ne.push_back((int64_t)v); }
        in.t.type = (int)r.u4();
        in.rel = u8();
        if (!r.ok || gt_bytes(in.t.type) == 0 || in.t.ne[0] % gt_block(in.t.type)) return fail("tensor " + in.name + " has an unusable type / shape");
    }
    const size_t data0 = (r.pos + (size_t)alignment - 1) & ~((size_t)alignment - 1);
    auto rename = [&](const std::string &g) -> std::string {
        if (g == "token_embd.weight") return "tok_embeddings.weight";
        if (g == "output_norm.weight") return "norm.weight";
        if (g == "output.weight") return "output.weight";
        if (g.compare(0, 4, "blk.") == 0) {
            const size_t dot = g.find('.', 4);
            if (dot == std::string::npos) return "";
            const std::string idx = g.substr(4, dot - 4), rest = g.substr(dot + 1);
            static const char *map[][2] = {{"attn_norm.weight", "attention_norm.weight"}, {"attn_q.weight", "attention.wq.weight"}, {"attn_k.weight", "attention.wk.weight"},
                {"attn_v.weight", "attention.wv.weight"}, {"attn_output.weight", "attention.wo.weight"}, {"ffn_norm.weight", "ffn_norm.weight"}, {"ffn_gate.weight", "feed_forward.w1.weight"},
                {"ffn_down.weight", "feed_forward.w2.weight"}, {"ffn_up.weight", "feed_forward.w3.weight"}};
            for (auto &m : map) if (rest == m[0]) return "layers." + idx + "." + m[1];
        }
        return "";
    };
    for (auto &in : infos) {
        TensorMeta t = in.t;
        t.nbytes = gt_nbytes(t.type, (size_t)t.nelements());
        if (in.rel % alignment || in.rel > mf.size || data0 > mf.size - in.rel || t.nbytes > mf.size - data0 - in.rel) return fail("tensor " + in.name + " runs past EOF");
        t.offset = data0 + (size_t)in.rel;
        t.name = rename(in.name);
        if (t.name.empty()) continue;                              // e.g. rope_freqs: not part of this graph
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
    if (text.empty()) return out;        // llama_tokenize (llama.cpp master-31cfbb1) returns before the BOS push for an empty text
    if (add_bos) out.push_back(1);
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
