// Readers for the two on-disk formats the reference consumes (host only, no GPU dependency).
//   format A: MiniGPT-4 vision container, "ggml" magic v1   (reference reader minigpt4.cpp:1478-1596, writer convert.py:74-180)
//   format B: Vicuna LLM file, GGJT v3                        (read by llama.cpp behind minigpt4.cpp:1783; layout SURVEY.md 2.5)
//   format B': the same model as a GGUF v2/v3 file (what current llama.cpp converters write) -- mapped onto the GGJT view below
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace mg4 {

struct MappedFile {
    int fd = -1;
    const uint8_t *data = nullptr;
    size_t size = 0;
    bool open(const std::string &path);
    void close();
    ~MappedFile() { close(); }
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

struct TensorMeta {
    std::string name;
    int type = -1;                 // ggml_type numbering
    std::vector<int64_t> ne;       // innermost first
    size_t offset = 0, nbytes = 0;
    int64_t nelements() const { int64_t n = 1; for (auto v : ne) n *= v; return n; }
};

struct VisionFile {
    MappedFile mf;
    int version = 0, ftype = 0;
    std::string config_json;
    std::vector<std::string> model_order;
    std::map<std::string, std::map<std::string, TensorMeta>> models;
    int load(const std::string &path);   // returns MiniGPT4Error
    const TensorMeta *find(const std::string &model, const std::string &name) const;
    int64_t config_int(const std::string &key, int64_t dflt) const;   // first occurrence of "key": <int> in the JSON config
};

struct LLMFile {
    MappedFile mf;
    uint32_t n_vocab = 0, n_embd = 0, n_mult = 0, n_head = 0, n_layer = 0, n_rot = 0, ftype = 0;
    std::vector<std::string> pieces;
    std::vector<float> scores;
    std::map<std::string, TensorMeta> tensors;
    uint32_t n_ff_explicit = 0;                                   // GGUF files state the feed-forward width; GGJT files derive it from n_mult
    int load(const std::string &path, bool vocab_only = false);   // returns MiniGPT4Error (LoadLanguageModel on malformed input); GGJT v3 or GGUF v2/v3 by magic
    int load_gguf(bool vocab_only);                               // mf already mapped
    uint32_t n_ff() const { return n_ff_explicit ? n_ff_explicit : ((2 * (4 * n_embd) / 3 + n_mult - 1) / n_mult) * n_mult; }
    const TensorMeta *find(const std::string &name) const { auto it = tensors.find(name); return it == tensors.end() ? nullptr : &it->second; }
};

// llama.cpp (master-31cfbb1) SentencePiece-style tokenizer over the file vocab: UTF-8 split, greedy best-score bigram merges,
// byte fallback (id = byte + 3).  Reference call site: minigpt4.cpp:2389.
struct Tokenizer {
    const std::vector<std::string> *pieces = nullptr;
    const std::vector<float> *scores = nullptr;
    std::unordered_map<std::string, int> token_to_id;
    void init(const LLMFile &f);
    std::vector<int> tokenize(const std::string &text, bool add_bos) const;
};

}  // namespace mg4
