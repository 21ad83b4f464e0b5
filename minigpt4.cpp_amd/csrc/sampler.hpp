// Host-side sampling chain behind minigpt4_end_chat(_image): restatement of the llama.cpp (master-31cfbb1) llama_sample_* calls made by
// MiniGPT4::sample_token (reference minigpt4.cpp:2425-2483).  Vocabulary-sized work (~32k floats): negligible next to a decode step.
#pragma once
#include <cstdint>
#include <random>
#include <vector>

namespace mg4 {

struct TokenData { int id; float logit; float p; };
struct Candidates { std::vector<TokenData> data; bool sorted = false; };

struct SampleParams {
    float temp = 0.8f; int32_t top_k = 40; float top_p = 0.9f; float tfs_z = 1.0f; float typical_p = 1.0f;
    int mirostat = 0; float mirostat_tau = 5.0f; float mirostat_eta = 1.0f;
};

struct Sampler {
    std::mt19937 rng;
    // The reference keeps mirostat_mu in function-local statics (process-wide, initialised from the first call's tau, minigpt4.cpp:2458,2465).
    // Here it is per context with the same initialisation rule.
    bool mu1_init = false, mu2_init = false; float mu1 = 0.0f, mu2 = 0.0f;
    void seed(int s);
    int sample(const float *logits, int n_vocab, const SampleParams &p);

    static void softmax(Candidates &c);
    static void top_k(Candidates &c, int k, size_t min_keep);
    static void top_p(Candidates &c, float p, size_t min_keep);
    static void tail_free(Candidates &c, float z, size_t min_keep);
    static void typical(Candidates &c, float p, size_t min_keep);
    static void temperature(Candidates &c, float t);
    int token(Candidates &c);
    static int greedy(const Candidates &c);
    int mirostat_v1(Candidates &c, float tau, float eta, int m, float *mu);
    int mirostat_v2(Candidates &c, float tau, float eta, float *mu);
};

}  // namespace mg4
