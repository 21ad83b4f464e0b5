#include "sampler.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <ctime>
#include <numeric>

namespace mg4 {

void Sampler::seed(int s) {
    if (s < 0) s = (int)time(nullptr);   // llama.cpp: seed < 0 -> time(NULL)
    rng.seed((uint32_t)s);
}

// All candidates by descending logit (llama_sample_softmax's std::sort over the whole vocabulary: what every mirostat step and an explicit top_k = 0 pay).  A comparison sort of
// 32000 entries costs ~1.9 ms per token on the GPU box's host -- more than half of the GPU's decode step; three stable 11-bit radix passes over (order-preserving key, position)
// cost ~0.15 ms.  Ties keep their token order (stable): the rule the CPU oracle's restatement uses (std::sort leaves the order of equal logits to the implementation).
static void sort_desc_by_logit(std::vector<TokenData> &v) {
    const size_t n = v.size();
    if (n < 1024) { std::stable_sort(v.begin(), v.end(), [](const TokenData &a, const TokenData &b) { return a.logit > b.logit; }); return; }
    for (const TokenData &t : v) if (t.logit != t.logit) {       // a NaN has no place in a radix order: the comparison sort's behaviour, whatever it is
        std::stable_sort(v.begin(), v.end(), [](const TokenData &a, const TokenData &b) { return a.logit > b.logit; }); return; }
    struct KP { uint32_t key, pos; };
    std::vector<KP> a(n), b(n);
    for (size_t i = 0; i < n; i++) {
        uint32_t u; const float f = v[i].logit == 0.0f ? 0.0f : v[i].logit;    // -0 and +0 compare equal: one key
        std::memcpy(&u, &f, 4);
        const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // ascending in the float's value
        a[i] = KP{~asc, (uint32_t)i};                                            // ascending key = descending logit
    }
    for (int pass = 0; pass < 3; pass++) {
        const int sh = 11 * pass;
        uint32_t cnt[2049] = {0};
        for (size_t i = 0; i < n; i++) cnt[((a[i].key >> sh) & 2047u) + 1]++;
        for (int k = 0; k < 2048; k++) cnt[k + 1] += cnt[k];
        for (size_t i = 0; i < n; i++) b[cnt[(a[i].key >> sh) & 2047u]++] = a[i];
        a.swap(b);
    }
    std::vector<TokenData> out(n);
    for (size_t i = 0; i < n; i++) out[i] = v[a[i].pos];
    v.swap(out);
}
void Sampler::softmax(Candidates &c) {
    if (c.data.empty()) return;
    if (!c.sorted) {
        sort_desc_by_logit(c.data);
        c.sorted = true;
    }
    const float max_l = c.data[0].logit;
    float cum = 0.0f;
    for (auto &t : c.data) { const float p = expf(t.logit - max_l); t.p = p; cum += p; }
    for (auto &t : c.data) t.p /= cum;
}
void Sampler::top_k(Candidates &c, int k, size_t min_keep) {
    k = std::max(k, (int)min_keep);
    k = std::min(k, (int)c.data.size());
    if (!c.sorted) {
        auto comp = [](const TokenData &a, const TokenData &b) { return a.logit > b.logit; };
        if (k == (int)c.data.size()) sort_desc_by_logit(c.data);
        else std::partial_sort(c.data.begin(), c.data.begin() + k, c.data.end(), comp);
        c.sorted = true;
    }
    c.data.resize((size_t)k);
}
void Sampler::top_p(Candidates &c, float p, size_t min_keep) {
    if (p >= 1.0f) return;
    softmax(c);
    float cum = 0.0f; size_t last = c.data.size();
    for (size_t i = 0; i < c.data.size(); i++) { cum += c.data[i].p; if (cum >= p && i + 1 >= min_keep) { last = i + 1; break; } }
    c.data.resize(last);
}
void Sampler::tail_free(Candidates &c, float z, size_t min_keep) {
    if (z >= 1.0f || c.data.size() <= 2) return;
    softmax(c);
    std::vector<float> d1(c.data.size() - 1), d2(c.data.size() - 2);
    for (size_t i = 0; i < d1.size(); i++) d1[i] = c.data[i].p - c.data[i + 1].p;
    for (size_t i = 0; i < d2.size(); i++) d2[i] = std::abs(d1[i] - d1[i + 1]);
    const float sum = std::accumulate(d2.begin(), d2.end(), 0.0f);
    if (sum > 1e-6f) for (float &v : d2) v /= sum; else for (float &v : d2) v = 1.0f / d2.size();
    float cum = 0.0f; size_t last = c.data.size();
    for (size_t i = 0; i < d2.size(); i++) { cum += d2[i]; if (cum > z && i >= min_keep) { last = i; break; } }
    c.data.resize(last);
}
void Sampler::typical(Candidates &c, float p, size_t min_keep) {
    if (p >= 1.0f) return;
    softmax(c);
    float entropy = 0.0f;
    for (auto &t : c.data) entropy += -t.p * logf(t.p);
    std::vector<float> shifted(c.data.size());
    for (size_t i = 0; i < c.data.size(); i++) shifted[i] = fabsf(-logf(c.data[i].p) - entropy);
    std::vector<size_t> idx(c.data.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return shifted[a] < shifted[b]; });
    float cum = 0.0f; size_t last = idx.size();
    for (size_t i = 0; i < idx.size(); i++) { cum += c.data[idx[i]].p; if (cum > p && i >= min_keep - 1) { last = i + 1; break; } }
    std::vector<TokenData> nd;
    for (size_t i = 0; i < last; i++) nd.push_back(c.data[idx[i]]);
    c.data.swap(nd);   // `sorted` is left as-is, like llama.cpp at this revision
}
void Sampler::temperature(Candidates &c, float t) { for (auto &d : c.data) d.logit /= t; }
int Sampler::token(Candidates &c) {
    softmax(c);
    std::vector<float> probs; probs.reserve(c.data.size());
    for (auto &t : c.data) probs.push_back(t.p);
    std::discrete_distribution<> dist(probs.begin(), probs.end());
    return c.data[(size_t)dist(rng)].id;
}
int Sampler::greedy(const Candidates &c) {
    auto it = std::max_element(c.data.begin(), c.data.end(), [](const TokenData &a, const TokenData &b) { return a.logit < b.logit; });
    return it->id;
}
int Sampler::mirostat_v1(Candidates &c, float tau, float eta, int m, float *mu) {
    const float N = (float)c.data.size();
    softmax(c);
    float s_hat = 0.0f, sum_ti_bi = 0.0f, sum_ti_sq = 0.0f;
    for (size_t i = 0; i < size_t(m - 1) && i < c.data.size() - 1; ++i) {
        const float t_i = logf(float(i + 2) / float(i + 1)), b_i = logf(c.data[i].p / c.data[i + 1].p);
        sum_ti_bi += t_i * b_i; sum_ti_sq += t_i * t_i;
    }
    s_hat = sum_ti_bi / sum_ti_sq;
    const float epsilon_hat = s_hat - 1;
    const float k = powf((epsilon_hat * powf(2, *mu)) / (1 - powf(N, -epsilon_hat)), 1 / s_hat);
    // int(k) of llama.cpp: k is inf / NaN for degenerate distributions (one candidate, all-equal or infinite logits); the x86 conversion the reference's build performs
    // yields INT_MIN there, which top_k then raises to min_keep -- made explicit instead of relying on undefined behaviour
    const int k_int = (k > -2147483648.0f && k < 2147483648.0f) ? (int)k : (-2147483647 - 1);
    top_k(c, k_int, 1);
    const int X = token(c);
    const size_t xi = std::distance(c.data.begin(), std::find_if(c.data.begin(), c.data.end(), [&](const TokenData &t) { return t.id == X; }));
    const float observed = -log2f(c.data[xi].p);
    *mu = *mu - eta * (observed - tau);
    return X;
}
int Sampler::mirostat_v2(Candidates &c, float tau, float eta, float *mu) {
    softmax(c);
    auto cut = std::find_if(c.data.begin(), c.data.end(), [&](const TokenData &t) { return -log2f(t.p) > *mu; });
    c.data.resize((size_t)std::distance(c.data.begin(), cut));
    if (c.data.empty()) { /* keep at least one, as later llama.cpp revisions do; unreachable for sane mu */ return 0; }
    softmax(c);
    const int X = token(c);
    const size_t xi = std::distance(c.data.begin(), std::find_if(c.data.begin(), c.data.end(), [&](const TokenData &t) { return t.id == X; }));
    const float observed = -log2f(c.data[xi].p);
    *mu = *mu - eta * (observed - tau);
    return X;
}

int Sampler::sample(const float *logits, int n_vocab, const SampleParams &p) {
    Candidates c; c.data.reserve((size_t)n_vocab);
    for (int i = 0; i < n_vocab; i++) c.data.push_back({i, logits[i], 0.0f});
    const int top_k_eff = p.top_k <= 0 ? n_vocab : p.top_k;   // minigpt4.cpp:2428
    if (p.temp <= 0) return greedy(c);
    if (p.mirostat == 1) {
        if (!mu1_init) { mu1 = 2.0f * p.mirostat_tau; mu1_init = true; }
        temperature(c, p.temp);
        return mirostat_v1(c, p.mirostat_tau, p.mirostat_eta, 100, &mu1);
    }
    if (p.mirostat == 2) {
        if (!mu2_init) { mu2 = 2.0f * p.mirostat_tau; mu2_init = true; }
        temperature(c, p.temp);
        return mirostat_v2(c, p.mirostat_tau, p.mirostat_eta, &mu2);
    }
    top_k(c, top_k_eff, 1);
    tail_free(c, p.tfs_z, 1);
    typical(c, p.typical_p, 1);
    top_p(c, p.top_p, 1);
    temperature(c, p.temp);
    return token(c);
}

}  // namespace mg4
