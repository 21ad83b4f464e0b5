#include "engine.hpp"
#include "dist.hpp"
#include "quantize.hpp"
#include "imageio.hpp"

#include <algorithm>
#include <optional>
#include <unistd.h>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstring>

namespace mg4 {

// ====================================================================================================================
// device helpers
// ====================================================================================================================
int device_count_noexcept() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
void DeviceArena::alloc(size_t bytes) {
    release();
    if (virt) base = reinterpret_cast<uint8_t *>((uintptr_t)1 << 40);      // never dereferenced
    // alignment gaps / plane padding are part of what arena checksums cover: make them deterministic
    else { HIP_CHECK(hipMalloc((void **)&base, bytes)); HIP_CHECK(hipMemset(base, 0, bytes)); HIP_CHECK(hipDeviceSynchronize()); }
    cap = bytes; used = 0; layout_hash = 1469598103934665603ull;
}
void DeviceArena::release() { if (base && !virt) HIP_IGNORE(hipFree(base)); base = nullptr; cap = used = 0; }
uint8_t *DeviceArena::take(size_t bytes, size_t align) {
    const size_t off = (used + align - 1) / align * align;
    if (off + bytes > cap) throw HipError{hipErrorOutOfMemory, "arena overflow", __FILE__, __LINE__};
    used = off + bytes;
    for (uint64_t v : {(uint64_t)off, (uint64_t)bytes}) for (int i = 0; i < 8; i++) { layout_hash ^= (v >> (8 * i)) & 0xFF; layout_hash *= 1099511628211ull; }
    return base + off;
}
template <typename T> T *Engine::upload_raw(DeviceArena &a, const void *src, size_t bytes) {
    uint8_t *d = a.take(bytes);
    if (moves_data()) HIP_CHECK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    return reinterpret_cast<T *>(d);
}

// ====================================================================================================================
// image preprocess (standalone: needs a device, not a loaded model)
// ====================================================================================================================
void preprocess_image_device(hipStream_t s, const uint8_t *rgb, int w, int h, float *out_chw) {
    constexpr int OUT = 224;                                                   // IMAGE_RESIZE (reference minigpt4.cpp:2620)
    const float mean[3] = {(float)0.48145466, (float)0.4578275, (float)0.40821073}, sd[3] = {(float)0.26862954, (float)0.26130258, (float)0.27577711};   // :2621-2622
    // Pillow skips a pass that keeps the size; an identity table (one tap of 2^22) is the same thing: (2^21 + p * 2^22) >> 22 == p
    auto table = [&](int in, ResampleCoeffs &c) {
        if (in != OUT) { precompute_bicubic_8bpc(in, OUT, c); return; }
        c.in_size = c.out_size = OUT; c.ksize = 1; c.first.resize(OUT); c.count.assign(OUT, 1); c.kk.assign(OUT, 1 << 22);
        for (int i = 0; i < OUT; i++) c.first[(size_t)i] = i;
    };
    ResampleCoeffs ch, cv;
    table(w, ch); table(h, cv);
    struct Dev { void *p = nullptr; ~Dev() { if (p) HIP_IGNORE(hipFree(p)); } };
    Dev d_src, d_tmp, d_out, d_tab;
    const size_t src_bytes = (size_t)w * h * 3, tmp_bytes = (size_t)h * OUT * 3, out_bytes = (size_t)3 * OUT * OUT * 4;
    std::vector<int> tab;                                                      // [first_h | count_h | kk_h | first_v | count_v | kk_v]
    tab.insert(tab.end(), ch.first.begin(), ch.first.end()); tab.insert(tab.end(), ch.count.begin(), ch.count.end()); tab.insert(tab.end(), ch.kk.begin(), ch.kk.end());
    const size_t voff = tab.size();
    tab.insert(tab.end(), cv.first.begin(), cv.first.end()); tab.insert(tab.end(), cv.count.begin(), cv.count.end()); tab.insert(tab.end(), cv.kk.begin(), cv.kk.end());
    HIP_CHECK(hipMalloc(&d_src.p, src_bytes)); HIP_CHECK(hipMalloc(&d_tmp.p, tmp_bytes)); HIP_CHECK(hipMalloc(&d_out.p, out_bytes)); HIP_CHECK(hipMalloc(&d_tab.p, tab.size() * 4));
    HIP_CHECK(hipMemcpyAsync(d_src.p, rgb, src_bytes, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    const int *t = static_cast<const int *>(d_tab.p);
    launch_resample_h(static_cast<const uint8_t *>(d_src.p), w, h, t, t + OUT, t + 2 * OUT, ch.ksize, static_cast<uint8_t *>(d_tmp.p), OUT, s);
    launch_resample_v_norm(static_cast<const uint8_t *>(d_tmp.p), OUT, t + voff, t + voff + OUT, t + voff + 2 * OUT, cv.ksize, static_cast<float *>(d_out.p), OUT, mean, sd, s);
    HIP_CHECK(hipMemcpyAsync(out_chw, d_out.p, out_bytes, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
}

void Engine::release_buffers() {
    for (Conversation &c : conv_) { if (c.graph) HIP_IGNORE(hipGraphExecDestroy(c.graph)); c.graph = nullptr; }
    for (hipGraphExec_t &g : batch_graph_) { if (g) HIP_IGNORE(hipGraphExecDestroy(g)); g = nullptr; }
    if (h_argmax_) HIP_IGNORE(hipHostFree(h_argmax_));
    if (h_bstage_) HIP_IGNORE(hipHostFree(h_bstage_));
    if (h_logits_) HIP_IGNORE(hipHostFree(h_logits_));
    h_argmax_ = nullptr; h_bstage_ = nullptr; h_logits_ = nullptr; logits_host_slot_ = -1;
    buf_arena_.release();
}
Engine::~Engine() {
    if (trace_file_) { fclose(trace_file_); trace_file_ = nullptr; }
    if (stream_hung_) {   // see engine.hpp: leak the device side, touch nothing that synchronises
        llm_arena_.abandon(); vis_arena_.abandon(); buf_arena_.abandon(); ri_arena_.abandon(); vgen_arena_.abandon();
        return;
    }
    if (stream_) HIP_IGNORE(hipStreamSynchronize(stream_));
    for (auto &e : site_events_) { HIP_IGNORE(hipEventDestroy(e.a)); HIP_IGNORE(hipEventDestroy(e.b)); }
    if (stage_) HIP_IGNORE(hipFree(stage_));
    release_buffers();
    if (stream_) HIP_IGNORE(hipStreamDestroy(stream_));
}
void Engine::sync() { (void)flush(); HIP_CHECK(hipStreamSynchronize(stream_)); }

// ====================================================================================================================
// init / load
// ====================================================================================================================
int Engine::init(const std::string &vision_path, const std::string &llm_path, int seed, int n_ctx, int n_batch) {
    const int ndev = device_count_noexcept();
    if (ndev <= 0) { set_last_error("no HIP device visible: the MI355X engine has no CPU fallback"); MG4_ERR("%s", last_error().c_str()); return E_LoadLanguageModel; }
    // native multi-GPU load (dist.hpp): MINIGPT4_WORLD_SIZE / MINIGPT4_RANK / MINIGPT4_NCCL_ID_FILE -> rank 0 reads the files, every other rank loads
    // headers only and receives both weight arenas by ncclBroadcast inside this call.  Parsed BEFORE the device is chosen: the stream, the arenas and
    // the communicator must all be created on the device the rank ends up on (round-5 advisor: the stream used to be created on device 0 first).
    DistEnv dist; { std::string derr; if (parse_dist_env(dist, derr)) { set_last_error(derr); MG4_ERR("%s", derr.c_str()); return E_LoadLanguageModel; } }
    const char *dv = getenv("MINIGPT4_DEVICE");
    if (!dv) dv = getenv("LOCAL_RANK");
    device_ = dv ? atoi(dv) % ndev : (dist.active() && dist.world > 1 ? dist.rank % ndev : 0);
    if (!dv && dist.active() && dist.world > 1) MG4_INFO("no MINIGPT4_DEVICE / LOCAL_RANK: rank %d takes device %d", dist.rank, device_);
    HIP_CHECK(hipSetDevice(device_));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device_));
    MG4_INFO("device %d: %s (%s), %d CUs, %.1f GiB", device_, prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.totalGlobalMem / 1073741824.0);
    // The library is built for gfx950 only, and its in-launch hand-offs (K-split tickets of k_matvec_ri, the key-split attention's arrival counters) rely on that part's cache
    // behaviour -- write-through agent-scope stores + vmcnt(0) before the ticket, cache-bypassing agent-scope loads after it, no fences (MI355X_MICROARCH.md "Valid forms"): refuse
    // any other device by name instead of failing at the first launch (or, worse, not failing)
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error(std::string("device ") + prop.gcnArchName + " is not gfx950 (MI355X): this library carries gfx950 code objects only"); MG4_ERR("%s", last_error().c_str());
        return E_LoadLanguageModel;
    }
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    n_ctx_ = n_ctx > 0 ? n_ctx : 2048;
    n_batch_ = n_batch > 0 ? n_batch : 512;
    max_rows_ = std::max(n_batch_, 32);
    // Run-time switches read here, once (no launcher calls getenv).  What is left after round 4's pruning: device / conversation count / load mode /
    // parity mode, the graph switch the profiling tools use, and the five arms the bit-identity tests compare against (tests/test_gpu_parity.py,
    // tests/test_gpu_batch.py).  Tile shapes, K splits and the other experiment parameters are reachable only through the test library's setters
    // (include/minigpt4_amd_test.h).
    use_graph_ = !(getenv("MINIGPT4_NO_GRAPH") && atoi(getenv("MINIGPT4_NO_GRAPH")));
    max_chunk_ = max_rows_;
    if (getenv("MINIGPT4_ATTN_PREFILL_W8")) set_attn_prefill_w8(atoi(getenv("MINIGPT4_ATTN_PREFILL_W8")));   // 0: prompt attention without the 8-wave loader / MFMA form
    if (const char *dc = getenv("MINIGPT4_DEFER_COMBINE")) defer_combine_ = atoi(dc) != 0;                    // 0: every K-split combine as its own launch
    // bit mask: 1 w1|w3 + silu*mul in one launch, 4 the attention kernel stores fp16 rows for wo
    if (const char *fp = getenv("MINIGPT4_F16_PAIR")) f16_pair_ = atoi(fp);
    // 0: wq|wk and wv of a mixed-type layer as two launches (batched step)
    batch_mix_ = !(getenv("MINIGPT4_BATCH_MIX") && atoi(getenv("MINIGPT4_BATCH_MIX")) == 0);
    if (const char *e = getenv("MINIGPT4_ATTN_SPLIT_T")) attn_split_t_ = atoi(e);   // cached keys from which the decode step uses the key-split attention (0 = never)
    n_cus_ = prop.multiProcessorCount;
    // 1: Q4_K / Q5_K prompt rows on the fp16-MFMA form with scaled operands (k_mmqh_q45k: measured SLOWER,
    // profiles/r05_prefill_fp16_scaled_operands.md; A/B)
    if (const char *e = getenv("MINIGPT4_MMQH")) set_mmqh(atoi(e));
    // 0: the decode step gathers exp / SiLU from ggml's fp16 tables like rounds 1-4 (A/B)
    if (const char *e = getenv("MINIGPT4_COMPUTED_TABLES")) computed_tables_ = atoi(e) != 0;
    // 1: the MFMA batched launches norm + quantise their rows themselves (measured SLOWER: 864 vs 987 tok/s at B = 4; A/B)
    if (const char *e = getenv("MINIGPT4_RI_FUSE")) ri_fuse_ = atoi(e) != 0;
    // 0: w2 of the 4-conversation step on the v_dot4 launch instead of the K-split MFMA launch (A/B: profiles/r05_batched_decode_inengine.log)
    if (const char *e = getenv("MINIGPT4_RI_W2")) ri_w2_ = atoi(e) != 0;
    // 1: wo of the 3- / 4-conversation step on the MFMA launch with the plain quantisation of the attention rows inside it (round 6, A/B)
    if (const char *e = getenv("MINIGPT4_RI_WO")) ri_wo_ = atoi(e) != 0;
    // 0: batched decode on the v_dot4 multi-row mat-vec (rounds 2-4), no row-interleaved image
    if (const char *e = getenv("MINIGPT4_RI")) use_ri_ = atoi(e) != 0;
    set_ri_cus(prop.multiProcessorCount);
    // 0: the Q-Former's image-independent head is recomputed per encode (round-4 form, A/B)
    if (const char *e = getenv("MINIGPT4_QF_FOLD")) qf_fold_ = atoi(e) != 0;
    // query tiles per workgroup of the ViT / Q-Former attention (k_attn_vit; unset / 0 = chosen per launch; 1 = the rounds 2-5 form; A/B, bit-identical)
    if (const char *e = getenv("MINIGPT4_ATTN_QT")) set_attn_vit_qt(atoi(e));
    // 0: split-K GEMM slices in grid.z, every XCD walks every slice (rounds 3-5; A/B, bit-identical)
    if (const char *e = getenv("MINIGPT4_SPLITK_XCD")) set_gemm_splitk_xcd(atoi(e));
    // K slices of the ViT's attention projection (1 = whole K + standalone LayerNorm: the default since round 3; 2..: split-K + the reduce that also normalises; A/B)
    if (const char *e = getenv("MINIGPT4_SPLITK_PROJ")) splitk_proj_ = std::max(1, std::min(SPLITK_MAX, atoi(e)));
    if (const char *e = getenv("MINIGPT4_SPLITK_FC2")) splitk_fc2_ = std::max(1, std::min(SPLITK_MAX, atoi(e)));     // K slices of the ViT's fc2 (default 4; A/B)
    // decode step: which row preparations ride in their consumer's prologue (bit 0 qkv, 1 wo, 2 w1|w3, 3 w2 <- SiLU * mul + quantisation, 4 output, 5 pair, 6 mixed qkv; default 87; A/B)
    if (const char *e = getenv("MINIGPT4_FUSE")) fuse_mask_ = atoi(e);
    if (const char *e = getenv("MINIGPT4_F16_KS")) set_gemm_tuning(-1, atoi(e));     // forced K split of the F16 language model's single-matrix prompt launches (wo, w2; 0 = choose; A/B)
    if (const char *e = getenv("MINIGPT4_PAIR_SILU_COMPUTED")) pair_silu_computed_ = atoi(e) != 0;   // 0: the F16 model's w1 | w3 pair epilogue gathers SiLU from the fp16 table (A/B)
    if (const char *e = getenv("MINIGPT4_COMPUTED_GELU")) computed_gelu_ = atoi(e) != 0;   // 0: the vision GEMMs' GELU epilogues gather from the fp16 table (A/B)
    if (const char *e = getenv("MINIGPT4_QKV_HEAD_MAJOR")) qkv_head_major_ = atoi(e) != 0;
    if (const char *e = getenv("MINIGPT4_QF_SPLITK")) qf_splitk_ = atoi(e) != 0;     // 0: the Q-Former's dense / output layers as whole-K launches + standalone LayerNorm (A/B)
    if (const char *e = getenv("MINIGPT4_KV_HOIST")) kv_hoist_ = atoi(e) != 0;     // 0: one K | V projection GEMM per cross-attention layer (round-4 form, A/B)
    set_matvec_tuning(0, 0, prop.multiProcessorCount);
    if (const char *ns = getenv("MINIGPT4_CONVERSATIONS")) conv_.assign((size_t)std::max(1, std::min(MAX_CONVERSATIONS, atoi(ns))), Conversation{});
    sampler_.seed(seed);
    if (const char *tf = getenv("MINIGPT4_PARITY_TRACE")) { if (*tf) trace_file_ = fopen(tf, "wb"); }
    // oracle-order fp32 accumulation (forward_ref): bit-identical to the CPU oracle, slow
    parity_ = trace_file_ || (getenv("MINIGPT4_PARITY") && atoi(getenv("MINIGPT4_PARITY")));
    if (const char *lm = getenv("MINIGPT4_LOAD")) load_mode_ = !strcmp(lm, "recv") ? LOAD_RECV : LOAD_FULL;
    // the exchange decides who reads the files: MINIGPT4_LOAD=recv on rank 0 would broadcast empty arenas
    if (dist.active()) load_mode_ = dist.rank != 0 ? LOAD_RECV : LOAD_FULL;
    // With the native exchange active a load error on THIS rank is not returned at once: the rank still joins the communicator and reports it there,
    // so that its peers fail with it instead of waiting for it inside RCCL (native_broadcast).
    int load_err = E_None;
    auto t0 = std::chrono::steady_clock::now();
    try {
        load_err = load_llm(llm_path);
        auto t1 = std::chrono::steady_clock::now();
        if (!load_err) MG4_INFO("LLM model init took %lld ms to complete", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count());
        if (!load_err) load_err = load_vision(vision_path);
        auto t2 = std::chrono::steady_clock::now();
        if (!load_err) MG4_INFO("Loading minigpt4 model took %lld ms to complete", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count());
        // LOAD_RECV: weights_received() derives both.  MINIGPT4_CONVERSATIONS > 1 sizes the context like set_conversations(n) does, row-interleaved image included
        if (!load_err) { alloc_buffers(); if (load_mode_ == LOAD_FULL) { fold_qformer_constants(); if (conv_.size() > 1) build_ri_planes(); } }
    } catch (const HipError &e) {
        if (!dist.active()) throw;
        set_last_error(std::string("load failed: ") + e.what + " (" + hipGetErrorString(e.code) + ")"); load_err = E_LoadLanguageModel;
    }
    if (stage_) { HIP_IGNORE(hipFree(stage_)); stage_ = nullptr; stage_cap_ = 0; }
    if (dist.active()) { if (int e = native_broadcast(dist.world, dist.rank, dist.id_file, dist.timeout_s, load_err)) return load_err ? load_err : e; }
    return load_err;
}

// The load-time exchange of a node's replicas, inside the C library: unique id through a file, communicator, layout agreement, both arenas from rank
// 0 in <= 1 GiB pieces, checksum agreement.  Any failure is an error of minigpt4_model_load (message in minigpt4_amd_last_error); there is no
// fallback to reading the files.
int Engine::native_broadcast(int world, int rank, const std::string &id_file, int timeout_s, int local_err) {
    std::string err;
    // round 5 (advisor): EVERY rank takes part in the same sequence of collectives and learns the same verdict, so a failure anywhere fails the load
    // everywhere instead of leaving the healthy ranks inside a collective for ever: (1) a rank whose own load failed (header error, unsupported
    // tensor) still joins and says so; (2) agreement is a symmetric all-reduce (element-wise max of {x, ~x} = max and min of every word, plus an "I
    // am fine" flag), not a one-way broadcast only the receivers compare; (3) ncclCommInitRank runs under a watchdog (dist.cpp) and the stream waits
    // below are bounded by MINIGPT4_DIST_TIMEOUT_S.
    auto fail = [&](const std::string &what) {
        if (rank == 0) unlink(id_file.c_str());        // never leave an id behind that a later job could pick up
        set_last_error("weight broadcast (rank " + std::to_string(rank) + " of " + std::to_string(world) + "): " + what); MG4_ERR("%s", last_error().c_str()); return (int)E_LoadLanguageModel; };
    // the Rccl object outlives a timed-out bootstrap on purpose (its helper thread is still inside the library): heap-allocated, leaked when poisoned
    struct Holder { Rccl *r = new Rccl; ~Holder() { if (r && !r->poisoned()) delete r; } } holder;
    Rccl &rccl = *holder.r;
    if (rccl.open(err)) return fail(err);
    uint8_t id[128];
    if (rank == 0) { unlink(id_file.c_str()); if (rccl.unique_id(id, err) || publish_unique_id(id_file, id, err)) return fail(err); }
    else if (await_unique_id(id_file, id, timeout_s, err, process_start_epoch_s())) return fail(err);
    if (rccl.init(world, rank, id, timeout_s, err)) return fail(err);
    if (rank == 0) unlink(id_file.c_str());            // every rank has joined: a later job must not pick this id up
    const auto t0 = std::chrono::steady_clock::now();
    auto wait_stream = [&](const char *what) -> bool {   // bounded hipStreamSynchronize: a peer that died inside a collective must not hang this rank
        const auto w0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(stream_);
            if (q == hipSuccess) return true;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); err = std::string(what) + ": " + hipGetErrorString(q); return false; }
            if (std::chrono::steady_clock::now() - w0 > std::chrono::seconds(timeout_s)) {
                // the collective (and the copies queued behind it) are STILL on the stream: nothing they touch may be freed, and the communicator must not be
                // destroyed under them -- poison the Rccl object (communicator and library leaked), leak the word buffers (Words below), fail the load
                rccl.poison(); stream_hung_ = true;
                err = std::string(what) + " did not complete within " + std::to_string(timeout_s) + " s (a peer left the exchange?)"; return false;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    };
    // device words of the agreements + a PINNED host image of them (the device-to-host copy behind a hung collective may land long after this frame is gone: never a
    // stack array).  Both are leaked when the exchange was poisoned: hipFree would synchronise with the hung stream.
    struct Words { Rccl &r; unsigned long long *d = nullptr, *h = nullptr; ~Words() { if (r.poisoned()) return; if (d) HIP_IGNORE(hipFree(d)); if (h) HIP_IGNORE(hipHostFree(h)); } } words{rccl};
    HIP_CHECK(hipMalloc((void **)&words.d, 128));
    HIP_CHECK(hipHostMalloc((void **)&words.h, 256, hipHostMallocDefault));
    unsigned long long *const d_words = words.d;
    // Symmetric agreement: true on EVERY rank iff every rank passed ok and all ranks hold the same four words; otherwise false on every rank, with a
    // message naming the cause.
    auto agree = [&](const unsigned long long mine[4], bool ok_here, const char *what) -> bool {
        unsigned long long *const v = words.h, *const r = words.h + 16;
        for (int i = 0; i < 4; i++) { v[i] = mine[i]; v[4 + i] = ~mine[i]; }
        v[8] = ok_here ? 0ull : 1ull;                    // max over ranks: 1 = somebody failed before this point
        v[9] = ok_here ? 0ull : (unsigned long long)(rank + 1);   // ... and (one of) who
        HIP_CHECK(hipMemcpyAsync(d_words, v, 80, hipMemcpyHostToDevice, stream_));
        if (rccl.allreduce_max_u64(d_words, 10, stream_, err)) return false;
        HIP_CHECK(hipMemcpyAsync(r, d_words, 80, hipMemcpyDeviceToHost, stream_));
        if (!wait_stream(what)) return false;
        if (r[8]) { err = std::string("rank ") + std::to_string((long long)r[9] - 1) + " failed before \"" + what + "\"" + (ok_here ? "" : " (this rank: " + last_error() + ")"); return false; }
        for (int i = 0; i < 4; i++) if (r[i] != ~r[4 + i]) {   // max != min: the ranks disagree
            char b[256]; snprintf(b, sizeof b, "%s differ between the ranks (word %d: max %llx, min %llx; this rank %llx)", what, i, r[i], ~r[4 + i], mine[i]); err = b; return false; }
        return true;
    };
    // test injection (tests/test_gpu_serve.py::test_native_broadcast_stream_timeout_leaks_and_returns): a bounded busy kernel in front of the first collective stands in
    // for a collective whose peer died -- the bounded wait below must give up, poison the exchange and return without touching the stream again
    if (const char *st = getenv("MINIGPT4_DIST_TEST_STALL_MS")) launch_stall((unsigned)std::max(0, atoi(st)), stream_);
    const ArenaPlan pl = arena_plan();
    const unsigned long long plan_words[4] = {(unsigned long long)pl.llm_bytes, (unsigned long long)pl.vision_bytes, (unsigned long long)pl.llm_hash, (unsigned long long)pl.vision_hash};
    if (!agree(plan_words, local_err == 0, "arena layouts (sizes / layout hashes)")) return fail(err);
    if (rccl.broadcast(llm_arena_.base, llm_arena_.used, 0, stream_, err) || rccl.broadcast(vis_arena_.base, vis_arena_.used, 0, stream_, err)) return fail(err);
    if (!wait_stream("the arena broadcast")) return fail(err);
    dist_bcast_ms_ = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    bool derived_ok = true;
    try { if (load_mode_ == LOAD_RECV) weights_received(); } catch (const HipError &e) { derived_ok = false; set_last_error(std::string("weights_received: ") + e.what); }
    unsigned long long sums[4] = {0ull, 0ull, 0ull, 0ull};
    if (derived_ok) { sums[0] = device_checksum(llm_arena_.base, llm_arena_.used, stream_); sums[1] = device_checksum(vis_arena_.base, vis_arena_.used, stream_); }
    if (!agree(sums, derived_ok, "arena contents (checksums) after the broadcast")) return fail(err);
    dist_world_ = world; dist_rank_ = rank;
    MG4_INFO("rank %d of %d: weight arenas %s in %.1f ms (%.2f GB, RCCL)", rank, world, rank ? "received" : "broadcast", dist_bcast_ms_, (llm_arena_.used + vis_arena_.used) / 1e9);
    return E_None;
}

// LOAD_RECV: the two weight arenas have been filled from rank 0 (broadcast): everything derived from them on the device
int Engine::weights_received() {
    if (load_mode_ != LOAD_RECV) return 0;
    load_mode_ = LOAD_FULL;
    const size_t NQ = (size_t)v_nq_;
    for (size_t b = 0; b < (size_t)VISION_BATCH_MAX; b++) HIP_CHECK(hipMemcpy(vi_qtok_rep_ + b * NQ * 768, v_qtok_, NQ * 768 * 4, hipMemcpyDeviceToDevice));
    fold_qformer_constants();
    if (conv_.size() > 1) build_ri_planes();               // set_conversations(n > 1) ran while the arenas were still empty
    return 0;
}
// Host only: the arena layout (sizes + layout hashes) the two files produce, without a device: what a receiving rank must reproduce
// (tests/test_cpu_dist.py).
int Engine::plan_arenas(const std::string &vision_path, const std::string &llm_path, ArenaPlan &out) {
    Engine e;
    e.load_mode_ = LOAD_PLAN;
    e.llm_arena_.virt = e.vis_arena_.virt = true;
    if (int err = e.load_llm(llm_path)) return err;
    if (int err = e.load_vision(vision_path)) return err;
    out = e.arena_plan();
    return 0;
}

// Tensor types the kernels do not stream natively but that have an exact image in one they do: Q3_K -> Q6_K (quantize.hpp).  The conversion runs on
// the host at load.
static int effective_type(int t) { return t == GT_Q3_K ? GT_Q6_K : t; }
static bool tensor_type_loadable(int t) { return qweight_supported(effective_type(t)); }
// bytes of a tensor as the GPU holds it
static size_t effective_nbytes(const TensorMeta &t) { return t.type == GT_Q3_K ? t.nbytes / 110 * 210 : t.nbytes; }

void Engine::upload_qweight(const TensorMeta &t0, const uint8_t *file_base0, QWeight &w) {
    TensorMeta t = t0;
    const uint8_t *file_base = file_base0;
    std::vector<uint8_t> conv;
    if (t0.type == GT_Q3_K) {
        conv.resize(moves_data() ? effective_nbytes(t0) : 0);
        if (moves_data()) q3k_to_q6k(file_base0 + t0.offset, conv.data(), t0.nbytes / 110);
        t.type = GT_Q6_K; t.nbytes = effective_nbytes(t0); t.offset = 0; file_base = conv.data();
    }
    const int cols = (int)t.ne[0], rows = (int)t.ne[1];
    QWeight plan;
    const size_t need = plan_qweight(t.type, rows, cols, plan, nullptr);
    uint8_t *base = llm_arena_.take(need);
    plan_qweight(t.type, rows, cols, w, base);
    if (!moves_data()) return;                                              // LOAD_RECV / LOAD_PLAN: layout only
    if (t.type == GT_F16 || t.type == GT_F32) { HIP_CHECK(hipMemcpy(base, file_base + t.offset, t.nbytes, hipMemcpyHostToDevice)); return; }
    if (t.nbytes > stage_cap_) throw HipError{hipErrorOutOfMemory, "staging buffer too small", __FILE__, __LINE__};
    HIP_CHECK(hipMemcpy(stage_, file_base + t.offset, t.nbytes, hipMemcpyHostToDevice));
    launch_repack(stage_, w, stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));
}

int Engine::load_llm(const std::string &path) {
    if (int e = llm_.load(path)) { MG4_ERR("failed to parse %s: %s", path.c_str(), last_error().c_str()); return e; }
    tok_.init(llm_);
    const int E = (int)llm_.n_embd, L = (int)llm_.n_layer, V = (int)llm_.n_vocab, F = (int)llm_.n_ff();
    const int hd = E / (int)llm_.n_head;
    if (!attn_head_size_supported(hd)) { set_last_error("unsupported head size (supported: 32, 64, 128)"); return E_LoadLanguageModel; }
    // k_attn_llm keeps one fp32 score + one fp16 probability per key of the context in LDS (6 bytes per key, 160 KiB per workgroup): refuse what
    // cannot launch
    if (parity_) attn_ref_prepare();
    auto ctx_too_long = [&](const char *which, int lim) {
        set_last_error(std::string(which) + "n_ctx " + std::to_string(n_ctx_) + " exceeds what the attention kernel's LDS score rows hold (" + std::to_string(lim) + ")");
        MG4_ERR("%s", last_error().c_str());
        return (int)E_LoadLanguageModel;
    };
    if (parity_ && n_ctx_ > attn_ref_max_ctx(hd)) return ctx_too_long("MINIGPT4_PARITY (oracle-order kernel): ", attn_ref_max_ctx(hd));
    if (n_ctx_ > attn_max_ctx(hd)) return ctx_too_long("", attn_max_ctx(hd));
    MG4_INFO("llm: n_vocab %d n_embd %d n_head %u n_layer %d n_ff %d n_ctx %d", V, E, llm_.n_head, L, F, n_ctx_);
    auto need = [&](const std::string &name, int64_t ne0, int64_t ne1) -> const TensorMeta * {
        const TensorMeta *t = llm_.find(name);
        if (!t) { set_last_error("LLM file: missing tensor " + name); return nullptr; }
        if (t->ne[0] != ne0 || (ne1 ? (t->ne.size() != 2 || t->ne[1] != ne1) : t->ne.size() != 1)) { set_last_error("LLM file: bad shape for " + name); return nullptr; }
        if (ne1 == 0 && t->type != GT_F32) { set_last_error("LLM file: " + name + " must be f32"); return nullptr; }
        if (ne1 && !tensor_type_loadable(t->type)) { set_last_error(std::string("LLM file: tensor type ") + gt_name(t->type) + " of " + name + " is not supported by the gfx950 kernels"); return nullptr; }
        return t;
    };
    // pass 1: validate + size the arena
    size_t total = 0, max_raw = 0;
    QWeight tmp;
    std::vector<std::pair<std::string, std::pair<int64_t, int64_t>>> mats;
    mats.push_back({"output.weight", {E, V}});
    for (int i = 0; i < L; i++) {
        const std::string p = "layers." + std::to_string(i) + ".";
        for (const char *w : {"attention.wq.weight", "attention.wk.weight", "attention.wv.weight", "attention.wo.weight"}) mats.push_back({p + w, {E, E}});
        mats.push_back({p + "feed_forward.w1.weight", {E, F}});
        mats.push_back({p + "feed_forward.w2.weight", {F, E}});
        mats.push_back({p + "feed_forward.w3.weight", {E, F}});
    }
    wbytes_token_ = 0;
    for (auto &m : mats) {
        const TensorMeta *t = need(m.first, m.second.first, m.second.second);
        if (!t) { MG4_ERR("%s", last_error().c_str()); return E_LoadLanguageModel; }
        total += plan_qweight(effective_type(t->type), (int)t->ne[1], (int)t->ne[0], tmp, nullptr) + 256;
        if (t->type != GT_F16 && t->type != GT_F32) max_raw = std::max(max_raw, effective_nbytes(*t));
        wbytes_token_ += effective_nbytes(*t);                             // what a decoded token streams from HBM
    }
    const TensorMeta *tt = llm_.find("tok_embeddings.weight");
    if (!tt || tt->ne.size() != 2 || tt->ne[0] != E || tt->ne[1] != V || !tensor_type_loadable(tt->type)) { set_last_error("LLM file: bad tok_embeddings.weight"); MG4_ERR("%s", last_error().c_str()); return E_LoadLanguageModel; }
    total += effective_nbytes(*tt) + 256;
    wbytes_token_ += gt_nbytes(effective_type(tt->type), (size_t)E);
    for (int i = 0; i < L; i++) for (const char *n : {"attention_norm.weight", "ffn_norm.weight"}) if (!need("layers." + std::to_string(i) + "." + n, E, 0)) { MG4_ERR("%s", last_error().c_str()); return E_LoadLanguageModel; }
    if (!need("norm.weight", E, 0)) { MG4_ERR("%s", last_error().c_str()); return E_LoadLanguageModel; }
    total += (size_t)(2 * L + 1) * ((size_t)E * 4 + 256);
    wbytes_token_ += (size_t)(2 * L + 1) * E * 4;
    llm_arena_.alloc(total + 4096);
    if (max_raw && moves_data()) { HIP_CHECK(hipMalloc((void **)&stage_, max_raw)); stage_cap_ = max_raw; }
    // pass 2: upload + repack
    const uint8_t *fb = llm_.mf.data;
    upload_qweight(*llm_.find("output.weight"), fb, output_);
    layers_.resize((size_t)L);
    for (int i = 0; i < L; i++) {
        const std::string p = "layers." + std::to_string(i) + ".";
        LayerW &lw = layers_[(size_t)i];
        upload_qweight(*llm_.find(p + "attention.wq.weight"), fb, lw.wq);
        upload_qweight(*llm_.find(p + "attention.wk.weight"), fb, lw.wk);
        upload_qweight(*llm_.find(p + "attention.wv.weight"), fb, lw.wv);
        upload_qweight(*llm_.find(p + "attention.wo.weight"), fb, lw.wo);
        upload_qweight(*llm_.find(p + "feed_forward.w1.weight"), fb, lw.w1);
        upload_qweight(*llm_.find(p + "feed_forward.w2.weight"), fb, lw.w2);
        upload_qweight(*llm_.find(p + "feed_forward.w3.weight"), fb, lw.w3);
        const TensorMeta *an = llm_.find(p + "attention_norm.weight"), *fn = llm_.find(p + "ffn_norm.weight");
        lw.attn_norm = upload_raw<float>(llm_arena_, fb + an->offset, an->nbytes);
        lw.ffn_norm = upload_raw<float>(llm_arena_, fb + fn->offset, fn->nbytes);
    }
    const TensorMeta *nt = llm_.find("norm.weight");
    norm_ = upload_raw<float>(llm_arena_, fb + nt->offset, nt->nbytes);
    if (tt->type == GT_Q3_K) {   // the embedding gather dequantises raw ggml rows: give it the (value-identical) Q6_K rows
        std::vector<uint8_t> conv(moves_data() ? effective_nbytes(*tt) : 0);
        if (moves_data()) q3k_to_q6k(fb + tt->offset, conv.data(), tt->nbytes / 110);
        tok_raw_ = upload_raw<uint8_t>(llm_arena_, conv.data(), effective_nbytes(*tt));
    } else tok_raw_ = upload_raw<uint8_t>(llm_arena_, fb + tt->offset, tt->nbytes);
    tok_type_ = effective_type(tt->type);
    MG4_INFO("llm weights: %.1f MB in HBM, %.3f GB streamed per decoded token", llm_arena_.used / 1048576.0, wbytes_token_ / 1e9);
    return E_None;
}

int Engine::load_vision(const std::string &path) {
    if (int e = vis_.load(path)) { MG4_ERR("failed to parse %s (%s)", path.c_str(), last_error().c_str()); return e; }
    auto fail = [&](const std::string &msg, int code) { set_last_error("vision file: " + msg); MG4_ERR("%s", last_error().c_str()); return code; };
    const TensorMeta *pos = vis_.find("visual_encoder", "pos_embed");
    if (!pos || pos->ne.size() != 2 || pos->type != GT_F32) return fail("missing/bad visual_encoder.pos_embed", E_LoadModelFileHeader);
    v_D_ = (int)pos->ne[0];
    if (pos->ne[1] != 257 || v_D_ % 88 || v_D_ % 16) return fail("pos_embed must be [D,257] with D a multiple of 88 and 16 (head size 88 is hard-coded in the reference, minigpt4.cpp:1271)", E_LoadModelFileHeader);
    v_heads_ = v_D_ / 88;
    if (vis_.config_int("encoder_width", v_D_) != v_D_) return fail("config Qformer.encoder_width disagrees with pos_embed", E_LoadModelFileHeader);
    v_depth_ = 0;
    while (vis_.find("visual_encoder", "blocks." + std::to_string(v_depth_) + ".norm1.weight")) v_depth_++;
    if (!v_depth_) return fail("no ViT blocks", E_LoadModelFileHeader);
    const TensorMeta *fc1 = vis_.find("visual_encoder", "blocks.0.mlp.fc1.weight");
    if (!fc1 || fc1->ne.size() != 2) return fail("missing blocks.0.mlp.fc1.weight", E_LoadModelFileHeader);
    v_M_ = (int)fc1->ne[1];
    const TensorMeta *qt = vis_.find("query_tokens", "weight");
    if (!qt || qt->ne.size() != 2 || qt->ne[0] != 768 || qt->type != GT_F32) return fail("bad query_tokens.weight", E_LoadModelFileHeader);
    v_nq_ = (int)qt->ne[1];
    // reference PANICs (minigpt4.cpp:2230)
    if (vis_.config_int("query_length", v_nq_) != v_nq_) return fail("query_embeds_length != query_length", E_LoadModelFileHeader);
    {   // sizes read from an untrusted header feed container sizes below
        const long long ql = vis_.config_int("num_hidden_layers", 12);
        if (ql < 1 || ql > 64 || v_nq_ < 1 || v_nq_ > 256) return fail("num_hidden_layers / query_length out of range", E_LoadModelFileHeader);
        v_ql_ = (int)ql;
    }
    const TensorMeta *lp = vis_.find("llama_proj", "weight");
    if (!lp || lp->ne.size() != 2 || lp->ne[0] != 768) return fail("bad llama_proj.weight", E_LoadModelFileHeader);
    v_out_ = (int)lp->ne[1];
    const TensorMeta *iq = vis_.find("Qformer", "bert.encoder.layer.0.intermediate_query.dense.weight");
    if (!iq || iq->ne.size() != 2) return fail("missing intermediate_query", E_LoadModelFileHeader);
    v_qi_ = (int)iq->ne[1];
    if (v_M_ % 16 || v_qi_ % 16) return fail("MLP widths must be multiples of 16", E_LoadModelFileHeader);

    // Linear weights: all F16 (the reference's published f16 files) -> the MFMA f16 GEMM path below; anything else -> the generic path
    // (engine_vision_generic.cpp)
    v_generic_ = false;
    for (auto &m : vis_.models) for (auto &t : m.second) {
        const TensorMeta &tm = t.second;
        const bool is_lin = tm.ne.size() == 2 && tm.name.size() >= 6 && tm.name.compare(tm.name.size() - 6, 6, "weight") == 0 && m.first != "query_tokens" && tm.name != "pos_embed" &&
                            tm.name.find("orm") == std::string::npos && tm.name != "patch_embed.proj.weight" && tm.name.find("embeddings") == std::string::npos;
        if (is_lin && tm.type != GT_F16) v_generic_ = true;
    }
    size_t total = 0;
    for (auto &m : vis_.models) for (auto &t : m.second) total += effective_nbytes(t.second) + 512 + (v_generic_ ? 2048 : 0);
    total += (size_t)v_D_ * 592 * 2 + (size_t)v_depth_ * 3 * v_D_ * 4 + (1 << 20);
    vis_arena_.alloc(total);
    const uint8_t *fb = vis_.mf.data;
    bool ok = true; std::string bad;
    auto f32v = [&](const std::string &model, const std::string &name, int64_t n) -> float * {
        const TensorMeta *t = vis_.find(model, name);
        if (!t || t->type != GT_F32 || t->nelements() != n) { ok = false; bad = model + "." + name; return nullptr; }
        return upload_raw<float>(vis_arena_, fb + t->offset, t->nbytes);
    };
    auto f16m = [&](const std::string &model, const std::string &name, int64_t n_in, int64_t n_out) -> const TensorMeta * {
        const TensorMeta *t = vis_.find(model, name);
        if (!t || t->ne.size() < 2 || t->nelements() != n_in * n_out) { ok = false; bad = model + "." + name; return nullptr; }
        if (t->type != GT_F16 && !v_generic_) { ok = false; bad = model + "." + name + " (unexpected type " + gt_name(t->type) + ")"; return nullptr; }
        return t;
    };
    // generic mode: the Linear weights are uploaded as QWeights by load_vision_generic(); the fp16 pointers of this path stay null
    auto up16 = [&](const TensorMeta *t) -> __half * { return t && !v_generic_ ? upload_raw<__half>(vis_arena_, fb + t->offset, t->nbytes) : nullptr; };
    auto concat16 = [&](const std::vector<const TensorMeta *> &ts) -> __half * {
        if (v_generic_) return nullptr;
        size_t bytes = 0; for (auto t : ts) { if (!t) return nullptr; bytes += t->nbytes; }
        uint8_t *d = vis_arena_.take(bytes); size_t off = 0;
        for (auto t : ts) { if (moves_data()) HIP_CHECK(hipMemcpy(d + off, fb + t->offset, t->nbytes, hipMemcpyHostToDevice)); off += t->nbytes; }
        return reinterpret_cast<__half *>(d);
    };
    auto concat32 = [&](const std::vector<std::pair<const char *, std::string>> &names, int64_t each) -> float * {
        uint8_t *d = vis_arena_.take((size_t)each * 4 * names.size()); size_t off = 0;
        for (auto &nm : names) { const TensorMeta *t = vis_.find(nm.first, nm.second);
            if (!t || t->type != GT_F32 || t->nelements() != each) { ok = false; bad = nm.second; return nullptr; }
            if (moves_data()) HIP_CHECK(hipMemcpy(d + off, fb + t->offset, t->nbytes, hipMemcpyHostToDevice)); off += t->nbytes; }
        return reinterpret_cast<float *>(d);
    };
    const int D = v_D_, M = v_M_;
    const char *VE = "visual_encoder";
    v_cls_ = f32v(VE, "cls_token", D);
    v_pos_ = f32v(VE, "pos_embed", (int64_t)D * 257);
    v_patch_b_ = f32v(VE, "patch_embed.proj.bias", D);
    const TensorMeta *pw = f16m(VE, "patch_embed.proj.weight", 588, D);
    if (pw && pw->type != GT_F16) { ok = false; bad = "visual_encoder.patch_embed.proj.weight (the conv kernel is always F16: convert.py:113-117, and minigpt4_quantize_model skips it)"; pw = nullptr; }
    if (pw) {   // [D][588] -> zero-padded [D][592] (MFMA K multiple of 16)
        std::vector<uint16_t> padded((size_t)D * 592, 0);
        const uint16_t *src = reinterpret_cast<const uint16_t *>(fb + pw->offset);
        for (int r = 0; r < D; r++) memcpy(&padded[(size_t)r * 592], src + (size_t)r * 588, 588 * 2);
        v_patch_w_ = upload_raw<__half>(vis_arena_, padded.data(), padded.size() * 2);
    }
    vblocks_.resize((size_t)v_depth_);
    for (int i = 0; i < v_depth_ && ok; i++) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        VBlock &b = vblocks_[(size_t)i];
        b.n1w = f32v(VE, p + "norm1.weight", D); b.n1b = f32v(VE, p + "norm1.bias", D);
        b.n2w = f32v(VE, p + "norm2.weight", D); b.n2b = f32v(VE, p + "norm2.bias", D);
        // qkv bias = [q_bias, 0, v_bias] (minigpt4.cpp:1259-1262)
        const TensorMeta *qb = vis_.find(VE, p + "attn.q_bias"), *vb = vis_.find(VE, p + "attn.v_bias");
        if (!qb || !vb || qb->type != GT_F32 || vb->type != GT_F32 || qb->nelements() != D || vb->nelements() != D) { ok = false; bad = p + "attn.q_bias/v_bias"; break; }
        std::vector<float> bias3((size_t)3 * D, 0.0f);
        const float *qsrc = reinterpret_cast<const float *>(fb + qb->offset), *vsrc = reinterpret_cast<const float *>(fb + vb->offset);
        for (int j = 0; j < D; j++) { bias3[(size_t)j] = 0.0f + qsrc[j]; bias3[(size_t)2 * D + j] = 0.0f + vsrc[j]; }
        b.qkv_b = upload_raw<float>(vis_arena_, bias3.data(), bias3.size() * 4);
        b.qkv_w = up16(f16m(VE, p + "attn.qkv.weight", D, 3 * (int64_t)D));
        b.proj_w = up16(f16m(VE, p + "attn.proj.weight", D, D)); b.proj_b = f32v(VE, p + "attn.proj.bias", D);
        b.fc1_w = up16(f16m(VE, p + "mlp.fc1.weight", D, M)); b.fc1_b = f32v(VE, p + "mlp.fc1.bias", M);
        b.fc2_w = up16(f16m(VE, p + "mlp.fc2.weight", M, D)); b.fc2_b = f32v(VE, p + "mlp.fc2.bias", D);
    }
    v_lnv_w_ = f32v("ln_vision", "weight", D); v_lnv_b_ = f32v("ln_vision", "bias", D);
    v_qtok_ = f32v("query_tokens", "weight", (int64_t)768 * v_nq_);
    const char *QF = "Qformer";
    v_qeln_w_ = f32v(QF, "bert.embeddings.LayerNorm.weight", 768); v_qeln_b_ = f32v(QF, "bert.embeddings.LayerNorm.bias", 768);
    qlayers_.resize((size_t)v_ql_);
    // the K | V projections of every cross-attention layer as ONE contiguous block [n_cross * 1536][D] (+ biases): one GEMM per image batch instead
    // of one per cross layer
    {
        std::vector<const TensorMeta *> kvw; std::vector<std::pair<const char *, std::string>> kvb;
        v_ncross_ = 0;
        for (int i = 0; i < v_ql_; i++) {
            const std::string a = "bert.encoder.layer." + std::to_string(i) + ".crossattention.";
            if (!vis_.find(QF, a + "self.query.weight")) continue;
            kvw.push_back(f16m(QF, a + "self.key.weight", D, 768)); kvw.push_back(f16m(QF, a + "self.value.weight", D, 768));
            kvb.push_back({QF, a + "self.key.bias"}); kvb.push_back({QF, a + "self.value.bias"});
            qlayers_[(size_t)i].cross_idx = v_ncross_++;
        }
        if (v_ncross_ && ok) { v_kv_all_w_ = concat16(kvw); v_kv_all_b_ = concat32(kvb, 768); }
    }
    for (int i = 0; i < v_ql_ && ok; i++) {
        const std::string p = "bert.encoder.layer." + std::to_string(i) + ".";
        QLayer &L = qlayers_[(size_t)i];
        {   // self attention: Q,K,V share their input -> one [2304][768] GEMM
            const std::string a = p + "attention.";
            L.self.q_w = concat16({f16m(QF, a + "self.query.weight", 768, 768), f16m(QF, a + "self.key.weight", 768, 768), f16m(QF, a + "self.value.weight", 768, 768)});
            L.self.q_b = concat32({{QF, a + "self.query.bias"}, {QF, a + "self.key.bias"}, {QF, a + "self.value.bias"}}, 768);
            L.self.dense_w = up16(f16m(QF, a + "output.dense.weight", 768, 768)); L.self.dense_b = f32v(QF, a + "output.dense.bias", 768);
            L.self.ln_w = f32v(QF, a + "output.LayerNorm.weight", 768); L.self.ln_b = f32v(QF, a + "output.LayerNorm.bias", 768);
        }
        L.has_cross = vis_.find(QF, p + "crossattention.self.query.weight") != nullptr;   // detected by name (minigpt4.cpp:2008)
        if (L.has_cross) {
            const std::string a = p + "crossattention.";
            L.cross.q_w = up16(f16m(QF, a + "self.query.weight", 768, 768)); L.cross.q_b = f32v(QF, a + "self.query.bias", 768);
            L.cross.kv_w = v_kv_all_w_ ? v_kv_all_w_ + (size_t)L.cross_idx * 1536 * D : nullptr;          // this layer's [1536][D] slice of the shared block
            L.cross.kv_b = v_kv_all_b_ ? v_kv_all_b_ + (size_t)L.cross_idx * 1536 : nullptr;
            L.cross.dense_w = up16(f16m(QF, a + "output.dense.weight", 768, 768)); L.cross.dense_b = f32v(QF, a + "output.dense.bias", 768);
            L.cross.ln_w = f32v(QF, a + "output.LayerNorm.weight", 768); L.cross.ln_b = f32v(QF, a + "output.LayerNorm.bias", 768);
        }
        L.inter_w = up16(f16m(QF, p + "intermediate_query.dense.weight", 768, v_qi_)); L.inter_b = f32v(QF, p + "intermediate_query.dense.bias", v_qi_);
        L.out_w = up16(f16m(QF, p + "output_query.dense.weight", v_qi_, 768)); L.out_b = f32v(QF, p + "output_query.dense.bias", 768);
        L.oln_w = f32v(QF, p + "output_query.LayerNorm.weight", 768); L.oln_b = f32v(QF, p + "output_query.LayerNorm.bias", 768);
    }
    v_proj_w_ = up16(f16m("llama_proj", "weight", 768, v_out_)); v_proj_b_ = f32v("llama_proj", "bias", v_out_);
    if (!ok) return fail("missing or unsupported tensor " + bad, E_LoadModelFileHeader);
    if (v_generic_) {
        if (load_mode_ != LOAD_FULL) return fail("quantised / f32 vision files keep a third arena that the multi-GPU receive path does not cover: load them with LOAD_FULL on every rank", E_LoadModelFileHeader);
        if (int e = load_vision_generic()) return e;
    }
    MG4_INFO("vision weights: %.1f MB in HBM (ViT dim %d x %d blocks, Q-Former %d layers, proj -> %d)", vis_arena_.used / 1048576.0, v_D_, v_depth_, v_ql_, v_out_);
    return E_None;
}

void Engine::alloc_buffers() {
    const size_t S = conv_.size();
    max_rows_ = std::max(std::max(n_batch_, 32), (int)S);
    const size_t E = llm_.n_embd, F = llm_.n_ff(), V = llm_.n_vocab, L = llm_.n_layer, B = (size_t)max_rows_, C = (size_t)n_ctx_;
    const size_t Kmax = std::max(E, F), hd = E / llm_.n_head;
    const size_t D = (size_t)v_D_, M = (size_t)v_M_, NQ = (size_t)v_nq_;
    size_t total = 0;
    auto sz = [&](size_t b) { total += (b + 255) / 256 * 256 + 256; };
    sz(S * L * C * E * 2); sz(S * L * C * E * 2); sz(2 * C * (hd / 2) * 4 * 2); sz(3 * 65536 * 2);
    sz(5 * B * E * 4); sz(2 * B * F * 4); sz(S * V * 4); sz(S * V * 4); sz(8 * 256 + 2 * B * 4);
    sz(2 * B * Kmax); sz(B * Kmax / 256 * 4 + 64); sz(B * Kmax / 16 * 2 + 64); sz(B * Kmax / 16 + 64); sz(B * Kmax * 2); sz(B * Kmax / 8 + 128); sz(4 * (B * Kmax / 32 * 4 + 64)); sz(B * Kmax * 2); sz(B * Kmax * 4);
    sz(8192); sz((size_t)64 << 20); sz(attn_split_workspace_bytes((int)llm_.n_head, (int)hd, n_ctx_, std::max(attn_splits_forced_, attn_split_count((int)llm_.n_head, n_cus_))));
    const size_t VB = (size_t)VISION_BATCH_MAX;                            // images encoded in one pass (minigpt4_amd_encode_images)
    sz(VB * 3 * 224 * 224 * 4); sz(VB * 256 * 592 * 2); sz(VB * 256 * D * 4); sz(VB * 257 * D * 4); sz(VB * 257 * 3 * D * 4); sz(VB * 3 * 257 * D * 2); sz(VB * 257 * M * 2);
    sz(VB * (size_t)SPLITK_MAX * 257 * D * 4);
    sz(VB * 9 * NQ * 2304 * 4); sz(VB * 257 * 1536 * 4 * (size_t)std::max(1, v_ncross_)); sz(VB * NQ * 768 * 4); sz(VB * NQ * 768 * 4); sz(VB * NQ * 768 * 2);
    sz(VB * NQ * (size_t)std::max(v_qi_, 768) * 2 * 4); sz(VB * NQ * (size_t)v_out_ * 4); sz(1 << 20);
    buf_arena_.alloc(total);
    auto takef = [&](size_t n) { return reinterpret_cast<float *>(buf_arena_.take(n * 4)); };
    auto takeh = [&](size_t n) { return reinterpret_cast<__half *>(buf_arena_.take(n * 2)); };
    kc_ = takeh(S * L * C * E); vc_ = takeh(S * L * C * E);                // [conversation][layer][n_ctx][n_embd]
    HIP_CHECK(hipMemset(kc_, 0, S * L * C * E * 2)); HIP_CHECK(hipMemset(vc_, 0, S * L * C * E * 2));
    // RoPE table, exactly ggml's iteration: theta = pos; theta *= theta_scale per pair (fp32), cosf/sinf
    {
        std::vector<float> c(C * (hd / 2)), s(C * (hd / 2));
        const float theta_scale = powf(10000.0f, -2.0f / (float)hd);
        for (size_t p = 0; p < C; p++) { float theta = (float)p; for (size_t i = 0; i < hd / 2; i++) { c[p * (hd / 2) + i] = cosf(theta); s[p * (hd / 2) + i] = sinf(theta); theta *= theta_scale; } }
        cos_ = takef(c.size()); sin_ = takef(s.size());
        HIP_CHECK(hipMemcpy(cos_, c.data(), c.size() * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(sin_, s.data(), s.size() * 4, hipMemcpyHostToDevice));
    }
    // fp16 lookup tables (ggml_table_gelu_f16 / silu_f16 / exp_f16), evaluated with the host libm like ggml does at init
    {
        std::vector<__half> g(65536), si(65536), ex(65536);
        for (int i = 0; i < 65536; i++) {
            const float x = __half2float(__ushort_as_half((unsigned short)i));
            g[(size_t)i] = __float2half_rn(0.5f * x * (1.0f + tanhf(0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x))));
            si[(size_t)i] = __float2half_rn(x / (1.0f + expf(-x)));
            ex[(size_t)i] = __float2half_rn(expf(x));
        }
        __half *dg = takeh(65536), *ds = takeh(65536), *de = takeh(65536);
        HIP_CHECK(hipMemcpy(dg, g.data(), 131072, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(ds, si.data(), 131072, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(de, ex.data(), 131072, hipMemcpyHostToDevice));
        tabs_.gelu = dg; tabs_.silu = ds; tabs_.exp = de;

        int last = 0;
        for (int i = 0; i < 0x7C00; i++) if (__half2float(ex[(size_t)(0x8000 + i)]) != 0.0f) last = i;
        tabs_.exp_neg_n = (last + 1 + 2047) / 2048 * 2048;
        tabs_dec_ = tabs_;
        if (computed_tables_) { tabs_dec_.exp = nullptr; tabs_dec_.silu = nullptr; }
        tabs_vis_ = tabs_;
        if (computed_tables_) tabs_vis_.exp = nullptr;          // the ViT / Q-Former attention of fast mode computes its exponentials (no 40 KB table DMA per workgroup)
        if (computed_tables_ && computed_gelu_) tabs_vis_.gelu = nullptr;     // ... and the GELU epilogues of its GEMMs compute the table's values (round 6)
    }
    x_ = takef(B * E); q_ = takef(B * E); k_ = takef(B * E); v_ = takef(B * E); att_ = takef(B * E);
    h1_ = takef(B * F); h3_ = takef(B * F); logits_ = takef(S * V); blogits_ = takef(S * V);
    act_.q8k = reinterpret_cast<int8_t *>(buf_arena_.take(B * Kmax)); act_.q80 = reinterpret_cast<int8_t *>(buf_arena_.take(B * Kmax));
    act_.dk = takef(B * Kmax / 256 + 16); act_.bsk = reinterpret_cast<int16_t *>(buf_arena_.take(B * Kmax / 16 * 2 + 64));
    act_.bsq = reinterpret_cast<int8_t *>(buf_arena_.take(B * Kmax / 16 + 64));
    // MINIGPT4_MMQH=1 only: fp16 images of the Q8_K rows for the fp16-MFMA prompt mat-mul (k_mmqh_q45k, not adopted)
    if (mmqh_enabled()) { act_.q16 = takeh(B * Kmax); act_.bs16 = takeh(B * Kmax / 16 + 64); }
    act_.d0 = takef(B * Kmax / 32 + 16); act_.d1 = takef(B * Kmax / 32 + 16); act_.s1 = takef(B * Kmax / 32 + 16);
    act_.sum0 = reinterpret_cast<int *>(buf_arena_.take(B * Kmax / 32 * 4 + 64));
    act_.xh = takeh(B * Kmax); act_.xf = takef(B * Kmax);
    static_assert(MAX_CONVERSATIONS * 4 <= 256, "per-conversation scalars live in 256-byte slabs");
    d_npast_ = reinterpret_cast<int *>(buf_arena_.take(256)); d_argmax_ = reinterpret_cast<int *>(buf_arena_.take(256)); d_feed_ = reinterpret_cast<int *>(buf_arena_.take(256));
    d_btok_ = reinterpret_cast<int *>(buf_arena_.take(768)); d_bslot_ = d_btok_ + MAX_CONVERSATIONS; d_bpos_ = d_bslot_ + MAX_CONVERSATIONS;
    batch_graph_.assign((size_t)MAX_CONVERSATIONS + 1, nullptr);
    d_tokens_ = reinterpret_cast<int *>(buf_arena_.take(B * 4));
    d_scratch_ = buf_arena_.take(8192);
    attn_splits_ = attn_splits_forced_ > 0 ? attn_splits_forced_ : attn_split_count((int)llm_.n_head, n_cus_);
    attn_ws_ = buf_arena_.take(attn_split_workspace_bytes((int)llm_.n_head, (int)hd, n_ctx_, attn_splits_));
    HIP_CHECK(hipMemset(attn_ws_, 0, attn_split_workspace_bytes((int)llm_.n_head, (int)hd, n_ctx_, attn_splits_)));   // the arrival counters start (and are left) at zero
    {   // K-split partial sums of the prefill mat-mul (only prompts short enough to need the extra parallelism use them)
        const size_t slab_floats = (size_t)16 << 20;
        hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, device_));
        act_.ws = reinterpret_cast<float *>(buf_arena_.take(slab_floats * 4)); act_.ws_floats = slab_floats;
        set_mmq2_cus(prop.multiProcessorCount);
    }
    HIP_CHECK(hipMemset(d_npast_, 0, 256)); HIP_CHECK(hipMemset(d_argmax_, 0, 256)); HIP_CHECK(hipMemset(d_feed_, 0, 256)); HIP_CHECK(hipMemset(d_btok_, 0, 768));
    HIP_CHECK(hipMemset(d_tokens_, 0, B * 4));
    HIP_CHECK(hipHostMalloc((void **)&h_argmax_, 256, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void **)&h_bstage_, 768, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void **)&h_logits_, V * 4, hipHostMallocDefault));
    memset(h_argmax_, 0, 256);
    // vision
    vi_img_ = takef(VB * 3 * 224 * 224); vi_patches_ = takeh(VB * 256 * 592); vi_pe_ = takef(VB * 256 * D); vi_x_ = takef(VB * 257 * D); vi_qkv_ = takef(VB * 257 * 3 * D);
    vi_slab_ = takef(VB * (size_t)SPLITK_MAX * 257 * D);
    vi_ln_h_ = takeh(VB * 257 * D); vi_att_h_ = takeh(VB * 257 * D); vi_img_h_ = takeh(VB * 257 * D); vi_mlp_h_ = takeh(VB * 257 * M);
    vi_hs_ = takef(VB * NQ * 768); vi_a1_ = takef(VB * NQ * 768); vi_a2_ = takef(VB * NQ * 768); vi_d_ = takef(VB * NQ * 768); vi_qq_ = takef(VB * NQ * 2304); vi_kv_ = takef(VB * 257 * 1536 * (size_t)std::max(1, v_ncross_));
    vi_hs_h_ = takeh(VB * NQ * 768); vi_a1_h_ = takeh(VB * NQ * 768); vi_a2_h_ = takeh(VB * NQ * 768); vi_ctx_h_ = takeh(VB * NQ * 768); vi_im_h_ = takeh(VB * NQ * (size_t)v_qi_);
    vi_out_ = takef(VB * NQ * (size_t)v_out_);
    vi_qtok_rep_ = takef(VB * NQ * 768);                                  // the query tokens once per image of a batch (every image starts from the same rows)
    // LOAD_RECV: weights_received()
    if (load_mode_ == LOAD_FULL) for (size_t b = 0; b < VB; b++) HIP_CHECK(hipMemcpy(vi_qtok_rep_ + b * NQ * 768, v_qtok_, NQ * 768 * 4, hipMemcpyDeviceToDevice));
    vi_c_a1_ = takef(VB * NQ * 768); vi_c_qq_ = takef(VB * NQ * 768); vi_c_a1_h_ = takeh(VB * NQ * 768);
    if (v_generic_) alloc_vision_generic();
    MG4_INFO("KV cache %.1f MB (fp16, n_ctx %d, %zu conversation%s), activation arena %.1f MB", 2.0 * S * L * C * E * 2 / 1048576.0, n_ctx_, S, S == 1 ? "" : "s", buf_arena_.used / 1048576.0);
}

// ====================================================================================================================
// language path
// ====================================================================================================================
// One launch for 1..3 same-shape matrices when decoding (v2 persistent-wave kernel); otherwise one k_mul_mat launch per matrix.
// prep != null: the activation row still has to be prepared (rms_norm*w | identity | silu(a)*b + quantisation); when decoding it is fused into
// the mat-vec prologue, otherwise the standalone preparation kernel runs first.
void Engine::site_begin(const char *site, double bytes, hipStream_t s) {
    if (!prof_on_) return;
    SiteEv ev{}; ev.site = site; ev.bytes = bytes;
    HIP_CHECK(hipEventCreate(&ev.a)); HIP_CHECK(hipEventCreate(&ev.b));
    reset_kernel_name();
    ev.p0 = ev.p1 = launch_probe_count();
    HIP_CHECK(hipEventRecord(ev.a, s));
    site_events_.push_back(ev);
}
void Engine::site_end(hipStream_t s) noexcept {   // called from SiteScope's destructor: must not throw (a failed record marks the site invalid instead)
    if (!prof_on_ || site_events_.empty()) return;
    SiteEv &ev = site_events_.back();
    if (hipEventRecord(ev.b, s) != hipSuccess) { (void)hipGetLastError(); ev.bytes = -1.0; }
    ev.kernel = last_kernel_name();
    ev.p1 = launch_probe_count();
}
// rms norm * w + quantisation of N rows of x; when x is the pending result of a deferred K-split combine (x = residual + slabs), x is formed by the
// same launch
void Engine::prep_rms(const float *x, const float *w, int N, int K, int mask, hipStream_t s) {
    if (pend_.ks > 1 && pend_.n == 1 && pend_.y[0] == x && pend_.stride == (long long)N * K) { launch_rms_quant_slabs(pend_, w, N, K, act_, mask, s); pend_ = SlabSrc{}; return; }
    flush_pending(s);
    launch_rms_quant(x, w, N, K, act_, mask, s);
}
void Engine::flush_pending(hipStream_t s) { if (pend_.ks > 1) { launch_slab_flush(pend_, s); } pend_ = SlabSrc{}; }
bool Engine::mul_mat_set(const QWeight *const *W, float *const *y, const float *const *res, int n, int N, int ldy, hipStream_t s, const Prep *prep, bool fuse, bool silu_pair, const char *site, bool defer_ok, bool keep_pending) {
    struct ClearOverride { const __half *&p; ~ClearOverride() { p = nullptr; } } clear_override{xh_override_};   // valid for exactly one call
    bool same = true;
    int mask = 0;
    for (int i = 0; i < n; i++) { mask |= act_mask_for(W[i]->type); if (i) same = same && W[i]->type == W[0]->type && W[i]->rows == W[0]->rows && W[i]->cols == W[0]->cols; }
    const bool v2 = N == 1 && use_v2_ && same;
    fuse = fuse && v2 && prep && matvec_prologue_supported(W[0]->type, W[0]->cols);
    silu_pair = silu_pair && v2 && n == 2 && !res && matvec_silu_pair_supported(W[0]->type, W[0]->cols) && (!fuse || prep->kind == 1);
    // standalone preparation -- also the consumer of a deferred split-K combine (pend_): x = residual + slabs / silu(h1) * h3 straight from the slabs
    if (prep && !fuse) {
        SiteScope sc(this, "prepare", 0.0, s);
        const int K = W[0]->cols;
        if (prep->kind == 1) prep_rms(prep->x, prep->w, N, K, mask, s);
        else if (pend_.ks > 1 && prep->kind == 3 && pend_.n == 2 && pend_.y[0] == prep->x && pend_.y[1] == prep->w && !pend_.res[0] && !pend_.res[1] && pend_.stride == (long long)N * K) {
            launch_silu_mul_quant_slabs(pend_, N, K, act_, mask, tabs_, s); pend_ = SlabSrc{};
        } else {
            flush_pending(s);
            launch_silu_mul_quant(prep->x, prep->kind == 3 ? prep->w : nullptr, N, K, act_, mask, N == 1 ? tabs_dec_ : tabs_, s);
        }
    }
    // keep_pending (wv after a deferred wq|wk): the earlier launch's slabs stay where they are, this launch's go behind them, and both are handed to
    // the consumer together
    SlabSrc kept;
    ActQ act_here = act_;
    if (keep_pending && !prep && pend_.ks > 1 && !pend_.mixed && pend_.n == 2 && n == 1 && (size_t)pend_.n * pend_.ks * (size_t)pend_.stride < act_.ws_floats) {
        kept = pend_; pend_ = SlabSrc{};
        const size_t used = (size_t)kept.n * kept.ks * (size_t)kept.stride;
        act_here.ws += used; act_here.ws_floats -= used;
    }
    flush_pending(s);      // (nothing left unless the preparation above was fused into a mat-vec prologue or absent)
    double wbytes = 0;
    for (int i = 0; i < n; i++) wbytes += (double)W[i]->bytes;
    SiteScope sc(this, site, wbytes, s);
    bool done = false;
    // prefill: one launch for the set, weights streamed once per <= 128 rows
    if (N >= 5 && same && mmq_enabled() >= 2) done = launch_mmq2_set(W, y, res, n, act_here, N, ldy, s, defer_ok && defer_combine_ ? &pend_ : nullptr);
    // the rows were left in fp16 by an earlier launch, act_ holds OTHER rows
    const bool staged_elsewhere = !prep && (xh_override_ != nullptr || (same && W[0]->type == GT_F16 && N >= 512));
    // unquantised weights at prompt sizes: the set in one launch of the big MFMA GEMM, split K for wo / w2
    if (!done && same && W[0]->type == GT_F16 && N >= 512 && act_.xh) {
        const __half *Wh[3]; for (int i = 0; i < n; i++) Wh[i] = reinterpret_cast<const __half *>(W[i]->qs);
        const __half *Ain = !prep && xh_override_ ? xh_override_ : act_.xh;     // rows some launch left in fp16 elsewhere (the feed-forward pair's epilogue)
        done = launch_gemm_f16_set(Ain, W[0]->cols, Wh, n, N, W[0]->rows, W[0]->cols, y, res, ldy, act_.ws, act_.ws_floats, n_cus_, s, defer_ok && defer_combine_ ? &pend_ : nullptr);
    }
    if (!done && staged_elsewhere && xh_override_) throw HipError{hipErrorInvalidValue, "F16 set launch refused rows that only exist in fp16 (no generic path may read act_ here)", __FILE__, __LINE__};
    if (!done && silu_pair) {   // the pair epilogue needs the two matrices equally spaced; launch_matvec_set refuses otherwise and the plain launch below runs
        done = fuse ? launch_matvec_set(W, y, res, n, act_, s, prep->kind, prep->x, prep->w, &tabs_, 1) : launch_matvec_set(W, y, res, n, act_, s, 0, nullptr, nullptr, &tabs_, 1);
        silu_pair = done;
    }
    if (!done) {
        if (fuse) done = launch_matvec_set(W, y, res, n, act_, s, prep->kind, prep->x, prep->w, &tabs_);
        else if (v2) done = launch_matvec_set(W, y, res, n, act_, s);
    }
    if (!done) {
        if (fuse) throw HipError{hipErrorInvalidValue, "fused mat-vec prologue rejected a supported shape", __FILE__, __LINE__};
        for (int i = 0; i < n; i++) {
            const QWeight *Wp[1] = {W[i]}; float *Yp[1] = {y[i]}; const float *Rp[1] = {res ? res[i] : nullptr};
            if (!(N == 1 && use_v2_ && launch_matvec_set(Wp, Yp, Rp, 1, act_, s))) launch_mul_mat(*W[i], act_, N, y[i], ldy, res ? res[i] : nullptr, s);
        }
    }
    if (kept.ks > 1) {   // q | k pending from the earlier launch + this launch's v (pending or already in place): one SlabSrc for launch_rope_kv_slabs / the flush
        SlabSrc m = kept; m.mixed = true; m.n = 3;
        m.y[2] = y[0]; m.res[2] = res ? res[0] : nullptr;
        if (pend_.ks > 1) { m.mbase[2] = pend_.mbase[0]; m.mks[2] = pend_.mks[0]; } else { m.mbase[2] = y[0]; m.mks[2] = 1; }
        pend_ = m;
    }
    return silu_pair;
}
void Engine::mul_mat(const QWeight &W, int N, float *y, int ldy, const float *residual, hipStream_t s, const Prep *prep, bool fuse, const char *site, bool defer_ok, bool keep_pending) {
    const QWeight *Wp[1] = {&W}; float *Yp[1] = {y}; const float *Rp[1] = {residual};
    mul_mat_set(Wp, Yp, Rp, 1, N, ldy, s, prep, fuse, false, site, defer_ok, keep_pending);
}

// wq|wk and a differently typed wv (k-quant "more bits" layers) in one launch; returns false when the shapes / types are outside the mixed kernel's
// range.
bool Engine::mixed_qkv(const LayerW &L, hipStream_t s, bool fuse) {
    if (!use_v2_) return false;
    const int E = (int)llm_.n_embd;
    const QWeight *W1[2] = {&L.wq, &L.wk}, *W2[1] = {&L.wv}; float *Y1[2] = {q_, k_}, *Y2[1] = {v_};
    if (act_mask_for(L.wq.type) != ACT_Q8K || act_mask_for(L.wv.type) != ACT_Q8K) return false;
    SiteScope sc(this, "qkv", (double)(L.wq.bytes + L.wk.bytes + L.wv.bytes), s);   // only once the launch is certain: a refused shape must not record an empty site
    if (fuse && matvec_prologue_supported(L.wq.type, E)) { if (launch_matvec_mixed(W1, Y1, 2, W2, Y2, 1, act_, s, 1, x_, L.attn_norm)) return true; }
    // standalone preparation, then the mixed launch; if that is refused the caller's per-type launches find the row already prepared
    launch_rms_quant(x_, L.attn_norm, 1, E, act_, ACT_Q8K, s);
    if (launch_matvec_mixed(W1, Y1, 2, W2, Y2, 1, act_, s)) return true;
    const bool keep = prof_on_; prof_on_ = false;                          // the enclosing site already covers these launches
    mul_mat_set(W1, Y1, nullptr, 2, 1, E, s, nullptr, false); mul_mat(L.wv, 1, v_, E, nullptr, s, nullptr, false);
    prof_on_ = keep;
    return true;
}

// MINIGPT4_PARITY=1 (or minigpt4_amd_set_parity): the same graph as forward() below with every fp32 accumulation in the CPU oracle's order --
// standalone row preparation (sum of squares in element order), k_mul_mat_ref (per-block terms added block after block), RoPE + cache append,
// k_attn_ref (sequential score / P.V chains), no fused prologues, no MFMA tiles.  Everything else (integer block dots, activation quantisation, fp16
// tables, RoPE table) is shared with the fast path and exact there, so this pass is BIT-IDENTICAL to oracle/refcpu.c: logits and greedy ids
// (tests/test_gpu_paritymode.py).  Slow by design (one wave per output).
void Engine::forward_ref(int N, bool from_tokens, hipStream_t s, bool feed) {
    const int E = (int)llm_.n_embd, F = (int)llm_.n_ff(), H = (int)llm_.n_head, hd = E / H, V = (int)llm_.n_vocab;
    const size_t C = (size_t)n_ctx_, sl = (size_t)cur_;
    int *const d_npast = d_npast_ + sl, *const d_argmax = d_argmax_ + sl, *const d_feed = d_feed_ + sl;
    float *const logits = logits_ + sl * (size_t)V;
    if (from_tokens) launch_get_rows(tok_type_, tok_raw_, E, feed ? d_feed : d_tokens_, N, x_, s);
    // MINIGPT4_PARITY_TRACE=<file>: every intermediate as a record {char name[32]; int64 n; float[n]} -- the oracle writes the same sequence
    // (orc_set_trace), and tools/trace_diff.py reports the first record that differs.  Synchronises after every launch; never inside a graph capture.
    auto tr = [&](const char *what, int il, const float *p, size_t n) {
        if (!trace_file_) return;
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone; HIP_IGNORE(hipStreamIsCapturing(s, &st));
        if (st != hipStreamCaptureStatusNone) return;
        std::vector<float> h(n);
        HIP_CHECK(hipMemcpyAsync(h.data(), p, n * 4, hipMemcpyDeviceToHost, s)); HIP_CHECK(hipStreamSynchronize(s));
        char name[32]; memset(name, 0, sizeof name); snprintf(name, sizeof name, "%s.%d", what, il);
        const long long cnt = (long long)n;
        fwrite(name, 1, 32, trace_file_); fwrite(&cnt, 8, 1, trace_file_); fwrite(h.data(), 4, n, trace_file_); fflush(trace_file_);
    };
    tr("embd", -1, x_, (size_t)N * E);
    const int t_max = n_ctx_;                                              // sizes k_attn_ref's LDS rows; a captured decode step is replayed at later positions
    // one row (decode): same-type matrices of a set in one launch of the prefetching row kernel; otherwise (prompt rows, other types) one generic
    // launch per matrix
    auto ref_set = [&](std::initializer_list<const QWeight *> Ws, std::initializer_list<float *> Ys, const float *res) {
        const QWeight *W[3]; float *Y[3]; const float *R[3]; int n = 0;
        for (const QWeight *w : Ws) W[n++] = w;
        n = 0; for (float *yv : Ys) { Y[n] = yv; R[n] = res; n++; }
        int done = 0;
        if (N == 1) {                                // split into runs of equal type / shape (wq | wk + a differently typed wv)
            while (done < n) {
                int run = 1; while (done + run < n && W[done + run]->type == W[done]->type && W[done + run]->rows == W[done]->rows && W[done + run]->cols == W[done]->cols) run++;
                if (!launch_mul_mat_ref_set(W + done, Y + done, res ? R + done : nullptr, run, act_, s))
                    for (int i = 0; i < run; i++) launch_mul_mat_ref(*W[done + i], act_, N, Y[done + i], W[done + i]->rows, res, s);
                done += run;
            }
        } else {
            // prompt rows: the int8-MFMA kernels WITHOUT a K split add every output's per-block terms block after block -- the oracle's order
            // (bit-identical: test_gpu_paritymode); runs of equal type / shape in one launch each, anything they refuse on the oracle-order row
            // kernels
            while (done < n) {
                int run = 1; while (done + run < n && W[done + run]->type == W[done]->type && W[done + run]->rows == W[done]->rows && W[done + run]->cols == W[done]->cols) run++;
                if (!(N >= 5 && launch_mmq2_set(W + done, Y + done, res ? R + done : nullptr, run, act_, N, W[done]->rows, s, nullptr, 1)))
                    for (int i = 0; i < run; i++) launch_mul_mat_ref(*W[done + i], act_, N, Y[done + i], W[done + i]->rows, res, s);
                done += run;
            }
        }
    };
    // One row without a trace (the decode step): the fast step's launch structure -- row preparation in the mat-vec prologues, wq | wk (| wv) and w1
    // | w3 as set launches -- on the oracle-order variants of the same kernels (MATVEC_EPI_REF); a set those kernels refuse (non-k-quant types) takes
    // the standalone preparation + row kernels below.
    const bool fused_row = N == 1 && !trace_file_;
    auto fused_set = [&](std::initializer_list<const QWeight *> Ws, std::initializer_list<float *> Ys, const float *res, int pro, const float *px, const float *pw) -> bool {
        if (!fused_row) return false;
        const QWeight *W[3]; float *Y[3]; const float *R[3]; int n = 0;
        for (const QWeight *w : Ws) W[n++] = w;
        n = 0; for (float *yv : Ys) { Y[n] = yv; R[n] = res; n++; }
        if (pro != 0 && !matvec_prologue_supported(W[0]->type, W[0]->cols)) return false;
        bool same = true;
        for (int i = 1; i < n; i++) same = same && W[i]->type == W[0]->type && W[i]->rows == W[0]->rows && W[i]->cols == W[0]->cols;
        if (same) return launch_matvec_set(W, Y, res ? R : nullptr, n, act_, s, pro, px, pw, &tabs_, MATVEC_EPI_REF);
        if (n == 3 && !res && W[1]->type == W[0]->type && W[1]->rows == W[0]->rows && W[1]->cols == W[0]->cols && W[2]->cols == W[0]->cols && (pro == 0 || pro == 1))
            return launch_matvec_mixed(W, Y, 2, W + 2, Y + 2, 1, act_, s, pro, px, pw, MATVEC_EPI_REF);     // wq | wk + a differently typed wv
        return false;
    };
    for (size_t il = 0; il < layers_.size(); il++) {
        const LayerW &L = layers_[il];
        __half *kc = kc_ + (sl * layers_.size() + il) * C * E, *vc = vc_ + (sl * layers_.size() + il) * C * E;
        if (!fused_set({&L.wq, &L.wk, &L.wv}, {q_, k_, v_}, nullptr, 1, x_, L.attn_norm)) {
            launch_rms_quant(x_, L.attn_norm, N, E, act_, act_mask_for(L.wq.type) | act_mask_for(L.wk.type) | act_mask_for(L.wv.type), s, true);
            ref_set({&L.wq, &L.wk, &L.wv}, {q_, k_, v_}, nullptr);
        }
        tr("q", (int)il, q_, (size_t)N * E); tr("k", (int)il, k_, (size_t)N * E); tr("v", (int)il, v_, (size_t)N * E);
        // decode: RoPE + cache append inside the attention launch
        if (N == 1 && !trace_file_) launch_attn_ref_fused(q_, k_, v_, kc, vc, H, hd, d_npast, t_max, cos_, sin_, tabs_, att_, s);
        else { launch_rope_kv(q_, k_, v_, N, H, hd, d_npast, cos_, sin_, kc, vc, s); launch_attn_ref(q_, kc, vc, N, H, hd, d_npast, t_max, tabs_, att_, s); }
        tr("q_rope", (int)il, q_, (size_t)N * E); tr("att", (int)il, att_, (size_t)N * E);
        if (!fused_set({&L.wo}, {x_}, x_, 2, att_, nullptr)) {
            launch_silu_mul_quant(att_, nullptr, N, E, act_, act_mask_for(L.wo.type), tabs_, s);
            ref_set({&L.wo}, {x_}, x_);
        }
        tr("x_attn", (int)il, x_, (size_t)N * E);
        if (!fused_set({&L.w1, &L.w3}, {h1_, h3_}, nullptr, 1, x_, L.ffn_norm)) {
            launch_rms_quant(x_, L.ffn_norm, N, E, act_, act_mask_for(L.w1.type) | act_mask_for(L.w3.type), s, true);
            ref_set({&L.w1, &L.w3}, {h1_, h3_}, nullptr);
        }
        tr("h1", (int)il, h1_, (size_t)N * F); tr("h3", (int)il, h3_, (size_t)N * F);
        launch_silu_mul_quant(h1_, h3_, N, F, act_, act_mask_for(L.w2.type), tabs_, s);
        if (!fused_set({&L.w2}, {x_}, x_, 0, nullptr, nullptr)) ref_set({&L.w2}, {x_}, x_);
        tr("x_ffn", (int)il, x_, (size_t)N * E);
    }
    if (!(N == 1 && fused_set({&output_}, {logits}, nullptr, 1, x_, norm_))) {
        launch_rms_quant(x_ + (size_t)(N - 1) * E, norm_, 1, E, act_, act_mask_for(output_.type), s, true);
        const QWeight *Wo[1] = {&output_}; float *Yo[1] = {logits};
        if (!launch_mul_mat_ref_set(Wo, Yo, nullptr, 1, act_, s)) launch_mul_mat_ref(output_, act_, 1, logits, V, nullptr, s);
    }
    launch_argmax(logits, V, d_argmax, d_scratch_, s);
    launch_advance(d_npast, N, d_feed, d_argmax, s);
    HIP_CHECK(hipMemcpyAsync(h_argmax_ + sl, d_argmax, 4, hipMemcpyDeviceToHost, s));
}
void Engine::set_parity(bool on) {
    if (on == parity_) return;
    if (on && llm_.n_head > 0) {   // (advisor, round 4) the oracle-order attention kernel holds fewer keys in LDS than the fast one: refuse here, not at the first launch
        const int hd = (int)(llm_.n_embd / llm_.n_head), lim = attn_ref_max_ctx(hd);
        if (n_ctx_ > lim) throw HipError{hipErrorInvalidValue, "parity mode: n_ctx exceeds what the oracle-order attention kernel's LDS rows hold (attn_ref_max_ctx)", __FILE__, __LINE__};
        attn_ref_prepare();
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    for (Conversation &c : conv_) { if (c.graph) HIP_IGNORE(hipGraphExecDestroy(c.graph)); c.graph = nullptr; }   // captured with the other mode's launches
    for (hipGraphExec_t &g : batch_graph_) { if (g) HIP_IGNORE(hipGraphExecDestroy(g)); g = nullptr; }
    parity_ = on;
}

// Enqueue one forward pass for N rows already described by d_tokens_ (from_tokens) or x_ (embeddings), at position *d_npast_.
void Engine::forward(int N, bool from_tokens, hipStream_t s, bool feed) {
    pend_ = SlabSrc{}; xh_override_ = nullptr;          // host-side state of a pass that threw half-way must not reach this one
    if (parity_) { forward_ref(N, from_tokens, s, feed); return; }
    const int E = (int)llm_.n_embd, F = (int)llm_.n_ff(), H = (int)llm_.n_head, hd = E / H, V = (int)llm_.n_vocab;
    const size_t C = (size_t)n_ctx_, sl = (size_t)cur_;                     // everything below addresses the selected conversation's cache / scalars
    int *const d_npast = d_npast_ + sl, *const d_argmax = d_argmax_ + sl, *const d_feed = d_feed_ + sl;
    float *const logits = logits_ + sl * (size_t)V;
    const bool dec = N == 1;
    // feed: the row is the decode token kept in d_feed (greedy feedback / set by eval_chunk); otherwise the rows are described by d_tokens_ (id, or
    // -1 = an embedding row already sitting in x_) -- a chunk of exactly ONE embedding row must not pick up the stale decode token
    if (from_tokens) { SiteScope sc(this, "embed", (double)gt_nbytes(tok_type_, (size_t)E) * N, s); launch_get_rows(tok_type_, tok_raw_, E, feed ? d_feed : d_tokens_, N, x_, s); }
    auto fz = [&](int bit) { return dec && (fuse_mask_ >> bit & 1); };
    for (size_t il = 0; il < layers_.size(); il++) {
        const LayerW &L = layers_[il];
        __half *kc = kc_ + (sl * layers_.size() + il) * C * E, *vc = vc_ + (sl * layers_.size() + il) * C * E;
        const Prep p_attn{1, x_, L.attn_norm}, p_att{2, att_, nullptr}, p_ffn{1, x_, L.ffn_norm}, p_silu{3, h1_, h3_};
        {
            const QWeight *W3[3] = {&L.wq, &L.wk, &L.wv}; float *Y3[3] = {q_, k_, v_};
            // a K-split combine is left to launch_rope_kv_slabs
            if (L.wv.type == L.wq.type) mul_mat_set(W3, Y3, nullptr, 3, N, E, s, &p_attn, fz(0), false, "qkv", !dec);
            else if (fz(6) && L.wk.type == L.wq.type && mixed_qkv(L, s, fz(0))) {}
            else if (act_mask_for(L.wv.type) == act_mask_for(L.wq.type) && !fz(0)) {   // one standalone preparation serves both launches
                prep_rms(x_, L.attn_norm, N, E, act_mask_for(L.wq.type), s);
                // both combines left to the rope launch
                mul_mat_set(W3, Y3, nullptr, 2, N, E, s, nullptr, false, false, "qk", !dec); mul_mat(L.wv, N, v_, E, nullptr, s, nullptr, false, "v", !dec, !dec);
            } else { mul_mat_set(W3, Y3, nullptr, 2, N, E, s, &p_attn, fz(0), false, "qk"); mul_mat(L.wv, N, v_, E, nullptr, s, &p_attn, fz(0), "v"); }
        }
        // algorithmic bytes of the attention site: the cached fp16 K and V rows of every head up to the current position
        bool att_in_xh = false;                                             // the attention kernel left fp16 rows in act_.xh
        std::optional<SiteScope> att_sc;
        if (prof_on_) att_sc.emplace(this, "attention", 4.0 * (double)E * (double)(conv_[sl].n_committed + N), s);
        if (dec && attn_split_now_) launch_attn_llm_split(q_, k_, v_, kc, vc, H, hd, d_npast, n_ctx_, cos_, sin_, tabs_dec_, att_, attn_ws_, attn_splits_, s);
        else if (dec) launch_attn_llm(q_, k_, v_, kc, vc, 1, H, hd, d_npast, n_ctx_, cos_, sin_, tabs_dec_, att_, true, s);
        else {
            if (pend_.ks > 1 && pend_.n == 3 && pend_.y[0] == q_ && pend_.y[1] == k_ && pend_.y[2] == v_ && !pend_.res[0] && !pend_.res[1] && !pend_.res[2] && pend_.stride == (long long)N * E) {
                launch_rope_kv_slabs(pend_, N, H, hd, d_npast, cos_, sin_, kc, vc, s); pend_ = SlabSrc{};
            } else { flush_pending(s); launch_rope_kv(q_, k_, v_, N, H, hd, d_npast, cos_, sin_, kc, vc, s); }
            // an F16 wo at prompt sizes multiplies fp16(attention output): let the attention kernel store those rows itself (act_.xh), no conversion
            // launch
            const bool want_h = (f16_pair_ & 4) && L.wo.type == GT_F16 && N >= 512 && act_.xh && E % 128 == 0;
            if (!(attn_prefill_ && launch_attn_prefill(q_, kc, vc, N, H, hd, d_npast, conv_[sl].n_committed + N, tabs_, att_, s, want_h ? act_.xh : nullptr, &att_in_xh)))
                launch_attn_llm(q_, k_, v_, kc, vc, N, H, hd, d_npast, n_ctx_, cos_, sin_, tabs_, att_, false, s);
        }
        att_sc.reset();
        mul_mat(L.wo, N, x_, E, x_, s, att_in_xh ? nullptr : &p_att, fz(1), "wo", !dec);             // combine left to the ffn norm's preparation
        bool paired = false;   // h1_ already holds silu(w1 x) * (w3 x)
        // F16 weights at prompt sizes: w1 | w3 in one launch whose epilogue stores fp16(silu(w1 x) * (w3 x)) -- the rows w2 multiplies -- into h1_'s
        // memory (round 3)
        bool pair16 = false;
        if (!dec && (f16_pair_ & 1) && N >= 512 && L.w1.type == GT_F16 && L.w3.type == GT_F16 && L.w1.rows == L.w3.rows && L.w1.cols == L.w3.cols && act_.xh &&
            L.w2.type == GT_F16 && E % 128 == 0 && F % 64 == 0) {   // (w2's set launch must take the fp16 rows: its own shape conditions)
            prep_rms(x_, L.ffn_norm, N, E, act_mask_for(GT_F16), s);
            SiteScope sc(this, "w1w3", (double)(L.w1.bytes + L.w3.bytes), s);
            pair16 = launch_gemm_f16_silu_pair(act_.xh, E, reinterpret_cast<const __half *>(L.w1.qs), reinterpret_cast<const __half *>(L.w3.qs), N, F, E, pair_silu_computed_ ? tabs_dec_ : tabs_, nullptr,
                                               reinterpret_cast<__half *>(h1_), F, n_cus_, s);
        }
        if (pair16) {
            xh_override_ = reinterpret_cast<const __half *>(h1_);
            mul_mat(L.w2, N, x_, E, x_, s, nullptr, false, "w2", true);
            continue;
        }
        // combine left to silu * mul
        if (L.w1.type == L.w3.type) { const QWeight *W2[2] = {&L.w1, &L.w3}; float *Y2[2] = {h1_, h3_}; paired = mul_mat_set(W2, Y2, nullptr, 2, N, F, s, &p_ffn, fz(2), fz(5), "w1w3", !dec); }
        else if (act_mask_for(L.w1.type) == act_mask_for(L.w3.type) && !fz(2)) {
            prep_rms(x_, L.ffn_norm, N, E, act_mask_for(L.w1.type), s);
            mul_mat(L.w1, N, h1_, F, nullptr, s, nullptr, false, "w1"); mul_mat(L.w3, N, h3_, F, nullptr, s, nullptr, false, "w3");
        } else { mul_mat(L.w1, N, h1_, F, nullptr, s, &p_ffn, fz(2), "w1"); mul_mat(L.w3, N, h3_, F, nullptr, s, &p_ffn, fz(2), "w3"); }
        const Prep p_h{2, h1_, nullptr};
        // combine left to the next layer's attention norm (or flushed in front of the output matrix)
        mul_mat(L.w2, N, x_, E, x_, s, paired ? &p_h : &p_silu, fz(3), "w2", !dec);
    }
    flush_pending(s);
    // only the last token's logits are kept (llama.cpp logits_all = false)
    const Prep p_out{1, x_ + (size_t)(N - 1) * E, norm_};
    mul_mat(output_, 1, logits, V, nullptr, s, &p_out, fz(4), "output");
    { SiteScope sc(this, "argmax", (double)V * 4, s); launch_argmax(logits, V, d_argmax, d_scratch_, s); }
    { SiteScope sc(this, "advance", 0.0, s); launch_advance(d_npast, N, d_feed, d_argmax, s); }
    HIP_CHECK(hipMemcpyAsync(h_argmax_ + sl, d_argmax, 4, hipMemcpyDeviceToHost, s));
}

// One decode step for B conversations at once: row r carries token d_btok_[r] of conversation d_bslot_[r].  The weights are streamed once for the B
// rows (k_mul_mat's 4-token tiles up to B = 4, the int8-MFMA kernels from B = 5); attention runs per row against its conversation's cache.  Same
// arithmetic per row as forward(1): per-row activation quantisation, exact integer block dots, the same attention kernel body.
void Engine::forward_batch(int B, hipStream_t s) {
    pend_ = SlabSrc{}; xh_override_ = nullptr;
    batch_path_ = BatchPath{}; batch_path_.rows = B;
    const int E = (int)llm_.n_embd, F = (int)llm_.n_ff(), H = (int)llm_.n_head, hd = E / H, V = (int)llm_.n_vocab;
    const size_t C = (size_t)n_ctx_, seq_stride = layers_.size() * C * (size_t)E;
    // y_m[t][r] = W_m[r] . act[t] (+ res_m[t][r]) for n matrices that share the prepared rows.  Up to batch_rows_max_ rows go through the pipelined
    // multi-row mat-vec in passes of 4 rows (weights streamed once per pass); larger batches / other types through launch_mul_mat (int8-MFMA tiles
    // from 5 rows). px != null: the launch prepares its rows itself (rms_norm(px_t) * pw, quantised) -- only called when rows_pro() said the shape /
    // type is in range Does the row-interleaved MFMA launch serve this set?  Every matrix of one k-quant type and shape with its image built, AND
    // where it was measured faster than the v_dot4 multi-row mat-vec (profiles/r05_batched_shapes.log, us per launch, dot4 / mfma at B = 2 | 3 | 4):
    // sets of >= 128 row groups -- qkv 17.2/15.2 | 19.8/15.2 | 23.2/15.6, wq|wk 12.4/12.9 | 14.7/12.9 | 17.1/13.2, w1|w3 25.7/22.2 | 29.6/22.1 |
    // 34.4/22.1, output 27.3/27.0 | 33.1/27.2 | 37.3/27.5 -- but not wo (80 groups: 12.3/10.7 at B = 4, less than the standalone quantisation of the
    // attention rows it would need) nor a lone wv; w2 only at B = 4 (below); and from 3 rows on: at B = 2 the v_dot4 launches prepare their rows in
    // their own prologue (two launches less per layer), which outweighs the 2-3.5 us the MFMA launch would save.
    auto ri_serves = [&](std::initializer_list<const QWeight *> Ws) {
        if (!ri_ready_ || B < 3 || B > 4) return false;
        const QWeight *w0 = *Ws.begin();
        int groups = 0;
        for (const QWeight *w : Ws) { if (!ri_of(w) || w->type != w0->type || w->rows != w0->rows || w->cols != w0->cols) return false; groups += w->rows / 64; }
        // w2 (13B: 80 row groups x 54 super-blocks): three or four workgroups share a row group, each a K range, last arriver adds the parts.  With
        // the first weight fetch ahead of the staging and batched staging loads the launch is 18.9 (Q5_K) / 20.8 us (Q6_K) against 22.8 / 22.4 for
        // the v_dot4 kernel at B = 4 (equal at B = 3): 1037 vs 1018 tok/s, alternating on one box (profiles/r05_batched_decode_inengine.log)
        if (ri_w2_ && B == 4 && Ws.size() == 1 && w0->cols >= 8192 && ri_ksplit(groups, w0->cols, ri_ws_) > 1) return true;
        return groups >= 128;
    };
    auto mm = [&](std::initializer_list<const QWeight *> Ws, std::initializer_list<float *> ys, const float *res0, int ld, const float *px = nullptr, const float *pw = nullptr) {
        const int n = (int)Ws.size();
        const QWeight *W[3]; float *y[3]; const float *r[3];
        int i = 0; for (const QWeight *w : Ws) W[i++] = w;
        i = 0; for (float *p : ys) { y[i] = p; r[i] = res0; i++; }
        bool same = true; for (int k = 1; k < n; k++) same = same && W[k]->type == W[0]->type && W[k]->rows == W[0]->rows && W[k]->cols == W[0]->cols;
        // wo at 3 / 4 rows (round 6, MINIGPT4_RI_WO): the attention rows are quantised as they are inside the MFMA launch (80 row groups: eight waves per workgroup split K, no
        // workgroup K split -- every workgroup stages the whole rows anyway)
        if (ri_wo_ && px && !pw && n == 1 && ri_ready_ && B >= 3 && B <= 4 && ri_of(W[0])) {
            const RiPlanes *rp[1] = {ri_of(W[0])};
            if (launch_matvec_ri(W, rp, y, res0 ? r : nullptr, 1, act_, B, ld, s, px, nullptr, W[0]->cols, RiWorkspace{})) { batch_path_.ri++; batch_path_.ri_plain++; return; }
        }
        if (ri_serves(Ws) && (!px || pw)) {                 // prepared rows, or rows this launch rms-norms and quantises itself (px, pw)
            const RiPlanes *rp[3]; for (int k = 0; k < n; k++) rp[k] = ri_of(W[k]);
            if (launch_matvec_ri(W, rp, y, res0 ? r : nullptr, n, act_, B, ld, s, px, pw, W[0]->cols, ri_ws_)) {
                batch_path_.ri++; if (ri_ksplit(n * (W[0]->rows / 64), W[0]->cols, ri_ws_) > 1) batch_path_.ri_ksplit++;
                return;
            }
        }
        if (same && B <= batch_rows_max_) {
            bool ok = true;
            for (int t0 = 0; t0 < B && ok; t0 += 4) {
                const int K = W[0]->cols;
                ActQ A = act_;
                A.q8k += (size_t)t0 * K; A.dk += (size_t)t0 * (K / 256); A.bsk += (size_t)t0 * (K / 16); A.bsq += (size_t)t0 * (K / 16); A.q80 += (size_t)t0 * K;
                A.d0 += (size_t)t0 * (K / 32); A.d1 += (size_t)t0 * (K / 32); A.s1 += (size_t)t0 * (K / 32); A.sum0 += (size_t)t0 * (K / 32);
                float *yo[3]; const float *ro[3];
                for (int k = 0; k < n; k++) { yo[k] = y[k] + (size_t)t0 * ld; ro[k] = r[k] ? r[k] + (size_t)t0 * ld : nullptr; }
                ok = launch_matvec_rows(W, yo, res0 ? ro : nullptr, n, A, std::min(4, B - t0), ld, s, px ? px + (size_t)t0 * K : nullptr, pw, K);
                if (!ok && t0) throw HipError{hipErrorInvalidValue, "multi-row mat-vec refused a pass it had accepted", __FILE__, __LINE__};
            }
            if (ok) { batch_path_.dot4++; return; }
        }
        // the launch that was to prepare its rows itself was refused (plane spacing, LDS): a performance choice must not fail the step -- prepare
        // them standalone
        if (px) {
            const int K = W[0]->cols; int mask = 0;
            for (int k = 0; k < n; k++) mask |= act_mask_for(W[k]->type);
            if (pw) launch_rms_quant(px, pw, B, K, act_, mask, s); else launch_silu_mul_quant(px, nullptr, B, K, act_, mask, tabs_, s);
        }
        for (int k = 0; k < n; k++) { launch_mul_mat(*W[k], act_, B, y[k], ld, r[k], s); batch_path_.mul_mat++; }
    };
    // may the rows of this set be prepared inside its launch?  (same conditions mm() takes the multi-row kernel under)
    auto rows_pro = [&](std::initializer_list<const QWeight *> Ws, bool plain = false) {
        // measured (profiles/r02j_batched_decode_ab.log): inside the launch the preparation pays at 2 rows (560 vs 540 tok/s) and loses at 4 (808 vs
        // 829: every fat workgroup repeats four rows' norm + quantisation); MINIGPT4_BATCH_FUSE=1 forces it for every B <= 4, =0 switches it off
        // plain = quantisation only (the attention output in front of wo: no norm, no double-precision sums): cheap enough to stay inside the launch
        // at 3 and 4 rows too
        // the MFMA launch norms + quantises its rows itself (not the plain quantisation in front of wo: wo is not served)
        if (ri_serves(Ws)) return !plain && ri_fuse_;
        if (batch_fuse_ == 0 || B > batch_rows_max_ || B > 4 || (batch_fuse_ < 0 && B > 2 && !plain)) return false;
        const QWeight *w0 = *Ws.begin();
        for (const QWeight *w : Ws) if (w->type != w0->type || w->rows != w0->rows || w->cols != w0->cols) return false;
        return matvec_rows_prologue_ok(w0->type, w0->cols);
    };
    launch_get_rows(tok_type_, tok_raw_, E, d_btok_, B, x_, s);
    for (size_t il = 0; il < layers_.size(); il++) {
        const LayerW &L = layers_[il];
        __half *kc = kc_ + il * C * E, *vc = vc_ + il * C * E;            // conversation 0's layer; the kernel adds slot * seq_stride
        // a "more bits" layer (wq|wk Q4_K / Q5_K + wv Q6_K): one launch over both sets, like the single-row step's k_matvec_mix
        auto qkv_mixed = [&](bool pro) {
            if (!batch_mix_ || B > batch_rows_max_ || B > 4 || L.wk.type != L.wq.type || L.wv.type == L.wq.type) return false;
            const QWeight *W1[2] = {&L.wq, &L.wk}, *W2[1] = {&L.wv};
            float *y1[2] = {q_, k_}, *y2[1] = {v_};
            // at 3 and 4 prepared rows on the matrix cores (k_matvec_ri_mix), under ri_serves' conditions: every image built, >= 128 row groups in
            // the set
            if (!pro && ri_ready_ && B >= 3 && ri_of(&L.wq) && ri_of(&L.wk) && ri_of(&L.wv) && (L.wq.rows + L.wk.rows + L.wv.rows) / 64 >= 128) {
                const RiPlanes *r1[2] = {ri_of(&L.wq), ri_of(&L.wk)}, *r2[1] = {ri_of(&L.wv)};
                if (launch_matvec_ri_mixed(W1, r1, y1, 2, W2, r2, y2, 1, act_, B, E, s)) { batch_path_.ri_mix++; return true; }
            }
            const bool ok = launch_matvec_rows_mixed(W1, y1, 2, W2, y2, 1, act_, B, E, s, pro ? x_ : nullptr, pro ? L.attn_norm : nullptr, E);
            if (ok) batch_path_.dot4_mix++;
            return ok;
        };
        if (B > batch_rows_max_ && batch_sets_) {
            batch_path_.sets++;
            // More conversations than the multi-row mat-vec takes (5 ... 32): the prompt pass's launches -- one int8-MFMA launch per matrix SET, row
            // preparations that also combine a K-split predecessor (pend_), i.e. 11 launches per layer instead of the 19 of one launch + one combine
            // per matrix (round 3).
            const Prep p_attn{1, x_, L.attn_norm}, p_att{2, att_, nullptr}, p_ffn{1, x_, L.ffn_norm}, p_silu{3, h1_, h3_};
            const QWeight *W3[3] = {&L.wq, &L.wk, &L.wv}; float *Y3[3] = {q_, k_, v_};
            if (L.wv.type == L.wq.type && L.wk.type == L.wq.type) mul_mat_set(W3, Y3, nullptr, 3, B, E, s, &p_attn, false, false, "qkv");
            else {
                prep_rms(x_, L.attn_norm, B, E, act_mask_for(L.wq.type) | act_mask_for(L.wk.type) | act_mask_for(L.wv.type), s);
                if (L.wk.type == L.wq.type) { mul_mat_set(W3, Y3, nullptr, 2, B, E, s, nullptr, false, false, "qk"); mul_mat(L.wv, B, v_, E, nullptr, s, nullptr, false, "v"); }
                else { mul_mat(L.wq, B, q_, E, nullptr, s, nullptr, false, "q"); mul_mat(L.wk, B, k_, E, nullptr, s, nullptr, false, "k"); mul_mat(L.wv, B, v_, E, nullptr, s, nullptr, false, "v"); }
            }
            flush_pending(s);
            launch_attn_llm_batched(q_, k_, v_, kc, vc, B, H, hd, d_npast_, d_bslot_, seq_stride, n_ctx_, cos_, sin_, tabs_dec_, att_, s);
            mul_mat(L.wo, B, x_, E, x_, s, &p_att, false, "wo", true);
            if (L.w1.type == L.w3.type) { const QWeight *W2[2] = {&L.w1, &L.w3}; float *Y2[2] = {h1_, h3_}; mul_mat_set(W2, Y2, nullptr, 2, B, F, s, &p_ffn, false, false, "w1w3", true); }
            else { prep_rms(x_, L.ffn_norm, B, E, act_mask_for(L.w1.type) | act_mask_for(L.w3.type), s); mul_mat(L.w1, B, h1_, F, nullptr, s, nullptr, false, "w1"); mul_mat(L.w3, B, h3_, F, nullptr, s, nullptr, false, "w3"); }
            mul_mat(L.w2, B, x_, E, x_, s, &p_silu, false, "w2", true);
            continue;
        }
        if (L.wv.type == L.wq.type && L.wk.type == L.wq.type && rows_pro({&L.wq, &L.wk, &L.wv})) mm({&L.wq, &L.wk, &L.wv}, {q_, k_, v_}, nullptr, E, x_, L.attn_norm);
        else if (L.wk.type == L.wq.type && rows_pro({&L.wq, &L.wk}) && rows_pro({&L.wv})) {
            if (!qkv_mixed(true)) { mm({&L.wq, &L.wk}, {q_, k_}, nullptr, E, x_, L.attn_norm); mm({&L.wv}, {v_}, nullptr, E, x_, L.attn_norm); }
        } else {
            launch_rms_quant(x_, L.attn_norm, B, E, act_, act_mask_for(L.wq.type) | act_mask_for(L.wk.type) | act_mask_for(L.wv.type), s);
            if (L.wv.type == L.wq.type && L.wk.type == L.wq.type) mm({&L.wq, &L.wk, &L.wv}, {q_, k_, v_}, nullptr, E);
            else if (qkv_mixed(false)) {}
            else if (L.wk.type == L.wq.type) { mm({&L.wq, &L.wk}, {q_, k_}, nullptr, E); mm({&L.wv}, {v_}, nullptr, E); }
            else { mm({&L.wq}, {q_}, nullptr, E); mm({&L.wk}, {k_}, nullptr, E); mm({&L.wv}, {v_}, nullptr, E); }
        }
        launch_attn_llm_batched(q_, k_, v_, kc, vc, B, H, hd, d_npast_, d_bslot_, seq_stride, n_ctx_, cos_, sin_, tabs_dec_, att_, s);
        if (rows_pro({&L.wo}, true)) mm({&L.wo}, {x_}, x_, E, att_, nullptr);     // the attention output rows are quantised inside the wo launch
        else { launch_silu_mul_quant(att_, nullptr, B, E, act_, act_mask_for(L.wo.type), tabs_, s); mm({&L.wo}, {x_}, x_, E); }
        if (L.w1.type == L.w3.type && rows_pro({&L.w1, &L.w3})) mm({&L.w1, &L.w3}, {h1_, h3_}, nullptr, F, x_, L.ffn_norm);
        else {
            launch_rms_quant(x_, L.ffn_norm, B, E, act_, act_mask_for(L.w1.type) | act_mask_for(L.w3.type), s);
            if (L.w1.type == L.w3.type) mm({&L.w1, &L.w3}, {h1_, h3_}, nullptr, F);
            else { mm({&L.w1}, {h1_}, nullptr, F); mm({&L.w3}, {h3_}, nullptr, F); }
        }
        launch_silu_mul_quant(h1_, h3_, B, F, act_, act_mask_for(L.w2.type), tabs_dec_, s);
        // (Round 6 built the next layer's row preparation into this launch's last-arriving workgroup -- the round-5 verdict's item 3 i -- and measured 834-841 vs 994-1023 tok/s:
        // profiles/r06_batched_producer_tail.log, commit 749954e.)
        mm({&L.w2}, {x_}, x_, E);
    }
    // final norm inside the output matrix's MFMA launch
    if (B <= batch_rows_max_ && ri_fuse_ && ri_serves({&output_})) mm({&output_}, {blogits_}, nullptr, V, x_, norm_);
    else {
        prep_rms(x_, norm_, B, E, act_mask_for(output_.type), s);             // (also combines the last layer's w2 slabs when its combine was deferred)
        mm({&output_}, {blogits_}, nullptr, V);
    }
    launch_batch_finish(blogits_, V, B, d_bslot_, d_npast_, d_argmax_, d_feed_, logits_, s);
}

// Evaluate one chunk of N rows of the selected conversation at its position n_committed.  row_tok[i] >= 0: token id; -1: the next packed embedding
// row of `embd`. LOAD_RECV: the arenas are allocated but hold nothing until the broadcast has landed and weights_received() ran -- every compute
// entry point refuses until then (a caller that skipped the hand-over, or an inherited MINIGPT4_LOAD=recv, must get an error, not text generated from
// uninitialised weights).
bool Engine::weights_missing() const {
    if (load_mode_ != LOAD_RECV) return false;
    set_last_error("the context was loaded in receive mode (MINIGPT4_LOAD=recv) and its weight arenas have not been filled: broadcast them, then call minigpt4_amd_weights_received");
    MG4_ERR("%s", last_error().c_str());
    return true;
}
int Engine::eval_chunk(const int *row_tok, int N, const float *embd) {
    if (N <= 0) return 0;
    if (weights_missing()) return 1;
    const int E = (int)llm_.n_embd;
    Conversation &cv = conv_[(size_t)cur_];
    if (logits_host_slot_ == cur_) logits_host_slot_ = -1;
    launch_set_int(d_npast_ + cur_, cv.n_committed, stream_);
    if (N == 1 && row_tok[0] >= 0) {
        launch_set_int(d_feed_ + cur_, row_tok[0], stream_);
        // long contexts: the decode step's attention shares every head's keys between workgroups (two launches instead of one: pays from a few
        // hundred keys on).  The choice is part of the captured graph, so a conversation that crosses the threshold gets its step re-captured (once).
        const bool split = attn_split_t_ > 0 && cv.n_committed + 1 > attn_split_t_;
        attn_split_now_ = split;
        if (use_graph_ && !prof_on_ && !trace_file_) {
            if (cv.graph && cv.graph_split != split) { HIP_IGNORE(hipGraphExecDestroy(cv.graph)); cv.graph = nullptr; }
            cv.graph_split = split;
            if (!cv.graph) {
                hipGraph_t g = nullptr;
                HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
                forward(1, true, stream_, true);
                HIP_CHECK(hipStreamEndCapture(stream_, &g));
                HIP_CHECK(hipGraphInstantiate(&cv.graph, g, nullptr, nullptr, 0));
                HIP_CHECK(hipGraphDestroy(g));
            }
            HIP_CHECK(hipGraphLaunch(cv.graph, stream_));
        } else {
            forward(1, true, stream_, true);
        }
    } else {
        HIP_CHECK(hipMemcpyAsync(d_tokens_, row_tok, (size_t)N * 4, hipMemcpyHostToDevice, stream_));
        size_t er = 0;
        for (int i = 0; i < N;) {   // contiguous runs of embedding rows go straight into the residual stream
            if (row_tok[i] >= 0) { i++; continue; }
            int j = i; while (j < N && row_tok[j] < 0) j++;
            HIP_CHECK(hipMemcpyAsync(x_ + (size_t)i * E, embd + er * E, (size_t)(j - i) * E * 4, hipMemcpyHostToDevice, stream_));
            er += (size_t)(j - i); i = j;
        }
        HIP_CHECK(hipStreamSynchronize(stream_));   // the (pageable) staging vectors may be reused right after this call
        forward(N, true, stream_);                   // k_get_rows skips rows whose id is negative
    }
    cv.n_committed += N;
    return 0;
}

// Deferred prefill batching.  The reference evaluates every prompt fragment separately (system prompt, "Human: <Img>", the 32 image rows,
// "</Img> ", the question, "### Assistant:" -- six passes over the weights for one image turn, minigpt4.cpp:2671-2702).  Token rows do not
// depend on how they are batched (causal attention, per-row quantisation), so fragments are queued and evaluated in one pass when logits are
// needed (sampling) -- the weights are streamed once.  Context overflow is still reported by the call that would overflow.
int Engine::flush() {
    Conversation &cv = conv_[(size_t)cur_];
    if (cv.pend_tok.empty()) return 0;
    const int E = (int)llm_.n_embd;
    size_t er = 0;
    for (size_t i = 0; i < cv.pend_tok.size(); i += (size_t)max_chunk_) {
        const int n = (int)std::min((size_t)max_chunk_, cv.pend_tok.size() - i);
        size_t ne = 0; for (int k = 0; k < n; k++) ne += cv.pend_tok[i + k] < 0;
        const int rc = eval_chunk(cv.pend_tok.data() + i, n, cv.pend_embd.data() + er * E);
        er += ne;
        if (rc) { cv.pend_tok.clear(); cv.pend_embd.clear(); cv.n_past = cv.n_committed; return rc; }
    }
    cv.pend_tok.clear(); cv.pend_embd.clear();
    return 0;
}

int Engine::add_tokens(const std::vector<int> &tokens, bool flush_now) {
    Conversation &cv = conv_[(size_t)cur_];
    if (cv.n_past + (int)tokens.size() > n_ctx_) { set_last_error("context overflow: n_past + n_tokens > n_ctx"); MG4_ERR("Failed to add string"); return E_FailedToAddString; }
    for (int t : tokens) if (t < 0 || t >= (int)llm_.n_vocab) { set_last_error("token id out of range"); MG4_ERR("Failed to add string"); return E_FailedToAddString; }
    cv.pend_tok.insert(cv.pend_tok.end(), tokens.begin(), tokens.end());
    cv.n_past += (int)tokens.size();
    if (flush_now || !defer_) { if (flush()) return E_FailedToAddString; }
    return E_None;
}
int Engine::add_string(const std::string &s) { return add_tokens(tok_.tokenize(s, true)); }
int Engine::add_embedding(const float *data, int n_rows) {
    Conversation &cv = conv_[(size_t)cur_];
    if (n_rows <= 0 || cv.n_past + n_rows > n_ctx_) { set_last_error("context overflow: n_past + n_rows > n_ctx"); MG4_ERR("Failed to add embedding"); return E_FailedToAddEmbedding; }
    cv.pend_tok.insert(cv.pend_tok.end(), (size_t)n_rows, -1);
    cv.pend_embd.insert(cv.pend_embd.end(), data, data + (size_t)n_rows * llm_.n_embd);
    cv.n_past += n_rows;
    if (!defer_) { if (flush()) return E_FailedToAddEmbedding; }
    return E_None;
}
const float *Engine::logits_host() {
    if (flush()) throw HipError{hipErrorUnknown, "deferred evaluation failed", __FILE__, __LINE__};
    if (logits_host_slot_ != cur_) {
        HIP_CHECK(hipMemcpyAsync(h_logits_, logits_ + (size_t)cur_ * llm_.n_vocab, (size_t)llm_.n_vocab * 4, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        logits_host_slot_ = cur_;
    }
    return h_logits_;
}
int Engine::sample_token(const SampleParams &p) {
    if (flush()) throw HipError{hipErrorUnknown, "deferred evaluation failed", __FILE__, __LINE__};
    if (p.temp <= 0) { HIP_CHECK(hipStreamSynchronize(stream_)); return h_argmax_[cur_]; }   // greedy: argmax computed on the device
    return sampler_.sample(logits_host(), (int)llm_.n_vocab, p);
}
const char *Engine::id_to_token(int id) const {
    if (id == 2) return "</s>";   // llama_token_eos()
    if (id < 0 || id >= (int)llm_.pieces.size()) return "";
    return llm_.pieces[(size_t)id].c_str();
}

int Engine::decode_loop(int steps, int *tokens_out, float *ms_total) {
    if (flush()) return 1;
    Conversation &cv = conv_[(size_t)cur_];
    if (steps <= 0 || cv.n_past + steps > n_ctx_) return 1;
    HIP_CHECK(hipStreamSynchronize(stream_));
    int first = h_argmax_[cur_];
    if (eval_chunk(&first, 1, nullptr)) return 1;   // builds the graph if needed, d_feed_[slot] <- greedy token afterwards (k_advance)
    cv.n_past += 1;
    HIP_CHECK(hipStreamSynchronize(stream_));
    if (tokens_out) tokens_out[0] = first;
    hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, stream_));
    for (int i = 1; i < steps; i++) {   // step i consumes the greedy token of step i-1, left in d_tokens_[0] by k_advance
        if (tokens_out) HIP_CHECK(hipMemcpyAsync(&tokens_out[i], d_feed_ + cur_, 4, hipMemcpyDeviceToHost, stream_));
        if (cv.graph && use_graph_) HIP_CHECK(hipGraphLaunch(cv.graph, stream_)); else forward(1, true, stream_, true);
    }
    HIP_CHECK(hipEventRecord(b, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms_total) *ms_total = ms;
    HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b));
    cv.n_past += steps - 1; cv.n_committed += steps - 1;
    if (logits_host_slot_ == cur_) logits_host_slot_ = -1;
    return 0;
}

int Engine::profile_sites(int steps, std::string &json) {
    if (flush()) return 1;
    Conversation &cv = conv_[(size_t)cur_];
    if (steps <= 0 || cv.n_past + steps > n_ctx_) return 1;
    HIP_CHECK(hipStreamSynchronize(stream_));
    prof_on_ = true; kernel_name_tracing(true);
    // Every step starts behind a gate kernel that holds the stream for ~8 ms: the host queues the step's ~250 launches and ~500 event records (6-7 us
    // of host time each, more than most of the kernels run) while the gate spins, and the GPU then drains them back to back -- the event pairs time
    // kernels, not the host.
    hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    int tok = h_argmax_[cur_];
    float tot = 0;
    for (int i = 0; i < steps; i++) {
        launch_delay(8000, stream_);
        HIP_CHECK(hipEventRecord(a, stream_));
        if (eval_chunk(&tok, 1, nullptr)) { prof_on_ = false; kernel_name_tracing(false); return 1; }
        cv.n_past += 1;
        HIP_CHECK(hipEventRecord(b, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b)); tot += ms;
        tok = h_argmax_[cur_];
    }
    prof_on_ = false; kernel_name_tracing(false);
    // us: dispatch begin..end (launch probes); mus: marker pairs
    struct Agg { std::string site, kernel; double us = 0, mus = 0, bytes = 0; long calls = 0; bool probed = true; };
    std::vector<Agg> agg;                                                    // first-seen order = launch order within the step
    for (auto &e : site_events_) {
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
        size_t k = 0;
        while (k < agg.size() && !(agg[k].site == e.site && agg[k].kernel == e.kernel)) k++;
        if (k == agg.size()) { agg.push_back(Agg{}); agg[k].site = e.site; agg[k].kernel = e.kernel; }
        const float pus = e.p1 > e.p0 ? launch_probe_us(e.p0, e.p1) : -1.0f;
        if (pus < 0.0f) agg[k].probed = false;
        agg[k].us += pus; agg[k].mus += ms * 1e3; agg[k].bytes += e.bytes; agg[k].calls++;
        HIP_IGNORE(hipEventDestroy(e.a)); HIP_IGNORE(hipEventDestroy(e.b));
    }
    site_events_.clear();
    HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b));
    char buf[768];
    json = "{\"steps\": " + std::to_string(steps) + ", \"eager_ms_per_step\": " + std::to_string(tot / steps) + ", \"sites\": [";
    for (size_t k = 0; k < agg.size(); k++) {
        // avg_us: the dispatches' own begin..end timestamps (what rocprofv3 reports); avg_us_markers: hipEventRecord pairs around the launch site
        // (adds packet processing)
        snprintf(buf, sizeof(buf), "%s{\"site\": \"%s\", \"kernel\": \"%s\", \"calls_per_step\": %.3f, \"avg_us\": %.3f, \"avg_us_markers\": %.3f, \"timing\": \"%s\", \"bytes_per_call\": %.1f}",
                 k ? ", " : "", agg[k].site.c_str(), agg[k].kernel.c_str(), (double)agg[k].calls / steps, (agg[k].probed ? agg[k].us : agg[k].mus) / (double)agg[k].calls,
                 agg[k].mus / (double)agg[k].calls, agg[k].probed ? "dispatch" : "markers", agg[k].bytes / (double)agg[k].calls);
        json += buf;
    }
    json += "]}";
    return 0;
}

// ====================================================================================================================
// several conversations per replica
// ====================================================================================================================
// The row-interleaved image of every k-quant matrix the batched step multiplies (ri_kernels.hip), built once, on the device, from the ordinary
// planes.
void Engine::build_ri_planes() {
    if (ri_ready_ || !use_ri_ || weights_missing()) return;
    std::vector<const QWeight *> ws;
    // only the sets the batched step serves this way (forward_batch: ri_serves, qkv_mixed): wq | wk | wv (of one type, or wq | wk + a Q6_K wv), w1 |
    // w3, the output matrix -- not the 80-group wo / w2
    for (const LayerW &L : layers_) {
        if (L.wk.type == L.wq.type && L.wv.type == L.wq.type) for (const QWeight *w : {&L.wq, &L.wk, &L.wv}) ws.push_back(w);
        // a "more bits" layer: k_matvec_ri_mix
        else if (L.wk.type == L.wq.type && L.wv.type == GT_Q6_K && batch_mix_) for (const QWeight *w : {&L.wq, &L.wk, &L.wv}) ws.push_back(w);
        if (L.w1.type == L.w3.type) for (const QWeight *w : {&L.w1, &L.w3}) ws.push_back(w);
    }
    if (ri_w2_) for (const LayerW &L : layers_) ws.push_back(&L.w2);
    if (ri_wo_) for (const LayerW &L : layers_) ws.push_back(&L.wo);
    ws.push_back(&output_);
    size_t total = 0;
    for (const QWeight *w : ws) { RiPlanes p; total += ri_plan(w->type, w->rows, w->cols, p, nullptr); }
    if (!total) return;
    ri_ws_.slab_floats = (size_t)1 << 18; ri_ws_.n_tickets = 1024;
    ri_arena_.alloc(total + ri_ws_.slab_floats * 4 + (size_t)ri_ws_.n_tickets * 4 + 8192);
    ri_ws_.slabs = reinterpret_cast<float *>(ri_arena_.take(ri_ws_.slab_floats * 4)); ri_ws_.tickets = reinterpret_cast<unsigned *>(ri_arena_.take((size_t)ri_ws_.n_tickets * 4));
    HIP_CHECK(hipMemset(ri_ws_.tickets, 0, (size_t)ri_ws_.n_tickets * 4));
    for (const QWeight *w : ws) {
        RiPlanes p; const size_t need = ri_plan(w->type, w->rows, w->cols, p, nullptr);
        if (!need) continue;
        ri_plan(w->type, w->rows, w->cols, p, ri_arena_.take(need));
        launch_ri_build(*w, p, stream_);
        ri_map_.emplace_back(w, p);
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    ri_ready_ = true;
    MG4_INFO("batched decode: row-interleaved image of %zu k-quant matrices, %.2f GB", ri_map_.size(), ri_arena_.used / 1e9);
}
int Engine::set_conversations(int n) {
    if (n < 1 || n > MAX_CONVERSATIONS) { set_last_error("conversation count out of range"); return 1; }
    HIP_CHECK(hipStreamSynchronize(stream_));
    release_buffers();
    conv_.assign((size_t)n, Conversation{});
    cur_ = 0;
    alloc_buffers();
    // the folded Q-Former constants live in the buffer arena alloc_buffers() has just re-taken (zeroed): evaluate them again, or every later encode
    // would run on all-zero layer-0 constants (round-5 advisor finding; tests/test_gpu_serve.py::test_encode_after_set_conversations_equals_fresh_context)
    if (load_mode_ == LOAD_FULL) fold_qformer_constants();
    if (n > 1 && load_mode_ == LOAD_FULL) build_ri_planes();                          // batched decode on the matrix cores needs the row-interleaved image (once per context)
    return 0;
}
int Engine::select_conversation(int slot) {
    if (slot < 0 || slot >= (int)conv_.size()) { set_last_error("conversation index out of range"); return 1; }
    cur_ = slot;
    return 0;
}
int Engine::decode_batch(const int *slots, int n, const SampleParams &p, int *ids_out, const int *forced) {
    if (!slots || !ids_out || n < 1 || n > (int)conv_.size()) { set_last_error("decode_batch: bad slot list"); return 1; }
    if (weights_missing()) return 1;
    bool seen[MAX_CONVERSATIONS] = {false};
    for (int i = 0; i < n; i++) { if (slots[i] < 0 || slots[i] >= (int)conv_.size() || seen[slots[i]]) { set_last_error("decode_batch: conversations must be distinct and in range"); return 1; } seen[slots[i]] = true; }
    if (forced) for (int i = 0; i < n; i++) if (forced[i] < 0 || forced[i] >= (int)llm_.n_vocab) { set_last_error("decode_batch: forced token id out of range"); return 1; }
    const int keep = cur_;
    struct Restore { Engine *e; int v; ~Restore() { e->cur_ = v; } } restore{this, keep};
    // 1. pending prompt rows of each conversation (its own prefill pass), then sample
    for (int i = 0; i < n; i++) { cur_ = slots[i]; ids_out[i] = sample_token(p); }
    // 2. one weight pass for the conversations that still have room
    HIP_CHECK(hipStreamSynchronize(stream_));                                // h_bstage_ may still feed the previous step's copies
    int B = 0;
    for (int i = 0; i < n; i++) {
        Conversation &cv = conv_[(size_t)slots[i]];
        if (cv.n_past + 1 > n_ctx_) continue;                               // context full: sampled, not advanced
        h_bstage_[B] = forced ? forced[i] : ids_out[i]; h_bstage_[MAX_CONVERSATIONS + B] = slots[i]; h_bstage_[2 * MAX_CONVERSATIONS + B] = cv.n_committed; B++;
    }
    if (!B) return 0;
    // oracle-order arithmetic exists for the single-conversation pass only: one pass per conversation (same results as the batched step is tested to
    // give)
    if (parity_) {
        for (int r = 0; r < B; r++) { cur_ = h_bstage_[MAX_CONVERSATIONS + r]; const int id = h_bstage_[r]; if (eval_chunk(&id, 1, nullptr)) return 1; conv_[(size_t)cur_].n_past += 1; }
        HIP_CHECK(hipStreamSynchronize(stream_));
        return 0;
    }
    // rows (token, conversation, position) travel in one copy; the device positions are normally current (k_advance / k_batch_finish keep them), but
    // a reset may have moved the host's view, so k_batch_begin writes them like eval_chunk does
    HIP_CHECK(hipMemcpyAsync(d_btok_, h_bstage_, 768, hipMemcpyHostToDevice, stream_));
    launch_batch_begin(d_npast_, d_bslot_, d_bpos_, B, stream_);
    if (use_graph_) {   // the step for B rows as one hipGraph: the rows live in device memory, so the same graph serves every set of B conversations
        hipGraphExec_t &ge = batch_graph_[(size_t)B];
        if (!ge) {
            hipGraph_t g = nullptr;
            HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
            forward_batch(B, stream_);
            HIP_CHECK(hipStreamEndCapture(stream_, &g));
            HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
        }
        HIP_CHECK(hipGraphLaunch(ge, stream_));
    } else forward_batch(B, stream_);
    for (int r = 0; r < B; r++) {
        const int sl = h_bstage_[MAX_CONVERSATIONS + r];
        Conversation &cv = conv_[(size_t)sl];
        cv.n_past += 1; cv.n_committed += 1;
        if (logits_host_slot_ == sl) logits_host_slot_ = -1;
    }
    HIP_CHECK(hipMemcpyAsync(h_argmax_, d_argmax_, conv_.size() * 4, hipMemcpyDeviceToHost, stream_));
    return 0;
}

// ====================================================================================================================
// image path
// ====================================================================================================================
// B images in ONE pass over the vision weights (B <= VISION_BATCH_MAX): every GEMM / LayerNorm runs on B x 257 (ViT) or B x 32 (Q-Former) rows,
// attention per image (grid z).  Per output element the arithmetic does not depend on B (one wave accumulates one 32x32 tile over K in a fixed
// order), so image b of a batch equals the same image encoded alone bit for bit
// (tests/test_gpu_parity.py::test_batched_image_encode_is_bit_identical).
int Engine::encode_images(const float *const *chw, int B, float *const *out) {
    if (B < 1 || B > VISION_BATCH_MAX) { set_last_error("encode_images: batch size out of range"); return E_ImageSize; }
    if (weights_missing()) return E_LoadModelFileHeader;
    // MINIGPT4_PARITY: the oracle's accumulation order, image embedding bit-identical to oracle/refcpu.c
    if (parity_) return encode_images_ref(chw, B, out);
    if (v_generic_) return encode_images_generic(chw, B, out);
    hipStream_t s = stream_;
    const int D = v_D_, M = v_M_, NQ = v_nq_, H = 768;
    const int R = B * 257, RQ = B * NQ;                                    // rows of the ViT / Q-Former activations
    hipEvent_t ea, eb; HIP_CHECK(hipEventCreate(&ea)); HIP_CHECK(hipEventCreate(&eb));
    for (int b = 0; b < B; b++) HIP_CHECK(hipMemcpyAsync(vi_img_ + (size_t)b * 3 * 224 * 224, chw[b], 3 * 224 * 224 * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipEventRecord(ea, s));
    // patch embedding: conv 14x14/14 as an f16 GEMM over im2col'd patches (ggml_conv_2d, minigpt4.cpp:1059) + bias
    launch_im2col(vi_img_, vi_patches_, 592, s, B);
    launch_gemm_f16(vi_patches_, 592, v_patch_w_, 592, B * 256, D, 592, v_patch_b_, nullptr, false, tabs_, vi_pe_, nullptr, D, s);
    launch_assemble_embeddings(v_cls_, vi_pe_, v_pos_, D, vi_x_, s, B);
    const float scale = 1.0f / sqrtf(88.0f);
    // ViT blocks.  attn.proj and mlp.fc2 (N = D: fewer 64x64 tiles than CUs, and fc2 has the longest K) run split-K; their deterministic reduce
    // also adds bias + residual and applies the LayerNorm that follows (norm2, the next block's norm1, ln_vision after the last block).
    const int sp = gemm_split_slices(D, splitk_proj_), sf = gemm_split_slices(M, splitk_fc2_);
    const size_t slab = (size_t)R * D;
    if (!vblocks_.empty()) launch_layernorm(vi_x_, vblocks_[0].n1w, vblocks_[0].n1b, R, D, nullptr, vi_ln_h_, s);
    for (size_t ib = 0; ib < vblocks_.size(); ib++) {
        const VBlock &b = vblocks_[ib];
        const bool last = ib + 1 == vblocks_.size();
        const float *nw = last ? v_lnv_w_ : vblocks_[ib + 1].n1w, *nb = last ? v_lnv_b_ : vblocks_[ib + 1].n1b;
        __half *nout = last ? vi_img_h_ : vi_ln_h_;
        if (qkv_head_major_) {   // q | k | v stored [3][head][R][88]: every head's rows contiguous for the attention kernel (round 6; MINIGPT4_QKV_HEAD_MAJOR=0: rows of 3 x D, A/B, bit-identical)
            launch_gemm_f16_head_major(vi_ln_h_, D, b.qkv_w, D, R, 3 * D, D, b.qkv_b, tabs_, vi_qkv_, D, 88, s);
            launch_attn_f32(vi_qkv_, 88, vi_qkv_ + (size_t)D * R, vi_qkv_ + (size_t)2 * D * R, 88, 257, 257, v_heads_, 88, scale, 0.0f, tabs_vis_, nullptr, vi_att_h_, D, s, B, R * 88, R * 88);
        } else {
        launch_gemm_f16(vi_ln_h_, D, b.qkv_w, D, R, 3 * D, D, b.qkv_b, nullptr, false, tabs_, vi_qkv_, nullptr, 3 * D, s);
        launch_attn_f32(vi_qkv_, 3 * D, vi_qkv_ + D, vi_qkv_ + 2 * D, 3 * D, 257, 257, v_heads_, 88, scale, 0.0f, tabs_vis_, nullptr, vi_att_h_, D, s, B);
        }
        if (sp > 1) {
            launch_gemm_f16_splitk(vi_att_h_, D, b.proj_w, D, R, D, D, sp, vi_slab_, slab, D, s);
            launch_splitk_reduce_ln(vi_slab_, sp, slab, b.proj_b, vi_x_, R, D, vi_x_, b.n2w, b.n2b, nullptr, vi_ln_h_, s);
        } else {
            launch_gemm_f16(vi_att_h_, D, b.proj_w, D, R, D, D, b.proj_b, vi_x_, false, tabs_, vi_x_, nullptr, D, s);
            launch_layernorm(vi_x_, b.n2w, b.n2b, R, D, nullptr, vi_ln_h_, s);
        }
        launch_gemm_f16(vi_ln_h_, D, b.fc1_w, D, R, M, D, b.fc1_b, nullptr, true, tabs_vis_, nullptr, vi_mlp_h_, M, s);
        if (sf > 1) {
            launch_gemm_f16_splitk(vi_mlp_h_, M, b.fc2_w, M, R, D, M, sf, vi_slab_, slab, D, s);
            launch_splitk_reduce_ln(vi_slab_, sf, slab, b.fc2_b, vi_x_, R, D, vi_x_, nw, nb, nullptr, nout, s);
        } else {
            launch_gemm_f16(vi_mlp_h_, M, b.fc2_w, M, R, D, M, b.fc2_b, vi_x_, false, tabs_, vi_x_, nullptr, D, s);
            launch_layernorm(vi_x_, nw, nb, R, D, nullptr, nout, s);
        }
    }
    if (vblocks_.empty()) launch_layernorm(vi_x_, v_lnv_w_, v_lnv_b_, R, D, nullptr, vi_img_h_, s);
    // Q-Former (masks are all-zero).  Its GEMMs have 32 rows per image: the skinny-M kernel (MINIGPT4_QF_SKINNY=0: the
    // 64x64-tile kernel, A/B)
    auto qgemm = [&](const __half *A, int lda, const __half *W, int ldw, int Mr, int N, int K, const float *bias, const float *residual, bool gelu, float *o, __half *oh, int ldo) {
        if (!(qf_skinny_ && launch_gemm_f16_skinny(A, lda, W, ldw, Mr, N, K, bias, residual, gelu, tabs_vis_, o, oh, ldo, s))) launch_gemm_f16(A, lda, W, ldw, Mr, N, K, bias, residual, gelu, tabs_vis_, o, oh, ldo, s);
    };
    // a 768-wide dense / output layer + residual + the LayerNorm behind it (NNSelfAttention's output block, NNBertEncoderLayer's query FFN output: minigpt4.cpp:1365-1400,
    // 1430-1461).  Round 6: K split over 2 (K = 768) or 4 (K = 3072) workgroups per tile with the LayerNorm in the slab reduce (qf_dense_ln below).
    auto dense_ln = [&](const __half *A, int lda, const __half *W, int K, const float *bias, const float *residual, const float *lw, const float *lb, float *o, __half *oh) {
        qf_dense_ln(A, lda, W, K, bias, residual, lw, lb, o, oh, RQ, s);
    };
    // the cross-attention K | V projections of every cross layer depend on the image features only (minigpt4.cpp:1148-1155): ONE [R x D] . [D x
    // n_cross * 1536] launch here instead of one per cross layer inside the loop; layer i reads its [R][1536] column slice (row stride n_cross *
    // 1536)
    const bool hoist = kv_hoist_ && v_ncross_ > 0 && v_kv_all_w_;
    const int ldkv = hoist ? v_ncross_ * 2 * H : 2 * H;
    if (hoist) launch_gemm_f16(vi_img_h_, D, v_kv_all_w_, D, R, v_ncross_ * 2 * H, D, v_kv_all_b_, nullptr, false, tabs_, vi_kv_, nullptr, ldkv, s);
    const bool folded = qf_fold_ && qf_folded_;                            // layer 0's image-independent head was evaluated at load time (fold_qformer_constants)
    if (!folded) launch_layernorm(vi_qtok_rep_, v_qeln_w_, v_qeln_b_, RQ, H, vi_hs_, vi_hs_h_, s);
    for (size_t il = 0; il < qlayers_.size(); il++) {
        const QLayer &L = qlayers_[il];
        const bool pre = folded && il == 0;
        const float *a1 = pre ? vi_c_a1_ : vi_a1_; const __half *a1_h = pre ? vi_c_a1_h_ : vi_a1_h_;
        if (!pre) {
            qgemm(vi_hs_h_, H, L.self.q_w, H, RQ, 3 * H, H, L.self.q_b, nullptr, false, vi_qq_, nullptr, 3 * H);
            launch_attn_f32(vi_qq_, 3 * H, vi_qq_ + H, vi_qq_ + 2 * H, 3 * H, NQ, NQ, 12, 64, 0.0f, 8.0f, tabs_vis_, nullptr, vi_ctx_h_, H, s, B);
            dense_ln(vi_ctx_h_, H, L.self.dense_w, H, L.self.dense_b, vi_hs_, L.self.ln_w, L.self.ln_b, vi_a1_, vi_a1_h_);
        }
        const float *ao = a1; const __half *ao_h = a1_h;
        if (L.has_cross) {
            const float *cq = pre ? vi_c_qq_ : vi_qq_;
            if (!pre) qgemm(vi_a1_h_, H, L.cross.q_w, H, RQ, H, H, L.cross.q_b, nullptr, false, vi_qq_, nullptr, H);
            const float *kv = hoist ? vi_kv_ + (size_t)L.cross_idx * 2 * H : vi_kv_;
            if (!hoist) launch_gemm_f16(vi_img_h_, D, L.cross.kv_w, D, R, 2 * H, D, L.cross.kv_b, nullptr, false, tabs_, vi_kv_, nullptr, 2 * H, s);
            launch_attn_f32(cq, H, kv, kv + H, ldkv, NQ, 257, 12, 64, 0.0f, 8.0f, tabs_vis_, nullptr, vi_ctx_h_, H, s, B);
            dense_ln(vi_ctx_h_, H, L.cross.dense_w, H, L.cross.dense_b, a1, L.cross.ln_w, L.cross.ln_b, vi_a2_, vi_a2_h_);
            ao = vi_a2_; ao_h = vi_a2_h_;
        }
        qgemm(ao_h, H, L.inter_w, H, RQ, v_qi_, H, L.inter_b, nullptr, true, nullptr, vi_im_h_, v_qi_);
        dense_ln(vi_im_h_, v_qi_, L.out_w, v_qi_, L.out_b, ao, L.oln_w, L.oln_b, vi_hs_, vi_hs_h_);
    }
    qgemm(vi_hs_h_, H, v_proj_w_, H, RQ, v_out_, H, v_proj_b_, nullptr, false, vi_out_, nullptr, v_out_);
    HIP_CHECK(hipEventRecord(eb, s));
    for (int b = 0; b < B; b++) HIP_CHECK(hipMemcpyAsync(out[b], vi_out_ + (size_t)b * NQ * v_out_, (size_t)NQ * v_out_ * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipEventElapsedTime(&last_encode_ms_, ea, eb));
    HIP_IGNORE(hipEventDestroy(ea)); HIP_IGNORE(hipEventDestroy(eb));
    return E_None;
}
int Engine::encode_image(const float *chw, float *out) { return encode_images(&chw, 1, &out); }

// out (fp32) / out_h (fp16) = LayerNorm(residual + (bias + A . W^T)) for a 768-wide layer of the Q-Former, `rows` rows.  The whole-K skinny launch is 48 workgroups on 256
// CUs, each pulling 32 rows of A and 16 of W over the full K through one CU's load path, followed by a standalone LayerNorm launch; here the K range is split over 2 (K = 768)
// or 4 (K >= 2048) workgroups per tile (raw partial sums into slabs) and the deterministic slab reduce adds bias + residual and normalises -- the same two launches, the
// first one 2-4 x wider.  The slice count depends on K only, never on the rows: image b of a batch equals the image encoded alone.  MINIGPT4_QF_SPLITK=0: the round 2-5 form.
void Engine::qf_dense_ln(const __half *A, int lda, const __half *W, int K, const float *bias, const float *residual, const float *ln_w, const float *ln_b, float *out, __half *out_h,
                         int rows, hipStream_t s) {
    const int H = 768, slices = K >= 2048 ? 4 : 2;
    if (qf_splitk_ && qf_skinny_ && launch_gemm_f16_skinny_splitk(A, lda, W, K, rows, H, K, slices, vi_slab_, (size_t)rows * H, H, s)) {
        launch_splitk_reduce_ln(vi_slab_, slices, (size_t)rows * H, bias, residual, rows, H, nullptr, ln_w, ln_b, out, out_h, s);
        return;
    }
    if (!(qf_skinny_ && launch_gemm_f16_skinny(A, lda, W, K, rows, H, K, bias, residual, false, tabs_, vi_d_, nullptr, H, s))) launch_gemm_f16(A, lda, W, K, rows, H, K, bias, residual, false, tabs_, vi_d_, nullptr, H, s);
    launch_layernorm(vi_d_, ln_w, ln_b, rows, H, out, out_h, s);
}

// What the Q-Former computes before it first looks at the image: hs = LayerNorm(query tokens) (minigpt4.cpp:2236-2246), layer 0's self-attention
// block on hs (NNSelfAttention + residual + LayerNorm, minigpt4.cpp:1365-1400) and, when layer 0 has cross-attention, its query projection.  None of
// it depends on the image, so it is evaluated ONCE here -- by the very launches encode_images would issue for one image (32 rows) -- and replicated
// for the images of a batch: six launches less per encode, bit-identical embeddings.  Fast f16 path only (parity mode and quantised / f32 vision
// files run their own chains).
void Engine::fold_qformer_constants() {
    qf_folded_ = false;
    if (!qf_fold_ || v_generic_ || qlayers_.empty() || !vi_c_a1_) return;
    hipStream_t s = stream_;
    const int H = 768, NQ = v_nq_;
    const QLayer &L = qlayers_[0];
    auto qgemm = [&](const __half *A, int lda, const __half *W, int ldw, int Mr, int N, int K, const float *bias, const float *residual, bool gelu, float *o, __half *oh, int ldo) {
        if (!(qf_skinny_ && launch_gemm_f16_skinny(A, lda, W, ldw, Mr, N, K, bias, residual, gelu, tabs_vis_, o, oh, ldo, s))) launch_gemm_f16(A, lda, W, ldw, Mr, N, K, bias, residual, gelu, tabs_vis_, o, oh, ldo, s);
    };
    launch_layernorm(vi_qtok_rep_, v_qeln_w_, v_qeln_b_, NQ, H, vi_hs_, vi_hs_h_, s);
    qgemm(vi_hs_h_, H, L.self.q_w, H, NQ, 3 * H, H, L.self.q_b, nullptr, false, vi_qq_, nullptr, 3 * H);
    launch_attn_f32(vi_qq_, 3 * H, vi_qq_ + H, vi_qq_ + 2 * H, 3 * H, NQ, NQ, 12, 64, 0.0f, 8.0f, tabs_vis_, nullptr, vi_ctx_h_, H, s, 1);
    qf_dense_ln(vi_ctx_h_, H, L.self.dense_w, H, L.self.dense_b, vi_hs_, L.self.ln_w, L.self.ln_b, vi_c_a1_, vi_c_a1_h_, NQ, s);   // the launches encode_images issues
    if (L.has_cross) qgemm(vi_c_a1_h_, H, L.cross.q_w, H, NQ, H, H, L.cross.q_b, nullptr, false, vi_c_qq_, nullptr, H);
    HIP_CHECK(hipStreamSynchronize(s));
    const size_t n = (size_t)NQ * H;
    for (size_t b = 1; b < (size_t)VISION_BATCH_MAX; b++) {
        HIP_CHECK(hipMemcpy(vi_c_a1_ + b * n, vi_c_a1_, n * 4, hipMemcpyDeviceToDevice)); HIP_CHECK(hipMemcpy(vi_c_a1_h_ + b * n, vi_c_a1_h_, n * 2, hipMemcpyDeviceToDevice));
        if (L.has_cross) HIP_CHECK(hipMemcpy(vi_c_qq_ + b * n, vi_c_qq_, n * 4, hipMemcpyDeviceToDevice));
    }
    qf_folded_ = true;
}

}  // namespace mg4
