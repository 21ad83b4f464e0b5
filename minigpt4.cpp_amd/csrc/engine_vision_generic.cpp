// Image path for vision files whose Linear weights are not all F16: `convert.py --ftype f32` files (F32 Linears) and files re-quantised by
// minigpt4_quantize_model (reference minigpt4.cpp:2817-2982: Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0 / k-quants for the ViT and Q-Former Linears, everything else
// untouched).  The reference evaluates such a file with the same graph (minigpt4.cpp:2094-2363); only `ggml_mul_mat` changes its arithmetic with the weight
// type: the activation rows are converted to the type's vec_dot_type (Q8_0 / Q8_1 / Q8_K blocks, fp16 for F16, nothing for F32) and block dots are exact
// integers.  That is exactly what the LLM kernels implement, so every Linear here is: activation conversion (k_rms_quant without a norm) -> launch_mul_mat
// (int8-MFMA tiles for Q4_0 / k-quants from 5 rows, v_dot4 tiles otherwise, the MFMA f16 GEMM for F16) -> bias / GELU / residual epilogue.  Activations stay fp32
// between kernels, like ggml's tensors.  LayerNorm, attention, the conv patch embedding and the batching over images are shared with the F16 path.
#include "engine.hpp"
#include "quantize.hpp"

#include <algorithm>
#include <cmath>

namespace mg4 {

int Engine::load_vision_generic() {
    auto fail = [&](const std::string &msg, int code) { set_last_error("vision file: " + msg); MG4_ERR("%s", last_error().c_str()); return code; };
    const uint8_t *fb = vis_.mf.data;
    size_t max_raw = 0;
    for (auto &m : vis_.models) for (auto &t : m.second) if (t.second.type != GT_F16 && t.second.type != GT_F32)
        max_raw = std::max(max_raw, t.second.type == GT_Q3_K ? t.second.nbytes / 110 * 210 : t.second.nbytes);
    std::vector<uint8_t> conv;
    struct Stage { void *p = nullptr; ~Stage() { if (p) HIP_IGNORE(hipFree(p)); } } stage;
    if (max_raw) HIP_CHECK(hipMalloc(&stage.p, max_raw));
    std::string bad; int bad_code = 0;
    auto lin = [&](const std::string &model, const std::string &name, int64_t n_in, int64_t n_out, GLin &L) {
        if (bad_code) return;
        const TensorMeta *t = vis_.find(model, name);
        if (!t || t->ne.size() != 2 || t->ne[0] != n_in || t->ne[1] != n_out) { bad = model + "." + name; bad_code = E_LoadModelFileHeader; return; }
        const int wt = t->type == GT_Q3_K ? GT_Q6_K : t->type;                  // Q3_K: exact Q6_K image built on the host (quantize.hpp), like the LLM loader
        if (!qweight_supported(wt) || n_in % gt_block(t->type)) {
            bad = model + "." + name + ": type " + gt_name(t->type) + " with " + std::to_string(n_in) + " columns is not supported by the gfx950 kernels"; bad_code = E_LoadModelMiniGPT4DataType; return; }
        QWeight plan;
        const size_t need = plan_qweight(wt, (int)n_out, (int)n_in, plan, nullptr);
        uint8_t *base = vis_arena_.take(need);
        plan_qweight(wt, (int)n_out, (int)n_in, L.w, base);
        if (t->type == GT_F16 || t->type == GT_F32) { HIP_CHECK(hipMemcpy(base, fb + t->offset, t->nbytes, hipMemcpyHostToDevice)); return; }
        if (t->type == GT_Q3_K) {
            conv.resize(t->nbytes / 110 * 210);
            q3k_to_q6k(fb + t->offset, conv.data(), t->nbytes / 110);
            HIP_CHECK(hipMemcpy(stage.p, conv.data(), conv.size(), hipMemcpyHostToDevice));
        } else
            HIP_CHECK(hipMemcpy(stage.p, fb + t->offset, t->nbytes, hipMemcpyHostToDevice));
        launch_repack(static_cast<const uint8_t *>(stage.p), L.w, stream_);
        HIP_CHECK(hipStreamSynchronize(stream_));
    };
    const int D = v_D_, M = v_M_;
    gblocks_.assign((size_t)v_depth_, GBlock{});
    for (int i = 0; i < v_depth_; i++) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        GBlock &b = gblocks_[(size_t)i];
        lin("visual_encoder", p + "attn.qkv.weight", D, 3 * (int64_t)D, b.qkv); lin("visual_encoder", p + "attn.proj.weight", D, D, b.proj);
        lin("visual_encoder", p + "mlp.fc1.weight", D, M, b.fc1); lin("visual_encoder", p + "mlp.fc2.weight", M, D, b.fc2);
    }
    gql_.assign((size_t)v_ql_, GQLayer{});
    for (int i = 0; i < v_ql_; i++) {
        const std::string p = "bert.encoder.layer." + std::to_string(i) + ".";
        GQLayer &L = gql_[(size_t)i];
        lin("Qformer", p + "attention.self.query.weight", 768, 768, L.self.q); lin("Qformer", p + "attention.self.key.weight", 768, 768, L.self.k);
        lin("Qformer", p + "attention.self.value.weight", 768, 768, L.self.v); lin("Qformer", p + "attention.output.dense.weight", 768, 768, L.self.dense);
        if (qlayers_[(size_t)i].has_cross) {
            lin("Qformer", p + "crossattention.self.query.weight", 768, 768, L.cross.q); lin("Qformer", p + "crossattention.self.key.weight", D, 768, L.cross.k);
            lin("Qformer", p + "crossattention.self.value.weight", D, 768, L.cross.v); lin("Qformer", p + "crossattention.output.dense.weight", 768, 768, L.cross.dense);
        }
        lin("Qformer", p + "intermediate_query.dense.weight", 768, v_qi_, L.inter); lin("Qformer", p + "output_query.dense.weight", v_qi_, 768, L.out);
    }
    lin("llama_proj", "weight", 768, v_out_, gproj_);
    if (bad_code) return fail(bad, bad_code);
    MG4_INFO("vision Linears are not all F16: generic path (activations converted to each weight type's vec_dot_type; int8 dot / MFMA kernels of the LLM path)");
    return E_None;
}

void Engine::alloc_vision_generic() {
    const size_t R = (size_t)VISION_BATCH_MAX * 257, RQ = (size_t)VISION_BATCH_MAX * (size_t)v_nq_, D = (size_t)v_D_, M = (size_t)v_M_;
    const size_t Kmax = std::max(std::max(D, M), (size_t)std::max(v_qi_, 768)), Nmax = std::max(std::max(3 * D, M), (size_t)std::max(std::max(v_qi_, v_out_), 2304));
    size_t total = 1 << 20;
    auto sz = [&](size_t b) { total += (b + 255) / 256 * 256 + 256; };
    sz(R * D * 4); sz(R * D * 4); sz(R * M * 4); sz(R * D * 4); sz(R * Nmax * 4); sz(RQ * 768 * 4); sz(RQ * (size_t)v_qi_ * 4);
    sz(2 * R * Kmax); sz(R * (Kmax / 256 + 1) * 4); sz(R * (Kmax / 16 + 1) * 2); sz(R * (Kmax / 16 + 16)); sz(4 * R * (Kmax / 32 + 1) * 4); sz(R * Kmax * 2); sz(R * Kmax * 4);
    vgen_arena_.alloc(total);
    auto takef = [&](size_t n) { return reinterpret_cast<float *>(vgen_arena_.take(n * 4)); };
    vg_ln_ = takef(R * D); vg_att_ = takef(R * D); vg_mlp_ = takef(R * M); vg_img_ = takef(R * D); vg_tmp_ = takef(R * Nmax); vg_ctx_ = takef(RQ * 768); vg_im_ = takef(RQ * (size_t)v_qi_);
    vact_.q8k = reinterpret_cast<int8_t *>(vgen_arena_.take(R * Kmax)); vact_.q80 = reinterpret_cast<int8_t *>(vgen_arena_.take(R * Kmax));
    vact_.dk = takef(R * (Kmax / 256 + 1)); vact_.bsk = reinterpret_cast<int16_t *>(vgen_arena_.take(R * (Kmax / 16 + 1) * 2));
    vact_.bsq = reinterpret_cast<int8_t *>(vgen_arena_.take(R * (Kmax / 16 + 16)));
    vact_.d0 = takef(R * (Kmax / 32 + 1)); vact_.d1 = takef(R * (Kmax / 32 + 1)); vact_.s1 = takef(R * (Kmax / 32 + 1));
    vact_.sum0 = reinterpret_cast<int *>(vgen_arena_.take(R * (Kmax / 32 + 1) * 4));
    vact_.xh = reinterpret_cast<__half *>(vgen_arena_.take(R * Kmax * 2)); vact_.xf = takef(R * Kmax);
}

// y = [residual +] gelu?(bias + W . convert(x)) for `rows` fp32 rows (NNLinear::forward, minigpt4.cpp:1020-1030, with the weight type's mul_mat arithmetic)
void Engine::glinear(const GLin &L, const float *x, int rows, const float *bias, bool gelu, const float *residual, float *out, hipStream_t s) {
    launch_rms_quant(x, nullptr, rows, L.w.cols, vact_, act_mask_for(L.w.type), s);
    launch_mul_mat(L.w, vact_, rows, vg_tmp_, L.w.rows, nullptr, s);
    launch_lin_epilogue(vg_tmp_, bias, residual, gelu, tabs_, rows, L.w.rows, out, nullptr, s);
}

int Engine::encode_images_generic(const float *const *chw, int B, float *const *out) {
    hipStream_t s = stream_;
    const int D = v_D_, NQ = v_nq_, H = 768;
    const int R = B * 257, RQ = B * NQ;
    hipEvent_t ea, eb; HIP_CHECK(hipEventCreate(&ea)); HIP_CHECK(hipEventCreate(&eb));
    for (int b = 0; b < B; b++) HIP_CHECK(hipMemcpyAsync(vi_img_ + (size_t)b * 3 * 224 * 224, chw[b], 3 * 224 * 224 * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipEventRecord(ea, s));
    // patch embedding: always the F16 conv kernel (ggml_conv_2d computes in fp16, minigpt4.cpp:1059)
    launch_im2col(vi_img_, vi_patches_, 592, s, B);
    launch_gemm_f16(vi_patches_, 592, v_patch_w_, 592, B * 256, D, 592, v_patch_b_, nullptr, false, tabs_, vi_pe_, nullptr, D, s);
    launch_assemble_embeddings(v_cls_, vi_pe_, v_pos_, D, vi_x_, s, B);
    const float scale = 1.0f / sqrtf(88.0f);
    for (size_t ib = 0; ib < vblocks_.size(); ib++) {
        const VBlock &b = vblocks_[ib]; const GBlock &g = gblocks_[ib];
        launch_layernorm(vi_x_, b.n1w, b.n1b, R, D, vg_ln_, nullptr, s);
        glinear(g.qkv, vg_ln_, R, b.qkv_b, false, nullptr, vi_qkv_, s);
        launch_attn_f32(vi_qkv_, 3 * D, vi_qkv_ + D, vi_qkv_ + 2 * D, 3 * D, 257, 257, v_heads_, 88, scale, 0.0f, tabs_, vg_att_, nullptr, D, s, B);
        glinear(g.proj, vg_att_, R, b.proj_b, false, vi_x_, vi_x_, s);
        launch_layernorm(vi_x_, b.n2w, b.n2b, R, D, vg_ln_, nullptr, s);
        glinear(g.fc1, vg_ln_, R, b.fc1_b, true, nullptr, vg_mlp_, s);
        glinear(g.fc2, vg_mlp_, R, b.fc2_b, false, vi_x_, vi_x_, s);
    }
    launch_layernorm(vi_x_, v_lnv_w_, v_lnv_b_, R, D, vg_img_, nullptr, s);
    // Q-Former
    launch_layernorm(vi_qtok_rep_, v_qeln_w_, v_qeln_b_, RQ, H, vi_hs_, nullptr, s);
    for (size_t il = 0; il < qlayers_.size(); il++) {
        const QLayer &L = qlayers_[il]; const GQLayer &G = gql_[il];
        // self attention: q | k | v written side by side ([RQ][3H]) so that the attention kernel sees the layout of the F16 path
        {
            launch_rms_quant(vi_hs_, nullptr, RQ, H, vact_, act_mask_for(G.self.q.w.type) | act_mask_for(G.self.k.w.type) | act_mask_for(G.self.v.w.type), s);
            const GLin *qkv[3] = {&G.self.q, &G.self.k, &G.self.v};
            for (int j = 0; j < 3; j++) {
                launch_mul_mat(qkv[j]->w, vact_, RQ, vg_tmp_, H, nullptr, s);
                launch_lin_epilogue(vg_tmp_, L.self.q_b + (size_t)j * H, nullptr, false, tabs_, RQ, H, vg_tmp_, nullptr, s);
                HIP_CHECK(hipMemcpy2DAsync(vi_qq_ + (size_t)j * H, (size_t)3 * H * 4, vg_tmp_, (size_t)H * 4, (size_t)H * 4, (size_t)RQ, hipMemcpyDeviceToDevice, s));
            }
        }
        launch_attn_f32(vi_qq_, 3 * H, vi_qq_ + H, vi_qq_ + 2 * H, 3 * H, NQ, NQ, 12, 64, 0.0f, 8.0f, tabs_, vg_ctx_, nullptr, H, s, B);
        glinear(G.self.dense, vg_ctx_, RQ, L.self.dense_b, false, vi_hs_, vi_d_, s);
        launch_layernorm(vi_d_, L.self.ln_w, L.self.ln_b, RQ, H, vi_a1_, nullptr, s);
        const float *ao = vi_a1_;
        if (L.has_cross) {
            glinear(G.cross.q, vi_a1_, RQ, L.cross.q_b, false, nullptr, vi_qq_, s);
            launch_rms_quant(vg_img_, nullptr, R, D, vact_, act_mask_for(G.cross.k.w.type) | act_mask_for(G.cross.v.w.type), s);
            const GLin *kv[2] = {&G.cross.k, &G.cross.v};
            for (int j = 0; j < 2; j++) {
                launch_mul_mat(kv[j]->w, vact_, R, vg_tmp_, H, nullptr, s);
                launch_lin_epilogue(vg_tmp_, L.cross.kv_b + (size_t)j * H, nullptr, false, tabs_, R, H, vg_tmp_, nullptr, s);
                HIP_CHECK(hipMemcpy2DAsync(vi_kv_ + (size_t)j * H, (size_t)2 * H * 4, vg_tmp_, (size_t)H * 4, (size_t)H * 4, (size_t)R, hipMemcpyDeviceToDevice, s));
            }
            launch_attn_f32(vi_qq_, H, vi_kv_, vi_kv_ + H, 2 * H, NQ, 257, 12, 64, 0.0f, 8.0f, tabs_, vg_ctx_, nullptr, H, s, B);
            glinear(G.cross.dense, vg_ctx_, RQ, L.cross.dense_b, false, vi_a1_, vi_d_, s);
            launch_layernorm(vi_d_, L.cross.ln_w, L.cross.ln_b, RQ, H, vi_a2_, nullptr, s);
            ao = vi_a2_;
        }
        glinear(G.inter, ao, RQ, L.inter_b, true, nullptr, vg_im_, s);
        glinear(G.out, vg_im_, RQ, L.out_b, false, ao, vi_d_, s);
        launch_layernorm(vi_d_, L.oln_w, L.oln_b, RQ, H, vi_hs_, nullptr, s);
    }
    glinear(gproj_, vi_hs_, RQ, v_proj_b_, false, nullptr, vi_out_, s);
    HIP_CHECK(hipEventRecord(eb, s));
    for (int b = 0; b < B; b++) HIP_CHECK(hipMemcpyAsync(out[b], vi_out_ + (size_t)b * NQ * v_out_, (size_t)NQ * v_out_ * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipEventElapsedTime(&last_encode_ms_, ea, eb));
    HIP_IGNORE(hipEventDestroy(ea)); HIP_IGNORE(hipEventDestroy(eb));
    return E_None;
}

// ====================================================================================================================
// MINIGPT4_PARITY image path: the generic graph above with every fp32 accumulation in the CPU oracle's order (oracle/refcpu.c orc_vision_encode) -- Linear = activation
// conversion + k_mul_mat_ref (one sequential chain per output: ggml_vec_dot_f16 / the block dots of a quantised file) + the bias / GELU / residual epilogue, LayerNorm with
// its two sums added in element order, attention through k_attn_vref.  Works for F16 files (the Linears are viewed as F16 QWeights in place: nothing is copied) and for
// the quantised / F32 files of the generic path.  The image embedding equals the oracle's bit for bit (tests/test_gpu_paritymode.py); slow by design.
// ====================================================================================================================
void Engine::build_vision_views() {
    if (v_generic_ || !gblocks_.empty()) return;                           // generic files already carry QWeights
    auto view = [](const __half *p, int n_in, int n_out) { GLin L; L.w.type = GT_F16; L.w.rows = n_out; L.w.cols = n_in; L.w.qs = reinterpret_cast<const uint8_t *>(p); L.w.bytes = (size_t)n_in * n_out * 2; return L; };
    const int D = v_D_, M = v_M_, H = 768;
    gblocks_.assign(vblocks_.size(), GBlock{});
    for (size_t i = 0; i < vblocks_.size(); i++) {
        const VBlock &b = vblocks_[i];
        gblocks_[i].qkv = view(b.qkv_w, D, 3 * D); gblocks_[i].proj = view(b.proj_w, D, D); gblocks_[i].fc1 = view(b.fc1_w, D, M); gblocks_[i].fc2 = view(b.fc2_w, M, D);
    }
    gql_.assign(qlayers_.size(), GQLayer{});
    for (size_t i = 0; i < qlayers_.size(); i++) {
        const QLayer &L = qlayers_[i]; GQLayer &G = gql_[i];
        G.self.q = view(L.self.q_w, H, H); G.self.k = view(L.self.q_w + (size_t)H * H, H, H); G.self.v = view(L.self.q_w + (size_t)2 * H * H, H, H);   // stored as one [2304][768] matrix
        G.self.dense = view(L.self.dense_w, H, H);
        if (L.has_cross) {
            G.cross.q = view(L.cross.q_w, H, H); G.cross.k = view(L.cross.kv_w, D, H); G.cross.v = view(L.cross.kv_w + (size_t)H * D, D, H);            // [1536][D]
            G.cross.dense = view(L.cross.dense_w, H, H);
        }
        G.inter = view(L.inter_w, H, v_qi_); G.out = view(L.out_w, v_qi_, H);
    }
    gproj_ = view(v_proj_w_, H, v_out_);
}

int Engine::encode_images_ref(const float *const *chw, int B, float *const *out) {
    hipStream_t s = stream_;
    build_vision_views();
    if (!vg_ln_) alloc_vision_generic();                                   // fp32 activation buffers + the activation planes of the generic path
    const int D = v_D_, NQ = v_nq_, H = 768;
    const int R = B * 257, RQ = B * NQ;
    auto rlinear = [&](const GLin &L, const float *x, int rows, const float *bias, bool gelu, const float *residual, float *o) {
        launch_rms_quant(x, nullptr, rows, L.w.cols, vact_, act_mask_for(L.w.type), s);
        launch_mul_mat_ref(L.w, vact_, rows, vg_tmp_, L.w.rows, nullptr, s);
        launch_lin_epilogue(vg_tmp_, bias, residual, gelu, tabs_, rows, L.w.rows, o, nullptr, s);
    };
    hipEvent_t ea, eb; HIP_CHECK(hipEventCreate(&ea)); HIP_CHECK(hipEventCreate(&eb));
    for (int b = 0; b < B; b++) HIP_CHECK(hipMemcpyAsync(vi_img_ + (size_t)b * 3 * 224 * 224, chw[b], 3 * 224 * 224 * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipEventRecord(ea, s));
    // patch embedding: fp16 im2col rows x the F16 conv kernel ([D][592], columns 588.. are zero on both sides), one sequential chain per output, + bias
    launch_im2col(vi_img_, vi_patches_, 592, s, B);
    {
        QWeight Wp; Wp.type = GT_F16; Wp.rows = D; Wp.cols = 592; Wp.qs = reinterpret_cast<const uint8_t *>(v_patch_w_); Wp.bytes = (size_t)D * 592 * 2;
        ActQ Ap{}; Ap.xh = vi_patches_;
        launch_mul_mat_ref(Wp, Ap, B * 256, vi_pe_, D, nullptr, s);
        launch_lin_epilogue(vi_pe_, v_patch_b_, nullptr, false, tabs_, B * 256, D, vi_pe_, nullptr, s);
    }
    launch_assemble_embeddings(v_cls_, vi_pe_, v_pos_, D, vi_x_, s, B);
    const float scale = 1.0f / sqrtf(88.0f);
    for (size_t ib = 0; ib < vblocks_.size(); ib++) {
        const VBlock &b = vblocks_[ib]; const GBlock &g = gblocks_[ib];
        launch_layernorm(vi_x_, b.n1w, b.n1b, R, D, vg_ln_, nullptr, s, true);
        rlinear(g.qkv, vg_ln_, R, b.qkv_b, false, nullptr, vi_qkv_);
        launch_attn_vref(vi_qkv_, 3 * D, vi_qkv_ + D, vi_qkv_ + 2 * D, 3 * D, 257, 257, v_heads_, 88, scale, 0.0f, tabs_, vg_att_, D, s, B);
        rlinear(g.proj, vg_att_, R, b.proj_b, false, vi_x_, vi_x_);
        launch_layernorm(vi_x_, b.n2w, b.n2b, R, D, vg_ln_, nullptr, s, true);
        rlinear(g.fc1, vg_ln_, R, b.fc1_b, true, nullptr, vg_mlp_);
        rlinear(g.fc2, vg_mlp_, R, b.fc2_b, false, vi_x_, vi_x_);
    }
    launch_layernorm(vi_x_, v_lnv_w_, v_lnv_b_, R, D, vg_img_, nullptr, s, true);
    launch_layernorm(vi_qtok_rep_, v_qeln_w_, v_qeln_b_, RQ, H, vi_hs_, nullptr, s, true);
    for (size_t il = 0; il < qlayers_.size(); il++) {
        const QLayer &L = qlayers_[il]; const GQLayer &G = gql_[il];
        {   // self attention: q | k | v side by side ([RQ][3H]), each its own Linear as in the reference graph
            const GLin *qkv[3] = {&G.self.q, &G.self.k, &G.self.v};
            for (int j = 0; j < 3; j++) {
                rlinear(*qkv[j], vi_hs_, RQ, L.self.q_b + (size_t)j * H, false, nullptr, vg_tmp_);
                HIP_CHECK(hipMemcpy2DAsync(vi_qq_ + (size_t)j * H, (size_t)3 * H * 4, vg_tmp_, (size_t)H * 4, (size_t)H * 4, (size_t)RQ, hipMemcpyDeviceToDevice, s));
            }
        }
        launch_attn_vref(vi_qq_, 3 * H, vi_qq_ + H, vi_qq_ + 2 * H, 3 * H, NQ, NQ, 12, 64, 0.0f, 8.0f, tabs_, vg_ctx_, H, s, B);
        rlinear(G.self.dense, vg_ctx_, RQ, L.self.dense_b, false, vi_hs_, vi_d_);
        launch_layernorm(vi_d_, L.self.ln_w, L.self.ln_b, RQ, H, vi_a1_, nullptr, s, true);
        const float *ao = vi_a1_;
        if (L.has_cross) {
            rlinear(G.cross.q, vi_a1_, RQ, L.cross.q_b, false, nullptr, vi_qq_);
            const GLin *kv[2] = {&G.cross.k, &G.cross.v};
            for (int j = 0; j < 2; j++) {
                rlinear(*kv[j], vg_img_, R, L.cross.kv_b + (size_t)j * H, false, nullptr, vg_tmp_);
                HIP_CHECK(hipMemcpy2DAsync(vi_kv_ + (size_t)j * H, (size_t)2 * H * 4, vg_tmp_, (size_t)H * 4, (size_t)H * 4, (size_t)R, hipMemcpyDeviceToDevice, s));
            }
            launch_attn_vref(vi_qq_, H, vi_kv_, vi_kv_ + H, 2 * H, NQ, 257, 12, 64, 0.0f, 8.0f, tabs_, vg_ctx_, H, s, B);
            rlinear(G.cross.dense, vg_ctx_, RQ, L.cross.dense_b, false, vi_a1_, vi_d_);
            launch_layernorm(vi_d_, L.cross.ln_w, L.cross.ln_b, RQ, H, vi_a2_, nullptr, s, true);
            ao = vi_a2_;
        }
        rlinear(G.inter, ao, RQ, L.inter_b, true, nullptr, vg_im_);
        rlinear(G.out, vg_im_, RQ, L.out_b, false, ao, vi_d_);
        launch_layernorm(vi_d_, L.oln_w, L.oln_b, RQ, H, vi_hs_, nullptr, s, true);
    }
    rlinear(gproj_, vi_hs_, RQ, v_proj_b_, false, nullptr, vi_out_);
    HIP_CHECK(hipEventRecord(eb, s));
    for (int b = 0; b < B; b++) HIP_CHECK(hipMemcpyAsync(out[b], vi_out_ + (size_t)b * NQ * v_out_, (size_t)NQ * v_out_ * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipEventElapsedTime(&last_encode_ms_, ea, eb));
    HIP_IGNORE(hipEventDestroy(ea)); HIP_IGNORE(hipEventDestroy(eb));
    return E_None;
}

}  // namespace mg4
