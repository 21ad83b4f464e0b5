// extern "C" hooks of libminigpt4_test.so ONLY (include/minigpt4_amd_test.h): single-kernel parity hooks, micro-benchmarks, hardware probes and host-only helpers for
// the CPU test tier.  The product library (libminigpt4.so) is linked without this file: none of these symbols ship.
#include <sys/stat.h>

#include <chrono>
#include <cstring>

#include "api_internal.hpp"
#include "imageio.hpp"
#include "quantize.hpp"
#include "minigpt4_amd.h"
#include "minigpt4_amd_test.h"

using namespace mg4;
using namespace mg4::apiutil;

namespace mg4 {   // probe_kernels.hip
float probe_valu_ns(int op, int waves_per_simd, int iters, int cus);
float probe_grid_barrier_us(int n_blocks, int iters, unsigned *errors_out);
int parse_dist_env_for_test(int *world, int *rank, char *id_file, size_t cap, char *err, size_t err_cap);
float probe_dma_GBps(int form, int policy, int waves, int fill, int depth, int deal, size_t total_bytes);
void launch_fill_random(void *p, size_t bytes, unsigned seed, hipStream_t s);
}

extern "C" {

int minigpt4_amd_copy_arenas(struct MiniGPT4Context *dst, struct MiniGPT4Context *src) {   // single-GPU stand-in for the broadcast (tests): both weight arenas, device to device
    if (!dst || !src) return 1;
    return guarded(3, [&]() -> int {
        Engine *d = E_(dst), *s = E_(src);
        if (d->llm_arena_bytes() != s->llm_arena_bytes() || d->vision_arena_bytes() != s->vision_arena_bytes()) { set_last_error("arena sizes differ"); return 2; }
        HIP_CHECK(hipMemcpy(d->llm_arena_ptr(), s->llm_arena_ptr(), s->llm_arena_bytes(), hipMemcpyDeviceToDevice));
        HIP_CHECK(hipMemcpy(d->vision_arena_ptr(), s->vision_arena_ptr(), s->vision_arena_bytes(), hipMemcpyDeviceToDevice));
        return 0;
    });
}

// ---- single-kernel hooks ---------------------------------------------------------------------------------------------------

int minigpt4_amd_timeline(unsigned long long *out, int max_workgroups) { return (out && max_workgroups > 0) ? read_matvec_timeline(out, max_workgroups) : -1; }

int minigpt4_amd_convert_q3k_q6k(const void *src, void *dst, int64_t n_blocks) {
    if (!src || !dst || n_blocks < 0) return 1;
    q3k_to_q6k(static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), (size_t)n_blocks);
    return 0;
}

static int test_mul_mat_impl(int ggml_type, const void *raw_w, int64_t n_in, int64_t n_out, const float *x, int64_t N, float *y, bool ref);
int minigpt4_amd_test_mul_mat(int ggml_type, const void *raw_w, int64_t n_in, int64_t n_out, const float *x, int64_t N, float *y) { return test_mul_mat_impl(ggml_type, raw_w, n_in, n_out, x, N, y, false); }
int minigpt4_amd_test_mul_mat_ref(int ggml_type, const void *raw_w, int64_t n_in, int64_t n_out, const float *x, int64_t N, float *y) { return test_mul_mat_impl(ggml_type, raw_w, n_in, n_out, x, N, y, true); }
static int test_mul_mat_impl(int ggml_type, const void *raw_w, int64_t n_in, int64_t n_out, const float *x, int64_t N, float *y, bool ref) {
    if (ggml_type == GT_Q3_K && raw_w && n_in > 0 && n_out > 0 && n_in % 256 == 0) {   // the engine's load path: exact Q6_K image (quantize.hpp)
        std::vector<uint8_t> q6((size_t)(n_in / 256 * n_out) * 210);
        q3k_to_q6k(static_cast<const uint8_t *>(raw_w), q6.data(), (size_t)(n_in / 256 * n_out));
        return test_mul_mat_impl(GT_Q6_K, q6.data(), n_in, n_out, x, N, y, ref);
    }
    if (!raw_w || !x || !y || n_in <= 0 || n_out <= 0 || N <= 0 || !qweight_supported(ggml_type) || n_in % gt_block(ggml_type)) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const size_t raw_bytes = gt_nbytes(ggml_type, (size_t)(n_in * n_out));
        QWeight W, plan;
        const size_t need = plan_qweight(ggml_type, (int)n_out, (int)n_in, plan, nullptr);
        DevBuf d_raw(raw_bytes), d_planes(need), d_x((size_t)(N * n_in) * 4), d_y((size_t)(N * n_out) * 4);
        plan_qweight(ggml_type, (int)n_out, (int)n_in, W, d_planes.as<uint8_t>());
        HIP_CHECK(hipMemcpy(d_raw.p, raw_w, raw_bytes, hipMemcpyHostToDevice));
        if (ggml_type == GT_F16 || ggml_type == GT_F32) HIP_CHECK(hipMemcpy(d_planes.p, raw_w, raw_bytes, hipMemcpyHostToDevice)); else launch_repack(d_raw.as<uint8_t>(), W, nullptr);
        HIP_CHECK(hipMemcpy(d_x.p, x, (size_t)(N * n_in) * 4, hipMemcpyHostToDevice));
        ActQ A; std::vector<std::unique_ptr<DevBuf>> keep; alloc_act(A, keep, (size_t)N, (size_t)n_in);
        launch_rms_quant(d_x.as<float>(), nullptr, (int)N, (int)n_in, A, act_mask_for(ggml_type), nullptr);
        if (ref) launch_mul_mat_ref(W, A, (int)N, d_y.as<float>(), (int)n_out, nullptr, nullptr);
        else launch_mul_mat(W, A, (int)N, d_y.as<float>(), (int)n_out, nullptr, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(y, d_y.p, (size_t)(N * n_out) * 4, hipMemcpyDeviceToHost));
        return 0;
    });
}

// The prefill launch of Engine::forward for N > 4 rows (mmq2_kernels.hip): n_mat equally shaped matrices (rows of `raw_w` back to back) against N rows in ONE launch, optional
// residual ([n_mat][N][n_out]), optional forced K split (ks > 1: partial sums combined in fixed order).  y: [n_mat][N][n_out].  Returns 4 when the kernels refuse the shape.
int minigpt4_amd_test_mmq2(int ggml_type, const void *raw_w, int n_mat, int64_t n_in, int64_t n_out, const float *x, int64_t N, const float *residual, int ks, int generation, float *y) {
    if (!raw_w || !x || !y || n_in <= 0 || n_out <= 0 || N <= 0 || n_mat < 1 || n_mat > 3 || !qweight_supported(ggml_type) || n_in % gt_block(ggml_type)) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const size_t raw_each = gt_nbytes(ggml_type, (size_t)(n_in * n_out)), out_each = (size_t)(N * n_out);
        QWeight W[3], plan;
        const size_t need = plan_qweight(ggml_type, (int)n_out, (int)n_in, plan, nullptr);
        DevBuf d_raw(raw_each), d_planes(need * (size_t)n_mat + 1024), d_x((size_t)(N * n_in) * 4), d_y(out_each * n_mat * 4), d_res(out_each * n_mat * 4), d_ws(out_each * n_mat * 16 * 4);
        for (int i = 0; i < n_mat; i++) {
            plan_qweight(ggml_type, (int)n_out, (int)n_in, W[i], d_planes.as<uint8_t>() + (size_t)i * need);
            HIP_CHECK(hipMemcpy(d_raw.p, static_cast<const uint8_t *>(raw_w) + (size_t)i * raw_each, raw_each, hipMemcpyHostToDevice));
            launch_repack(d_raw.as<uint8_t>(), W[i], nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        }
        HIP_CHECK(hipMemcpy(d_x.p, x, (size_t)(N * n_in) * 4, hipMemcpyHostToDevice));
        if (residual) HIP_CHECK(hipMemcpy(d_res.p, residual, out_each * n_mat * 4, hipMemcpyHostToDevice));
        ActQ A; std::vector<std::unique_ptr<DevBuf>> keep; alloc_act(A, keep, (size_t)N, (size_t)n_in);
        launch_rms_quant(d_x.as<float>(), nullptr, (int)N, (int)n_in, A, act_mask_for(ggml_type), nullptr);
        const QWeight *Wp[3]; float *Yp[3]; const float *Rp[3];
        for (int i = 0; i < n_mat; i++) { Wp[i] = &W[i]; Yp[i] = d_y.as<float>() + (size_t)i * out_each; Rp[i] = d_res.as<float>() + (size_t)i * out_each; }
        A.ws = d_ws.as<float>(); A.ws_floats = out_each * n_mat * 16;
        bool ok;
        if (ggml_type == GT_F16) {   // the F16 language-model set path (big MFMA GEMM, several matrices per launch / split K)
            const __half *Wh[3]; for (int i = 0; i < n_mat; i++) Wh[i] = reinterpret_cast<const __half *>(W[i].qs);
            struct KsScope { KsScope(int k) { set_gemm_tuning(-1, k); } ~KsScope() { set_gemm_tuning(-1, 0); } } scope(std::max(0, ks));   // restored on every exit path
            ok = launch_gemm_f16_set(A.xh, (int)n_in, Wh, n_mat, (int)N, (int)n_out, (int)n_in, Yp, residual ? Rp : nullptr, (int)n_out, A.ws, A.ws_floats, 256, nullptr);
        } else {
            struct KsScope { KsScope(int k) { set_mmq2_tuning(-1, -1, k); set_mmq3_tuning(-1, k); } ~KsScope() { set_mmq2_tuning(-1, -1, 0); set_mmq3_tuning(-1, 0); } } scope(std::max(0, ks));
            if (generation == 3) {   // load-time digit planes (mmq3_kernels.hip)
                if (!mmq3_supported(ggml_type, (int)n_out, (int)n_in)) return 4;
                const size_t pb = mmq3_plane_bytes((int)n_out, (int)n_in);
                DevBuf d_pl(pb * (size_t)n_mat);
                const uint8_t *Pp[3];
                for (int i = 0; i < n_mat; i++) { Pp[i] = d_pl.as<uint8_t>() + (size_t)i * pb; launch_mmq3_build(W[i], d_pl.as<uint8_t>() + (size_t)i * pb, nullptr); }
                ok = launch_mmq3_set(Wp, Pp, Yp, residual ? Rp : nullptr, n_mat, A, (int)N, (int)n_out, nullptr);
                HIP_CHECK(hipDeviceSynchronize());
            } else {
                struct HScope { int keep; HScope(int v) : keep(mmqh_enabled()) { set_mmqh(v); } ~HScope() { set_mmqh(keep); } } h_scope(generation == 4);   // 4: k_mmqh_q45k (fp16-MFMA form, not adopted)
                ok = launch_mmq2_set(Wp, Yp, residual ? Rp : nullptr, n_mat, A, (int)N, (int)n_out, nullptr);
            }
        }
        HIP_CHECK(hipDeviceSynchronize());
        if (!ok) return 4;
        HIP_CHECK(hipMemcpy(y, d_y.p, out_each * n_mat * 4, hipMemcpyDeviceToHost));
        return 0;
    });
}

// The decode mat-vec launches exactly as Engine::forward issues them: n1 equally spaced matrices of type1 (+ optionally n2 of type2 in the same,
// mixed-type launch), activation preparation either standalone (fuse = 0) or in the kernel prologue, optional residual, optional SiLU row-pair
// epilogue.  prep: 1 = rms_norm(x) * x2, 2 = x, 3 = silu(x) * x2.  y: (n1 + n2) * n_out floats (epi = 1: n_out floats).
int minigpt4_amd_test_matvec(int type1, const void *raw1, int n1, int type2, const void *raw2, int n2, int64_t n_in, int64_t n_out, const float *x, const float *x2, int prep,
                             int fuse, int epi, const float *residual, float *y) {
    if (!raw1 || !x || !y || n_in <= 0 || n_out <= 0 || n1 < 1 || n1 > 3 || n2 < 0 || n2 > 1 || prep < 1 || prep > 3 || (prep != 2 && !x2)) return 1;
    if (!qweight_supported(type1) || n_in % gt_block(type1) || (n2 && (!raw2 || !qweight_supported(type2) || n_in % gt_block(type2)))) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const int K = (int)n_in, R = (int)n_out, nt = n1 + n2;
        std::vector<std::unique_ptr<DevBuf>> keep;
        std::vector<QWeight> W((size_t)nt);
        auto upload = [&](int type, const void *raw, int n, QWeight *dst) {
            QWeight plan; const size_t need = plan_qweight(type, R, K, plan, nullptr), raw_bytes = gt_nbytes(type, (size_t)R * K);
            keep.emplace_back(new DevBuf(need * (size_t)n)); uint8_t *base = (uint8_t *)keep.back()->p;   // one allocation: equal spacing
            DevBuf d_raw(raw_bytes);
            for (int m = 0; m < n; m++) {
                plan_qweight(type, R, K, dst[m], base + (size_t)m * need);
                HIP_CHECK(hipMemcpy(d_raw.p, (const uint8_t *)raw + (size_t)m * raw_bytes, raw_bytes, hipMemcpyHostToDevice));
                if (type == GT_F16 || type == GT_F32) HIP_CHECK(hipMemcpy(base + (size_t)m * need, d_raw.p, raw_bytes, hipMemcpyDeviceToDevice)); else launch_repack(d_raw.as<uint8_t>(), dst[m], nullptr);
                HIP_CHECK(hipDeviceSynchronize());
            }
        };
        upload(type1, raw1, n1, W.data());
        if (n2) upload(type2, raw2, n2, W.data() + n1);
        DevBuf d_x((size_t)K * 4), d_x2((size_t)K * 4), d_y((size_t)nt * R * 4), d_res((size_t)nt * R * 4), d_tab(65536 * 2);
        HIP_CHECK(hipMemcpy(d_x.p, x, (size_t)K * 4, hipMemcpyHostToDevice));
        if (x2) HIP_CHECK(hipMemcpy(d_x2.p, x2, (size_t)K * 4, hipMemcpyHostToDevice));
        if (residual) HIP_CHECK(hipMemcpy(d_res.p, residual, (size_t)nt * R * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(d_y.p, 0xFF, (size_t)nt * R * 4));
        Tables tb;
        { std::vector<__half> si(65536);
          for (int i = 0; i < 65536; i++) { const float v = __half2float(__ushort_as_half((unsigned short)i)); si[(size_t)i] = __float2half_rn(v / (1.0f + expf(-v))); }
          HIP_CHECK(hipMemcpy(d_tab.p, si.data(), 131072, hipMemcpyHostToDevice)); tb.silu = d_tab.as<__half>(); }
        ActQ A; alloc_act(A, keep, 1, (size_t)K);
        int mask = 0; for (int m = 0; m < nt; m++) mask |= act_mask_for(W[(size_t)m].type);
        if (!fuse) {
            if (prep == 1) launch_rms_quant(d_x.as<float>(), d_x2.as<float>(), 1, K, A, mask, nullptr, epi == MATVEC_EPI_REF);   // oracle-order launch: the oracle's rms mean
            else launch_silu_mul_quant(d_x.as<float>(), prep == 3 ? d_x2.as<float>() : nullptr, 1, K, A, mask, tb, nullptr);
        }
        const QWeight *Wp[4]; float *Yp[4]; const float *Rp[4];
        for (int m = 0; m < nt; m++) { Wp[m] = &W[(size_t)m]; Yp[m] = d_y.as<float>() + (size_t)m * R; Rp[m] = d_res.as<float>() + (size_t)m * R; }
        bool ok;
        if (n2) ok = launch_matvec_mixed(Wp, Yp, n1, Wp + n1, Yp + n1, n2, A, nullptr, fuse ? prep : 0, d_x.as<float>(), d_x2.as<float>(), epi);
        else ok = launch_matvec_set(Wp, Yp, residual ? Rp : nullptr, n1, A, nullptr, fuse ? prep : 0, d_x.as<float>(), d_x2.as<float>(), &tb, epi);
        if (!ok) { set_last_error("shape / type outside the decode mat-vec kernel's range"); return 4; }
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(y, d_y.p, (size_t)(epi == 1 ? 1 : nt) * R * 4, hipMemcpyDeviceToHost));
        return 0;
    });
}

// The batched-decode mat-vec (k_matvec_tn) as Engine::forward_batch issues it: N = 1..4 rows against n_mat equally spaced matrices (raw_w = their file bytes
// back to back), optional residual.  x: [N][n_in]; y / residual: [n_mat][N][n_out].  Returns 4 when the shape is outside the kernel's range.
int minigpt4_amd_test_matvec_rows(int ggml_type, const void *raw_w, int n_mat, int64_t n_in, int64_t n_out, const float *x, int N, const float *residual, float *y) {
    if (!raw_w || !x || !y || n_in <= 0 || n_out <= 0 || n_mat < 1 || n_mat > 3 || N < 1 || N > 4 || !qweight_supported(ggml_type) || n_in % gt_block(ggml_type)) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const int K = (int)n_in, R = (int)n_out;
        QWeight plan; const size_t need = plan_qweight(ggml_type, R, K, plan, nullptr), raw_bytes = gt_nbytes(ggml_type, (size_t)R * K);
        DevBuf planes(need * (size_t)n_mat), d_raw(raw_bytes), d_x((size_t)N * K * 4), d_y((size_t)n_mat * N * R * 4), d_res((size_t)n_mat * N * R * 4);
        std::vector<QWeight> W((size_t)n_mat);
        for (int m = 0; m < n_mat; m++) {
            plan_qweight(ggml_type, R, K, W[(size_t)m], planes.as<uint8_t>() + (size_t)m * need);
            HIP_CHECK(hipMemcpy(d_raw.p, (const uint8_t *)raw_w + (size_t)m * raw_bytes, raw_bytes, hipMemcpyHostToDevice));
            if (ggml_type == GT_F16 || ggml_type == GT_F32) HIP_CHECK(hipMemcpy(planes.as<uint8_t>() + (size_t)m * need, d_raw.p, raw_bytes, hipMemcpyDeviceToDevice)); else launch_repack(d_raw.as<uint8_t>(), W[(size_t)m], nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        }
        HIP_CHECK(hipMemcpy(d_x.p, x, (size_t)N * K * 4, hipMemcpyHostToDevice));
        if (residual) HIP_CHECK(hipMemcpy(d_res.p, residual, (size_t)n_mat * N * R * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(d_y.p, 0xFF, (size_t)n_mat * N * R * 4));
        ActQ A; std::vector<std::unique_ptr<DevBuf>> keep; alloc_act(A, keep, (size_t)N, (size_t)K);
        launch_rms_quant(d_x.as<float>(), nullptr, N, K, A, act_mask_for(ggml_type), nullptr);
        const QWeight *Wp[3]; float *Yp[3]; const float *Rp[3];
        for (int m = 0; m < n_mat; m++) { Wp[m] = &W[(size_t)m]; Yp[m] = d_y.as<float>() + (size_t)m * N * R; Rp[m] = d_res.as<float>() + (size_t)m * N * R; }
        if (!launch_matvec_rows(Wp, Yp, residual ? Rp : nullptr, n_mat, A, N, R, nullptr)) { set_last_error("shape / type outside the multi-row mat-vec kernel's range"); return 4; }
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(y, d_y.p, (size_t)n_mat * N * R * 4, hipMemcpyDeviceToHost));
        return 0;
    });
}

// The batched decode's MFMA launch (ri_kernels.hip): n_mat equally shaped k-quant matrices (rows of raw_w back to back) as ordinary planes -> row-interleaved image -> N = 1..4
// prepared rows; y [n_mat][N][n_out].  4: the kernel refuses the shape / type.
int minigpt4_amd_test_matvec_ri(int ggml_type, const void *raw_w, int n_mat, int64_t n_in, int64_t n_out, const float *x, int N, const float *residual, const float *rms_w, float *y) {
    if (!raw_w || !x || !y || n_in <= 0 || n_out <= 0 || n_mat < 1 || n_mat > 3 || N < 1 || N > 4 || !qweight_supported(ggml_type) || n_in % gt_block(ggml_type)) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const int K = (int)n_in, R = (int)n_out;
        if (!ri_supported(ggml_type, R, K)) { set_last_error("type / shape outside the row-interleaved kernel's range"); return 4; }
        QWeight plan; const size_t need = plan_qweight(ggml_type, R, K, plan, nullptr), raw_bytes = gt_nbytes(ggml_type, (size_t)R * K);
        RiPlanes rplan; const size_t rneed = ri_plan(ggml_type, R, K, rplan, nullptr);
        DevBuf planes(need * (size_t)n_mat), rplanes(rneed * (size_t)n_mat), d_raw(raw_bytes), d_x((size_t)N * K * 4), d_y((size_t)n_mat * N * R * 4), d_res((size_t)n_mat * N * R * 4);
        std::vector<QWeight> W((size_t)n_mat); std::vector<RiPlanes> P((size_t)n_mat);
        for (int m = 0; m < n_mat; m++) {
            plan_qweight(ggml_type, R, K, W[(size_t)m], planes.as<uint8_t>() + (size_t)m * need);
            HIP_CHECK(hipMemcpy(d_raw.p, (const uint8_t *)raw_w + (size_t)m * raw_bytes, raw_bytes, hipMemcpyHostToDevice));
            launch_repack(d_raw.as<uint8_t>(), W[(size_t)m], nullptr);
            ri_plan(ggml_type, R, K, P[(size_t)m], rplanes.as<uint8_t>() + (size_t)m * rneed);
            launch_ri_build(W[(size_t)m], P[(size_t)m], nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        }
        HIP_CHECK(hipMemcpy(d_x.p, x, (size_t)N * K * 4, hipMemcpyHostToDevice));
        if (residual) HIP_CHECK(hipMemcpy(d_res.p, residual, (size_t)n_mat * N * R * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(d_y.p, 0xFF, (size_t)n_mat * N * R * 4));
        ActQ A; std::vector<std::unique_ptr<DevBuf>> keep; alloc_act(A, keep, (size_t)N, (size_t)K);
        DevBuf d_w((size_t)K * 4);
        if (rms_w) HIP_CHECK(hipMemcpy(d_w.p, rms_w, (size_t)K * 4, hipMemcpyHostToDevice));
        else launch_rms_quant(d_x.as<float>(), nullptr, N, K, A, ACT_Q8K, nullptr);                   // rms_w given: the launch norms + quantises the rows itself
        const QWeight *Wp[3]; const RiPlanes *Pp[3]; float *Yp[3]; const float *Rp[3];
        for (int m = 0; m < n_mat; m++) { Wp[m] = &W[(size_t)m]; Pp[m] = &P[(size_t)m]; Yp[m] = d_y.as<float>() + (size_t)m * N * R; Rp[m] = d_res.as<float>() + (size_t)m * N * R; }
        hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0)); set_ri_cus(prop.multiProcessorCount);
        DevBuf d_ws(((size_t)1 << 18) * 4 + 4096);                           // K-split workspace (few groups, long K): slabs + zeroed tickets
        HIP_CHECK(hipMemset(d_ws.p, 0, ((size_t)1 << 18) * 4 + 4096));
        const RiWorkspace ws{d_ws.as<float>(), (size_t)1 << 18, reinterpret_cast<unsigned *>(d_ws.as<float>() + ((size_t)1 << 18)), 1024};
        if (!launch_matvec_ri(Wp, Pp, Yp, residual ? Rp : nullptr, n_mat, A, N, R, nullptr, rms_w ? d_x.as<float>() : nullptr, rms_w ? d_w.as<float>() : nullptr, K, ws)) { set_last_error("launch_matvec_ri refused"); return 4; }
        HIP_CHECK(hipDeviceSynchronize());
        if (!launch_matvec_ri(Wp, Pp, Yp, residual ? Rp : nullptr, n_mat, A, N, R, nullptr, rms_w ? d_x.as<float>() : nullptr, rms_w ? d_w.as<float>() : nullptr, K, ws)) { set_last_error("launch_matvec_ri refused"); return 4; }   // twice: the tickets must be back at zero
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(y, d_y.p, (size_t)n_mat * N * R * 4, hipMemcpyDeviceToHost));
        return 0;
    });
}

// the mixed-type launch of a "more bits" layer: n_a matrices of type_a (Q4_K / Q5_K) + n_b of type_b (Q6_K), same shape -> y [n_a + n_b][N][n_out]
int minigpt4_amd_test_matvec_ri_mixed(int type_a, const void *raw_a, int n_a, int type_b, const void *raw_b, int n_b, int64_t n_in, int64_t n_out, const float *x, int N, float *y) {
    if (!raw_a || !raw_b || !x || !y || n_in <= 0 || n_out <= 0 || n_a < 1 || n_b < 1 || n_a + n_b > 3 || N < 1 || N > 4 || !qweight_supported(type_a) || !qweight_supported(type_b)) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const int K = (int)n_in, R = (int)n_out, n = n_a + n_b;
        if (!ri_supported(type_a, R, K) || !ri_supported(type_b, R, K)) { set_last_error("type / shape outside the row-interleaved kernel's range"); return 4; }
        std::vector<QWeight> W((size_t)n); std::vector<RiPlanes> P((size_t)n);
        std::vector<std::unique_ptr<DevBuf>> hold;
        for (int m = 0; m < n; m++) {
            const int ty = m < n_a ? type_a : type_b;
            QWeight plan; const size_t need = plan_qweight(ty, R, K, plan, nullptr), raw_bytes = gt_nbytes(ty, (size_t)R * K);
            RiPlanes rplan; const size_t rneed = ri_plan(ty, R, K, rplan, nullptr);
            hold.emplace_back(new DevBuf(need)); DevBuf &pl = *hold.back();
            hold.emplace_back(new DevBuf(rneed)); DevBuf &rp = *hold.back();
            DevBuf d_raw(raw_bytes);
            const uint8_t *src = m < n_a ? (const uint8_t *)raw_a + (size_t)m * raw_bytes : (const uint8_t *)raw_b + (size_t)(m - n_a) * raw_bytes;
            plan_qweight(ty, R, K, W[(size_t)m], pl.as<uint8_t>());
            HIP_CHECK(hipMemcpy(d_raw.p, src, raw_bytes, hipMemcpyHostToDevice));
            launch_repack(d_raw.as<uint8_t>(), W[(size_t)m], nullptr);
            ri_plan(ty, R, K, P[(size_t)m], rp.as<uint8_t>());
            launch_ri_build(W[(size_t)m], P[(size_t)m], nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        }
        DevBuf d_x((size_t)N * K * 4), d_y((size_t)n * N * R * 4);
        HIP_CHECK(hipMemcpy(d_x.p, x, (size_t)N * K * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(d_y.p, 0xFF, (size_t)n * N * R * 4));
        ActQ A; std::vector<std::unique_ptr<DevBuf>> keep; alloc_act(A, keep, (size_t)N, (size_t)K);
        launch_rms_quant(d_x.as<float>(), nullptr, N, K, A, ACT_Q8K, nullptr);
        const QWeight *Wa[2], *Wb[2]; const RiPlanes *Pa[2], *Pb[2]; float *Ya[2], *Yb[2];
        for (int m = 0; m < n; m++) {
            float *yo = d_y.as<float>() + (size_t)m * N * R;
            if (m < n_a) { Wa[m] = &W[(size_t)m]; Pa[m] = &P[(size_t)m]; Ya[m] = yo; } else { Wb[m - n_a] = &W[(size_t)m]; Pb[m - n_a] = &P[(size_t)m]; Yb[m - n_a] = yo; }
        }
        hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0)); set_ri_cus(prop.multiProcessorCount);
        if (!launch_matvec_ri_mixed(Wa, Pa, Ya, n_a, Wb, Pb, Yb, n_b, A, N, R, nullptr)) { set_last_error("launch_matvec_ri_mixed refused"); return 4; }
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(y, d_y.p, (size_t)n * N * R * 4, hipMemcpyDeviceToHost));
        return 0;
    });
}

// microseconds per launch of the row-interleaved MFMA mat-vec for one set of n_mat rows x cols matrices against N prepared rows; n_sets weight sets are rotated
int minigpt4_amd_bench_matvec_ri(int ggml_type, int rows, int cols, int n_mat, int N, int iters, int n_sets, float *us_per_launch) {
    if (!ri_supported(ggml_type, rows, cols) || n_mat < 1 || n_mat > 3 || N < 1 || N > 4 || iters < 1 || n_sets < 1) return 1;
    if (device_count_noexcept() <= 0) return 2;
    return guarded(3, [&]() -> int {
        hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0)); set_ri_cus(prop.multiProcessorCount);
        QWeight plan; const size_t need = plan_qweight(ggml_type, rows, cols, plan, nullptr);
        RiPlanes rplan; const size_t rneed = ri_plan(ggml_type, rows, cols, rplan, nullptr);
        std::vector<std::unique_ptr<DevBuf>> keep;
        std::vector<QWeight> W((size_t)n_sets * n_mat); std::vector<RiPlanes> P((size_t)n_sets * n_mat);
        DevBuf tmp(need);
        for (size_t i = 0; i < W.size(); i++) {
            plan_qweight(ggml_type, rows, cols, W[i], tmp.as<uint8_t>());
            launch_fill_random(tmp.p, need, (unsigned)(i * 7919 + 13), nullptr);
            const size_t n = (size_t)rows * cols;
            if (ggml_type == GT_Q4_K || ggml_type == GT_Q5_K) launch_fill_u16((void *)W[i].sc, n / 256 * 16 / 2, 0x1C00, nullptr);
            if (W[i].d) launch_fill_u16((void *)W[i].d, n / 256, 0x1C00, nullptr);
            keep.emplace_back(new DevBuf(rneed));
            ri_plan(ggml_type, rows, cols, P[i], (uint8_t *)keep.back()->p);
            launch_ri_build(W[i], P[i], nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        }
        ActQ A; alloc_act(A, keep, (size_t)N, (size_t)cols);
        DevBuf dx((size_t)N * cols * 4), dy((size_t)rows * 4 * 3 * N);
        { std::vector<float> hx((size_t)N * cols); for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)((int)(i * 37 % 201) - 100) / 64.0f; HIP_CHECK(hipMemcpy(dx.p, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); }
        launch_rms_quant(dx.as<float>(), nullptr, N, cols, A, ACT_Q8K, nullptr);
        DevBuf d_ws(((size_t)1 << 18) * 4 + 4096);
        HIP_CHECK(hipMemset(d_ws.p, 0, ((size_t)1 << 18) * 4 + 4096));
        const RiWorkspace ws = getenv("MG4_RI_NOSPLIT") ? RiWorkspace{} : RiWorkspace{d_ws.as<float>(), (size_t)1 << 18, reinterpret_cast<unsigned *>(d_ws.as<float>() + ((size_t)1 << 18)), 1024};
        auto run = [&](int set) {
            const QWeight *Wp[3]; const RiPlanes *Pp[3]; float *Yp[3];
            for (int m = 0; m < n_mat; m++) { Wp[m] = &W[(size_t)set * n_mat + m]; Pp[m] = &P[(size_t)set * n_mat + m]; Yp[m] = dy.as<float>() + (size_t)m * rows * N; }
            if (!launch_matvec_ri(Wp, Pp, Yp, nullptr, n_mat, A, N, rows, nullptr, nullptr, nullptr, 0, ws)) throw HipError{hipErrorInvalidValue, "launch_matvec_ri refused", __FILE__, __LINE__};
        };
        for (int i = 0; i < std::min(n_sets, 4); i++) run(i);
        HIP_CHECK(hipDeviceSynchronize());
        hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; i++) run(i % n_sets);
        HIP_CHECK(hipEventRecord(b, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        if (us_per_launch) *us_per_launch = ms * 1e3f / (float)iters;
        HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b));
        return 0;
    });
}

int minigpt4_amd_test_quantize(const float *x, const float *rms_w, int64_t N, int64_t K, int8_t *q8k, float *dk, int16_t *bsums, int8_t *q80, float *d0) {
    if (!x || N <= 0 || K <= 0 || K % 256) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        DevBuf d_x((size_t)(N * K) * 4), d_w((size_t)K * 4);
        HIP_CHECK(hipMemcpy(d_x.p, x, (size_t)(N * K) * 4, hipMemcpyHostToDevice));
        if (rms_w) HIP_CHECK(hipMemcpy(d_w.p, rms_w, (size_t)K * 4, hipMemcpyHostToDevice));
        ActQ A; std::vector<std::unique_ptr<DevBuf>> keep; alloc_act(A, keep, (size_t)N, (size_t)K);
        launch_rms_quant(d_x.as<float>(), rms_w ? d_w.as<float>() : nullptr, (int)N, (int)K, A, ACT_Q8K | ACT_Q80, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        if (q8k) HIP_CHECK(hipMemcpy(q8k, A.q8k, (size_t)(N * K), hipMemcpyDeviceToHost));
        if (dk) HIP_CHECK(hipMemcpy(dk, A.dk, (size_t)(N * K / 256) * 4, hipMemcpyDeviceToHost));
        if (bsums) HIP_CHECK(hipMemcpy(bsums, A.bsk, (size_t)(N * K / 16) * 2, hipMemcpyDeviceToHost));
        if (q80) HIP_CHECK(hipMemcpy(q80, A.q80, (size_t)(N * K), hipMemcpyDeviceToHost));
        if (d0) HIP_CHECK(hipMemcpy(d0, A.d0, (size_t)(N * K / 32) * 4, hipMemcpyDeviceToHost));
        return 0;
    });
}

static int test_gemm_impl(const float *A, const float *W, const float *bias, int M, int N, int K, int gelu, float *C, bool skinny);
int minigpt4_amd_test_gemm_f16(const float *A, const float *W, const float *bias, int M, int N, int K, int gelu, float *C) { return test_gemm_impl(A, W, bias, M, N, K, gelu, C, false); }
int minigpt4_amd_test_gemm_f16_skinny(const float *A, const float *W, const float *bias, int M, int N, int K, int gelu, float *C) { return test_gemm_impl(A, W, bias, M, N, K, gelu, C, true); }
static int test_gemm_impl(const float *A, const float *W, const float *bias, int M, int N, int K, int gelu, float *C, bool skinny) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || K % 16) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        DevBuf dA((size_t)M * K * 4), dW((size_t)N * K * 4), dAh((size_t)M * K * 2), dWh((size_t)N * K * 2), dC((size_t)M * N * 4), db((size_t)N * 4), dtab(65536 * 2);
        HIP_CHECK(hipMemcpy(dA.p, A, (size_t)M * K * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dW.p, W, (size_t)N * K * 4, hipMemcpyHostToDevice));
        if (bias) HIP_CHECK(hipMemcpy(db.p, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        Tables tb;
        if (gelu) { std::vector<__half> g(65536);
            for (int i = 0; i < 65536; i++) { const float x = __half2float(__ushort_as_half((unsigned short)i)); g[(size_t)i] = __float2half_rn(0.5f * x * (1.0f + tanhf(0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x)))); }
            HIP_CHECK(hipMemcpy(dtab.p, g.data(), 131072, hipMemcpyHostToDevice)); tb.gelu = dtab.as<__half>(); }
        launch_f32_to_f16(dA.as<float>(), dAh.as<__half>(), (size_t)M * K, nullptr); launch_f32_to_f16(dW.as<float>(), dWh.as<__half>(), (size_t)N * K, nullptr);
        if (skinny) { if (!launch_gemm_f16_skinny(dAh.as<__half>(), K, dWh.as<__half>(), K, M, N, K, bias ? db.as<float>() : nullptr, nullptr, gelu != 0, tb, dC.as<float>(), nullptr, N, nullptr)) return 4; }
        else launch_gemm_f16(dAh.as<__half>(), K, dWh.as<__half>(), K, M, N, K, bias ? db.as<float>() : nullptr, nullptr, gelu != 0, tb, dC.as<float>(), nullptr, N, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(C, dC.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        return 0;
    });
}

// Micro-benchmark of the fp16 GEMM of the image path on synthetic operands (tools/timeline_gemm.py): `iters` launches of C[M][N] = A[M][K] . W[N][K]^T, cycling through
// `n_sets` weight matrices so that the Infinity Cache cannot serve repeats (as in the encoder, where every block has its own weights).  flags: 1 = GELU epilogue,
// 2 = residual epilogue, 4 = fp16 output as well.  variant 0 = launch_gemm_f16 (the dispatcher: 128x128 tiles from M = 512, 64x64 below), 1 = the skinny-M kernel,
// 2 = split K in `slices` + k_splitk_reduce_ln (the ViT's fc2 form; `slices` in bits 8.. of `variant`).
int minigpt4_amd_bench_gemm_f16(int M, int N, int K, int flags, int variant, int iters, int n_sets, float *us_per_launch) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 16 || iters < 1 || n_sets < 1) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const int slices = std::max(1, (variant >> 8) & 255), sk_arm = variant >> 16; variant &= 255;
        DevBuf dA((size_t)M * K * 2), dC((size_t)M * N * 4), dCh((size_t)M * N * 2), dR((size_t)M * N * 4), db((size_t)N * 4), dtab(65536 * 2), dslab((size_t)std::max(slices, 8) * M * N * 4), dln((size_t)N * 4);
        std::vector<std::unique_ptr<DevBuf>> W;
        for (int i = 0; i < n_sets; i++) { W.emplace_back(new DevBuf((size_t)N * K * 2)); launch_fill_u16(W.back()->p, (size_t)N * K, (unsigned short)(0x2E66 + i), nullptr); }
        launch_fill_u16(dA.p, (size_t)M * K, 0x3266, nullptr); launch_fill_u16(dtab.p, 65536, 0x3800, nullptr);
        HIP_CHECK(hipMemset(dR.p, 0, (size_t)M * N * 4)); HIP_CHECK(hipMemset(db.p, 0, (size_t)N * 4)); HIP_CHECK(hipMemset(dln.p, 0, (size_t)N * 4));
        Tables tb; tb.gelu = dtab.as<__half>();
        const bool gelu = flags & 1, res = flags & 2;
        auto run = [&](int set) -> bool {
            const __half *Wh = W[(size_t)set]->as<__half>();
            if (variant == 1) return launch_gemm_f16_skinny(dA.as<__half>(), K, Wh, K, M, N, K, db.as<float>(), res ? dR.as<float>() : nullptr, gelu, tb, dC.as<float>(), (flags & 4) ? dCh.as<__half>() : nullptr, N, nullptr);
            if (variant == 2) {
                int sl = gemm_split_slices(K, slices);
                if (sk_arm) { sl = slices; if (!launch_gemm_f16_splitk_arm(sk_arm, dA.as<__half>(), K, Wh, K, M, N, K, sl, dslab.as<float>(), (size_t)M * N, N, nullptr)) return false; }
                else launch_gemm_f16_splitk(dA.as<__half>(), K, Wh, K, M, N, K, sl, dslab.as<float>(), (size_t)M * N, N, nullptr);
                if (N > 2048) launch_slab_reduce(dslab.as<float>(), sl, (long long)M * N, dR.as<float>(), dC.as<float>(), (size_t)M * N, nullptr);   // the language model's combine
                else launch_splitk_reduce_ln(dslab.as<float>(), sl, (size_t)M * N, db.as<float>(), dR.as<float>(), M, N, dC.as<float>(), dln.as<float>(), dln.as<float>(), nullptr, dCh.as<__half>(), nullptr);
                return true;
            }
            if (variant >= 101 && variant <= 103) {   // the F16 language model's set launch (n = variant - 100 equally spaced matrices of N columns each; wo / w2: n = 1 with a residual)
                const int n = variant - 100;
                const __half *Wp[3]; float *Yp[3]; const float *Rp[3];
                for (int m = 0; m < n; m++) { Wp[m] = Wh + (size_t)m * (N / n) * K; Yp[m] = dC.as<float>() + (size_t)m * M * (N / n); Rp[m] = dR.as<float>() + (size_t)m * M * (N / n); }
                return launch_gemm_f16_set(dA.as<__half>(), K, Wp, n, M, N / n, K, Yp, res ? Rp : nullptr, N / n, dslab.as<float>(), (size_t)8 * M * N, 256, nullptr);
            }
            if (variant >= 3 && variant < 100) return launch_gemm_f16_arm(variant, dA.as<__half>(), K, Wh, K, M, N, K, db.as<float>(), res ? dR.as<float>() : nullptr, gelu, tb, dC.as<float>(), (flags & 4) ? dCh.as<__half>() : nullptr, N, nullptr);
            launch_gemm_f16(dA.as<__half>(), K, Wh, K, M, N, K, db.as<float>(), res ? dR.as<float>() : nullptr, gelu, tb, dC.as<float>(), (flags & 4) ? dCh.as<__half>() : nullptr, N, nullptr);
            return true;
        };
        for (int i = 0; i < std::min(n_sets, 4); i++) if (!run(i)) return 4;
        HIP_CHECK(hipDeviceSynchronize());
        hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; i++) run(i % n_sets);
        HIP_CHECK(hipEventRecord(b, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b));
        if (us_per_launch) *us_per_launch = ms * 1e3f / (float)iters;
        return 0;
    });
}
// Micro-benchmark of the ViT / Q-Former attention (launch_attn_f32) on synthetic q|k|v rows: heads x hd wide, nq queries against nk keys, `iters` launches.
int minigpt4_amd_bench_attn_f32(int heads, int hd, int nq, int nk, int iters, float *us_per_launch) {
    if (heads < 1 || (hd != 88 && hd != 64) || nq < 1 || nk < 1 || nk > 320 || iters < 1) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const int D = heads * hd, n = std::max(nq, nk);
        DevBuf dq((size_t)n * 3 * D * 4), dout((size_t)nq * D * 4), douth((size_t)nq * D * 2), dtab(65536 * 2);
        { std::vector<float> h((size_t)n * 3 * D); for (size_t i = 0; i < h.size(); i++) h[i] = (float)((int)(i * 2654435761u % 2001u) - 1000) / 1000.0f; HIP_CHECK(hipMemcpy(dq.p, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
        { std::vector<__half> e(65536); for (int i = 0; i < 65536; i++) e[(size_t)i] = __float2half_rn(expf(__half2float(__ushort_as_half((unsigned short)i)))); HIP_CHECK(hipMemcpy(dtab.p, e.data(), 131072, hipMemcpyHostToDevice)); }
        Tables tb; tb.exp = dtab.as<__half>();
        { int nneg = 0; for (int c = 0x8000; c < 0xFC00; c++) { if (__half2float(__float2half_rn(expf(__half2float(__ushort_as_half((unsigned short)c))))) == 0.0f) break; nneg++; } tb.exp_neg_n = (nneg + 2047) / 2048 * 2048; }
        const float scale = 1.0f / sqrtf((float)hd);
        auto run = [&]() { launch_attn_f32(dq.as<float>(), 3 * D, dq.as<float>() + D, dq.as<float>() + 2 * D, 3 * D, nq, nk, heads, hd, scale, 0.0f, tb, nullptr, douth.as<__half>(), D, nullptr, 1); };
        for (int i = 0; i < 3; i++) run();
        HIP_CHECK(hipDeviceSynchronize());
        hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; i++) run();
        HIP_CHECK(hipEventRecord(b, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b));
        if (us_per_launch) *us_per_launch = ms * 1e3f / (float)iters;
        return 0;
    });
}
// the same with `batch` images per launch, computed exponentials (what the engine's fast mode runs) and a forced number of query tiles per workgroup (0 = the launcher's choice)
int minigpt4_amd_bench_attn_f32_b(int heads, int hd, int nq, int nk, int batch, int qt, int iters, float *us_per_launch) {
    if (heads < 1 || (hd != 88 && hd != 64) || nq < 1 || nk < 1 || nk > 320 || iters < 1 || batch < 1 || batch > 16 || nq != nk) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const int D = heads * hd;
        DevBuf dq((size_t)batch * nq * 3 * D * 4), douth((size_t)batch * nq * D * 2);
        { std::vector<float> h((size_t)batch * nq * 3 * D); for (size_t i = 0; i < h.size(); i++) h[i] = (float)((int)(i * 2654435761u % 2001u) - 1000) / 1000.0f; HIP_CHECK(hipMemcpy(dq.p, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
        Tables tb;                                                          // exp == null: computed exponentials
        const float scale = 1.0f / sqrtf((float)hd);
        struct QtScope { QtScope(int q) { set_attn_vit_qt(q); } ~QtScope() { set_attn_vit_qt(0); } } scope(qt);
        // MG4_ATTN_HEAD_MAJOR=1 (experiment): q | k | v as [3][head][batch x rows][hd] instead of [rows][3 x heads x hd]
        const bool hm = getenv("MG4_ATTN_HEAD_MAJOR") && atoi(getenv("MG4_ATTN_HEAD_MAJOR"));
        const int R = batch * nq;
        auto run = [&]() {
            if (hm) launch_attn_f32(dq.as<float>(), hd, dq.as<float>() + (size_t)D * R, dq.as<float>() + (size_t)2 * D * R, hd, nq, nk, heads, hd, scale, 0.0f, tb, nullptr, douth.as<__half>(), D, nullptr, batch, R * hd, R * hd);
            else launch_attn_f32(dq.as<float>(), 3 * D, dq.as<float>() + D, dq.as<float>() + 2 * D, 3 * D, nq, nk, heads, hd, scale, 0.0f, tb, nullptr, douth.as<__half>(), D, nullptr, batch); };
        for (int i = 0; i < 3; i++) run();
        HIP_CHECK(hipDeviceSynchronize());
        hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; i++) run();
        HIP_CHECK(hipEventRecord(b, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b));
        if (us_per_launch) *us_per_launch = ms * 1e3f / (float)iters;
        return 0;
    });
}
void minigpt4_amd_test_set_attn_qt(int qt) { set_attn_vit_qt(qt); }
// The F16 feed-forward pair launch (launch_gemm_f16_silu_pair): x [N][n_in] fp32 (rounded to fp16 as the engine's row preparation does), w = w1 then w3, each [n_out][n_in]
// fp16; out_h [N][n_out] = fp16(silu_table(w1 x) * (w3 x)) as uint16 bit patterns, out_f (optional) the fp32 product before the rounding
int minigpt4_amd_test_f16_silu_pair(const float *x, const void *w_f16, int64_t N, int64_t n_in, int64_t n_out, unsigned short *out_h, float *out_f) {
    if (!x || !w_f16 || !out_h || N <= 0 || n_in <= 0 || n_out <= 0) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const size_t nx = (size_t)(N * n_in), nw = (size_t)(n_in * n_out), no = (size_t)(N * n_out);
        DevBuf dx(nx * 4), dxh(nx * 2), dw(nw * 2 * 2), dh(no * 2), df(no * 4), dtab(65536 * 2);
        HIP_CHECK(hipMemcpy(dx.p, x, nx * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dw.p, w_f16, nw * 4, hipMemcpyHostToDevice));
        { std::vector<__half> si(65536); for (int i = 0; i < 65536; i++) { const float v = __half2float(__ushort_as_half((unsigned short)i)); si[(size_t)i] = __float2half_rn(v / (1.0f + expf(-v))); }
          HIP_CHECK(hipMemcpy(dtab.p, si.data(), 131072, hipMemcpyHostToDevice)); }
        launch_f32_to_f16(dx.as<float>(), dxh.as<__half>(), nx, nullptr);
        Tables tb; tb.silu = dtab.as<__half>();
        hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0));
        if (!launch_gemm_f16_silu_pair(dxh.as<__half>(), (int)n_in, dw.as<__half>(), dw.as<__half>() + nw, (int)N, (int)n_out, (int)n_in, tb, df.as<float>(), dh.as<__half>(), (int)n_out,
                                       prop.multiProcessorCount, nullptr)) return 4;
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out_h, dh.p, no * 2, hipMemcpyDeviceToHost));
        if (out_f) HIP_CHECK(hipMemcpy(out_f, df.p, no * 4, hipMemcpyDeviceToHost));
        return 0;
    });
}
// Micro-benchmark of the prompt-row attention (launch_attn_prefill): N query rows at positions n_past .. n_past + N - 1 of an fp16 K / V cache filled with synthetic rows
int minigpt4_amd_bench_attn_prefill(int n_head, int hd, int N, int n_past, int iters, float *us_per_launch) {
    if (const char *w8 = getenv("MINIGPT4_ATTN_PREFILL_W8")) set_attn_prefill_w8(atoi(w8));        // micro-benchmark only (no engine in this process)
    if (n_head < 1 || !attn_head_size_supported(hd) || N < 2 || n_past < 0 || iters < 1) return 1;
    if (device_count_noexcept() <= 0) { set_last_error("no HIP device"); return 2; }
    return guarded(3, [&]() -> int {
        const int E = n_head * hd, T = n_past + N;
        DevBuf dq((size_t)N * E * 4), dk((size_t)T * E * 2), dv((size_t)T * E * 2), dout((size_t)N * E * 4), dtab(65536 * 2), dnp(4);
        { std::vector<float> h((size_t)N * E); for (size_t i = 0; i < h.size(); i++) h[i] = (float)((int)(i * 2654435761u % 2001u) - 1000) / 4000.0f; HIP_CHECK(hipMemcpy(dq.p, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
        { std::vector<__half> h((size_t)T * E); for (size_t i = 0; i < h.size(); i++) h[i] = __float2half_rn((float)((int)(i * 40503u % 1001u) - 500) / 500.0f);
          HIP_CHECK(hipMemcpy(dk.p, h.data(), h.size() * 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dv.p, h.data(), h.size() * 2, hipMemcpyHostToDevice)); }
        { std::vector<__half> e(65536); for (int i = 0; i < 65536; i++) e[(size_t)i] = __float2half_rn(expf(__half2float(__ushort_as_half((unsigned short)i)))); HIP_CHECK(hipMemcpy(dtab.p, e.data(), 131072, hipMemcpyHostToDevice)); }
        HIP_CHECK(hipMemcpy(dnp.p, &n_past, 4, hipMemcpyHostToDevice));
        Tables tb; tb.exp = dtab.as<__half>();
        { int nneg = 0; for (int c = 0x8000; c < 0xFC00; c++) { if (__half2float(__float2half_rn(expf(__half2float(__ushort_as_half((unsigned short)c))))) == 0.0f) break; nneg++; } tb.exp_neg_n = (nneg + 2047) / 2048 * 2048; }
        auto run = [&]() { return launch_attn_prefill(dq.as<float>(), dk.as<__half>(), dv.as<__half>(), N, n_head, hd, dnp.as<int>(), T, tb, dout.as<float>(), nullptr); };
        for (int i = 0; i < 3; i++) if (!run()) return 4;
        HIP_CHECK(hipDeviceSynchronize());
        hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; i++) run();
        HIP_CHECK(hipEventRecord(b, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b));
        if (us_per_launch) *us_per_launch = ms * 1e3f / (float)iters;
        return 0;
    });
}
int minigpt4_amd_timeline_attn(unsigned long long *out, int max_workgroups) { return (out && max_workgroups > 0) ? read_attn_timeline(out, max_workgroups) : -1; }
void minigpt4_amd_test_set_gemm_arm(int arm, int sk_arm) { set_gemm_tuning(-1, 0, arm, sk_arm); }
void minigpt4_amd_test_set_splitk_xcd(int on) { set_gemm_splitk_xcd(on); }
int minigpt4_amd_timeline_vision(unsigned long long *out, int max_workgroups) { return (out && max_workgroups > 0) ? read_vision_timeline(out, max_workgroups) : -1; }

// Micro-benchmark of the decode mat-vec kernels on synthetic planes (random quant bytes, sane fp16 scales).  `n_sets` distinct weight sets
// are cycled so the 256 MiB Infinity Cache cannot serve repeats.  variant 0 = k_mul_mat per matrix, 1 = persistent-wave v2 (fused set).
int minigpt4_amd_bench_matvec(int ggml_type, int rows, int cols, int n_mat, int variant, int iters, int n_sets, int waves_per_cu, float *us_per_launch, double *bytes_per_launch) {
    if (!qweight_supported(ggml_type) || rows <= 0 || cols <= 0 || cols % gt_block(ggml_type) || n_mat < 1 || n_mat > 3 || iters < 1 || n_sets < 1) return 1;
    if (device_count_noexcept() <= 0) return 2;
    return guarded(3, [&]() -> int {
        hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0));
        set_matvec_tuning(waves_per_cu, 0, prop.multiProcessorCount);
        QWeight plan; const size_t need = plan_qweight(ggml_type, rows, cols, plan, nullptr);
        std::vector<std::unique_ptr<DevBuf>> keep;
        std::vector<QWeight> W((size_t)n_sets * n_mat);
        for (size_t i = 0; i < W.size(); i++) {
            keep.emplace_back(new DevBuf(need));
            uint8_t *base = (uint8_t *)keep.back()->p;
            plan_qweight(ggml_type, rows, cols, W[i], base);
            launch_fill_random(base, need, (unsigned)(i * 7919 + 13), nullptr);
            const size_t n = (size_t)rows * cols;
            if (W[i].sc) { const size_t sc_bytes = (size_t)((W[i].d ? W[i].d : base + need) - W[i].sc);
                if (ggml_type == GT_Q4_K || ggml_type == GT_Q5_K) { /* header: d, dmin fp16 then 12 scale bytes: make d/dmin sane, keep scales random */
                    launch_fill_u16((void *)W[i].sc, std::min(sc_bytes, n / 256 * 16) / 2, 0x1C00, nullptr); }
                else if (ggml_type != GT_Q6_K) launch_fill_u16((void *)W[i].sc, std::min(sc_bytes, n / 32 * 4) / 2, 0x1C00, nullptr); }
            if (W[i].d) launch_fill_u16((void *)W[i].d, n / 256, 0x1C00, nullptr);
            if (ggml_type == GT_F16) launch_fill_u16(base, n, 0x2E66, nullptr);
            if (ggml_type == GT_F32) { std::vector<float> h(n, 0.01f); HIP_CHECK(hipMemcpy(base, h.data(), n * 4, hipMemcpyHostToDevice)); }
        }
        // variants 3..6: the batched decode's multi-row launch (weights streamed once): 3 = 4 prepared rows, 4 = 2 prepared rows, 5 = 2 rows prepared inside the launch, 6 = 4 rows inside
        // 10 + N / 20 + N: the multi-row launch with N = 1..4 prepared rows / rows prepared inside the launch (N = 1 and 3 run the 2- and 4-row kernels with a row to spare)
        const int NR = (variant > 10 && variant <= 14) ? variant - 10 : (variant > 20 && variant <= 24) ? variant - 20 : (variant == 3 || variant == 6) ? 4 : ((variant == 4 || variant == 5) ? 2 : 1);
        ActQ A; alloc_act(A, keep, (size_t)NR, (size_t)cols);
        DevBuf dx((size_t)NR * cols * 4), dy((size_t)rows * 4 * 3 * NR);
        { std::vector<float> hx((size_t)NR * cols); for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)((int)(i * 37 % 201) - 100) / 64.0f; HIP_CHECK(hipMemcpy(dx.p, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); }
        launch_rms_quant(dx.as<float>(), nullptr, NR, cols, A, act_mask_for(ggml_type), nullptr);
        auto run = [&](int set) {
            const QWeight *Wp[3]; float *Yp[3];
            for (int m = 0; m < n_mat; m++) { Wp[m] = &W[(size_t)set * n_mat + m]; Yp[m] = dy.as<float>() + (size_t)m * rows * NR; }
            if (variant == 1 && launch_matvec_set(Wp, Yp, nullptr, n_mat, A, nullptr)) return;
            if ((variant == 3 || variant == 4 || (variant > 10 && variant <= 14)) && launch_matvec_rows(Wp, Yp, nullptr, n_mat, A, NR, rows, nullptr)) return;
            if ((variant == 5 || variant == 6 || (variant > 20 && variant <= 24)) && launch_matvec_rows(Wp, Yp, nullptr, n_mat, A, NR, rows, nullptr, dx.as<float>(), dx.as<float>(), cols)) return;
            if (variant == 2 && launch_matvec_set(Wp, Yp, nullptr, n_mat, A, nullptr, 1, dx.as<float>(), dx.as<float>())) return;   // rms-norm prologue, as the decode's qkv / w1|w3 launches
            for (int m = 0; m < n_mat; m++) launch_mul_mat(*Wp[m], A, 1, Yp[m], rows, nullptr, nullptr);
        };
        for (int i = 0; i < std::min(n_sets, 4); i++) run(i);
        HIP_CHECK(hipDeviceSynchronize());
        hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; i++) run(i % n_sets);
        HIP_CHECK(hipEventRecord(b, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b));
        if (us_per_launch) *us_per_launch = ms * 1e3f / (float)iters;
        if (bytes_per_launch) *bytes_per_launch = (double)gt_nbytes(ggml_type, (size_t)rows * cols) * n_mat;
        return 0;
    });
}

// Prefill mat-mul micro-benchmark (tools/mmq2_bench.py): n_mat matrices of random blocks against N random rows, `iters` launches of the engine's prefill launch
// (generation 2: mmq2_kernels.hip; generation 1: the round-1 kernels, one launch per matrix); ks: forced K split (0 = the launcher's choice).
int minigpt4_amd_bench_mmq(int ggml_type, int rows, int cols, int n_mat, int N, int iters, int ks, int generation, float *us_per_launch) {
    if (!qweight_supported(ggml_type) || rows <= 0 || cols <= 0 || cols % gt_block(ggml_type) || n_mat < 1 || n_mat > 3 || iters < 1 || N < 1) return 1;
    if (device_count_noexcept() <= 0) return 2;
    return guarded(3, [&]() -> int {
        hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0));
        QWeight plan; const size_t need = plan_qweight(ggml_type, rows, cols, plan, nullptr);
        std::vector<std::unique_ptr<DevBuf>> keep;
        QWeight W[3];
        for (int i = 0; i < n_mat; i++) {
            keep.emplace_back(new DevBuf(need));
            uint8_t *base = (uint8_t *)keep.back()->p;
            plan_qweight(ggml_type, rows, cols, W[i], base);
            launch_fill_random(base, need, (unsigned)(i * 7919 + 13), nullptr);
            const size_t n = (size_t)rows * cols;
            if (ggml_type == GT_Q4_K || ggml_type == GT_Q5_K) launch_fill_u16((void *)W[i].sc, n / 256 * 16 / 2, 0x1C00, nullptr);
            if (W[i].d) launch_fill_u16((void *)W[i].d, n / 256, 0x1C00, nullptr);
        }
        ActQ A; alloc_act(A, keep, (size_t)N, (size_t)cols);
        const size_t out_each = (size_t)N * rows;
        DevBuf dx((size_t)N * cols * 4), dy(out_each * n_mat * 4), dws(out_each * n_mat * 16 * 4);
        { std::vector<float> hx((size_t)N * cols); for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)((int)(i * 37 % 201) - 100) / 64.0f; HIP_CHECK(hipMemcpy(dx.p, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); }
        launch_rms_quant(dx.as<float>(), nullptr, N, cols, A, act_mask_for(ggml_type), nullptr);
        A.ws = dws.as<float>(); A.ws_floats = out_each * n_mat * 16;
        set_mmq2_cus(prop.multiProcessorCount);
        struct KsScope { KsScope(int k) { set_mmq2_tuning(-1, -1, k); set_mmq3_tuning(-1, k); } ~KsScope() { set_mmq2_tuning(-1, -1, 0); set_mmq3_tuning(-1, 0); } } ks_scope(std::max(0, ks));
        set_mmq3_tuning(prop.multiProcessorCount, -1);
        { const char *e = getenv("MMQ3_NW"); set_mmq3_waves(e ? atoi(e) : 8); }
        const uint8_t *Pp[3] = {nullptr, nullptr, nullptr};
        if (generation == 3) {
            if (!mmq3_supported(ggml_type, rows, cols)) return 4;
            const size_t pb = mmq3_plane_bytes(rows, cols);
            for (int i = 0; i < n_mat; i++) { keep.emplace_back(new DevBuf(pb)); Pp[i] = (const uint8_t *)keep.back()->p; launch_mmq3_build(W[i], (uint8_t *)keep.back()->p, nullptr); }
        }
        const int keep_gen = mmq_enabled();
        set_mmq_enabled(std::min(generation, 2));
        struct HScope { int keep; HScope(int v) : keep(mmqh_enabled()) { set_mmqh(v); } ~HScope() { set_mmqh(keep); } } h_scope(generation == 4);   // 4: the fp16-MFMA form of the Q4_K / Q5_K launches (k_mmqh_q45k), 2: the int8 kernels
        const QWeight *Wp[3]; float *Yp[3];
        for (int m = 0; m < n_mat; m++) { Wp[m] = &W[m]; Yp[m] = dy.as<float>() + (size_t)m * out_each; }
        auto run = [&]() {
            if (generation == 3) { if (!launch_mmq3_set(Wp, Pp, Yp, nullptr, n_mat, A, N, rows, nullptr)) throw HipError{hipErrorInvalidValue, "mmq3 refused the shape", __FILE__, __LINE__}; return; }
            if (generation >= 2 && launch_mmq2_set(Wp, Yp, nullptr, n_mat, A, N, rows, nullptr)) return;
            for (int m = 0; m < n_mat; m++) launch_mul_mat(*Wp[m], A, N, Yp[m], rows, nullptr, nullptr);
        };
        run(); run();
        HIP_CHECK(hipDeviceSynchronize());
        hipEvent_t ea, eb; HIP_CHECK(hipEventCreate(&ea)); HIP_CHECK(hipEventCreate(&eb));
        HIP_CHECK(hipEventRecord(ea, nullptr));
        for (int i = 0; i < iters; i++) run();
        HIP_CHECK(hipEventRecord(eb, nullptr));
        HIP_CHECK(hipDeviceSynchronize());
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ea, eb));
        HIP_IGNORE(hipEventDestroy(ea)); HIP_IGNORE(hipEventDestroy(eb));
        set_mmq_enabled(keep_gen);
        if (us_per_launch) *us_per_launch = ms * 1e3f / (float)iters;
        return 0;
    });
}

// Kernels of directions that were measured and closed (mmq3_kernels.hip: the digit-plane prompt mat-mul of round 4; tn_mfma_probe.hip: the batched-decode MFMA probe of round 5)
// are built into the test library only by `make test-extras` (-DMG4_TEST_EXTRAS); the default test library carries these refusing stubs, and minigpt4_amd_test_extras() says which.
#ifndef MG4_TEST_EXTRAS
extern "C++" {
namespace mg4 {
bool mmq3_supported(int, int, int) { return false; }
size_t mmq3_plane_bytes(int, int, size_t *) { return 0; }
void launch_mmq3_build(const QWeight &, uint8_t *, hipStream_t) {}
bool launch_mmq3_set(const QWeight *const *, const uint8_t *const *, float *const *, const float *const *, int, const ActQ &, int, int, hipStream_t, SlabSrc *) { return false; }
void set_mmq3_waves(int) {}
void set_mmq3_tuning(int, int) {}
int probe_tn_mfma(int, int, int, int, int, int, int, float *, float *) { return 5; }
}
}  // extern "C++"
int minigpt4_amd_test_extras(void) { return 0; }
#else
int minigpt4_amd_test_extras(void) { return 1; }
#endif
int minigpt4_amd_probe_tn_mfma(int rows, int cols, int TN, int iters, int n_sets, int check, float *us_per_launch, float *rel_diff) {
    if (device_count_noexcept() <= 0) return 2;
    return guarded(3, [&]() -> int { hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0)); return probe_tn_mfma(rows, cols, TN, iters, n_sets, check, prop.multiProcessorCount, us_per_launch, rel_diff); });
}
float minigpt4_amd_probe_valu(int op, int waves_per_simd, int iters) {
    if (device_count_noexcept() <= 0 || iters < 1) return -1.0f;
    float ns = -1.0f;
    guarded(1, [&] { hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0)); set_matvec_tuning(0, 0, prop.multiProcessorCount); ns = probe_valu_ns(op, waves_per_simd, iters, prop.multiProcessorCount); return 0; });
    return ns;
}
int minigpt4_amd_dist_env(int *world, int *rank, char *id_file, size_t cap, char *err, size_t err_cap) { return parse_dist_env_for_test(world, rank, id_file, cap, err, err_cap); }
float minigpt4_amd_probe_dma(int form, int policy, int waves, int fill, int depth, int deal, double total_gb) {
    float r = -1.0f;
    guarded(1, [&] { r = probe_dma_GBps(form, policy, waves, fill, depth, deal, (size_t)(total_gb * 1e9)); return 0; });
    return r;
}
float minigpt4_amd_probe_grid_barrier(int n_blocks, int iters, unsigned *errors) {
    if (n_blocks < 1 || n_blocks > 1024 || iters < 1 || device_count_noexcept() <= 0) return -1.0f;
    float us = -1.0f;
    guarded(1, [&] { us = probe_grid_barrier_us(n_blocks, iters, errors); return 0; });
    return us;
}

// ---- host-only logic -----------------------------------------------------------------------------------------------------------
struct MiniGPT4Vocab { LLMFile f; Tokenizer t; };
struct MiniGPT4Vocab *minigpt4_amd_vocab_load(const char *llm_path) {
    if (!file_exists(llm_path)) return nullptr;
    MiniGPT4Vocab *v = new (std::nothrow) MiniGPT4Vocab();
    if (!v) return nullptr;
    if (v->f.load(llm_path, true)) { delete v; return nullptr; }
    v->t.init(v->f);
    return v;
}
void minigpt4_amd_vocab_free(struct MiniGPT4Vocab *v) { delete v; }
int minigpt4_amd_vocab_size(struct MiniGPT4Vocab *v) { return v ? (int)v->f.n_vocab : 0; }
const char *minigpt4_amd_vocab_piece(struct MiniGPT4Vocab *v, int id, int *len) {
    if (!v || id < 0 || id >= (int)v->f.pieces.size()) return nullptr;
    if (len) *len = (int)v->f.pieces[(size_t)id].size();
    return v->f.pieces[(size_t)id].c_str();
}
int minigpt4_amd_vocab_tokenize(struct MiniGPT4Vocab *v, const char *text, int add_bos, int32_t *out, int cap) {
    if (!v || !text) return -1;
    const std::vector<int> t = v->t.tokenize(text, add_bos != 0);
    for (int i = 0; i < (int)t.size() && i < cap; i++) out[i] = t[(size_t)i];
    return (int)t.size();
}
int minigpt4_amd_inspect_files(const char *vision_path, const char *llm_path, int *n_vision_tensors, int *n_llm_tensors, int64_t *llm_weight_bytes_per_token) {
    if (vision_path) {
        if (!file_exists(vision_path)) return E_PathDoesNotExist;
        VisionFile vf; if (int e = vf.load(vision_path)) return e;
        int n = 0; for (auto &m : vf.models) n += (int)m.second.size();
        if (n_vision_tensors) *n_vision_tensors = n;
    }
    if (llm_path) {
        if (!file_exists(llm_path)) return E_PathDoesNotExist;
        LLMFile lf; if (int e = lf.load(llm_path)) return e;
        if (n_llm_tensors) *n_llm_tensors = (int)lf.tensors.size();
        if (llm_weight_bytes_per_token) { int64_t b = 0; for (auto &t : lf.tensors) b += t.first == "tok_embeddings.weight" ? (int64_t)gt_nbytes(t.second.type, (size_t)t.second.ne[0]) : (int64_t)t.second.nbytes; *llm_weight_bytes_per_token = b; }
    }
    return E_None;
}
int minigpt4_amd_resample_coeffs(int in_size, int out_size, int *ksize, int *first, int *count, int *kk, size_t kk_cap) {
    if (in_size <= 0 || out_size <= 0 || in_size > (1 << 24) || out_size > (1 << 16)) return -1;
    ResampleCoeffs c; precompute_bicubic_8bpc(in_size, out_size, c);
    if (ksize) *ksize = c.ksize;
    if (first) memcpy(first, c.first.data(), (size_t)out_size * 4);
    if (count) memcpy(count, c.count.data(), (size_t)out_size * 4);
    if (kk) { if (kk_cap < c.kk.size()) return -2; memcpy(kk, c.kk.data(), c.kk.size() * 4); }
    return 0;
}
// FNV-1a digest of everything the engine takes from an LLM file: hyper-parameters, vocabulary (pieces + scores) and every tensor (name, type, shape, bytes),
// tensors in name order.  A GGUF file and the GGJT v3 file of the same model must give the same digest.
int minigpt4_amd_llm_file_digest(const char *llm_path, uint64_t *digest, int with_data) {
    if (!llm_path || !digest) return E_LoadLanguageModel;
    if (!file_exists(llm_path)) return E_PathDoesNotExist;
    LLMFile f;
    if (int e = f.load(llm_path)) return e;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) { const uint8_t *b = static_cast<const uint8_t *>(p); for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    const uint32_t hp[5] = {f.n_vocab, f.n_embd, f.n_head, f.n_layer, f.n_ff()};
    mix(hp, sizeof(hp));
    for (size_t i = 0; i < f.pieces.size(); i++) { const uint32_t n = (uint32_t)f.pieces[i].size(); mix(&n, 4); mix(f.pieces[i].data(), n); mix(&f.scores[i], 4); }
    for (auto &kv : f.tensors) {
        const TensorMeta &t = kv.second;
        mix(kv.first.data(), kv.first.size()); mix(&t.type, 4);
        for (int64_t d : t.ne) mix(&d, 8);
        if (with_data) mix(f.mf.data + t.offset, t.nbytes);
    }
    *digest = h;
    return E_None;
}
int64_t minigpt4_amd_quantize_chunk(int ggml_type, const float *x, void *dst, int64_t n) {
    if (!x || !dst || n <= 0) return 0;
    return (int64_t)quantize_chunk(ggml_type, x, static_cast<uint8_t *>(dst), (size_t)n);
}
int minigpt4_amd_sample_logits(const float *logits, int n_vocab, int seed, float temp, int32_t top_k, float top_p, float tfs_z, float typical_p, int mirostat, float mirostat_tau, float mirostat_eta) {
    if (!logits || n_vocab <= 0) return -1;
    Sampler s; s.seed(seed);
    SampleParams p; p.temp = temp; p.top_k = top_k; p.top_p = top_p; p.tfs_z = tfs_z; p.typical_p = typical_p; p.mirostat = mirostat; p.mirostat_tau = mirostat_tau; p.mirostat_eta = mirostat_eta;
    return s.sample(logits, n_vocab, p);
}

}  // extern "C"
