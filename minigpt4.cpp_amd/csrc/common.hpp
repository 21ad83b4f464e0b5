// Shared host/device declarations of the MI355X MiniGPT-4 engine.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace mg4 {

// MiniGPT4Error (reference minigpt4.cpp:97-119) -- values are part of the ABI.
enum Err : int {
    E_None, E_LoadModelFileHeader, E_LoadModelFileVersion, E_LoadModelMiniGPT4DataType, E_LoadLanguageModel, E_OpenImage, E_ImageSize,
    E_MmapSupport, E_FailedToAddString, E_LLamaProjectionEmbeddingInvalidSize, E_FailedToAddEmbedding, E_EosToken, E_Eos,
    E_ImageNot224_244_3, E_ImageNotF32, E_ImageChannelsExpectedRGB, E_ImageFormatExpectedU8, E_PathDoesNotExist, E_DumpModelFileOpen,
    E_OpenCVNotLinked,
};

// ggml_type numbering (LLM file); the vision file's MiniGPT4DataType is mapped onto it by the loader.
enum GType : int { GT_F32 = 0, GT_F16 = 1, GT_Q4_0 = 2, GT_Q4_1 = 3, GT_Q5_0 = 6, GT_Q5_1 = 7, GT_Q8_0 = 8, GT_Q8_1 = 9, GT_Q2_K = 10,
                   GT_Q3_K = 11, GT_Q4_K = 12, GT_Q5_K = 13, GT_Q6_K = 14, GT_Q8_K = 15, GT_I32 = 18, GT_I64 = 19 };

inline int gt_block(int t) { switch (t) { case GT_F32: case GT_F16: case GT_I32: case GT_I64: return 1; case GT_Q2_K: case GT_Q3_K: case GT_Q4_K: case GT_Q5_K: case GT_Q6_K: case GT_Q8_K: return 256; default: return 32; } }
inline int gt_bytes(int t) {
    switch (t) { case GT_F32: case GT_I32: return 4; case GT_F16: return 2; case GT_I64: return 8; case GT_Q4_0: return 18; case GT_Q4_1: return 20; case GT_Q5_0: return 22;
        case GT_Q5_1: return 24; case GT_Q8_0: return 34; case GT_Q8_1: return 40; case GT_Q2_K: return 84; case GT_Q3_K: return 110; case GT_Q4_K: return 144;
        case GT_Q5_K: return 176; case GT_Q6_K: return 210; case GT_Q8_K: return 292; default: return 0; }
}
inline size_t gt_nbytes(int t, size_t n) { return n / (size_t)gt_block(t) * (size_t)gt_bytes(t); }
const char *gt_name(int t);

extern int g_verbosity;  // MiniGPT4Verbosity
void set_last_error(const std::string &s);
const std::string &last_error();
void log_msg(int level, const char *tag, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
#define MG4_DEBUG(...) ::mg4::log_msg(3, "DEBUG", __VA_ARGS__)
#define MG4_INFO(...) ::mg4::log_msg(2, "INFO", __VA_ARGS__)
#define MG4_ERR(...) ::mg4::log_msg(1, "ERROR", __VA_ARGS__)

struct HipError { hipError_t code; const char *what; const char *file; int line; };
#define HIP_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw ::mg4::HipError{_e, #expr, __FILE__, __LINE__}; } while (0)
// A call whose failure is acceptable (attribute hints, teardown): the runtime's sticky per-thread "last error" is cleared right away, so that whatever IS left in that
// slot when an entry point returns comes from a kernel launch (which has no return value) and is reported (api.cpp: guarded()).
#define HIP_IGNORE(expr) do { (void)(expr); (void)hipGetLastError(); } while (0)

// The > 64 KiB dynamic-LDS opt-in of a kernel.  The request must leave room for the kernel's STATIC shared arrays: asking for the CU's full 160 KiB is refused (invalid
// argument) for any kernel that has some -- which rounds 1-4 did, behind HIP_IGNORE, so those opt-ins never took effect (found in round 5 when the call became checked).
template <typename K> inline hipError_t lds_optin_max(K kernel) {
    hipFuncAttributes fa;
    const hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kernel));
    if (e != hipSuccess) return e;
    const int dyn = (160 * 1024 - (int)fa.sharedSizeBytes) & ~255;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
}

// ------------------------------------------------------------------------------------------------------------
// Device-side views
// ------------------------------------------------------------------------------------------------------------

// One 2-D weight matrix in HBM, repacked at load time from ggml's array-of-blocks into planes (DESIGN.md "HBM layout").
// A "unit" is 16 bytes of the main quant plane = what one lane fetches with one dwordx4 load:
//   Q4_0/Q4_1/Q5_0/Q5_1 : 1 unit = 1 block (32 weights)        Q8_0 : 1 unit = half a block (16 weights)
//   Q4_K/Q5_K/Q6_K      : 1 unit = 32 weights (8 units / super-block)
//   F16 : 8 weights, F32 : 4 weights
struct QWeight {
    int type = -1;
    int rows = 0, cols = 0;          // n_out, n_in
    const uint8_t *qs = nullptr;     // main plane  [rows][units][16]
    const uint8_t *qh = nullptr;     // high-bit plane, pre-transposed for v_dot4: Q5_*: 4 B/unit, Q6_K: 8 B/unit
    const uint8_t *sc = nullptr;     // scale plane: Q4_0/Q5_0/Q8_0: f16 d per block; Q4_1/Q5_1: {d,m}; Q4_K/Q5_K: 16 B {d,dmin,scales12} per
                                     //   super-block; Q6_K: 2 int8 per unit
    const uint8_t *d = nullptr;      // Q6_K: f16 d per super-block
    size_t bytes = 0;                // HBM bytes of all planes (== file bytes of the tensor)
};

// Row-interleaved second image of a k-quant matrix for the batched decode step on the int8 matrix cores (ri_kernels.hip): per group of 64 rows and per unit the 64 rows'
// pieces back to back, so that a LANE owns a weight row and a wave load is still 1 KiB of whole cache lines.  Built on the device when a context gets > 1 conversation.
struct RiPlanes { const uint8_t *qs = nullptr, *qh = nullptr, *sc = nullptr, *d = nullptr; };

// Quantised activations ("vec_dot_type" of ggml) for N rows of width K.
struct ActQ {
    int8_t *q8k = nullptr;   // [N][K]     Q8_K values
    float *dk = nullptr;     // [N][K/256]
    int16_t *bsk = nullptr;  // [N][K/16]  Q8_K bsums
    int8_t *bsq = nullptr;   // [N][K/256][16] (optional) per-32 sums of the Q8_K values split for the int8 matrix cores: bytes 0..7 = s & 127, bytes 8..15 = s >> 7 (s = 128 hi + lo)
    // (optional, round 5) fp16 images of the SAME Q8_K row for the fp16-MFMA prompt mat-mul (k_mmqh_q45k): the int8 values as fp16, in the kernel's fragment order --
    // per super-block 32 chunks of 8 halfs, chunk c = 8 p + 4 uu + d holds elements 64 p + 16 uu + 4 d + {0, 2, 1, 3} and the same + 32 -- and the digit-split per-32 sums as fp16
    __half *q16 = nullptr;   // [N][K]
    __half *bs16 = nullptr;  // [N][K/256][16]: halfs 0..7 = s & 127, 8..15 = s >> 7
    int8_t *q80 = nullptr;   // [N][K]     Q8_0 / Q8_1 values (identical)
    float *d0 = nullptr;     // [N][K/32]  Q8_0 d, fp16-rounded
    float *d1 = nullptr;     // [N][K/32]  Q8_1 d (float)
    float *s1 = nullptr;     // [N][K/32]  Q8_1 s = d * sum(q)
    int *sum0 = nullptr;     // [N][K/32]  sum(q) per block
    __half *xh = nullptr;    // [N][K]     fp16-rounded activations (F16 weights)
    float *xf = nullptr;     // [N][K]     fp32 activations (F32 weights)
    float *ws = nullptr; size_t ws_floats = 0;   // (optional) workspace that travels with the planes: K-split partial sums of the prefill mat-mul (mmq2_kernels.hip)
};
enum ActMask : int { ACT_Q8K = 1, ACT_Q80 = 2, ACT_F16 = 4, ACT_F32 = 8 };
__host__ __device__ inline int act_mask_for(int wtype) {
    switch (wtype) { case GT_Q4_0: case GT_Q5_0: case GT_Q8_0: case GT_Q4_1: case GT_Q5_1: return ACT_Q80; case GT_Q2_K: case GT_Q4_K: case GT_Q5_K: case GT_Q6_K: return ACT_Q8K;
        case GT_F16: return ACT_F16; case GT_F32: return ACT_F32; default: return 0; }
}

}  // namespace mg4
