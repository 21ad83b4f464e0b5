// Device-side wave64 primitives on the DPP crossbar (no LDS round trips, no lgkmcnt waits) shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

namespace mg4 {

// wave64 sum on the DPP crossbar (no LDS round trips): quad_perm x2, row_half_mirror, row_mirror leave every lane of a 16-lane row with
// the row sum; the four row sums are then combined through v_readlane.  The result is wave-uniform.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);   // row_half_mirror
    v += dpp_f<0x140>(v);   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
// fp32 -> fp16, IEEE round-to-nearest-even of the fp32 VALUE (what ggml's F16C / software conversion does to a float that sits in memory).  The empty asm makes the value
// opaque: otherwise hipcc folds the multiply that produced it into v_fma_mixlo_f16, which rounds the EXACT product once -- the CPU rounds twice (to fp32, then to fp16) and the
// two differ for ~6e-5 of random operands (tools/probe_mixlo.hip: 1045 of 2^24).  Found by MINIGPT4_PARITY as a 1-ulp softmax-probability difference (round 3).
__device__ __forceinline__ __half f2h_rn(float x) { asm("" : "+v"(x)); return __float2half_rn(x); }
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)dpp_i<CTRL>((int)(unsigned)b), hi = (unsigned)dpp_i<CTRL>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<0xB1>(v); v += dpp_d<0x4E>(v); v += dpp_d<0x141>(v); v += dpp_d<0x140>(v);
    const long long b = __builtin_bit_cast(long long, v);
    double r = 0.0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
        r += __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
    }
    return r;
}

}  // namespace mg4
