// Hardware probes and benchmark helpers of libminigpt4_test.so ONLY (never linked into the product library): a device-wide barrier latency probe, the vector-ALU
// issue-rate probe of tools/probe_valu.py, and the random fill the mat-vec micro-benchmarks use for their synthetic weight planes.
#include "kernels.hpp"
#include "devutil.hpp"

#include <algorithm>

namespace mg4 {

static int g_probe_cus = 256;
__global__ void k_fill_random(unsigned *p, size_t n_words, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; p[i] = x; }
}
void launch_fill_random(void *p, size_t bytes, unsigned seed, hipStream_t s) { hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, s, (unsigned *)p, bytes / 4, seed); }
// Device-wide barrier latency probe (is a persistent multi-phase decode kernel worth building?): `n_blocks` co-resident workgroups pass `iters` barriers.
// Between barriers every workgroup writes one word and reads its neighbour's (so the fences have something to make visible).
__device__ __forceinline__ void grid_sync(unsigned *bar, unsigned &target, unsigned n_blocks) {
    target += n_blocks;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(bar, 1u);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}
__global__ __launch_bounds__(512) void k_barrier_probe(unsigned *bar, unsigned *words, int iters, unsigned *errors) {
    unsigned target = 0;
    const unsigned nb = gridDim.x, me = blockIdx.x, nxt = (me + 1) % nb;
    unsigned bad = 0;
    for (int i = 0; i < iters; i++) {
        if (threadIdx.x == 0) words[me] = (unsigned)i * 2654435761u + me;
        grid_sync(bar, target, nb);
        if (threadIdx.x == 0) bad += words[nxt] != (unsigned)i * 2654435761u + nxt;
        grid_sync(bar, target, nb);
    }
    if (threadIdx.x == 0 && bad) atomicAdd(errors, bad);
}
float probe_grid_barrier_us(int n_blocks, int iters, unsigned *errors_out) {
    unsigned *d = nullptr;
    HIP_CHECK(hipMalloc((void **)&d, (size_t)(n_blocks + 2) * 4));
    HIP_CHECK(hipMemset(d, 0, (size_t)(n_blocks + 2) * 4));
    hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_barrier_probe, dim3((unsigned)n_blocks), dim3(512), 0, nullptr, d, d + 2, 4, d + 1);       // warm-up
    HIP_CHECK(hipMemset(d, 0, 8));
    HIP_CHECK(hipEventRecord(a, nullptr));
    hipLaunchKernelGGL(k_barrier_probe, dim3((unsigned)n_blocks), dim3(512), 0, nullptr, d, d + 2, iters, d + 1);
    HIP_CHECK(hipEventRecord(b, nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned err = 0; HIP_CHECK(hipMemcpy(&err, d + 1, 4, hipMemcpyDeviceToHost));
    if (errors_out) *errors_out = err;
    HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b)); HIP_IGNORE(hipFree(d));
    return ms * 1e3f / (float)(2 * iters);
}
// Vector-ALU issue-rate probe (tools/probe_valu.py): every wave issues `iters` x 64 instructions of ONE kind over 8 independent accumulators (no dependent-issue
// stalls); with 1 / 2 / 3 waves per SIMD the wall time per instruction tells the issue cost of that instruction relative to v_and_b32.
template <int OP> __device__ __forceinline__ void valu_probe_op(int &acc, int a, int b) {
    if constexpr (OP == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(acc) : "v"(a));
    else if constexpr (OP == 1) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (OP == 2) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(acc) : "v"(a));
    else if constexpr (OP == 3) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (OP == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (OP == 5) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (OP == 6) asm volatile("v_bfe_u32 %0, %0, 4, 6" : "+v"(acc));
    else if constexpr (OP == 7) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(acc));
    else if constexpr (OP == 8) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (OP == 9) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*reinterpret_cast<long long *>(&acc)) : "v"(a), "v"(b) : "vcc");
    else asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(acc));
}
template <int OP> __global__ __launch_bounds__(1024) void k_valu_probe(int iters, int *sink) {
    int acc[8];
    long long wide[8];                                       // OP 9 works on register pairs
#pragma unroll
    for (int i = 0; i < 8; i++) { acc[i] = (int)threadIdx.x + i; wide[i] = acc[i]; }
    const int a = (int)threadIdx.x * 0x01010101, b = 0x01020304 + (int)blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) { if constexpr (OP == 9) valu_probe_op<OP>(*reinterpret_cast<int *>(&wide[i]), a, b); else valu_probe_op<OP>(acc[i], a, b); }
    }
    int x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= acc[i] ^ (int)wide[i];
    if (x == 0x7FFFFFFF) *sink = x;
}
template <int OP> static float valu_probe_run(int threads, int iters) {
    int *d = nullptr; HIP_CHECK(hipMalloc((void **)&d, 4));
    hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_valu_probe<OP>, dim3((unsigned)g_probe_cus), dim3((unsigned)threads), 0, nullptr, 16, d);
    HIP_CHECK(hipEventRecord(a, nullptr));
    hipLaunchKernelGGL(k_valu_probe<OP>, dim3((unsigned)g_probe_cus), dim3((unsigned)threads), 0, nullptr, iters, d);
    HIP_CHECK(hipEventRecord(b, nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b)); HIP_IGNORE(hipFree(d));
    return ms * 1e6f / ((float)iters * 64.0f);              // ns per instruction of one wave
}
// ns per issued instruction and wave; threads = 256 x waves per SIMD (one workgroup per CU)
float probe_valu_ns(int op, int waves_per_simd, int iters, int cus) {
    g_probe_cus = cus > 0 ? cus : 256;
    const int threads = 256 * std::max(1, std::min(4, waves_per_simd));
    switch (op) {
    case 0: return valu_probe_run<0>(threads, iters); case 1: return valu_probe_run<1>(threads, iters); case 2: return valu_probe_run<2>(threads, iters);
    case 3: return valu_probe_run<3>(threads, iters); case 4: return valu_probe_run<4>(threads, iters); case 5: return valu_probe_run<5>(threads, iters);
    case 6: return valu_probe_run<6>(threads, iters); case 7: return valu_probe_run<7>(threads, iters); case 8: return valu_probe_run<8>(threads, iters);
    case 9: return valu_probe_run<9>(threads, iters); case 10: return valu_probe_run<10>(threads, iters); default: return -1.0f;
    }
}

}  // namespace mg4
