// Hardware probes and benchmark helpers of libminigpt4_test.so ONLY (never linked into the product library): a device-wide barrier latency probe, the vector-ALU
// issue-rate probe of tools/probe_valu.py, and the random fill the mat-vec micro-benchmarks use for their synthetic weight planes.
#include "kernels.hpp"
#include "devutil.hpp"

#include <algorithm>

namespace mg4 {

static int g_probe_cus = 256;
typedef unsigned v4u_pr __attribute__((ext_vector_type(4)));
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

// LDS-DMA stream probe (tools/probe_dma.py): how fast can loader waves alone pull a large buffer into an LDS ring?  One workgroup per CU, `waves` loader waves, each
// keeping `depth` fills of `fill` bytes (whole KiB) in flight into its own slots of the ring; nobody reads the data.  form 0: scalar base + lane offset + instruction
// offsets, four pieces per M0 write (the decode engine's statement); form 1: per-lane 64-bit address, one piece per M0 write.  policy 0: nt, 1: default.  deal 0: every
// workgroup streams its own contiguous region, 1: fills dealt round-robin over the workgroups.  Compared with form 2: the same bytes through ordinary register loads
// (8 x dwordx4 per lane in flight per wave, what k_matvec_v2's pipeline does), and form 3: register loads + ds_write into the ring.
typedef __attribute__((address_space(3))) void *pr_lds_t;
template <int FORM, int POLICY>
__global__ __launch_bounds__(512) void k_probe_dma(const unsigned char *__restrict__ src, size_t bytes_per_wg, int fill, int depth, int deal, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
    const int n_fills = (int)(bytes_per_wg / (size_t)fill);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)ring);
    const int pieces = fill >> 10;
    unsigned acc = 0;
    int issued = 0;                                     // fills this wave has issued
    for (int f = wave; f < n_fills; f += n_waves, issued++) {
        const size_t gfill = deal ? (size_t)f * gridDim.x + blockIdx.x : (size_t)blockIdx.x * n_fills + f;
        const unsigned char *p = src + gfill * (size_t)fill;
        const unsigned slot = (unsigned)((wave * depth + issued % depth) * fill);
        if (FORM <= 1) {
            if (issued >= depth) {                      // the fill that used this slot must have landed: at most (depth - 1) fills' pieces may remain in flight
                const int left = (depth - 1) * pieces;
                switch (left) { case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break; case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
                    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break; case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break; case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
                    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break; case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
                    case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break; case 56: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; }
            }
            if (FORM == 0) {
                const unsigned long long a = (unsigned long long)(size_t)p;
                unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
                unsigned ld = lds0 + slot; const unsigned voff = (unsigned)lane * 16u;
                for (int k = 0; k + 4 <= pieces; k += 4, sb += 4096, ld += 4096) {
                    unsigned keep;
                    if (POLICY == 0) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:0 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sb), "s"(ld) : "memory");
                    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sb), "s"(ld) : "memory");
                }
            } else {
                for (int k = 0; k < pieces; k++) {
                    const unsigned char *g = p + (size_t)k * 1024 + lane * 16;
                    const unsigned ld = lds0 + slot + (unsigned)k * 1024u;
                    unsigned keep;
                    if (POLICY == 0) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(ld) : "memory");
                    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(ld) : "memory");
                }
            }
        } else {                                        // register loads: `pieces` x dwordx4 per lane, consumed (xor) one fill later so that two fills are in flight
            for (int k = 0; k < pieces; k += 8) {
                v4u_pr v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = POLICY == 0 ? __builtin_nontemporal_load(reinterpret_cast<const v4u_pr *>(p + (size_t)(k + j) * 1024 + lane * 16)) : *reinterpret_cast<const v4u_pr *>(p + (size_t)(k + j) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (FORM == 3) *reinterpret_cast<v4u_pr *>(ring + slot + (size_t)(k + j) * 1024 + lane * 16) = v[j];
                    else acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) *sink = acc;
}
// GB/s over the whole chip; < 0 on a bad argument
float probe_dma_GBps(int form, int policy, int waves, int fill, int depth, int deal, size_t total_bytes) {
    if (fill % 1024 || fill <= 0 || (form <= 1 && (fill >> 10) % 4) || waves < 1 || waves > 8 || depth < 1 || (size_t)waves * depth * fill > 150 * 1024) return -1.0f;
    hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t per_wg = total_bytes / cus / fill * fill, total = per_wg * cus;
    unsigned char *buf = nullptr; unsigned *sink = nullptr;
    HIP_CHECK(hipMalloc((void **)&buf, total + 4096)); HIP_CHECK(hipMalloc((void **)&sink, 4));
    launch_fill_random(buf, total, 7u, nullptr);
    const size_t lds = (size_t)waves * depth * fill;
    auto launch = [&](size_t bytes_per_wg) {
        const dim3 g((unsigned)cus), b((unsigned)waves * 64);
#define PR_GO(F, P) do { HIP_IGNORE(lds_optin_max(&k_probe_dma<F, P>)); \
                         hipLaunchKernelGGL((k_probe_dma<F, P>), g, b, lds, nullptr, buf, bytes_per_wg, fill, depth, deal, sink); } while (0)
        switch (form * 2 + policy) { case 0: PR_GO(0, 0); break; case 1: PR_GO(0, 1); break; case 2: PR_GO(1, 0); break; case 3: PR_GO(1, 1); break;
                                     case 4: PR_GO(2, 0); break; case 5: PR_GO(2, 1); break; case 6: PR_GO(3, 0); break; default: PR_GO(3, 1); break; }
#undef PR_GO
    };
    hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    launch((size_t)fill * waves * 4);                   // warm-up (code object, attributes)
    HIP_CHECK(hipEventRecord(a, nullptr));
    launch(per_wg);
    HIP_CHECK(hipEventRecord(b, nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    HIP_IGNORE(hipEventDestroy(a)); HIP_IGNORE(hipEventDestroy(b)); HIP_IGNORE(hipFree(buf)); HIP_IGNORE(hipFree(sink));
    return (float)((double)total / (ms * 1e-3) / 1e9);
}

}  // namespace mg4
