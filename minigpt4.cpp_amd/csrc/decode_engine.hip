// Persistent decode engine (round 4): ONE launch runs a chain of dependent decode mat-vecs -- attention output -> wo -> (+residual, ffn norm) -> w1|w3 -> silu * mul -> w2
// -> (+residual, next layer's attention norm) -> wq, wk, wv (or final norm -> output matrix) -- replacing five launches of k_matvec_v2 / k_matvec_mix / k_silu_mul_quant per
// layer of the reference's per-token ggml graph (llama_eval behind minigpt4.cpp:2373).  Structure (MI355X_MICROARCH.md, persistent-kernel price list, row engine-vs-launches):
//
//   * one workgroup per CU: wave 0 = LOADER, waves 1..8 = CONSUMERS.  The loader streams this CU's share of every matrix of the chain -- whole row groups, all planes of a
//     row group into one 17 KiB ring slot -- global -> LDS by LDS-DMA (global_load_lds, non-temporal), 8 slots deep, and never waits for a dependency: while the consumers
//     sit in a hand-off the ring fills with the NEXT op's weights (the guide's prefetch-credit), which is what the launch-per-op form cannot do.
//   * a consumer wave owns every 8th fill: it waits for the slot's `filled` word (LDS), multiplies the rows of the fill against the activation units it keeps in registers --
//     the SAME unit traits, the same lane -> unit map (u = lane + 64 i), the same per-lane fma order and the same DPP wave reduction as k_matvec_v2, so every output is
//     bit-identical to the launch-per-op path -- and gives the slot back through the `done` word.
//   * hand-offs between ops: an op's output vector is published as 8-byte {tag, value} granules (ONE agent-scope store each; the data is the flag, cdna_hip_programming.md
//     Guideline 16 R2); every workgroup's consumer waves sweep the whole vector (relaxed agent-scope loads, re-reading a round until its tags match), then prepare the next
//     op's activation row exactly as k_matvec_v2's fused prologue does at 512 threads (rms-norm with the double sum in that order, Q8_K / Q8_0 quantisation) into an LDS
//     image.  Tags are (step counter from device memory, layer, buffer): nothing needs zeroing between launches or graph replays.
//   * no s_barrier after the entry: consumer-only barriers are an LDS counter; every spin is bounded (EG_TIMEOUT_TICKS) and gives up with an error word instead of hanging.
//
// Attention stays its own launch between two engine launches (the chain is cut at qkv -> attention: the ring cannot cover that seam anyway, and the key-split long-context
// kernels keep working unchanged).
#include "kernels.hpp"
#include "devutil.hpp"
#include "qtraits.hpp"

#include <algorithm>
#include <cstring>
#include <hip/hip_ext.h>

namespace mg4 {

#ifndef EG_NL
#define EG_NL 1                                  // loader waves (tools/probe_dma.py: one wave alone streams 5.3 TB/s chip-wide, two 5.8)
#endif
constexpr int EG_NC = 8 - EG_NL;                 // consumer waves (with the loaders: 8 waves = two per SIMD, so a wave may hold 256 VGPRs -- the widest rows keep 77 of activations)
constexpr int EG_NV = 8;                         // VIRTUAL waves of the row preparation: k_matvec_v2's fused prologue runs at 512 threads, and its summation order is kept
constexpr int EG_THREADS = (EG_NC + EG_NL) * 64;  // waves 0 .. EG_NL-1 = loaders
constexpr int EG_NS = 8;                         // ring slots
constexpr int EG_SLOT = ENG_SLOT_BYTES;          // bytes per slot
constexpr int EG_IMG = 17408;                    // activation image
#ifndef EG_BEHIND
#define EG_BEHIND 2                              // fills in flight behind the one the loader is about to publish (x ipf <= 63 - ipf: vmcnt is 6 bits)
#endif
constexpr int EG_XRES = 256;                     // own output rows kept for a later residual
constexpr unsigned long long EG_TIMEOUT_TICKS = 3000000ull;   // 30 ms of the 100 MHz constant clock

struct EgShared {
    unsigned filled[EG_NS];     // fill index + 1 whose DMA has landed in the slot
    unsigned done[EG_NS];       // fill index + 1 whose rows have been consumed
    unsigned part[EG_NS];       // rows of the slot's current fill consumed so far
    unsigned bar;               // consumer barrier: arrivals so far
    unsigned fin;               // consumer waves that have finished their rows, summed over the ops so far (the last one of an op publishes the workgroup's sentinel)
    unsigned ready;             // hand-offs whose sentinels consumer wave 0 has seen complete
    unsigned abort;             // a spin gave up: every wait returns at once from now on
    double red[EG_NV];
    float xres[EG_XRES];
};
constexpr size_t EG_LDS_BYTES = (size_t)EG_NS * EG_SLOT + EG_IMG + sizeof(EgShared);
static_assert(EG_LDS_BYTES <= 160 * 1024, "ring + image + control words must fit one CU's LDS");

// In-kernel timeline (diagnostic builds, -DMG4_TIMELINE; tools/timeline_engine.py): 64 stamps of the 100 MHz clock per workgroup, kept for launches of layer g_etl_layer only.
//   0 consumer entry | per op o: 1+4o preparation begins (consumer wave 0), 2+4o its row is gathered, 3+4o image complete, 4+4o the LATEST consumer wave has finished the op's fills
//   32 loader entry | 33+o first fill of op o issued, 44+o last fill of op o issued | 43 everything landed | 56+o time the loader spent waiting for free slots during op o (ticks)
#ifdef MG4_TIMELINE
__device__ unsigned long long g_etl[512 * 64];
__device__ int g_etl_layer = 1;
__device__ unsigned long long g_efl[4 * 128];      // workgroup 7's fills: [0] issued by the loader, [1] published (`filled`), [2] its consumer starts on it, [3] the consumer is done
#define EG_TL(i) do { if (tl_on && lane == 0 && blockIdx.x < 512 && (i) < 64) g_etl[blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define EG_TL_MAX(i) do { if (tl_on && lane == 0 && blockIdx.x < 512 && (i) < 64) atomicMax(&g_etl[blockIdx.x * 64 + (i)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define EG_TL_ADD(i, v) do { if (tl_on && lane == 0 && blockIdx.x < 512 && (i) < 64) g_etl[blockIdx.x * 64 + (i)] += (v); } while (0)
#define EG_TL_ZERO() do { if (tl_on && blockIdx.x < 512 && threadIdx.x < 64) g_etl[blockIdx.x * 64 + threadIdx.x] = 0; } while (0)
#else
#define EG_TL(i) do {} while (0)
#define EG_TL_MAX(i) do {} while (0)
#define EG_TL_ADD(i, v) do {} while (0)
#define EG_TL_ZERO() do {} while (0)
#endif

typedef __attribute__((address_space(3))) void *eg_lds_ptr_t;
typedef __attribute__((address_space(1))) const void *eg_glb_ptr_t;

#define EG_RLX_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP
__device__ __forceinline__ void eg_give_up(EgShared *sh, unsigned *err, unsigned code) {
    __hip_atomic_store(&sh->abort, 1u, EG_RLX_WG);
    __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wave-uniform wait for an LDS word to reach `want`; false = gave up (timeout here or elsewhere in the workgroup)
__device__ __forceinline__ bool eg_wait_ge(unsigned *flag, unsigned want, EgShared *sh, unsigned *err, unsigned code) {
    unsigned spins = 0; unsigned long long t0 = 0;
    for (;;) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int)(v - want) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 127) == 0) {
            if (__hip_atomic_load(&sh->abort, EG_RLX_WG)) return false;
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if (!t0) t0 = t; else if (t - t0 > EG_TIMEOUT_TICKS) { eg_give_up(sh, err, code); return false; }
        }
    }
}
__device__ __forceinline__ void eg_wait_vm(int n) {   // at most n vector-memory operations of this wave still in flight
    switch (n) {
#define EG_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    EG_VM(0) EG_VM(1) EG_VM(2) EG_VM(3) EG_VM(4) EG_VM(5) EG_VM(6) EG_VM(7) EG_VM(8) EG_VM(9) EG_VM(10) EG_VM(11) EG_VM(12) EG_VM(13) EG_VM(14) EG_VM(15) EG_VM(16) EG_VM(17) EG_VM(18) EG_VM(19)
    EG_VM(20) EG_VM(21) EG_VM(22) EG_VM(23) EG_VM(24) EG_VM(25) EG_VM(26) EG_VM(27) EG_VM(28) EG_VM(29) EG_VM(30) EG_VM(31) EG_VM(32) EG_VM(33) EG_VM(34) EG_VM(35) EG_VM(36) EG_VM(37) EG_VM(38) EG_VM(39)
    EG_VM(40) EG_VM(41) EG_VM(42) EG_VM(43) EG_VM(44) EG_VM(45) EG_VM(46) EG_VM(47) EG_VM(48)
#undef EG_VM
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;   // n > 48: waiting for fewer in flight is always safe
    }
}

// LDS-DMA as inline asm: hipcc must NOT know that LDS writes are in flight -- it otherwise puts s_waitcnt vmcnt(0) in front of every LDS access of the loader (the `done` /
// `filled` words), which drained the whole queue once per fill (first timeline of this kernel: 2.4 us per 14 KiB fill).  The loader counts its own completions (eg_wait_vm).
// M0 = LDS byte address of lane 0's element; lane l lands at M0 + l * size (cdna_hip_programming.md 5.7).
__device__ __forceinline__ unsigned eg_lds_addr(const void *p) { return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const void *)p); }
__device__ __forceinline__ unsigned long long eg_rfl64(unsigned long long a) {   // a wave-uniform value the compiler may have parked in vector registers -> scalar
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);   // (the builtin returns int)
}
__device__ __forceinline__ unsigned long long eg_uniform64(const void *p) { return eg_rfl64((unsigned long long)(size_t)p); }
// N (1..4) whole 1-KiB pieces from the scalar base `src` (+ the lane's constant offset voff = lane * 16): the instruction offset moves the global AND the LDS address, so
// the pieces share one M0 and one address.  ~6 scalar instructions + N DMA issues per statement.
#define EG_DMA_HEAD "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
#ifndef EG_POLICY
#define EG_POLICY " nt"                       // streamed once by one CU: non-temporal (MI355X_MICROARCH.md, row nt-weights); -DEG_POLICY='""' builds the default-policy arm
#endif
#define EG_DMA16(OFF) "global_load_lds_dwordx4 %1, %2 offset:" #OFF EG_POLICY "\n\t"
__device__ __forceinline__ void eg_dma16_n(int n, unsigned voff, unsigned long long src, unsigned lds) {
    unsigned keep;
    src = eg_rfl64(src); lds = __builtin_amdgcn_readfirstlane(lds);
    switch (n) {
    case 1: asm volatile(EG_DMA_HEAD EG_DMA16(0) "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds) : "memory"); break;
    case 2: asm volatile(EG_DMA_HEAD EG_DMA16(0) EG_DMA16(1024) "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds) : "memory"); break;
    case 3: asm volatile(EG_DMA_HEAD EG_DMA16(0) EG_DMA16(1024) EG_DMA16(2048) "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds) : "memory"); break;
    default: asm volatile(EG_DMA_HEAD EG_DMA16(0) EG_DMA16(1024) EG_DMA16(2048) EG_DMA16(3072) "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds) : "memory"); break;
    }
}
// one piece with only the lanes of `mask` active (the tail of a plane: nothing is read or written beyond it)
__device__ __forceinline__ void eg_dma16_masked(unsigned long long mask, unsigned voff, unsigned long long src, unsigned lds) {
    unsigned keep; unsigned long long keepx;
    src = eg_rfl64(src); lds = __builtin_amdgcn_readfirstlane(lds); mask = eg_rfl64(mask);
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %5\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" EG_POLICY "\n\ts_mov_b32 m0, %0\n\ts_mov_b64 exec, %1"
                 : "=&s"(keep), "=&s"(keepx) : "v"(voff), "s"(src), "s"(lds), "s"(mask) : "memory");
}

// Partition units are dealt CYCLICALLY: unit u belongs to workgroup u % n_cus, so at any moment the 256 loaders read one compact, advancing window of the image (neighbouring
// workgroups read neighbouring fills) instead of 256 separate streams.  Returns this workgroup's number of units; its j-th unit is cu + j * n_cus.
__device__ __forceinline__ int eg_share(const EngOp &op, int cu, int n_cus) {
    const int U = op.rows / op.unit;
    return cu < U ? (U - cu + n_cus - 1) / n_cus : 0;
}
__device__ __forceinline__ int eg_log2(int v) { return v >= 4 ? 2 : v >= 2 ? 1 : 0; }   // fills per unit: 1, 2 or 4
// global fill index (= position in the image) of this workgroup's i-th fill of the op
__device__ __forceinline__ int eg_gfill(int i, int fsh, int cu, int n_cus) { return (((i >> fsh) * n_cus + cu) << fsh) + (i & ((1 << fsh) - 1)); }

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------------
// loader wave
// ---------------------------------------------------------------------------------------------------------------------------------------------------------------------
// Loader wave l of EG_NL: fills l, l + EG_NL, l + 2 EG_NL, ... of the workgroup's fill sequence (one wave alone sustains only ~15 GB/s of LDS-DMA whatever its queue depth --
// first timelines of this kernel -- so the stream is split over two waves on two SIMDs).
__device__ __forceinline__ void eg_loader(const EngOp *__restrict__ ops, const int n_ops, unsigned char *smem, EgShared *sh, unsigned *err, const int cu, const int n_cus, const int l, const bool tl_on) {
    const int lane = threadIdx.x & 63; (void)tl_on;
    const unsigned voff16 = (unsigned)lane * 16u, lds0 = eg_lds_addr(smem);
    int gf = l, gr = l;                     // global index of this wave's next fill to issue / oldest fill not yet marked `filled`
    // DMA instructions of the issued, not yet published fills, newest first: vmcnt counts instructions, so "fill g has landed" = "at most the instructions issued behind g
    // are still in flight".
    int nq0 = 0, nq1 = 0, nq2 = 0, nq3 = 0, nq4 = 0;
    bool alive = true;
    auto behind = [&](int pend) { return pend >= 6 ? nq0 + nq1 + nq2 + nq3 + nq4 : pend == 5 ? nq0 + nq1 + nq2 + nq3 : pend == 4 ? nq0 + nq1 + nq2 : pend == 3 ? nq0 + nq1 : pend == 2 ? nq0 : 0; };
    auto pending = [&]() { return (gf - gr) / EG_NL; };
    auto retire_one = [&]() { eg_wait_vm(behind(pending()));
                              asm volatile("" ::: "memory"); __hip_atomic_store(&sh->filled[gr % EG_NS], (unsigned)(gr + 1), EG_RLX_WG);
#ifdef MG4_TIMELINE
                              if (tl_on && blockIdx.x == 7 && lane == 0 && gr < 128) g_efl[128 + gr] = __builtin_amdgcn_s_memrealtime();
#endif
                              gr += EG_NL; };
    if (l == 0) EG_TL(32);
    int f0 = 0;                             // global index of the op's first fill
    for (int o = 0; o < n_ops && alive; o++) {
        const EngOp &op = ops[o];
        const int fsh = eg_log2(op.unit / op.G), nf = eg_share(op, cu, n_cus) << fsh, n_instr = op.ipf;
        const unsigned fill_bytes = (unsigned)op.fill_bytes, whole = fill_bytes >> 10, tail = (fill_bytes & 1023u) >> 4;
        const unsigned long long tmask = tail ? (1ull << tail) - 1ull : 0ull;
        const unsigned long long base = eg_uniform64(op.image);
        while (gf < f0 + nf && alive) {
            const int slot = gf % EG_NS;
            const unsigned long long src = base + (unsigned long long)(unsigned)eg_gfill(gf - f0, fsh, cu, n_cus) * fill_bytes;
            while (gr < gf && behind(pending() + 1) + n_instr > 63) retire_one();          // (an op with longer fills than its predecessor's)
            if (gf >= EG_NS) {                                                          // the slot's previous fill must have been consumed
                const unsigned want = (unsigned)(gf - EG_NS + 1);
                if ((int)(__hip_atomic_load(&sh->done[slot], EG_RLX_WG) - want) < 0) {
                    while (gr < gf) retire_one();                                       // about to sleep: everything issued must become visible first
#ifdef MG4_TIMELINE
                    const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
                    if (!eg_wait_ge(&sh->done[slot], want, sh, err, 0x100u + (unsigned)o)) { alive = false; break; }
#ifdef MG4_TIMELINE
                    if (l == 0) EG_TL_ADD(56 + o, __builtin_amdgcn_s_memrealtime() - tw0);
#endif
                }
            }
            const unsigned dst = lds0 + (unsigned)slot * EG_SLOT;
            unsigned k = 0;
            for (; k + 4 <= whole; k += 4) eg_dma16_n(4, voff16, src + ((unsigned long long)k << 10), dst + (k << 10));
            if (k < whole) eg_dma16_n((int)(whole - k), voff16, src + ((unsigned long long)k << 10), dst + (k << 10));
            if (tmask) eg_dma16_masked(tmask, voff16, src + ((unsigned long long)whole << 10), dst + (whole << 10));
#ifdef MG4_TIMELINE
            if (tl_on && blockIdx.x == 7 && lane == 0 && gf < 128) g_efl[gf] = __builtin_amdgcn_s_memrealtime();          // per-fill stamps of ONE workgroup: issued
            if (gf - f0 < EG_NL) EG_TL(33 + o); if (gf - f0 >= nf - EG_NL) EG_TL_MAX(44 + o);
#endif
            nq4 = nq3; nq3 = nq2; nq2 = nq1; nq1 = nq0; nq0 = n_instr;
            gf += EG_NL;
            // publish what must have landed before this wave's NEXT fill may be issued: at most EG_BEHIND fills behind the oldest, and never more than 60 instructions in flight
            while (pending() > EG_BEHIND + 1 || (pending() > 1 && behind(pending() + 1) + n_instr > 60)) retire_one();
        }
        f0 += nf;
    }
    while (gr < gf) retire_one();
    EG_TL_MAX(43);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------------
// consumer waves
// ---------------------------------------------------------------------------------------------------------------------------------------------------------------------
struct EgCtx {
    unsigned char *smem; EgShared *sh; unsigned *err; unsigned long long *gbuf; int gstride; unsigned tag0; Tables tb;
    int cu, n_cus, cw /*consumer wave 0..EG_NC-1*/, lane; bool tl_on; int op_idx;
    unsigned bar_gen;
    unsigned n_gather;     // gathered hand-offs so far
    int f0, it0;           // global index of the current op's first fill (of this workgroup) / first item
};
__device__ __forceinline__ bool eg_cbarrier(EgCtx &c, unsigned code) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (c.lane == 0) __hip_atomic_fetch_add(&c.sh->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    c.bar_gen++;
    return eg_wait_ge(&c.sh->bar, c.bar_gen * EG_NC, c.sh, c.err, code);
}
__device__ __forceinline__ unsigned long long eg_gload(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void eg_publish(unsigned long long *g, unsigned tag, float v) { __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// LDS image of the quantised activation row (one value plane: a launch's ops read either the Q8_K or the Q8_0 form)
__device__ __forceinline__ ActQ eg_image(unsigned char *img, int K) {
    ActQ L{};
    const int o_dk = K, o_bsk = K + (((K / 256 + 1) * 4 + 15) & ~15), q = (K / 32) * 4;   // Q8_K: values | dk | bsk      Q8_0: values | d0 | d1 | s1 | sum0 (the two never coexist)
    L.q8k = reinterpret_cast<int8_t *>(img); L.q80 = reinterpret_cast<int8_t *>(img);
    L.dk = reinterpret_cast<float *>(img + o_dk); L.bsk = reinterpret_cast<int16_t *>(img + o_bsk);
    L.d0 = reinterpret_cast<float *>(img + K); L.d1 = reinterpret_cast<float *>(img + K + q); L.s1 = reinterpret_cast<float *>(img + K + 2 * q); L.sum0 = reinterpret_cast<int *>(img + K + 3 * q);
    return L;
}

typedef __attribute__((address_space(1))) const float eg_gcf_t;
typedef __attribute__((address_space(1))) float eg_gf_t;
__device__ __forceinline__ float4 eg_ldg4(eg_gcf_t *p) { const v4f_t v = *reinterpret_cast<const __attribute__((address_space(1))) v4f_t *>(p); return make_float4(v.x, v.y, v.z, v.w); }
// Prepare the activation row of an op: gather (or load) the fp32 row, rms-norm it if asked, quantise it into the LDS image.  Work unit = one 256-element block (a wave's 64
// lanes x 4 consecutive elements: quant_emit4's layout).
//   RMS rows: the arithmetic is k_matvec_v2's fused prologue at 512 threads -- virtual thread t of round r owns elements (512 r + t) * 4 ..., its partial sum of squares runs
//             over its rounds in order, then wave sums, then the 8 wave sums in order.  Consumer wave cw plays virtual wave cw, consumer wave 0 virtual wave 7 as well
//             (NT = 2 RND tasks, the second half empty except in wave 0).
//   plain rows: blocks are independent (per-block quantisation), so they are dealt round-robin: wave cw takes blocks cw, cw + 7, ... (NT = ceil(8 RND / 7) tasks).
// Gathered rows: every granule request of the wave goes out before the first check; a task is re-polled until all four granules of every lane carry this hand-off's tag.
template <int RND, bool RMS>
__device__ __forceinline__ bool eg_prepare(EgCtx &c, const EngOp &op, unsigned char *img) {
    const int K = op.K, lane = c.lane;
    const bool gather = op.in_kind == ENG_IN_GATHER_RMS || op.in_kind == ENG_IN_GATHER;
    constexpr int NT = RMS ? 2 * RND : (EG_NV * RND + EG_NC - 1) / EG_NC;
    float4 xv[NT], yv[RMS ? NT : 1];
    bool in[NT];
    int idx[NT];
    eg_gcf_t *in_x = (eg_gcf_t *)op.in_x, *in_w = (eg_gcf_t *)op.in_w;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int i; bool valid;
        if (RMS) { const int v = t / RND, r = t % RND, vw = c.cw + EG_NC * v; i = (r * (EG_NV * 64) + vw * 64 + lane) * 4; valid = vw < EG_NV; }
        else { i = (c.cw + EG_NC * t) * 256 + lane * 4; valid = true; }
        in[t] = valid && i < K; idx[t] = in[t] ? i : 0;
    }
    const bool second = c.cw + EG_NC < EG_NV;            // RMS: does this wave play a second virtual wave?
    auto live = [&](int t) { return !RMS || t < RND || second; };
    if (RMS) {
#pragma unroll
        for (int t = 0; t < NT; t++) if (live(t)) yv[t] = eg_ldg4(in_w + idx[t]);
    }
    if (!gather) {
#pragma unroll
        for (int t = 0; t < NT; t++) if (live(t)) xv[t] = eg_ldg4(in_x + idx[t]);
    } else {
        const unsigned long long *g = c.gbuf + (size_t)op.in_g * c.gstride;
        const unsigned tag = c.tag0 + (unsigned)op.in_g;
        // phase 1: consumer wave 0 polls the producers' sentinels (one granule per workgroup), the other waves sleep on an LDS word; phase 2: everybody sweeps the row once
        // (every granule's tag is still checked: a sentinel ahead of its workgroup's data just costs a re-read)
        c.n_gather++;
        if (c.cw == 0) {
            const unsigned long long *sg = g + (c.gstride - 1024);
            unsigned spins = 0; unsigned long long t0 = 0;
            for (;;) {
                bool ok = true;
                for (int w0 = 0; w0 < c.n_cus; w0 += 64) { const int w = w0 + lane; const unsigned long long v = eg_gload(sg + (w < c.n_cus ? w : 0)); ok = ok && (unsigned)(v >> 32) == tag; }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(8);
                if ((++spins & 31) == 0) {
                    const unsigned long long tn = __builtin_amdgcn_s_memrealtime();
                    if (__hip_atomic_load(&c.sh->abort, EG_RLX_WG)) break;
                    if (!t0) t0 = tn; else if (tn - t0 > EG_TIMEOUT_TICKS) { eg_give_up(c.sh, c.err, 0x280u + (unsigned)op.in_g); break; }
                }
            }
            if (lane == 0) __hip_atomic_store(&c.sh->ready, c.n_gather, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (!eg_wait_ge(&c.sh->ready, c.n_gather, c.sh, c.err, 0x290u)) return false;
        constexpr int HALF = NT > 6 ? (NT + 1) / 2 : NT;     // the widest rows in two passes: 8 tasks' granules at once do not fit the register file next to everything else
        unsigned spins = 0; unsigned long long t0 = 0;
        bool gave_up = false;                            // (no return from inside the unrolled polling loops: the exits would keep every granule register alive across them)
#pragma unroll
        for (int h0 = 0; h0 < NT; h0 += HALF) {
            unsigned long long gv[HALF][4];
#pragma unroll
            for (int t = h0; t < h0 + HALF && t < NT; t++) {
                if (!live(t)) continue;
#pragma unroll
                for (int e = 0; e < 4; e++) gv[t - h0][e] = eg_gload(g + idx[t] + e);
            }
#pragma unroll
            for (int t = h0; t < h0 + HALF && t < NT; t++) {
                if (!live(t)) continue;
                unsigned long long(&q)[4] = gv[t - h0];
                while (!gave_up) {
                    const bool ok = !in[t] || ((unsigned)(q[0] >> 32) == tag && (unsigned)(q[1] >> 32) == tag && (unsigned)(q[2] >> 32) == tag && (unsigned)(q[3] >> 32) == tag);
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(2);
#pragma unroll
                    for (int e = 0; e < 4; e++) q[e] = eg_gload(g + idx[t] + e);
                    if ((++spins & 63) == 0) {
                        const unsigned long long tn = __builtin_amdgcn_s_memrealtime();
                        if (__hip_atomic_load(&c.sh->abort, EG_RLX_WG)) gave_up = true;
                        else if (!t0) t0 = tn;
                        else if (tn - t0 > EG_TIMEOUT_TICKS) { eg_give_up(c.sh, c.err, 0x200u + (unsigned)op.in_g); gave_up = true; }
                    }
                }
                xv[t] = make_float4(__uint_as_float((unsigned)q[0]), __uint_as_float((unsigned)q[1]), __uint_as_float((unsigned)q[2]), __uint_as_float((unsigned)q[3]));
            }
        }
        if (gave_up) return false;
    }
    if (RMS) {
#pragma unroll
        for (int v = 0; v < 2; v++) {
            if (v && !second) break;
            double sum = 0.0;
#pragma unroll
            for (int r = 0; r < RND; r++) {
                const float4 x4 = xv[v * RND + r];
                double q = 0.0;
                q += (double)(x4.x * x4.x); q += (double)(x4.y * x4.y); q += (double)(x4.z * x4.z); q += (double)(x4.w * x4.w);
                sum += in[v * RND + r] ? q : 0.0;
            }
            sum = wave_sum_d(sum);
            if (lane == 0) c.sh->red[c.cw + EG_NC * v] = sum;
        }
    }
#ifdef MG4_TIMELINE
    { const bool tl_on = c.tl_on && c.cw == 0; EG_TL(2 + 4 * c.op_idx); }
#endif
    if (!eg_cbarrier(c, 0x300u)) return false;          // everybody has left the previous image (and the partial sums are visible)
    float scale = 1.0f;
    if (RMS) {
        double tot = 0.0;
        for (int w = 0; w < EG_NV; w++) tot += c.sh->red[w];
        const float mean = (float)(tot / (double)K);
        scale = 1.0f / sqrtf(mean + 1e-6f);
    }
    const ActQ L = eg_image(img, K);
    const int mask = act_mask_for(op.type);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (!live(t)) continue;
        int i;
        if (RMS) { const int v = t / RND, r = t % RND; i = (r * (EG_NV * 64) + (c.cw + EG_NC * v) * 64 + lane) * 4; } else i = (c.cw + EG_NC * t) * 256 + lane * 4;
        const float4 x4 = xv[t];
        float q4[4];
        if (RMS) { const float4 y4 = yv[t]; q4[0] = (x4.x * scale) * y4.x; q4[1] = (x4.y * scale) * y4.y; q4[2] = (x4.z * scale) * y4.z; q4[3] = (x4.w * scale) * y4.w; }
        else { q4[0] = x4.x; q4[1] = x4.y; q4[2] = x4.z; q4[3] = x4.w; }
        if (!in[t]) { q4[0] = 0.0f; q4[1] = 0.0f; q4[2] = 0.0f; q4[3] = 0.0f; }
        if (RMS || __any(in[t])) quant_emit4(q4, in[t], i, 0, K, L, mask);       // (a plain block past the end of the row has nothing to emit; the wave decides together)
    }
    return eg_cbarrier(c, 0x301u);                      // the image is complete
}

// weight units out of a ring slot (plane p of matrix m starts pl[m * PM + p].loff bytes into the slot; rows of a fill are consecutive)
template <int T> struct EgLd;
template <> struct EgLd<GT_Q4_0> { static constexpr int PM = 2;
    static __device__ __forceinline__ void ld(const unsigned char *s, const int *lo, const int *rb, int r, int u, Tr<GT_Q4_0>::WU &w) {
        w.q = *reinterpret_cast<const int4 *>(s + lo[0] + r * rb[0] + u * 16); w.dh = *reinterpret_cast<const unsigned short *>(s + lo[1] + r * rb[1] + u * 2); } };
template <> struct EgLd<GT_Q4_K> { static constexpr int PM = 2;
    static __device__ __forceinline__ void ld(const unsigned char *s, const int *lo, const int *rb, int r, int u, Tr<GT_Q4_K>::WU &w) {
        w.q = *reinterpret_cast<const int4 *>(s + lo[0] + r * rb[0] + u * 16); w.h = *reinterpret_cast<const int4 *>(s + lo[1] + r * rb[1] + (u >> 3) * 16); } };
template <> struct EgLd<GT_Q5_K> { static constexpr int PM = 3;
    static __device__ __forceinline__ void ld(const unsigned char *s, const int *lo, const int *rb, int r, int u, Tr<GT_Q5_K>::WU &w) {
        w.q = *reinterpret_cast<const int4 *>(s + lo[0] + r * rb[0] + u * 16); w.P = *reinterpret_cast<const unsigned *>(s + lo[1] + r * rb[1] + u * 4);
        w.h = *reinterpret_cast<const int4 *>(s + lo[2] + r * rb[2] + (u >> 3) * 16); } };
template <> struct EgLd<GT_Q6_K> { static constexpr int PM = 4;
    static __device__ __forceinline__ void ld(const unsigned char *s, const int *lo, const int *rb, int r, int u, Tr<GT_Q6_K>::WU &w) {
        w.q = *reinterpret_cast<const int4 *>(s + lo[0] + r * rb[0] + u * 16); const uint2 p = *reinterpret_cast<const uint2 *>(s + lo[1] + r * rb[1] + u * 8); w.Plo = p.x; w.Phi = p.y;
        w.sc = *reinterpret_cast<const unsigned short *>(s + lo[2] + r * rb[2] + u * 2); w.dh = *reinterpret_cast<const unsigned short *>(s + lo[3] + r * rb[3] + (u >> 3) * 2); } };

template <int T, int NU>
__device__ __forceinline__ bool eg_run_op(EgCtx &c, const EngOp &op, unsigned char *img) {
    using X = Tr<T>;
    constexpr int PM = EgLd<T>::PM;
    const int lane = c.lane, K = op.K, U = K / X::EPU;
    int uc[NU]; bool ok[NU];
#pragma unroll
    for (int i = 0; i < NU; i++) { const int u = lane + 64 * i; ok[i] = u < U; uc[i] = ok[i] ? u : 0; }
    typename X::AU a[NU];
    {
        const ActQ L = eg_image(img, K);
#pragma unroll
        for (int i = 0; i < NU; i++) X::loada(L, 0, K, uc[i], a[i]);
    }
    const int fsh = eg_log2(op.unit / op.G), nf = eg_share(op, c.cu, c.n_cus) << fsh;
    // the op's description in registers (scalar): the loop below stores to global memory, after which the compiler would re-read every field
    const int G = op.G, pair = op.pair, res_kind = op.res_kind, save_res = op.save_res, out_kind = op.out_kind, nm = pair ? 2 : 1;
    static_assert(EG_NC >= 4, "a wave holds at most one row of a fill");
    eg_gcf_t *const resp = (eg_gcf_t *)op.res; eg_gf_t *const yp = (eg_gf_t *)op.y;
    int lo[2 * PM], rb[2 * PM];
#pragma unroll
    for (int p = 0; p < 2 * PM; p++) { const int ix = op.pidx[(p / PM) * 4 + p % PM]; lo[p] = op.pl[ix].loff; rb[p] = op.pl[ix].rb; }

    const unsigned tag = c.tag0 + (unsigned)op.out_g;
    unsigned long long *gout = c.gbuf + (size_t)op.out_g * c.gstride;
    // A fill's G rows (row pairs) are its ITEMS; item number `it` (counted over the whole launch) belongs to consumer wave it % EG_NC, so the rows of one fill are multiplied by
    // G different waves at once and the slot is free again after ONE row's time (a whole fill per wave kept a slot for 2.2 us: 8 slots / (landing + 2.2 us) capped the stream
    // at 0.8 us per fill -- timelines in profiles/r04_engine_*).  G <= 4 < EG_NC: a wave has at most one item per fill.
    for (int i = 0; i < nf; i++) {
        const int q = ((c.cw - (c.it0 + i * G)) % EG_NC + EG_NC) % EG_NC;            // my item of this fill, if < G
        if (q >= G) continue;
        const int f = c.f0 + i, slot = f % EG_NS, row = eg_gfill(i, fsh, c.cu, c.n_cus) * G + q, loc = i * G + q;   // loc: the row among this workgroup's own rows
        float res = 0.0f;
        if (res_kind == ENG_RES_GLOBAL) res = resp[row];
        if (!eg_wait_ge(&c.sh->filled[slot], (unsigned)(f + 1), c.sh, c.err, 0x400u)) return false;
#ifdef MG4_TIMELINE
        if (c.tl_on && blockIdx.x == 7 && lane == 0 && f < 128 && q == 0) g_efl[256 + f] = __builtin_amdgcn_s_memrealtime();
#endif
        const unsigned char *s = c.smem + (size_t)slot * EG_SLOT;
        float out[2] = {0.0f, 0.0f};
#ifndef EG_SKIP_COMPUTE                                  // (diagnostic arm: rows are not multiplied at all -- what the stream alone does with consumers that only hand slots back)
#pragma unroll
        for (int m = 0; m < 2; m++) {
            if (m < nm) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < NU; j++) { typename X::WU w; EgLd<T>::ld(s, lo + m * PM, rb + m * PM, q, uc[j], w); float cc = acc; X::dot(w, a[j], cc); acc = ok[j] ? cc : acc;
                    if (NU >= 6 && (j & 1)) __builtin_amdgcn_sched_barrier(0); }     // widest rows: at most two units' weights in registers beside the 77 of activations
                out[m] = wave_sum(acc);
            }
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                           // every read of the slot has returned
        if (lane == 0) {                                                                // the last of the fill's G items hands the slot back
            const unsigned before = __hip_atomic_fetch_add(&c.sh->part[slot], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (before + 1u == (unsigned)G) {
                __hip_atomic_store(&c.sh->part[slot], 0u, EG_RLX_WG);
                __hip_atomic_store(&c.sh->done[slot], (unsigned)(f + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef MG4_TIMELINE
                if (c.tl_on && blockIdx.x == 7 && f < 128) g_efl[384 + f] = __builtin_amdgcn_s_memrealtime();
#endif
            }
        }
        if (pair) {                                                                     // h[g] = silu_table(w1[g] . x) * (w3[g] . x), k_matvec_v2's EPI_SILU_PAIR
            const unsigned short t = reinterpret_cast<const unsigned short *>(c.tb.silu)[f2h_bits(out[0])];
            if (lane == 0) {
                const float h = h2f_bits(t) * out[1];
                if (out_kind & ENG_OUT_GRANULE) eg_publish(gout + row, tag, h);
                if (out_kind & ENG_OUT_PLAIN) yp[row] = h;
            }
        } else if (lane == 0) {
            float v = out[0];
            if (res_kind == ENG_RES_GLOBAL) v = v + res;
            else if (res_kind == ENG_RES_SAVED) v = v + c.sh->xres[loc];
            if (save_res) c.sh->xres[loc] = v;
            if (out_kind & ENG_OUT_GRANULE) eg_publish(gout + row, tag, v);
            if (out_kind & ENG_OUT_PLAIN) yp[row] = v;
        }
    }
    c.f0 += nf; c.it0 += nf * G;
    // This wave's rows of the op are published.  The LAST consumer wave of the workgroup to get here stores the workgroup's sentinel granule of the hand-off: consumers poll
    // the n_cus sentinels (2 KiB) instead of the whole row (40 ... 110 KiB per poll and workgroup: the early finishers' polling slowed the stragglers' weight streams).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                               // (waits for this wave's granule stores)
    if (lane == 0) {
        const unsigned before = __hip_atomic_fetch_add(&c.sh->fin, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((out_kind & ENG_OUT_GRANULE) && (before + 1u) % EG_NC == 0) eg_publish(gout + (c.gstride - 1024) + c.cu, tag, 1.0f);
    }
    return true;
}

template <int T>
__device__ __forceinline__ bool eg_run_type(EgCtx &c, const EngOp &op, unsigned char *img) {
    switch (op.nu) {
    case 1: return eg_run_op<T, 1>(c, op, img);
    case 2: return eg_run_op<T, 2>(c, op, img);
    case 3: return eg_run_op<T, 3>(c, op, img);
    case 6: return eg_run_op<T, 6>(c, op, img);
    case 7: return eg_run_op<T, 7>(c, op, img);
    default: return false;
    }
}

__global__ __launch_bounds__(EG_THREADS) void k_decode_engine(const EngOp *__restrict__ ops, const int n_ops, unsigned long long *gbuf, const int gstride, const int *__restrict__ epoch,
                                                               const int layer, const Tables tb, unsigned *err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_eg[];
    unsigned char *img = smem_eg + (size_t)EG_NS * EG_SLOT;
    EgShared *sh = reinterpret_cast<EgShared *>(img + EG_IMG);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x < EG_NS) { sh->filled[threadIdx.x] = 0; sh->done[threadIdx.x] = 0; sh->part[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { sh->bar = 0; sh->abort = 0; sh->fin = 0; sh->ready = 0; }
    __syncthreads();                                   // the only s_barrier of the kernel
    const int cu = blockIdx.x, n_cus = gridDim.x;
#ifdef MG4_TIMELINE
    const bool tl_on = layer == g_etl_layer;
    EG_TL_ZERO();
    __syncthreads();
#else
    const bool tl_on = false;
#endif
    if (wave < EG_NL) { eg_loader(ops, n_ops, smem_eg, sh, err, cu, n_cus, wave, tl_on); return; }
    EgCtx c;
    c.smem = smem_eg; c.sh = sh; c.err = err; c.gbuf = gbuf; c.gstride = gstride; c.tb = tb; c.cu = cu; c.n_cus = n_cus; c.cw = wave - EG_NL; c.lane = lane; c.tl_on = tl_on;
    c.bar_gen = 0; c.f0 = 0; c.it0 = 0; c.n_gather = 0;
    c.tag0 = ((((unsigned)*epoch) * 64u + (unsigned)layer) << 2) + 1u;
    if (c.cw == 0) EG_TL(0);
    for (int o = 0; o < n_ops; o++) {
        const EngOp &op = ops[o];
        bool okk = true;
        c.op_idx = o;
        if (op.in_kind != ENG_IN_KEEP) {
            if (c.cw == 0) EG_TL(1 + 4 * o);
            const bool rms = op.in_kind == ENG_IN_GATHER_RMS || op.in_kind == ENG_IN_PLAIN_RMS;
            switch ((op.K + 2047) / 2048 + (rms ? 8 : 0)) {
            case 1: okk = eg_prepare<1, false>(c, op, img); break;
            case 2: okk = eg_prepare<2, false>(c, op, img); break;
            case 3: okk = eg_prepare<3, false>(c, op, img); break;
            case 6: okk = eg_prepare<6, false>(c, op, img); break;
            case 7: okk = eg_prepare<7, false>(c, op, img); break;
            case 9: okk = eg_prepare<1, true>(c, op, img); break;
            case 10: okk = eg_prepare<2, true>(c, op, img); break;
            case 11: okk = eg_prepare<3, true>(c, op, img); break;
            default: okk = false;                       // (the host never queues a normed row wider than 3 rounds: Engine::build_engine_ops)
            }
            if (c.cw == 0) EG_TL(3 + 4 * o);
        }
        if (okk) switch (op.type) {
        case GT_Q4_0: okk = eg_run_type<GT_Q4_0>(c, op, img); break;
        case GT_Q4_K: okk = eg_run_type<GT_Q4_K>(c, op, img); break;
        case GT_Q5_K: okk = eg_run_type<GT_Q5_K>(c, op, img); break;
        case GT_Q6_K: okk = eg_run_type<GT_Q6_K>(c, op, img); break;
        default: okk = false;
        }
        EG_TL_MAX(4 + 4 * o);
        if (!okk) {                                    // gave up (or an op the host should never have queued): let the loader drain instead of waiting for consumers
            eg_give_up(sh, err, 0x500u + (unsigned)o);
            for (int sl = 0; sl < EG_NS; sl++) __hip_atomic_store(&sh->done[sl], 0x3fffffffu, EG_RLX_WG);
            return;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------------------------------------------
static int eng_planes_of(const QWeight &W, EngPlane *pl) {   // planes of ONE matrix, loff left for the caller
    const int K = W.cols;
    auto set = [&](EngPlane &p, const uint8_t *base, int rb) { p = EngPlane{}; p.base = base; p.rb = rb; };
    switch (W.type) {
    case GT_Q4_0: set(pl[0], W.qs, K / 2); set(pl[1], W.sc, K / 16); return 2;
    case GT_Q4_K: set(pl[0], W.qs, K / 2); set(pl[1], W.sc, K / 16); return 2;
    case GT_Q5_K: set(pl[0], W.qs, K / 2); set(pl[1], W.qh, K / 8); set(pl[2], W.sc, K / 16); return 3;
    case GT_Q6_K: set(pl[0], W.qs, K / 2); set(pl[1], W.qh, K / 4); set(pl[2], W.sc, K / 16); set(pl[3], W.d, K / 128); return 4;
    default: return 0;
    }
}
// Describes one mat-vec of a chain (W1 != null: the w1 | w3 pair).  false: outside the engine's range (the caller keeps the launch-per-op path).
bool eng_make_op(EngOp &op, const QWeight &W0, const QWeight *W1, bool matched_partition) {
    op = EngOp{};
    const int K = W0.cols;
    if (K % 256 || K > 7 * 2048 || W0.rows <= 0) return false;
    if (W1 && (W1->type != W0.type || W1->rows != W0.rows || W1->cols != W0.cols)) return false;
    const int nu = (K / 32 + 63) / 64;
    if (nu != 1 && nu != 2 && nu != 3 && nu != 6 && nu != 7) return false;
    if ((K + 2047) / 2048 != nu) return false;           // (the prologue's rounds and the units per lane coincide for these K)
    EngPlane one[4];
    const int pm = eng_planes_of(W0, one);
    if (!pm) return false;
    const int nm = W1 ? 2 : 1;
    int G = 0;
    for (int g : {4, 2, 1}) {                              // the largest row group whose planes fit one ring slot
        if (W1 && g == 4) continue;
        if (W0.rows % g) continue;
        size_t bytes = 0; bool okp = true;
        for (int p = 0; p < pm; p++) { const int b = g * one[p].rb; if (b % 4) okp = false; bytes += (size_t)((b + 15) & ~15); }
        if (okp && bytes * nm <= (size_t)ENG_SLOT_BYTES) { G = g; break; }
    }
    if (!G) return false;
    op.type = W0.type; op.K = K; op.rows = W0.rows; op.G = G; op.pair = W1 ? 1 : 0; op.nu = nu;
    op.unit = matched_partition ? 4 : G;
    if (op.rows % op.unit) return false;
    int off = 0;
    for (int m = 0; m < nm; m++) {
        EngPlane pl[4];
        eng_planes_of(m ? *W1 : W0, pl);
        for (int p = 0; p < pm; p++) {
            op.pidx[m * 4 + p] = op.n_planes;
            EngPlane &d = op.pl[op.n_planes++];
            d = pl[p]; d.loff = off;
            off += (G * d.rb + 15) & ~15;
        }
    }
    op.fill_bytes = off; op.ipf = (off + 1023) / 1024;
    return op.ipf <= 21;                                   // two fills in flight behind the one being published: vmcnt counts to 63
}
// planes -> fill-major image: workgroup = one fill, 4-byte words (a plane's G-row piece is a whole number of words, its rows are 4-byte aligned in the arena)
__global__ __launch_bounds__(256) void k_eng_repack(const EngOp op, uint8_t *__restrict__ image) {
    const size_t fill = blockIdx.x;
    unsigned *dst = reinterpret_cast<unsigned *>(image + fill * (size_t)op.fill_bytes);
    for (int p = 0; p < op.n_planes; p++) {
        const int words = op.G * op.pl[p].rb / 4;
        const unsigned *src = reinterpret_cast<const unsigned *>(op.pl[p].base + fill * (size_t)(op.G * op.pl[p].rb));
        unsigned *d = dst + op.pl[p].loff / 4;
        for (int i = threadIdx.x; i < words; i += 256) d[i] = src[i];
    }
}
void launch_eng_repack(const EngOp &op, uint8_t *image, hipStream_t s) { k_eng_repack<<<dim3((unsigned)(op.rows / op.G)), dim3(256), 0, s>>>(op, image); }
size_t decode_engine_lds_bytes() { return EG_LDS_BYTES; }
bool launch_decode_engine(const EngOp *d_ops, int n_ops, unsigned long long *gbuf, int gstride, const int *d_epoch, int layer, const Tables &tb, unsigned *d_err, int n_cus, hipStream_t s) {
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode_engine), hipFuncAttributeMaxDynamicSharedMemorySize, (int)EG_LDS_BYTES) != hipSuccess) { (void)hipGetLastError(); return false; } attr = true; }
    hipEvent_t e0, e1;
    if (kernel_probe_begin("k_decode_engine", &e0, &e1)) {
        Tables tbc = tb;
        void *args[] = {&d_ops, &n_ops, &gbuf, &gstride, &d_epoch, &layer, &tbc, &d_err};
        HIP_IGNORE(hipExtLaunchKernel(reinterpret_cast<const void *>(&k_decode_engine), dim3((unsigned)n_cus), dim3(EG_THREADS), args, EG_LDS_BYTES, s, e0, e1, 0));
        return true;
    }
    k_decode_engine<<<dim3((unsigned)n_cus), dim3(EG_THREADS), EG_LDS_BYTES, s>>>(d_ops, n_ops, gbuf, gstride, d_epoch, layer, tb, d_err);
    return true;
}
int read_engine_timeline(unsigned long long *out, int max_workgroups, int layer) {
#ifdef MG4_TIMELINE
    if (layer >= 0) { if (hipMemcpyToSymbol(HIP_SYMBOL(g_etl_layer), &layer, sizeof(int)) != hipSuccess) return -1; if (!out) return 0; }
    const int n = std::min(max_workgroups, 512);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_etl), (size_t)n * 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (max_workgroups >= 520 && hipMemcpyFromSymbol(out + 512 * 64, HIP_SYMBOL(g_efl), sizeof(g_efl)) != hipSuccess) return -1;   // (callers that pass room for it: per-fill stamps behind the table)
    return n;
#else
    (void)out; (void)max_workgroups; (void)layer; return 0;
#endif
}

}  // namespace mg4
