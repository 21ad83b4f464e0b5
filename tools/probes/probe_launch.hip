// Probe (round 3): what does one dependent kernel boundary cost on this box, as wall time per kernel of a long chain replayed from a hipGraph?
// Variants: trivial 1-thread kernel (read-modify-write of one word, like k_advance), 256 workgroups x 256 threads trivial, the same with a 600-byte kernarg block,
// and a two-round-trip kernel (dependent loads).  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_launch.hip -o tools/probes/probe_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Big { int pad[150]; };
__global__ void k_one(int *p) { *p += 1; }
__global__ void k_wide(int *p) { if (blockIdx.x == 0 && threadIdx.x == 0) *p += 1; }
__global__ void k_wide_big(int *p, Big b) { if (blockIdx.x == 0 && threadIdx.x == 0) *p += b.pad[3]; }
__global__ void k_chase(int *p, const int *idx) { int i = idx[0]; int j = idx[i]; if (threadIdx.x == 0 && blockIdx.x == 0) *p += j; }
__global__ void k_empty() {}
template <typename F> static int run(const char *name, F launch, hipStream_t s, int n) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; i++) launch();
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 20;
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(a, s));
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    auto t1 = std::chrono::steady_clock::now();
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s graph of %d: %.3f us per kernel (events), %.3f us (host wall)\n", name, n, ms * 1e3 / (reps * n), std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * n));
    // eager
    for (int i = 0; i < n; i++) launch();
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; r++) for (int i = 0; i < n; i++) launch();
    CK(hipStreamSynchronize(s));
    t1 = std::chrono::steady_clock::now();
    printf("%-28s eager        : %.3f us per kernel (host wall)\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / (5 * n));
    return 0;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *p, *idx; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096)); CK(hipMalloc(&idx, 4096)); int h[2] = {1, 1}; CK(hipMemcpy(idx, h, 8, hipMemcpyHostToDevice));
    Big big{}; big.pad[3] = 1;
    const int n = 240;
    if (run("empty <<<1,64>>>", [&] { k_empty<<<1, 64, 0, s>>>(); }, s, n)) return 1;
    if (run("rmw one thread", [&] { k_one<<<1, 1, 0, s>>>(p); }, s, n)) return 1;
    if (run("rmw 256 x 256", [&] { k_wide<<<256, 256, 0, s>>>(p); }, s, n)) return 1;
    if (run("rmw 256 x 256, 600 B kernarg", [&] { k_wide_big<<<256, 256, 0, s>>>(p, big); }, s, n)) return 1;
    if (run("rmw 256 x 512", [&] { k_wide<<<256, 512, 0, s>>>(p); }, s, n)) return 1;
    if (run("two dependent loads", [&] { k_chase<<<256, 256, 0, s>>>(p, idx); }, s, n)) return 1;
    return 0;
}
