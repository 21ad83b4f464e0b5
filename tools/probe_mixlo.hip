// Probe (round 3): does `__float2half_rn(a * b)` equal the two-step IEEE result (fp32 product, then fp32 -> fp16)?  hipcc folds the multiply into v_fma_mixlo_f16, which rounds
// the exact product once: 1045 of 2^24 random operand pairs differ on gfx950 (profiles/r03_probe_mixlo.log); with the product made opaque (asm "+v") all equal the host.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_mixlo.hip -o /tmp/probe_mixlo
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(const float *a, const float *b, unsigned short *fused, unsigned short *split, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    fused[i] = __half_as_ushort(__float2half_rn(a[i] * b[i]));
    float p = a[i] * b[i]; asm volatile("" : "+v"(p));
    split[i] = __half_as_ushort(__float2half_rn(p));
}
int main() {
    const int n = 1 << 24; float *ha = (float *)malloc(n * 4), *hb = (float *)malloc(n * 4);
    srand(1); for (int i = 0; i < n; i++) { ha[i] = (float)(rand() % 2048 + 1) / 2048.0f; hb[i] = 0.02f + (float)rand() / RAND_MAX; }
    float *da, *db; unsigned short *df, *ds; hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&df, n * 2); hipMalloc(&ds, n * 2);
    hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, df, ds, n);
    unsigned short *hf = (unsigned short *)malloc(n * 2), *hs = (unsigned short *)malloc(n * 2);
    hipMemcpy(hf, df, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, n * 2, hipMemcpyDeviceToHost);
    int diff = 0, host_bad = 0; for (int i = 0; i < n; i++) { if (hf[i] != hs[i]) { if (diff < 5) printf("a=%.9g b=%.9g fused=0x%04x split=0x%04x\n", ha[i], hb[i], hf[i], hs[i]); diff++; }
        volatile float p = ha[i] * hb[i]; __half hh = __float2half_rn(p); if (__half_as_ushort(hh) != hs[i]) host_bad++; }
    printf("differences fused vs split: %d of %d; split vs host: %d\n", diff, n, host_bad);
}
