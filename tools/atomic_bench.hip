// Micro-benchmark behind DESIGN.md section 5 ("one cursor, 12 ns per reservation"): how fast can workgroups
// reserve ranges from ONE 64-bit device counter, and how much faster are distinct addresses.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o tools/atomic_bench && ./tools/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_atomic(unsigned long long* ctr, int per_wg, int spin, unsigned long long* sink) {
    __shared__ unsigned long long b;
    unsigned long long acc = 0;
    for (int r = 0; r < per_wg; ++r) {
        if (threadIdx.x == 0) b = atomicAdd(ctr, 1ull + (unsigned long long)(blockIdx.x & 7));
        __syncthreads();
        acc += b;
        for (int i = 0; i < spin; ++i) acc = acc * 6364136223846793005ull + 1442695040888963407ull;
        __syncthreads();
    }
    if (acc == 12345) sink[0] = acc;
}
__global__ void k_atomic_multi(unsigned long long* ctr, int stride, unsigned long long* sink) {
    __shared__ unsigned long long b;
    if (threadIdx.x == 0) b = atomicAdd(ctr + (blockIdx.x % 64) * stride, 1ull);
    __syncthreads();
    if (b == 12345678901ull) sink[0] = b;
}
int main() {
    unsigned long long *ctr, *sink; hipMalloc(&ctr, 1 << 20); hipMalloc(&sink, 64); hipMemset(ctr, 0, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wgs : {97656, 195312, 390624}) for (int spin : {0, 2000}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a); hipLaunchKernelGGL(k_atomic, dim3(wgs), dim3(256), 0, 0, ctr, 1, spin, sink); hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("same-address: wgs %d spin %d  %.3f ms  %.1f ns/atomic\n", wgs, spin, ms, ms * 1e6 / wgs);
    }
    for (int stride : {1, 8, 16, 512}) {
        for (int rep = 0; rep < 2; ++rep) { hipEventRecord(a); hipLaunchKernelGGL(k_atomic_multi, dim3(195312), dim3(256), 0, 0, ctr, stride, sink); hipEventRecord(b); hipEventSynchronize(b); }
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("64 addresses stride %d words: 195312 wgs %.3f ms %.1f ns/atomic\n", stride, ms, ms * 1e6 / 195312);
    }
    // empty-ish kernel baseline
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(a); hipLaunchKernelGGL(k_atomic, dim3(195312), dim3(256), 0, 0, ctr, 0, 0, sink); hipEventRecord(b); hipEventSynchronize(b); }
    float ms; hipEventElapsedTime(&ms, a, b); printf("no atomic: 195312 wgs %.3f ms\n", ms);
    return 0;
}
