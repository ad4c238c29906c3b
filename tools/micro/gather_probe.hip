// gather_probe.hip -- micro-benchmark behind DESIGN section 8: what does ONE random record gather per probe cost on MI355X, by table
// size (L2-resident ... beyond the Infinity Cache) and record width?  Probe columns are streamed and a 12-byte result is written per
// probe, as a per-probe kernel of this library does.  build: hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// WORDS = record width in 16-byte words; DEP = a second, dependent gather (index taken from the first record)
template <int WORDS, bool DEP>
__global__ __launch_bounds__(256) void k_gather(const int4* __restrict__ tab, uint32_t nrec, const int32_t* __restrict__ ps, const int32_t* __restrict__ pe,
                                                int64_t n, int32_t* __restrict__ out_a, long long* __restrict__ out_b) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t s = __builtin_nontemporal_load(ps + i), e = __builtin_nontemporal_load(pe + i);
    uint32_t idx = (uint32_t)(((unsigned long long)mix((uint32_t)s * 2654435761u + (uint32_t)e) * nrec) >> 32);
    int4 acc = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int w = 0; w < WORDS; ++w) { const int4 v = tab[(size_t)idx * WORDS + w]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    if (DEP) {
        idx = (uint32_t)(((unsigned long long)mix((uint32_t)acc.x + idx) * nrec) >> 32);
#pragma unroll
        for (int w = 0; w < WORDS; ++w) { const int4 v = tab[(size_t)idx * WORDS + w]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    }
    __builtin_nontemporal_store(acc.x ^ acc.z, out_a + i);
    __builtin_nontemporal_store((long long)acc.y + acc.w, out_b + i);
}

template <int WORDS, bool DEP>
static float run(const int4* tab, uint32_t nrec, const int32_t* ps, const int32_t* pe, int64_t n, int32_t* oa, long long* ob) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)((n + 255) / 256);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_gather<WORDS, DEP>), dim3(grid), dim3(256), 0, 0, tab, nrec, ps, pe, n, oa, ob);
    CK(hipEventRecord(a));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_gather<WORDS, DEP>), dim3(grid), dim3(256), 0, 0, tab, nrec, ps, pe, n, oa, ob);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5;
}

__global__ void k_fill(int32_t* p, int64_t n, uint32_t salt) { const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = (int32_t)(mix((uint32_t)i + salt) >> 4); }

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 50000000;
    int32_t *ps, *pe, *oa; long long* ob; int4* tab;
    const size_t tab_max = (size_t)1 << 30;
    CK(hipMalloc(&ps, n * 4)); CK(hipMalloc(&pe, n * 4)); CK(hipMalloc(&oa, n * 4)); CK(hipMalloc(&ob, n * 8)); CK(hipMalloc(&tab, tab_max));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ps, n, 1u);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, pe, n, 7u);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((tab_max / 4 + 255) / 256)), dim3(256), 0, 0, (int32_t*)tab, (int64_t)(tab_max / 4), 3u);
    CK(hipDeviceSynchronize());
    std::printf("# %lld probes; one random record gather per probe (+ 8 B read, 12 B written per probe); ms per launch | ps per probe\n", (long long)n);
    std::printf("%10s %8s %12s %12s %12s %14s\n", "table", "record", "1 gather", "ps/probe", "2 dependent", "ps/probe");
    const size_t sizes[] = {(size_t)2 << 20, (size_t)3 << 20, (size_t)4 << 20, (size_t)6 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20, (size_t)512 << 20, (size_t)1 << 30};
    for (size_t sz : sizes) {
        for (int words : {1, 2, 4, 8}) {
            const uint32_t nrec = (uint32_t)(sz / (16 * (size_t)words));
            float m1 = 0, m2 = 0;
            switch (words) {
                case 1: m1 = run<1, false>(tab, nrec, ps, pe, n, oa, ob); m2 = run<1, true>(tab, nrec, ps, pe, n, oa, ob); break;
                case 2: m1 = run<2, false>(tab, nrec, ps, pe, n, oa, ob); m2 = run<2, true>(tab, nrec, ps, pe, n, oa, ob); break;
                case 4: m1 = run<4, false>(tab, nrec, ps, pe, n, oa, ob); m2 = run<4, true>(tab, nrec, ps, pe, n, oa, ob); break;
                default: m1 = run<8, false>(tab, nrec, ps, pe, n, oa, ob); m2 = run<8, true>(tab, nrec, ps, pe, n, oa, ob); break;
            }
            std::printf("%7zu MB %6d B %9.3f ms %9.1f ps %9.3f ms %11.1f ps\n", sz >> 20, words * 16, m1, m1 * 1e9 / (double)n, m2, m2 * 1e9 / (double)n);
        }
    }
    return 0;
}
