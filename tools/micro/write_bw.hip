// Write bandwidth of the join's copy-out shapes (round 6): what do 4-byte-per-lane non-temporal stores reach against 16-byte-per-lane
// ones, streaming and in the join's form (each wavefront writes ranges of ~2 KB to TWO columns at positions handed out by a cursor)?
// build: hipcc --offload-arch=gfx950 -O3 -o write_bw write_bw.hip ; prints GB/s per kernel (best of 5)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int W>   // W = dwords per lane (1 or 4); grid-stride streaming stores, 1024-thread workgroups, one per CU x 4
__global__ __launch_bounds__(1024) void k_stream(int32_t* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 1024 * W;
    for (int64_t i = ((int64_t)blockIdx.x * 1024 + threadIdx.x) * W; i < n; i += stride) {
        if (W == 1) __builtin_nontemporal_store((int32_t)i, out + i);
        else { v4i v = {(int)i, 1, 2, 3}; __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(out + i)); }
    }
}
// the join's form: every wavefront takes ranges of `len` elements and writes
// them to two columns, W dwords per lane per store
template <int W>
__global__ __launch_bounds__(1024) void k_ranges2(int32_t* __restrict__ a, int32_t* __restrict__ b, int64_t n, int len, unsigned long long* cursor) {
    const int lane = threadIdx.x & 63;
    // (ranges dealt out round robin over the wavefronts in flight -- a cursor's same-address atomics, 1.5 ns each, would be what is measured;
    // neighbouring ranges are written by different CUs at about the same time, as the join's tiles are)
    const int64_t nw = (int64_t)gridDim.x * 16;
    for (int64_t r = (int64_t)(threadIdx.x >> 6) * gridDim.x + blockIdx.x; ; r += nw) {
        const int64_t base = r * len;
        if (base >= n) return;
        const int m = (int64_t)base + len <= n ? len : (int)(n - (int64_t)base);
        for (int i = lane * W; i < m; i += 64 * W) {
            if (W == 1) { __builtin_nontemporal_store(i, a + base + i); __builtin_nontemporal_store(i, b + base + i); }
            else if (i + 4 <= m) {
                v4i v = {i, 1, 2, 3};
                __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(a + base + i)); __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(b + base + i));
            } else for (int u = i; u < m; ++u) { a[base + u] = u; b[base + u] = u; }
        }
    }
}

int main() {
    const int64_t n = (int64_t)200 << 20;                 // 200 M elements per column = 0.8 GB per column
    int32_t *a, *b; unsigned long long* cur;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&cur, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, double bytes, auto&& launch) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            CK(hipMemset(cur, 0, 8));
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
        }
        std::printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
    };
    run("k_stream<1> one column", n * 4.0, [&] { hipLaunchKernelGGL((k_stream<1>), dim3(1024), dim3(1024), 0, 0, a, n); });
    run("k_stream<4> one column", n * 4.0, [&] { hipLaunchKernelGGL((k_stream<4>), dim3(1024), dim3(1024), 0, 0, a, n); });
    for (int len : {508, 512, 2032, 2048, 8128}) {
        char nm[96];
        std::snprintf(nm, 96, "k_ranges2<1> two columns, ranges of %d", len);
        run(nm, n * 8.0, [&] { hipLaunchKernelGGL((k_ranges2<1>), dim3(256), dim3(1024), 0, 0, a, b, n, len, cur); });
        std::snprintf(nm, 96, "k_ranges2<4> two columns, ranges of %d", len);
        run(nm, n * 8.0, [&] { hipLaunchKernelGGL((k_ranges2<4>), dim3(256), dim3(1024), 0, 0, a, b, n, len, cur); });
    }
    return 0;
}
