// write_calib.hip -- calibration of the HBM WRITE counters on this stack (guide, section HBM: "calibrate on a known
// byte count in your own access pattern").  Every kernel below writes a KNOWN number of bytes in one of the store patterns this library
// produces; run under `rocprofv3 --kernel-trace --pmc WRITE_SIZE` and `--pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum` (tools/write_calib.sh)
// and compare per kernel.  The kernel NAME carries the pattern, stdout the known bytes.
//   k_stream<NT>         full 64-byte lines, 4 bytes per lane, consecutive (the join's pair columns, the per-probe result columns)
//   k_runs<LEN8>         runs of LEN8 x 8 bytes at a random 8-byte offset inside their own 256-byte slot, consecutive lanes of a wavefront
//                        write consecutive records of consecutive runs (the scatter's bucket runs: 62 bytes on average in round 5)
//   k_random8            one 8-byte store per lane at a random, line-distinct address (the round-5 per-probe scatter)
//   k_ranges<ALIGNED>    a wavefront copies ranges of ~480 elements x 4 bytes that start at a random element; ALIGNED: the loop starts at
//                        lane - (misalignment), so that every store instruction after the first covers whole lines (the join's copy-out)
// build: hipcc --offload-arch=gfx950 -O3 -o write_calib write_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <bool NT>
__global__ __launch_bounds__(256) void k_stream(int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (NT) __builtin_nontemporal_store((int32_t)i, out + i); else out[i] = (int32_t)i;
}

// run r lives in slot r (256 bytes); it starts at byte 8 * (mix(r) % 24) and holds LEN8 records of 8 bytes: any alignment against the
// 64-byte lines, never leaving the slot (8 * 23 + 8 * 16 = 312 > 256 -> offsets are taken mod (32 - LEN8 + 1))
template <int LEN8>
__global__ __launch_bounds__(256) void k_runs(int2* __restrict__ out, int64_t n_rec) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rec) return;
    const int64_t r = i / LEN8;
    const int k = (int)(i - r * LEN8);
    const uint32_t off = mix((uint32_t)r) % (uint32_t)(32 - LEN8 + 1);
    out[r * 32 + off + k] = make_int2((int)i, k);
}

__global__ __launch_bounds__(256) void k_random8(int2* __restrict__ out, int64_t n, uint32_t lines_mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // a bijection of the low bits keeps the stores line-distinct: line = bit-mixed i, word inside the line = mix(i) & 7
    const uint32_t line = (uint32_t)(((uint64_t)(uint32_t)i * 2654435761ull) & lines_mask) ;
    out[(size_t)line * 8 + (mix((uint32_t)i) & 7u)] = make_int2((int)i, 1);
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_ranges(int32_t* __restrict__ out, int64_t n_ranges, int len_base) {
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;          // one range per wavefront
    if (w >= n_ranges) return;
    const int len = len_base + (int)(mix((uint32_t)w) & 63u);                  // 480 .. 543 elements
    int32_t* op = out + w * 640 + (mix((uint32_t)w * 7u) & 15u);              // random element inside the first line of a 2560-byte slot
    const int ca = ALIGNED ? (int)((reinterpret_cast<uintptr_t>(op) >> 2) & 15u) : 0;
    for (int i = lane - ca; i < len; i += 64)
        if ((unsigned)i < (unsigned)len) __builtin_nontemporal_store(i, op + i);
}

int main(int argc, char** argv) {
    const int64_t n = (int64_t)256 << 20;                                      // 1 GiB of int32
    int32_t* buf;
    CK(hipMalloc(&buf, (size_t)4 << 30));
    CK(hipMemset(buf, 0, (size_t)4 << 30));
    CK(hipDeviceSynchronize());
    auto grid = [](int64_t t) { return dim3((unsigned)((t + 255) / 256)); };
    std::printf("# kernel known_bytes\n");
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_stream<false>), grid(n), dim3(256), 0, 0, buf, n);
        hipLaunchKernelGGL((k_stream<true>), grid(n), dim3(256), 0, 0, buf, n);
        if (!rep) std::printf("k_stream<false> %lld\nk_stream<true> %lld\n", (long long)n * 4, (long long)n * 4);
        const int64_t runs = (int64_t)8 << 20;                                 // 8 M runs in 8 M slots of 256 bytes = 2 GiB of address range
#define RUNS(L) hipLaunchKernelGGL((k_runs<L>), grid(runs * L), dim3(256), 0, 0, (int2*)buf, runs * L); if (!rep) std::printf("k_runs<%d> %lld\n", L, (long long)runs * L * 8);
        RUNS(4) RUNS(7) RUNS(8) RUNS(12) RUNS(15) RUNS(16)
        const int64_t nr = (int64_t)64 << 20;                                  // 64 M stores over 2^26 lines (4 GiB)
        hipLaunchKernelGGL(k_random8, grid(nr), dim3(256), 0, 0, (int2*)buf, nr, (uint32_t)((1u << 26) - 1));
        if (!rep) std::printf("k_random8 %lld\n", (long long)nr * 8);
        const int64_t ranges = (int64_t)1 << 20;
        long long known = 0;
        for (int64_t w = 0; w < ranges; ++w) { uint32_t x = (uint32_t)w; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; known += 480 + (x & 63u); }
        hipLaunchKernelGGL((k_ranges<false>), grid(ranges * 64), dim3(256), 0, 0, buf, ranges, 480);
        hipLaunchKernelGGL((k_ranges<true>), grid(ranges * 64), dim3(256), 0, 0, buf, ranges, 480);
        if (!rep) std::printf("k_ranges<false> %lld\nk_ranges<true> %lld\n", known * 4, known * 4);
        CK(hipDeviceSynchronize());
    }
    CK(hipFree(buf));
    return 0;
}
