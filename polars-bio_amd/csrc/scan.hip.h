// scan.hip.h -- device-wide prefix scans for gfx950 (wave64), three launches:
// tile reduce -> scan of tile partials (one workgroup) -> tile apply.
// Used for: output tile bases (i64 sum), cluster ids and merged lengths of the sort-scan family; the sort and the
// partition tables use the single-launch look-back scans of onesweep.hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ivj {

constexpr int kWave = 64;

// ---- wavefront helpers shared by the partition / sort kernels ----
// Lanes of this wavefront that are valid and hold the same 8-bit digit (match-any from eight 64-bit ballots).
__device__ __forceinline__ uint64_t wave_match8(uint32_t digit, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    return (1ull << (threadIdx.x & (kWave - 1))) - 1ull;
}
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

struct SumOp {
    template <class T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
};
struct MaxOp {
    template <class T> __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
};

template <class T, class Op>
__device__ __forceinline__ T wave_inclusive_scan(T v, Op op) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        T o = __shfl_up(v, d, kWave);
        if (lane >= d) v = op(o, v);
    }
    return v;
}

// Exclusive scan of one value per thread across a 256-thread workgroup.
// lds must hold SCAN_THREADS / kWave elements.  Returns the exclusive prefix
// of this thread; *total receives the workgroup aggregate (all threads).
template <class T, class Op>
__device__ __forceinline__ T block_exclusive_scan(T v, Op op, T identity, T* lds, T* total) {
    constexpr int NW = SCAN_THREADS / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    T inc = wave_inclusive_scan(v, op);
    if (lane == kWave - 1) lds[w] = inc;
    __syncthreads();
    T wprefix = identity, tot = identity;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        T x = lds[i];
        if (i < w) wprefix = op(wprefix, x);
        tot = op(tot, x);
    }
    T exc = __shfl_up(inc, 1, kWave);
    if (lane == 0) exc = identity;
    __syncthreads();
    *total = tot;
    return op(wprefix, exc);
}

template <class T, class Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const T* __restrict__ in, int64_t n, T identity,
                                                              T* __restrict__ partials) {
    __shared__ T lds[SCAN_THREADS / kWave];
    Op op;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    T acc = identity;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        int64_t i = base + j;
        if (i < n) acc = op(acc, in[i]);
    }
    T tot;
    block_exclusive_scan(acc, op, identity, lds, &tot);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// One workgroup: exclusive scan of the partials in place; optional grand total.
template <class T, class Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_partials(T* __restrict__ partials, int64_t m, T identity,
                                                                T* __restrict__ total_out) {
    __shared__ T lds[SCAN_THREADS / kWave];
    Op op;
    T carry = identity;
    for (int64_t base = 0; base < m; base += SCAN_THREADS) {
        int64_t i = base + threadIdx.x;
        T v = i < m ? partials[i] : identity;
        T tot;
        T exc = block_exclusive_scan(v, op, identity, lds, &tot);
        if (i < m) partials[i] = op(carry, exc);
        carry = op(carry, tot);
    }
    if (total_out && threadIdx.x == 0) *total_out = carry;
}

template <class T, class Op, bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const T* __restrict__ in, T* __restrict__ out, int64_t n,
                                                             T identity, const T* __restrict__ partials) {
    __shared__ T lds[SCAN_THREADS / kWave];
    Op op;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T acc = identity;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        int64_t i = base + j;
        v[j] = i < n ? in[i] : identity;
        acc = op(acc, v[j]);
    }
    T tot;
    T prefix = block_exclusive_scan(acc, op, identity, lds, &tot);
    prefix = op(partials[blockIdx.x], prefix);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        int64_t i = base + j;
        T inc = op(prefix, v[j]);
        if (i < n) out[i] = INCLUSIVE ? inc : prefix;
        prefix = inc;
    }
}

inline int64_t scan_num_tiles(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

}  // namespace ivj
