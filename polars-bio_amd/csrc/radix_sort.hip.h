// radix_sort.hip.h -- hand-written stable LSD radix sort of (u32 key, u32 value)
// pairs for gfx950.  8-bit digits; one pass = histogram -> scan -> scatter.
//
// The ranking is wave64-native: each wavefront finds, with eight 64-bit
// ballots, the set of lanes that hold the same digit (match-any), ranks a key
// by the popcount of its peers in lower lanes, and only the leader lane of a
// peer group touches the LDS counters.  Four wavefronts of a 256-thread
// workgroup are ordered through a [wave][digit] LDS table, so a pass is stable
// for any key distribution (the contig pass has ~24 distinct digits, the
// coordinate passes 256).
#pragma once
#include "scan.hip.h"

namespace ivj {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / kWave;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
constexpr int RS_RADIX = 256;

// Lanes of this wavefront that are valid and hold the same 8-bit digit.
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

// blk_hist is digit-major: blk_hist[digit * nblocks + block].
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                        uint32_t* __restrict__ blk_hist, int nblocks) {
    __shared__ uint32_t h[RS_RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    const uint64_t lt = lanemask_lt();
#pragma unroll 4
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = base + (int64_t)j * RS_THREADS + threadIdx.x;
        const bool valid = i < n;
        const uint32_t d = valid ? ((keys[i] >> shift) & 0xFFu) : 0u;
        const uint64_t peers = wave_match8(d, valid);
        if (valid && (peers & lt) == 0) atomicAdd(&h[d], (uint32_t)__popcll(peers));
    }
    __syncthreads();
    blk_hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// blk_off = exclusive scan of blk_hist over the digit-major layout.
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                           const uint32_t* __restrict__ blk_off, int nblocks) {
    __shared__ uint32_t run[RS_RADIX];             // next free slot of each digit for this workgroup
    __shared__ uint32_t wcnt[RS_WAVES][RS_RADIX];  // per-round, per-wave digit counts
    const int tid = threadIdx.x;
    const int w = tid / kWave;
    run[tid] = blk_off[(int64_t)tid * nblocks + blockIdx.x];
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) wcnt[k][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    const uint64_t lt = lanemask_lt();
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = base + (int64_t)j * RS_THREADS + tid;
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0u;
        const uint32_t val = valid ? vals_in[i] : 0u;
        const uint32_t d = (key >> shift) & 0xFFu;
        const uint64_t peers = wave_match8(d, valid);
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        if (valid && rank == 0) wcnt[w][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        uint32_t dst = 0;
        if (valid) {
            dst = run[d] + rank;
#pragma unroll
            for (int k = 0; k < RS_WAVES; ++k)
                if (k < w) dst += wcnt[k][d];
        }
        __syncthreads();
        {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < RS_WAVES; ++k) { s += wcnt[k][tid]; wcnt[k][tid] = 0; }
            run[tid] += s;
        }
        __syncthreads();
        if (valid) { keys_out[dst] = key; vals_out[dst] = val; }
    }
}

inline int rs_num_blocks(int64_t n) { return (int)((n + RS_TILE - 1) / RS_TILE); }

}  // namespace ivj
