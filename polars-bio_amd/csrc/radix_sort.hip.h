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
// Wavefront w owns the contiguous quarter [w*1024, (w+1)*1024) of the tile and ranks its keys in
// sixteen rounds against its PRIVATE counter row wcnt[w][*] (no workgroup barrier inside the
// rounds: LDS operations of one wavefront execute in order).  One barrier, a per-digit exclusive
// prefix over the four wavefronts, and every key knows its global slot.  Stable: the order inside a
// digit is (wavefront, round, lane) = ascending input position.
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                           const uint32_t* __restrict__ blk_off, int nblocks) {
    __shared__ uint32_t wcnt[RS_WAVES][RS_RADIX];
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) wcnt[k][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * (RS_ITEMS * kWave);
    const uint64_t lt = lanemask_lt();
    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = base + (int64_t)j * kWave + lane;
        const bool valid = i < n;
        key[j] = valid ? keys_in[i] : 0u;
        val[j] = valid ? vals_in[i] : 0u;
    }
    uint32_t* my = wcnt[w];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const bool valid = base + (int64_t)j * kWave + lane < n;
        const uint32_t d = (key[j] >> shift) & 0xFFu;
        const uint64_t peers = wave_match8(d, valid);
        const uint32_t rk = (uint32_t)__popcll(peers & lt);
        const uint32_t before = valid ? my[d] : 0u;
        rank[j] = before + rk;
        __builtin_amdgcn_wave_barrier();
        if (valid && rk == 0) my[d] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // digit `tid`: exclusive prefix over the wavefronts, shifted by the global offset of (digit, block)
        uint32_t x = blk_off[(int64_t)tid * nblocks + blockIdx.x];
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) { const uint32_t t = wcnt[k][tid]; wcnt[k][tid] = x; x += t; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        if (base + (int64_t)j * kWave + lane < n) {
            const uint32_t dst = my[(key[j] >> shift) & 0xFFu] + rank[j];
            keys_out[dst] = key[j];
            vals_out[dst] = val[j];
        }
    }
}

inline int rs_num_blocks(int64_t n) { return (int)((n + RS_TILE - 1) / RS_TILE); }

}  // namespace ivj
