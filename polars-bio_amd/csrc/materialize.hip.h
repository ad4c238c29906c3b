// materialize.hip.h -- row materialisation on the device: the step right after the join
// (reference: the SELECT that renames and gathers every column of both sides for each emitted pair,
// src/operation.rs:272-301).  Output columns are Arrow-layout value buffers (plain little-endian
// fixed-width values, optional validity bitmap), so they can be handed over with the Arrow C Data
// interface without another copy.
#pragma once
#include "index_view.hip.h"

namespace ivj {

constexpr int MAT_THREADS = 256;
constexpr int MAT_ITEMS = 4;     // pairs per thread: 16-byte index reads and column writes

// The five key columns of both sides in one pass over the pair list: the contig id (equal on both
// sides by the join condition), probe start/end, build start/end.  Null output pointers are skipped.
__global__ __launch_bounds__(MAT_THREADS) void k_materialize_keys(
        const int32_t* __restrict__ p_contig, const int32_t* __restrict__ p_start, const int32_t* __restrict__ p_end,
        const int32_t* __restrict__ b_start, const int32_t* __restrict__ b_end,
        const int32_t* __restrict__ probe_idx, const int32_t* __restrict__ build_idx, int64_t n, bool vec_ok,
        int32_t* __restrict__ o_contig, int32_t* __restrict__ o_s1, int32_t* __restrict__ o_e1,
        int32_t* __restrict__ o_s2, int32_t* __restrict__ o_e2) {
    const int64_t i0 = ((int64_t)blockIdx.x * MAT_THREADS + threadIdx.x) * MAT_ITEMS;
    if (i0 >= n) return;
    int32_t p[MAT_ITEMS], b[MAT_ITEMS];
    load_items(probe_idx, i0, n, vec_ok, 0, p);
    load_items(build_idx, i0, n, vec_ok, 0, b);
    int32_t v[MAT_ITEMS];
    if (o_contig) {
#pragma unroll
        for (int k = 0; k < MAT_ITEMS; ++k) v[k] = (i0 + k < n) ? p_contig[p[k]] : 0;
        store_items(o_contig, i0, n, vec_ok, v);
    }
    if (o_s1) {
#pragma unroll
        for (int k = 0; k < MAT_ITEMS; ++k) v[k] = (i0 + k < n) ? p_start[p[k]] : 0;
        store_items(o_s1, i0, n, vec_ok, v);
    }
    if (o_e1) {
#pragma unroll
        for (int k = 0; k < MAT_ITEMS; ++k) v[k] = (i0 + k < n) ? p_end[p[k]] : 0;
        store_items(o_e1, i0, n, vec_ok, v);
    }
    if (o_s2) {
#pragma unroll
        for (int k = 0; k < MAT_ITEMS; ++k) v[k] = (i0 + k < n) ? b_start[b[k]] : 0;
        store_items(o_s2, i0, n, vec_ok, v);
    }
    if (o_e2) {
#pragma unroll
        for (int k = 0; k < MAT_ITEMS; ++k) v[k] = (i0 + k < n) ? b_end[b[k]] : 0;
        store_items(o_e2, i0, n, vec_ok, v);
    }
}

// Arrow `take` of one fixed-width column (T = 4- or 8-byte values): dst[i] = idx[i] >= 0 ? src[idx[i]] : 0.
// With `validity` (one bit per row, LSB first, Arrow layout) a negative index yields a null; the bitmap words
// are assembled with a wavefront ballot, so a workgroup writes whole 64-bit words.
template <class T>
__global__ __launch_bounds__(MAT_THREADS) void k_take(const T* __restrict__ src, const int32_t* __restrict__ idx, int64_t n,
                                                       T* __restrict__ dst, unsigned long long* __restrict__ validity) {
    const int64_t i = (int64_t)blockIdx.x * MAT_THREADS + threadIdx.x;
    int32_t j = -1;
    if (i < n) j = idx[i];
    if (i < n) dst[i] = j >= 0 ? src[j] : T(0);
    if (validity) {
        const unsigned long long m = __ballot(j >= 0);
        if ((threadIdx.x & (kWave - 1)) == 0 && i < n) validity[i >> 6] = m;
    }
}

}  // namespace ivj
