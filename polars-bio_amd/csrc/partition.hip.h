// partition.hip.h -- 256-way stable, LDS-staged bucketing of the probe side (deterministic).
#pragma once
#include "index_view.hip.h"

// =================================================================== probe bucketing
// One 256-way, stable, LDS-staged radix partition of the probe side by the direct-address table
// index of q.end.  After it, consecutive probes touch one narrow slice of bins / b_start / ep /
// b_row, so the random gathers of the count and fill passes hit the XCD's L2 instead of going to
// the fabric.  Output: permuted copies of the three probe columns plus the original (or global)
// row id of every permuted probe; the count and fill kernels then run unchanged on those columns.
namespace ivj {

constexpr int PART_THREADS = 1024;
constexpr int PART_WAVES = PART_THREADS / kWave;
constexpr int PART_ITEMS = 4;
constexpr int PART_TILE = PART_THREADS * PART_ITEMS;
constexpr int PART_BUCKETS = 256;   // bucket 255 = probes without any candidate row

// dynamic LDS of k_part_scatter
constexpr size_t PART_LDS_BYTES = (size_t)PART_TILE * 4 /* one column at a time */ + (size_t)PART_TILE /* bucket ids */ +
                                  (size_t)PART_WAVES * PART_BUCKETS * 4 + 3 * PART_BUCKETS * 4 + 16;

// `bshift` = table-slot shift of the bucket id: 254 buckets over the table slots; bucket 255 = probes without
// any candidate row (they end up at the very end).
__device__ __forceinline__ uint32_t part_digit(uint32_t slot, int bshift) {
    const uint32_t id = slot >> bshift;
    return id < (uint32_t)(PART_BUCKETS - 2) ? id : (uint32_t)(PART_BUCKETS - 2);
}

template <bool STRICT>
__device__ __forceinline__ uint32_t probe_bucket(const IndexView& ix, int32_t c, int32_t qe, int bshift) {
    if ((uint32_t)c >= (uint32_t)ix.n_contigs) return PART_BUCKETS - 1;
    const int4 m0 = ix.cmeta[2 * c], m1 = ix.cmeta[2 * c + 1];
    if (m0.y <= m0.x) return PART_BUCKETS - 1;
    const uint32_t ulo = (uint32_t)m0.z, uhi = (uint32_t)m0.w;
    const unsigned long long tu = (unsigned long long)flip(qe) + (STRICT ? 0ull : 1ull);
    uint32_t j;
    if (tu <= ulo) j = 0;
    else if (tu > uhi) j = ((uhi - ulo) >> m1.x) + 1u;
    else j = ((uint32_t)tu - ulo) >> m1.x;
    return part_digit((uint32_t)m1.y + j, bshift);
}

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch); give every XCD a
// contiguous range of tiles so the partial 64-byte lines two neighbouring tiles write into the
// same bucket meet in ONE L2.  Placement only affects speed, never the result.
__device__ __forceinline__ int xcd_tile(int block, int ntiles) {
    const int per = (ntiles + 7) / 8;
    const int t = (block & 7) * per + (block >> 3);
    return t;
}

constexpr int PART_LDS_CONTIGS = 1024;   // per-contig metadata is staged in LDS up to this many contigs

template <bool STRICT>
__device__ __forceinline__ uint32_t probe_bucket_m(const int4& m0, const int4& m1, int32_t qe, int bshift) {
    if (m0.y <= m0.x) return PART_BUCKETS - 1;
    const uint32_t ulo = (uint32_t)m0.z, uhi = (uint32_t)m0.w;
    const unsigned long long tu = (unsigned long long)flip(qe) + (STRICT ? 0ull : 1ull);
    uint32_t j;
    if (tu <= ulo) j = 0;
    else if (tu > uhi) j = ((uhi - ulo) >> m1.x) + 1u;
    else j = ((uint32_t)tu - ulo) >> m1.x;
    return part_digit((uint32_t)m1.y + j, bshift);
}

template <bool STRICT>
__global__ __launch_bounds__(PART_THREADS) void k_part_hist(IndexView ix, const int32_t* __restrict__ pc,
                                                            const int32_t* __restrict__ pe, int64_t n, int bshift,
                                                            uint32_t* __restrict__ blk_hist, int ntiles, bool vec_ok) {
    __shared__ uint32_t h[PART_BUCKETS];
    __shared__ int4 l_meta[2 * PART_LDS_CONTIGS];
    const int tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    if (threadIdx.x < PART_BUCKETS) h[threadIdx.x] = 0;
    const bool lmeta = ix.n_contigs <= PART_LDS_CONTIGS;
    if (lmeta) for (int k = threadIdx.x; k < 2 * ix.n_contigs; k += PART_THREADS) l_meta[k] = ix.cmeta[k];
    __syncthreads();
    const int64_t base = (int64_t)tile * PART_TILE;
    // each thread takes two groups of four consecutive probes (16-byte loads)
#pragma unroll
    for (int g = 0; g < PART_ITEMS / 4; ++g) {
        const int64_t i0 = base + (int64_t)g * (PART_THREADS * 4) + (int64_t)threadIdx.x * 4;
        int32_t c[4], e[4];
        load_items_nt(pc, i0, n, vec_ok, -1, c);
        load_items_nt(pe, i0, n, vec_ok, 0, e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k >= n) continue;
            uint32_t d = PART_BUCKETS - 1;
            if ((uint32_t)c[k] < (uint32_t)ix.n_contigs) {
                const int4 m0 = lmeta ? l_meta[2 * c[k]] : ix.cmeta[2 * c[k]];
                const int4 m1 = lmeta ? l_meta[2 * c[k] + 1] : ix.cmeta[2 * c[k] + 1];
                d = probe_bucket_m<STRICT>(m0, m1, e[k], bshift);
            }
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < PART_BUCKETS) blk_hist[(int64_t)threadIdx.x * ntiles + tile] = h[threadIdx.x];
}

// blk_off = exclusive scan of blk_hist in bucket-major order.
template <bool STRICT>
__global__ __launch_bounds__(PART_THREADS) void k_part_scatter(IndexView ix, const int32_t* __restrict__ pc,
                                                               const int32_t* __restrict__ ps,
                                                               const int32_t* __restrict__ pe,
                                                               const int32_t* __restrict__ row_id, int64_t n, int bshift,
                                                               const uint32_t* __restrict__ blk_off, int ntiles,
                                                               int32_t* __restrict__ oc, int32_t* __restrict__ os,
                                                               int32_t* __restrict__ oe, int32_t* __restrict__ orow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char part_lds[];
    int32_t* l_buf = reinterpret_cast<int32_t*>(part_lds);                   // one column of the tile
    uint32_t* wcnt = reinterpret_cast<uint32_t*>(l_buf + PART_TILE);         // [PART_WAVES][PART_BUCKETS]
    uint32_t* run = wcnt + PART_WAVES * PART_BUCKETS;                        // running count per bucket
    uint32_t* lstart = run + PART_BUCKETS;                                   // tile-local start of each bucket
    uint32_t* goff = lstart + PART_BUCKETS;                                  // global offset of (bucket, tile)
    unsigned char* l_d = reinterpret_cast<unsigned char*>(goff + PART_BUCKETS);
    uint32_t* wtot = reinterpret_cast<uint32_t*>(l_d + PART_TILE);           // 4 wavefront totals of the 256-value scan

    const int tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    if (tid < PART_BUCKETS) goff[tid] = blk_off[(int64_t)tid * ntiles + tile];
    for (int k = tid; k < PART_WAVES * PART_BUCKETS; k += PART_THREADS) wcnt[k] = 0;
    __syncthreads();
    const int64_t base = (int64_t)tile * PART_TILE;
    const int tile_n = (int)((n - base) < (int64_t)PART_TILE ? (n - base) : (int64_t)PART_TILE);
    const uint64_t lt = lanemask_lt();
    int32_t c[PART_ITEMS], s[PART_ITEMS], e[PART_ITEMS], r[PART_ITEMS];
    uint32_t d[PART_ITEMS], rank[PART_ITEMS];
    // wavefront w owns the contiguous chunk [w*512, (w+1)*512) of the tile: item j of lane l is
    // tile element w*512 + j*64 + l (every load is one contiguous 256-byte segment).
    const int chunk0 = w * (PART_ITEMS * kWave);
#pragma unroll
    for (int j = 0; j < PART_ITEMS; ++j) {
        const int il = chunk0 + j * kWave + lane;
        const int64_t i = base + il;
        const bool valid = il < tile_n;
        c[j] = valid ? pc[i] : -1; s[j] = valid ? ps[i] : 0; e[j] = valid ? pe[i] : 0;
        r[j] = valid ? (row_id ? row_id[i] : (int32_t)i) : -1;
    }
    // rank inside (wavefront chunk, bucket): the row wcnt[w][*] is private to wavefront w, so the
    // eight rounds need no workgroup barrier (LDS operations of one wavefront execute in order).
    uint32_t* my = wcnt + w * PART_BUCKETS;
#pragma unroll
    for (int j = 0; j < PART_ITEMS; ++j) {
        const bool valid = chunk0 + j * kWave + lane < tile_n;
        d[j] = valid ? probe_bucket<STRICT>(ix, c[j], e[j], bshift) : 0u;
        const uint64_t peers = wave_match8(d[j], valid);
        const uint32_t rk = (uint32_t)__popcll(peers & lt);
        const uint32_t before = valid ? my[d[j]] : 0u;
        rank[j] = before + rk;
        __builtin_amdgcn_wave_barrier();
        if (valid && rk == 0) my[d[j]] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // per bucket: exclusive prefix over the wavefronts (in place) and the tile total
    if (tid < PART_BUCKETS) {
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < PART_WAVES; ++k) { const uint32_t t = wcnt[k * PART_BUCKETS + tid]; wcnt[k * PART_BUCKETS + tid] = x; x += t; }
        run[tid] = x;
        // tile-local exclusive scan of the bucket totals (256 values: four full wavefronts)
        const uint32_t inc = wave_inclusive_scan(x, SumOp());
        lstart[tid] = inc - x;
        if (lane == kWave - 1) wtot[tid / kWave] = inc;
    }
    __syncthreads();
    if (tid < PART_BUCKETS) {
        uint32_t add = 0;
        for (int k = 0; k < tid / kWave; ++k) add += wtot[k];
        lstart[tid] += add;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PART_ITEMS; ++j) rank[j] += wcnt[w * PART_BUCKETS + d[j]];
    // Columns are exchanged ONE AT A TIME through a single LDS buffer (4 KiB-threads x 4 B): small
    // LDS footprint -> four workgroups per CU overlap their load / rank / store phases.
    uint32_t pos[PART_ITEMS];
#pragma unroll
    for (int j = 0; j < PART_ITEMS; ++j) {
        pos[j] = lstart[d[j]] + rank[j];
        if (chunk0 + j * kWave + lane < tile_n) l_d[pos[j]] = (unsigned char)d[j];
    }
    __syncthreads();
    // destination of the sorted tile element il = j*PART_THREADS + tid (consecutive threads ->
    // consecutive elements of one bucket run -> coalesced stores)
    uint32_t g[PART_ITEMS];
#pragma unroll
    for (int j = 0; j < PART_ITEMS; ++j) {
        const int il = j * PART_THREADS + tid;
        g[j] = 0;
        if (il < tile_n) { const uint32_t dd = l_d[il]; g[j] = goff[dd] + ((uint32_t)il - lstart[dd]); }
    }
#define IVJ_PART_EXCHANGE(SRC, DST)                                                          \
    do {                                                                                      \
        _Pragma("unroll") for (int j = 0; j < PART_ITEMS; ++j)                                \
            if (chunk0 + j * kWave + lane < tile_n) l_buf[pos[j]] = SRC[j];                   \
        __syncthreads();                                                                      \
        _Pragma("unroll") for (int j = 0; j < PART_ITEMS; ++j) {                              \
            const int il = j * PART_THREADS + tid;                                            \
            if (il < tile_n) DST[g[j]] = l_buf[il];                                           \
        }                                                                                     \
        __syncthreads();                                                                      \
    } while (0)
    IVJ_PART_EXCHANGE(s, os);
    IVJ_PART_EXCHANGE(e, oe);
    IVJ_PART_EXCHANGE(c, oc);
    IVJ_PART_EXCHANGE(r, orow);                      // (k_unpermute places every result by this column: it is not optional)
#undef IVJ_PART_EXCHANGE
}

// ---- inverse of the one-level partition for per-probe results -------------------------------------------
// A kernel that ran over the bucketed probes leaves its per-probe results in bucket order.  Scattering them
// to the original rows costs one partial-line HBM write per value (measured: ~2 ms per 50 M values and
// column).  The partition is stable, so inside every bucket the original row ids ascend: the values of the
// output rows [r0, r0 + UNP_TILE) are 256 short CONTIGUOUS runs (one per bucket, found by two bound searches
// on the row-id column).  A workgroup reads those runs coalesced, places them in LDS by row and writes the
// tile out coalesced.  Up to three columns of 4- or 8-byte values share the searches.
constexpr int UNP_THREADS = 1024;   // 16 wavefronts per 72-KiB workgroup, two workgroups per CU: the gathers of a tile are latency-bound
constexpr int UNP_TILE = 8192;
constexpr int UNP_PER = UNP_TILE / UNP_THREADS;

struct UnpermuteCols {
    const void* src[3];
    void* dst[3];
    int bytes[3];     // 4 or 8
    int ncols;
    int32_t* flag_dst;   // optional: flag_dst[row] = (int32 value of column 0 >= 0), written with column 0
};

// exclusive scan over the UNP_THREADS threads of the workgroup (lds: UNP_THREADS / 64 entries)
template <class Op>
__device__ __forceinline__ int unp_block_exclusive_scan(int v, Op op, int identity, int* lds, int* total) {
    constexpr int NW = UNP_THREADS / kWave;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const int inc = wave_inclusive_scan(v, op);
    if (lane == kWave - 1) lds[w] = inc;
    __syncthreads();
    int wprefix = identity, tot = identity;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int x = lds[i];
        if (i < w) wprefix = op(wprefix, x);
        tot = op(tot, x);
    }
    int exc = __shfl_up(inc, 1, kWave);
    if (lane == 0) exc = identity;
    __syncthreads();
    *total = tot;
    return op(wprefix, exc);
}

__global__ __launch_bounds__(UNP_THREADS) void k_unpermute(const int32_t* __restrict__ rows, const uint32_t* __restrict__ bstart,
                                                           const uint32_t* __restrict__ tile_off, int ntiles,
                                                           int64_t n, UnpermuteCols cols) {
    __shared__ int l_lo[PART_BUCKETS];
    __shared__ int l_pre[PART_BUCKETS + 1];
    __shared__ int lds_i[UNP_THREADS / kWave];
    __shared__ __align__(16) uint8_t l_bucket[UNP_TILE];      // bucket of the t-th element of the concatenated runs
    __shared__ unsigned long long stage[UNP_TILE];
    static_assert(PART_BUCKETS <= UNP_THREADS && PART_BUCKETS <= 256, "one thread per bucket, bucket ids in a byte");
    static_assert(UNP_TILE == UNP_THREADS * UNP_PER && UNP_PER == 8, "8 elements per thread (one 8-byte LDS word of bucket ids)");
    static_assert(UNP_TILE % PART_TILE == 0, "an output tile is a whole number of partition tiles");
    const long long r0 = (long long)blockIdx.x * UNP_TILE;
    const long long r1 = (r0 + UNP_TILE) < n ? (r0 + UNP_TILE) : n;
    const int t_all = (int)(r1 - r0);                          // every row of the tile sits in exactly one bucket
    int cnt = 0;
    if (threadIdx.x < PART_BUCKETS) {
        // the rows [r0, r1) of bucket b are one run of the bucket-ordered arrays (the partition is stable), and the run
        // starts where the partition's scanned histogram put the first of the partition tiles this output tile covers:
        // tile_off[b * ntiles + t] -- two reads instead of two bound searches over the row ids
        const int b = threadIdx.x;
        const int t0 = (int)(blockIdx.x * (UNP_TILE / PART_TILE)), t1 = t0 + UNP_TILE / PART_TILE;
        const int first = (int)tile_off[(int64_t)b * ntiles + t0];
        const int last = t1 < ntiles ? (int)tile_off[(int64_t)b * ntiles + t1] : (int)bstart[b + 1];
        cnt = last - first;
        l_lo[b] = first;
    }
    // element t of the concatenated runs belongs to the last bucket whose prefix is <= t: mark + max-scan
    int tot;
    const int pre = unp_block_exclusive_scan(cnt, SumOp(), 0, lds_i, &tot);
    if (threadIdx.x < PART_BUCKETS) l_pre[threadIdx.x] = pre;
    reinterpret_cast<unsigned long long*>(l_bucket)[threadIdx.x] = 0ull;
    __syncthreads();
    if (cnt > 0) l_bucket[pre] = (uint8_t)threadIdx.x;
    __syncthreads();
    {
        // thread owns 8 consecutive entries; the running max starts from the buckets before them
        uint32_t own = 0;
        uint8_t* mine = l_bucket + threadIdx.x * UNP_PER;
#pragma unroll
        for (int j = 0; j < UNP_PER; ++j) own = own > mine[j] ? own : mine[j];
        int dummy;
        uint32_t run = (uint32_t)unp_block_exclusive_scan((int)own, MaxOp(), 0, lds_i, &dummy);
#pragma unroll
        for (int j = 0; j < UNP_PER; ++j) { run = run > mine[j] ? run : mine[j]; mine[j] = (uint8_t)run; }
    }
    __syncthreads();
    for (int c = 0; c < cols.ncols; ++c) {
        const bool wide = cols.bytes[c] == 8;
        const unsigned long long* s8 = reinterpret_cast<const unsigned long long*>(cols.src[c]);
        const uint32_t* s4 = reinterpret_cast<const uint32_t*>(cols.src[c]);
#pragma unroll 4
        for (int t = threadIdx.x; t < t_all; t += UNP_THREADS) {
            const int b = l_bucket[t];
            const int p = l_lo[b] + (t - l_pre[b]);
            const int r = rows[p] - (int)r0;
            stage[r] = wide ? s8[p] : (unsigned long long)s4[p];
        }
        __syncthreads();
        if (wide) for (int i = threadIdx.x; i < t_all; i += UNP_THREADS) __builtin_nontemporal_store(stage[i], reinterpret_cast<unsigned long long*>(cols.dst[c]) + r0 + i);
        else for (int i = threadIdx.x; i < t_all; i += UNP_THREADS) __builtin_nontemporal_store((uint32_t)stage[i], reinterpret_cast<uint32_t*>(cols.dst[c]) + r0 + i);
        if (c == 0 && cols.flag_dst)
            for (int i = threadIdx.x; i < t_all; i += UNP_THREADS) __builtin_nontemporal_store(((int32_t)(uint32_t)stage[i] >= 0) ? 1 : 0, cols.flag_dst + r0 + i);
        __syncthreads();
    }
}

}  // namespace ivj
