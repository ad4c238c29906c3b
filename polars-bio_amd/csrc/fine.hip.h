// fine.hip.h -- "fine" overlap path: 8192-way probe bucketing + LDS-resident index slices.
//
// The 256-way bucketing of probe.hip.h makes the index slice of a bucket L2-resident, but every
// gather of a probe still pulls its own 64-byte line through the CU's L1 (measured: ~10 L1 line
// lookups per probe, the kernel runs at the one-line-per-clock L1 ceiling).  Here the probe side is
// split into 8192 buckets of <= 2048 table slots (~1200 build rows), so the slice of the index a
// bucket needs -- bin table, starts, (end,pmax) pairs, build rows: ~50 KB -- fits in LDS.  One
// workgroup handles one tile of ONE bucket: it streams the slices into LDS once (coalesced) and
// answers all lookups of its 2048 probes from LDS; rows outside the cached range (crowded buckets,
// windows reaching further down than the margin) transparently fall back to global memory.
//
// The 8192-way split cannot rank with private per-wavefront counters (LDS), so ranks come from LDS
// atomics and per-(tile,bucket) ranges from global atomics: the order of the probes inside a
// bucket is not reproducible from run to run (the pairs of one probe stay contiguous and ordered).
// The deterministic path is the 256-way partition + count/fill pair.
#pragma once
#include "probe.hip.h"

namespace ivj {

constexpr int FINE_BUCKETS = 8192;       // bucket FINE_BUCKETS-1 = probes without any candidate row
constexpr int FINE_SLOT_BITS = 11;       // table slots per bucket <= 2048
constexpr int FINE_SLOTS = 1 << FINE_SLOT_BITS;
constexpr int FINE_ROWS = 2560;          // build rows of a bucket cached in LDS
constexpr int FINE_MARGIN = 128;         // rows below the bucket cached for the windows
constexpr int FINE_THREADS = 512;
constexpr int FINE_TILE = FINE_THREADS * PROBE_ITEMS;   // probes per workgroup
constexpr int FINE_STAGE = 3072;         // pairs per output window

template <bool STRICT>
__device__ __forceinline__ uint32_t fine_bucket(const IndexView& ix, int32_t c, int32_t qe, int bshift) {
    if ((uint32_t)c >= (uint32_t)ix.n_contigs) return FINE_BUCKETS - 1;
    const int4 m0 = ix.cmeta[2 * c], m1 = ix.cmeta[2 * c + 1];
    if (m0.y <= m0.x) return FINE_BUCKETS - 1;
    const uint32_t ulo = (uint32_t)m0.z, uhi = (uint32_t)m0.w;
    const unsigned long long tu = (unsigned long long)flip(qe) + (STRICT ? 0ull : 1ull);
    uint32_t j;
    if (tu <= ulo) j = 0;
    else if (tu > uhi) j = ((uhi - ulo) >> m1.x) + 1u;
    else j = ((uint32_t)tu - ulo) >> m1.x;
    const uint32_t bkt = ((uint32_t)m1.y + j) >> bshift;
    return bkt < (uint32_t)(FINE_BUCKETS - 2) ? bkt : (uint32_t)(FINE_BUCKETS - 2);
}

// global histogram of the buckets: LDS histogram per workgroup, one global atomic per non-empty bin
template <bool STRICT>
__global__ __launch_bounds__(1024) void k_fine_hist(IndexView ix, const int32_t* __restrict__ pc,
                                                    const int32_t* __restrict__ pe, int64_t n, int bshift, bool vec_ok,
                                                    uint32_t* __restrict__ gcount) {
    __shared__ uint32_t h[FINE_BUCKETS];
    for (int k = threadIdx.x; k < FINE_BUCKETS; k += 1024) h[k] = 0;
    __syncthreads();
    const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 3) & ~3ll;     // multiple of 4 rows per workgroup
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    for (int64_t i0 = lo + (int64_t)threadIdx.x * 4; i0 < hi; i0 += 1024 * 4) {
        int32_t c[4], e[4];
        load_items(pc, i0, hi, vec_ok, -1, c);
        load_items(pe, i0, hi, vec_ok, 0, e);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < hi) atomicAdd(&h[fine_bucket<STRICT>(ix, c[k], e[k], bshift)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < FINE_BUCKETS; k += 1024) {
        const uint32_t v = h[k];
        if (v) atomicAdd(&gcount[k], v);
    }
}

// one workgroup: bucket offsets (exclusive scan of the counts), scatter cursors, and the prefix of
// the number of FINE_TILE-sized tiles per bucket (workgroup -> (bucket, tile) map of the join kernel)
__global__ __launch_bounds__(1024) void k_fine_offsets(const uint32_t* __restrict__ gcount, uint32_t* __restrict__ gstart,
                                                       uint32_t* __restrict__ cursor, uint32_t* __restrict__ tile_prefix) {
    __shared__ uint32_t lds_a[1024 / kWave], lds_b[1024 / kWave];
    constexpr int PER = FINE_BUCKETS / 1024;
    uint32_t cnt[PER], tl[PER];
    uint32_t s0 = 0, s1 = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        cnt[k] = gcount[threadIdx.x * PER + k];
        // the last bucket holds probes without candidates: it needs no join tiles
        tl[k] = (threadIdx.x * PER + k == FINE_BUCKETS - 1) ? 0u : (cnt[k] + FINE_TILE - 1) / FINE_TILE;
        s0 += cnt[k]; s1 += tl[k];
    }
    // two workgroup scans (1024 threads = 16 wavefronts)
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    uint32_t i0 = wave_inclusive_scan(s0, SumOp()), i1 = wave_inclusive_scan(s1, SumOp());
    if (lane == kWave - 1) { lds_a[w] = i0; lds_b[w] = i1; }
    __syncthreads();
    uint32_t p0 = 0, p1 = 0;
    for (int k = 0; k < w; ++k) { p0 += lds_a[k]; p1 += lds_b[k]; }
    uint32_t e0 = p0 + i0 - s0, e1 = p1 + i1 - s1;         // exclusive prefixes of this thread
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int b = threadIdx.x * PER + k;
        gstart[b] = e0; cursor[b] = e0; tile_prefix[b] = e1;
        e0 += cnt[k]; e1 += tl[k];
    }
    if (threadIdx.x == 1023) { gstart[FINE_BUCKETS] = e0; tile_prefix[FINE_BUCKETS] = e1; }
}

// scatter: rank inside (tile, bucket) by LDS atomics, range of the (tile, bucket) group by one global
// atomic on the bucket cursor, then direct stores (groups are ~1 element: nothing to coalesce; the
// partial lines of neighbouring tiles merge in the L2).
template <bool STRICT>
__global__ __launch_bounds__(1024) void k_fine_scatter(IndexView ix, const int32_t* __restrict__ pc,
                                                       const int32_t* __restrict__ ps, const int32_t* __restrict__ pe,
                                                       const int32_t* __restrict__ row_id, int64_t n, int bshift, bool vec_ok,
                                                       uint32_t* __restrict__ cursor, int4* __restrict__ orec) {
    __shared__ uint32_t lcnt[FINE_BUCKETS];
    __shared__ uint32_t lbase[FINE_BUCKETS];
    for (int k = threadIdx.x; k < FINE_BUCKETS; k += 1024) lcnt[k] = 0;
    __syncthreads();
    constexpr int G = 2;                                   // two groups of four probes per thread
    const int64_t base = (int64_t)blockIdx.x * (1024 * 4 * G);
    int32_t c[G][4], s[G][4], e[G][4];
    uint32_t d[G][4], r[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t i0 = base + (int64_t)g * (1024 * 4) + (int64_t)threadIdx.x * 4;
        load_items(pc, i0, n, vec_ok, -1, c[g]);
        load_items(ps, i0, n, vec_ok, 0, s[g]);
        load_items(pe, i0, n, vec_ok, 0, e[g]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d[g][k] = 0; r[g][k] = 0;
            if (i0 + k < n) { d[g][k] = fine_bucket<STRICT>(ix, c[g][k], e[g][k], bshift); r[g][k] = atomicAdd(&lcnt[d[g][k]], 1u); }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < FINE_BUCKETS; k += 1024) {
        const uint32_t v = lcnt[k];
        lbase[k] = v ? atomicAdd(&cursor[k], v) : 0u;
    }
    __syncthreads();
    // one 16-byte record {contig, start, end, row} per probe: a quarter of the store requests of four
    // separate columns, and four records of neighbouring tiles fill one 64-byte line in the L2
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t i0 = base + (int64_t)g * (1024 * 4) + (int64_t)threadIdx.x * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) {
                const uint32_t dst = lbase[d[g][k]] + r[g][k];
                orec[dst] = make_int4(c[g][k], s[g][k], e[g][k], row_id ? row_id[i0 + k] : (int32_t)(i0 + k));
            }
        }
    }
}

// per bucket: the build-row range of its table slots, and the bucket of every join tile
__global__ void k_fine_tilemap(const uint32_t* __restrict__ bins, long long bins_len, int bshift,
                               const uint32_t* __restrict__ tile_prefix, int2* __restrict__ brange,
                               uint32_t* __restrict__ tile_bucket) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= FINE_BUCKETS) return;
    const long long s0 = (long long)b << bshift, s1 = (long long)(b + 1) << bshift;
    brange[b] = make_int2((int)bins[s0 < bins_len ? s0 : bins_len - 1], (int)bins[s1 < bins_len ? s1 : bins_len - 1]);
    for (uint32_t t = tile_prefix[b]; t < tile_prefix[b + 1]; ++t) tile_bucket[t] = (uint32_t)b;
}

// LDS-cached view of the index slice of one bucket; indices outside the cached ranges read global memory
struct FineCache {
    const uint32_t* l_bins; int slot0, nslot;      // table slots [slot0, slot0 + nslot]
    const int32_t* l_start; int s_lo, s_hi;        // rows [s_lo, s_hi)
    const int2* l_ep; const int32_t* l_brow; int e_lo, e_hi;   // rows [e_lo, e_hi)
};
// The global-memory fallback sits behind a wavefront-uniform test (ballot), so the vector-memory
// instruction is not even issued while every lane of the wavefront hits the cached range -- a plain
// per-lane select lets the compiler issue BOTH loads for every access.
__device__ __forceinline__ uint32_t fc_bins(const IndexView& ix, const FineCache& fc, int slot, bool active) {
    const int o = slot - fc.slot0;
    const bool in = (uint32_t)o <= (uint32_t)fc.nslot;
    uint32_t v = (active && in) ? fc.l_bins[o] : 0u;
    if (__ballot(active && !in)) { if (active && !in) v = ix.bins[slot]; }
    return v;
}
__device__ __forceinline__ int32_t fc_start(const IndexView& ix, const FineCache& fc, int p, bool active) {
    const bool in = p >= fc.s_lo && p < fc.s_hi;
    int32_t v = (active && in) ? fc.l_start[p - fc.s_lo] : 0;
    if (__ballot(active && !in)) { if (active && !in) v = ix.b_start[p]; }
    return v;
}
__device__ __forceinline__ int2 fc_ep(const IndexView& ix, const FineCache& fc, int p, bool active) {
    const bool in = p >= fc.e_lo && p < fc.e_hi;
    int2 v = make_int2(0, 0);
    if (active && in) v = fc.l_ep[p - fc.e_lo];
    if (__ballot(active && !in)) { if (active && !in) v = ix.ep[p]; }
    return v;
}
struct FineRow {
    const IndexView* ix; const FineCache* fc;
    // called under divergent control flow (per-lane emission): no ballot here; rows of a mask window
    // are within 32 rows below hi, i.e. inside the cached range except for crowded buckets
    __device__ __forceinline__ int32_t operator()(int p) const {
        if (p >= fc->e_lo && p < fc->e_hi) return fc->l_brow[p - fc->e_lo];
        return ix->b_row[p];
    }
};

// One workgroup = one tile (<= FINE_TILE probes) of ONE bucket of the fine partition.
// state[0] = output cursor (= total on exit), state[1] = 1 when the capacity was exceeded.
template <bool STRICT>
__global__ __launch_bounds__(FINE_THREADS) void k_overlap_fused_fine(IndexView ix, const int4* __restrict__ prec,
                                                                     const uint32_t* __restrict__ gstart,
                                                                     const uint32_t* __restrict__ tile_prefix,
                                                                     const uint32_t* __restrict__ tile_bucket,
                                                                     const int2* __restrict__ brange, int bshift,
                                                                     long long bins_len, long long capacity,
                                                                     unsigned long long* __restrict__ state,
                                                                     int32_t* __restrict__ out_probe,
                                                                     int32_t* __restrict__ out_build) {
    __shared__ long long lds[FINE_THREADS / kWave];
    __shared__ long long s_base;
    __shared__ uint32_t l_bins[FINE_SLOTS + 1];
    __shared__ int32_t l_start[FINE_ROWS];
    __shared__ int2 l_ep[FINE_ROWS + FINE_MARGIN];
    __shared__ int32_t l_brow[FINE_ROWS + FINE_MARGIN];
    __shared__ int32_t st_p[FINE_STAGE];
    __shared__ int32_t st_b[FINE_STAGE];

    // workgroup -> (bucket, tile): two dependent reads, then everything else is requested at once
    const uint32_t blk = blockIdx.x;
    if (blk >= tile_prefix[FINE_BUCKETS]) return;
    const int bucket = (int)tile_bucket[blk];
    const uint32_t t = blk - tile_prefix[bucket];
    const int64_t qlo = (int64_t)gstart[bucket];
    const int64_t q1e = (int64_t)gstart[bucket + 1];
    const int2 rr = brange[bucket];
    const int64_t q0 = qlo + (int64_t)t * FINE_TILE;
    const int64_t q1 = q0 + FINE_TILE < q1e ? q0 + FINE_TILE : q1e;

    // probe records of this thread (requested together with the slices)
    const int64_t i0 = q0 + (int64_t)threadIdx.x * PROBE_ITEMS;
    int32_t c[PROBE_ITEMS], s[PROBE_ITEMS], e[PROBE_ITEMS], row[PROBE_ITEMS];
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) {
        int4 v = make_int4(-1, 0, 0, 0);
        if (i0 + k < q1) v = prec[i0 + k];
        c[k] = v.x; s[k] = v.y; e[k] = v.z; row[k] = v.w;
    }
    // stream the bucket's slices of the index into LDS
    FineCache fc;
    fc.slot0 = bucket << bshift; fc.nslot = 1 << bshift;
    const int r_lo = rr.x, r_hi = rr.y;
    fc.l_bins = l_bins; fc.l_start = l_start; fc.l_ep = l_ep; fc.l_brow = l_brow;
    fc.s_lo = r_lo; fc.s_hi = r_hi < r_lo + FINE_ROWS ? r_hi : r_lo + FINE_ROWS;
    fc.e_lo = r_lo > FINE_MARGIN ? r_lo - FINE_MARGIN : 0; fc.e_hi = fc.s_hi;
    for (int k = threadIdx.x; k <= fc.nslot; k += FINE_THREADS) {
        const long long sl = (long long)fc.slot0 + k;
        l_bins[k] = ix.bins[sl < bins_len ? sl : bins_len - 1];
    }
    for (int p = fc.s_lo + threadIdx.x; p < fc.s_hi; p += FINE_THREADS) l_start[p - fc.s_lo] = ix.b_start[p];
    for (int p = fc.e_lo + threadIdx.x; p < fc.e_hi; p += FINE_THREADS) { l_ep[p - fc.e_lo] = ix.ep[p]; l_brow[p - fc.e_lo] = ix.b_row[p]; }
    __syncthreads();

    int hi[PROBE_ITEMS], x[PROBE_ITEMS], cnt[PROBE_ITEMS], a[PROBE_ITEMS];
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) {
        hi[k] = 0; a[k] = 0;
        const bool ok = i0 + k < q1 && (uint32_t)c[k] < (uint32_t)ix.n_contigs;
        int4 m0 = make_int4(0, 0, 0, 0), m1 = make_int4(0, 0, 0, 0);
        if (ok) { m0 = ix.cmeta[2 * c[k]]; m1 = ix.cmeta[2 * c[k] + 1]; }
        a[k] = m0.x;
        const int b = m0.y;
        const uint32_t ulo = (uint32_t)m0.z, uhi = (uint32_t)m0.w;
        const unsigned long long tu = (unsigned long long)flip(e[k]) + (STRICT ? 0ull : 1ull);
        bool search = false;
        int slot = 0;
        if (!ok || b <= a[k] || tu <= ulo) hi[k] = a[k];
        else if (tu > uhi) hi[k] = b;
        else { search = true; slot = m1.y + (int)(((uint32_t)tu - ulo) >> m1.x); }
        // every loop below runs until no lane of the wavefront needs it (uniform), so the accessors'
        // ballots are legal
        int l = (int)fc_bins(ix, fc, slot, search), h = (int)fc_bins(ix, fc, slot + 1, search);
        if (!search) { l = 0; h = 0; }
        while (__ballot(l < h)) {
            const bool act = l < h;
            const int m = l + ((h - l) >> 1);
            const int32_t v = fc_start(ix, fc, m, act);
            if (act) { if ((unsigned long long)flip(v) < tu) l = m + 1; else h = m; }
        }
        if (search) hi[k] = l;
        // window below hi: one LDS read per row
        uint32_t mask = 0;
        bool small = true, open = true;
        const int top = hi[k] - 1;
        int p = top;
        while (__ballot(open && p >= a[k])) {
            const bool act = open && p >= a[k];
            const int2 v = fc_ep(ix, fc, p, act);
            if (act) {
                if (!lt_op<STRICT>(s[k], v.y)) open = false;
                else if (top - p >= 32) { small = false; open = false; }
                else { if (lt_op<STRICT>(s[k], v.x)) mask |= 1u << (top - p); --p; }
            }
        }
        int cn = small ? __popc(mask) : 0;
        x[k] = (int)mask;
        unsigned long long todo = __ballot(!small);        // long windows: whole wavefront, global memory
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int ca = __shfl(a[k], src, kWave), chi = __shfl(hi[k], src, kWave);
            const int32_t cqs = __shfl(s[k], src, kWave);
            const int cc = wave_count_window<STRICT>(ix, ca, chi, cqs);
            if (lane == src) { cn = cc; x[k] = cc; }
        }
        if (!small) hi[k] |= (int)0x80000000;
        cnt[k] = cn;
    }
    long long tsum = 0;
#pragma unroll
    for (int k = 0; k < PROBE_ITEMS; ++k) tsum += cnt[k];
    // workgroup scan over FINE_THREADS threads (8 wavefronts)
    long long inc = wave_inclusive_scan(tsum, SumOp());
    const int w = threadIdx.x / kWave;
    if (lane == kWave - 1) lds[w] = inc;
    __syncthreads();
    long long wprefix = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < FINE_THREADS / kWave; ++k) { const long long v = lds[k]; if (k < w) wprefix += v; tot += v; }
    const long long loc0 = wprefix + inc - tsum;
    if (threadIdx.x == 0) {
        const long long base = tot ? (long long)atomicAdd(&state[0], (unsigned long long)tot) : 0ll;
        if (base + tot > capacity) { atomicExch(&state[1], 1ull); s_base = -1; }
        else s_base = base;
    }
    __syncthreads();
    const long long tbase = s_base;
    if (tbase < 0 || tot == 0) return;                     // uniform
    FineRow rowof{&ix, &fc};
    emit_tile_rows<STRICT, FINE_THREADS, FINE_STAGE>(ix, rowof, PairOut{out_probe, out_build}, hi, x, cnt, row, s, loc0, tot, tbase, st_p, st_b);
}

}  // namespace ivj
