// ixsort3.hip.h -- index build, round 5: ONE balanced most-significant pass + an LDS-resident sort per bucket that writes the index.
//
// The round-2 build (onesweep.hip.h) is three least-significant 11-bit passes over 16-byte records, each a histogram, a look-back scan
// and a scatter, then k_ix_final: 14 launches and ~0.45 ms for 5 M rows -- a fifth of a config-3 step, all of it latency (every pass
// walks ~5 sub-tiles per workgroup behind seven barriers).  Here the records cross HBM ONCE between the caller's columns and the index:
//
//   k_v3_stats    per-contig min / max of the starts (LDS-privatised), inverted-row flag
//   k_v3_hist     key geometry (every workgroup derives it from the extremes, workgroup 0 records it): the contigs' start ranges laid
//                 end to end give a DENSE linear key lin = base[c] + (start - min_c) < span, and bucket(lin) = (lin * M) >> 32 with
//                 M = floor(2^43 / span) cuts [0, span) into 2048 equal ranges -- for rows spread evenly over their contigs every bucket
//                 holds n / 2048 rows, whatever the contig lengths (a bit field of the (contig, start) key would leave a human genome's
//                 24 contigs in 739 of 2048 buckets); then the per-(bucket, chunk) histogram
//   k_scan_lb_u32 its exclusive scan (onesweep.hip.h)
//   k_v3_check    rows of the largest bucket (one workgroup)
//   -- 8 bytes to the host, read while the next kernel runs: a bucket above V3_CAP rows (clustered build sides) or a span beyond
//      32 bits hands the build to the round-2 sort, so exactness never rests on the balance --
//   k_v3_scatter  stable scatter of the 16-byte records {start, end, row, contig} into their buckets (the pass kernel of the round-2
//                 sort with the bucket function above)
//   k_v3_local    one workgroup per bucket: rows -> LDS; the (contig, end) maximum of the bucket is published at once (it does not
//                 depend on the order), then ONE counting pass inside LDS -- 4096 bins over the bucket-local key (~0.6 rows per bin),
//                 exclusive scan, rows dropped into their bin's range, and every row ranks itself among the rows of its bin by
//                 (key, slot): equal keys keep their input order -- then the index arrays straight from LDS in sorted order:
//                 b_start / (end, prefix max) / b_row / b_contig, the prefix max carried across buckets by a wavefront-wide decoupled
//                 look-back over the composites (buckets are handed out by ticket, so every predecessor is running), segment offsets.
//
// Six launches, two of them one-workgroup kernels.  Order and content of the index are exactly the round-2 build's: rows sorted by
// (contig, start) with equal keys in input order, rows outside the dictionary parked last under contig id n_contigs.
#pragma once
#include "onesweep.hip.h"

namespace ivj {

constexpr int V3_BUCKETS = OS_RADIX;                   // 2048: the pass kernel's digit count
constexpr int V3_ITEMS = 4;
constexpr int V3_CAP = OS_THREADS * V3_ITEMS;          // rows of one bucket the local kernel holds in LDS
constexpr int V3_MAX_KEYS = 256;                       // contig keys 0 .. n_contigs (the last one: rows outside the dictionary)
constexpr int V3_MAX_BIN_BITS = 12;                    // bins of the local counting pass: up to 4096 (>= the bucket capacity: below one row per bin)

struct V3Meta {                                        // device-resident, zero-initialised before every build
    uint32_t inverted;                                 // some row of the dictionary has start > end
    uint32_t pad0;
    uint32_t bad;                                      // the linear keys do not fit 32 bits: the host takes the round-2 sort
    uint32_t max_bucket;                               // rows of the largest bucket (k_v3_scatter)
    uint32_t ticket;                                   // bucket tickets of k_v3_local
    uint32_t wbits;                                    // bits of a bucket-local key
    uint32_t pad[2];
    unsigned long long M;                              // bucket(lin) = (lin * M) >> 32
    unsigned long long span;                           // sum of the contigs' start ranges
    uint32_t cmax[V3_MAX_KEYS];                        // per contig key: max of flip(start)         (both grow from 0:
    uint32_t cimn[V3_MAX_KEYS];                        //                 max of ~flip(start)          a key has rows iff cmax | cimn != 0)
    uint32_t base[V3_MAX_KEYS];                        // linear key of the contig's smallest start
};

__device__ __forceinline__ uint32_t v3_bucket(uint32_t lin, unsigned long long M) { return (uint32_t)(((unsigned long long)lin * M) >> 32); }

__device__ __forceinline__ void v3_load_tables(const V3Meta* __restrict__ meta, int nk, uint32_t* l_base, uint32_t* l_cmin) {
    for (int k = threadIdx.x; k < nk; k += OS_THREADS) { l_base[k] = meta->base[k]; l_cmin[k] = ~meta->cimn[k]; }
}

// ---- statistics ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OS_THREADS) void k_v3_stats(const int32_t* __restrict__ start, const int32_t* __restrict__ end,
                                                        const int32_t* __restrict__ contig, int64_t n, int32_t n_contigs, V3Meta* __restrict__ meta) {
    __shared__ uint32_t l_mx[V3_MAX_KEYS], l_imn[V3_MAX_KEYS];
    __shared__ uint32_t l_inv;
    const int tid = threadIdx.x, nk = n_contigs + 1;
    for (int k = tid; k < V3_MAX_KEYS; k += OS_THREADS) { l_mx[k] = 0u; l_imn[k] = 0u; }
    if (tid == 0) l_inv = 0u;
    __syncthreads();
    uint32_t inv = 0;
    for (int64_t i = (int64_t)blockIdx.x * OS_THREADS + tid; i < n; i += (int64_t)gridDim.x * OS_THREADS) {
        const int32_t s = start[i], c0 = contig[i];
        const int c = (uint32_t)c0 < (uint32_t)n_contigs ? c0 : n_contigs;
        const uint32_t u = flip(s);
        // (an atomic only when it can change the value: after the first few rows of a contig nearly none is issued)
        if (u > l_mx[c]) atomicMax(&l_mx[c], u);
        if (~u > l_imn[c]) atomicMax(&l_imn[c], ~u);
        if (s > end[i] && c < n_contigs) inv = 1u;
    }
    if (inv) l_inv = 1u;
    __syncthreads();
    // (no fence, no "last workgroup" here: a device-scope release costs an L2 write-back per workgroup on this part -- the geometry
    // is derived by the next kernel, behind the kernel boundary)
    for (int k = tid; k < nk; k += OS_THREADS) {
        const uint32_t a = l_mx[k], b = l_imn[k];
        if (a > os_ld(&meta->cmax[k])) atomicMax(&meta->cmax[k], a);
        if (b > os_ld(&meta->cimn[k])) atomicMax(&meta->cimn[k], b);
    }
    if (tid == 0 && l_inv && !os_ld(&meta->inverted)) atomicOr(&meta->inverted, 1u);
}

// key geometry from the per-contig extremes: every workgroup of k_v3_hist derives it (a 256-entry scan and one 64-bit division),
// workgroup 0 records it for the kernels behind.  -> false: unusable (the linear keys do not fit 32 bits)
struct V3Geom { unsigned long long M; uint32_t wbits; bool bad; };
__device__ __forceinline__ V3Geom v3_geometry(V3Meta* __restrict__ meta, int nk, uint32_t* l_base, uint32_t* l_cmin, unsigned long long* l_ws /* OS_WAVES + 2 */, bool record) {
    const int tid = threadIdx.x;
    unsigned long long sp = 0;
    uint32_t cmn = 0xffffffffu;
    if (tid < nk) {
        const uint32_t cmx = meta->cmax[tid], cim = meta->cimn[tid];
        cmn = ~cim;
        if ((cmx | cim) != 0u) sp = (unsigned long long)(cmx - cmn) + 1ull;
    }
    unsigned long long total;
    const unsigned long long pre = sl_block_exclusive_sum<unsigned long long>(sp, l_ws, &total);
    if (tid < nk) { l_base[tid] = (uint32_t)pre; l_cmin[tid] = cmn; if (record) meta->base[tid] = (uint32_t)pre; }
    if (tid == 0) {
        const bool bad = total == 0ull || total > 0xffffffffull;
        const unsigned long long M = bad ? 1ull : (1ull << 43) / total;
        const unsigned long long wmax = (1ull << 32) / M;                       // bucket-local keys lie in [0, 2^32 / M]
        const uint32_t wbits = (uint32_t)os_bits_for(wmax > 0xffffffffull ? 0xffffffffu : (uint32_t)wmax);
        l_ws[OS_WAVES] = M;
        l_ws[OS_WAVES + 1] = (unsigned long long)wbits | (bad ? (1ull << 32) : 0ull);
        if (record) { meta->bad = bad ? 1u : 0u; meta->span = total; meta->M = M; meta->wbits = wbits; }
    }
    __syncthreads();
    V3Geom g;
    g.M = l_ws[OS_WAVES];
    g.wbits = (uint32_t)l_ws[OS_WAVES + 1];
    g.bad = (l_ws[OS_WAVES + 1] >> 32) != 0ull;
    return g;
}

// ---- geometry + per-(bucket, chunk) histogram (bucket-major: hist[b * nchunks + g]) ------------------------------------------------
__global__ __launch_bounds__(OS_THREADS) void k_v3_hist(const int32_t* __restrict__ contig, const int32_t* __restrict__ start, int64_t n,
                                                       int32_t n_contigs, V3Meta* __restrict__ meta, int chunk, int nchunks,
                                                       uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[V3_BUCKETS];
    __shared__ uint32_t l_base[V3_MAX_KEYS], l_cmin[V3_MAX_KEYS];
    __shared__ unsigned long long l_ws[OS_WAVES + 2];
    const V3Geom g = v3_geometry(meta, n_contigs + 1, l_base, l_cmin, l_ws, blockIdx.x == 0);
    if (g.bad) return;                                                          // uniform
    const unsigned long long M = g.M;
    for (int k = threadIdx.x; k < V3_BUCKETS; k += OS_THREADS) h[k] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * chunk;
    const int64_t end = base + chunk < n ? base + chunk : n;
    for (int64_t i = base + threadIdx.x; i < end; i += OS_THREADS) {
        const int32_t c0 = contig[i];
        const int c = (uint32_t)c0 < (uint32_t)n_contigs ? c0 : n_contigs;
        const uint32_t lin = l_base[c] + (flip(start[i]) - l_cmin[c]);
        atomicAdd(&h[v3_bucket(lin, M)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < V3_BUCKETS; k += OS_THREADS) hist[(int64_t)k * nchunks + blockIdx.x] = h[k];
}

// Rows of the largest bucket: bucket b = [off[b * nchunks], off[(b + 1) * nchunks]) of the scanned, bucket-major histogram -- and of the
// largest MERGED bucket for every merge shift ms = 1 .. V3_MAX_MERGE (2^ms adjacent buckets: contiguous in the scattered records, their
// key ranges adjacent, so the local kernel can take them as one).  A small build side spread over 2048 buckets paid the local kernel's
// fixed cost 2048 times (1 M rows: 0.056 ms; as 256 merged buckets 0.031 ms): the device picks the LARGEST shift whose largest merged
// bucket still fits the local kernel (<= V3_CAP rows; ms_max pins an upper bound), so clustered build sides keep the fine buckets.
// meta->max_bucket = rows of the largest bucket at the chosen shift | shift << 24 (n <= 2^23).  One workgroup.
constexpr int V3_MAX_MERGE = 5;
__global__ __launch_bounds__(OS_THREADS) void k_v3_check(const uint32_t* __restrict__ off, int nchunks, int64_t n, int ms_max, V3Meta* __restrict__ meta,
                                                        uint32_t* __restrict__ hw, uint32_t hw_seq) {
    __shared__ uint32_t wmx[V3_MAX_MERGE + 1][OS_WAVES];
    if (meta->bad) {
        if (threadIdx.x == 0 && hw) { hw_store(hw + 0, 1u); hw_store(hw + 1, 0u); hw_store(hw + 2, hw_seq); }   // host words: {bad, largest bucket | shift << 24, seq}
        return;
    }
    const int tid = threadIdx.x;
    uint32_t sz[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int b = 2 * tid + q;
        const uint32_t a = off[(int64_t)b * nchunks];
        const uint32_t z = b + 1 < V3_BUCKETS ? off[(int64_t)(b + 1) * nchunks] : (uint32_t)n;
        sz[q] = z - a;
    }
    uint32_t mx[V3_MAX_MERGE + 1];
    mx[0] = sz[0] > sz[1] ? sz[0] : sz[1];
    uint32_t sum = sz[0] + sz[1];                                               // shift 1: this thread's pair
    mx[1] = sum;
#pragma unroll
    for (int k = 2; k <= V3_MAX_MERGE; ++k) { sum += __shfl_xor(sum, 1 << (k - 2), kWave); mx[k] = sum; }   // 2^(k-1) adjacent threads (inside a wavefront)
#pragma unroll
    for (int k = 0; k <= V3_MAX_MERGE; ++k) {
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(mx[k], d, kWave); mx[k] = o > mx[k] ? o : mx[k]; }
        if ((tid & (kWave - 1)) == 0) wmx[k][tid / kWave] = mx[k];
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t best = 0, best_ms = 0;
#pragma unroll
        for (int k = 0; k <= V3_MAX_MERGE; ++k) {
            uint32_t m = 0;
#pragma unroll
            for (int w = 0; w < OS_WAVES; ++w) m = wmx[k][w] > m ? wmx[k][w] : m;
            if (k == 0 || (k <= ms_max && m <= (uint32_t)V3_CAP)) { best = m; best_ms = (uint32_t)k; }
        }
        meta->max_bucket = best | (best_ms << 24);
        if (hw) { hw_store(hw + 0, 0u); hw_store(hw + 1, best | (best_ms << 24)); hw_store(hw + 2, hw_seq); }
    }
}

// ---- the one pass over HBM: stable scatter into the buckets (k_os_scatter<true> with the bucket function above) ---------------------
struct V3PassLds { OsPassLds P; int tbase, tcmin, total; };
__host__ __device__ inline V3PassLds v3_pass_lds() {
    V3PassLds L;
    L.P = os_pass_lds();
    int o = (L.P.total + 15) & ~15;
    L.tbase = o; o += 4 * V3_MAX_KEYS;
    L.tcmin = o; o += 4 * V3_MAX_KEYS;
    L.total = o;
    return L;
}

__global__ __launch_bounds__(OS_THREADS) void k_v3_scatter(const int32_t* __restrict__ contig, const int32_t* __restrict__ start,
                                                          const int32_t* __restrict__ end, const int32_t* __restrict__ row_id,
                                                          int4* __restrict__ dst, int64_t n, int32_t n_contigs, V3Meta* __restrict__ meta,
                                                          int chunk, int nchunks, const uint32_t* __restrict__ off /* exclusive scan of the histogram */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char os_lds[];
    const V3PassLds LL = v3_pass_lds();
    const OsPassLds& L = LL.P;
    int4* l_rec = reinterpret_cast<int4*>(os_lds + L.rec);
    unsigned short* wcnt = reinterpret_cast<unsigned short*>(os_lds + L.wcnt);
    unsigned short* l_d = reinterpret_cast<unsigned short*>(os_lds + L.d);
    uint32_t* base = reinterpret_cast<uint32_t*>(os_lds + L.base);
    uint32_t* lstart = reinterpret_cast<uint32_t*>(os_lds + L.lstart);
    uint32_t* wsum = reinterpret_cast<uint32_t*>(os_lds + L.wsum);
    uint32_t* l_base = reinterpret_cast<uint32_t*>(os_lds + LL.tbase);
    uint32_t* l_cmin = reinterpret_cast<uint32_t*>(os_lds + LL.tcmin);
    if (meta->bad) return;                                                      // uniform
    const unsigned long long M = meta->M;
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    v3_load_tables(meta, n_contigs + 1, l_base, l_cmin);
    for (int k = tid; k < OS_RADIX; k += OS_THREADS) base[k] = off[(int64_t)k * nchunks + blockIdx.x];
    for (int k = tid; k < OS_RADIX * OS_WAVES / 2; k += OS_THREADS) reinterpret_cast<uint32_t*>(wcnt)[k] = 0;
    __syncthreads();
    const int64_t cbase = (int64_t)blockIdx.x * chunk;
    const int64_t cend = cbase + chunk < n ? cbase + chunk : n;
    const uint64_t lt = lanemask_lt();
    unsigned short* my = wcnt + w * OS_RADIX;
    const int el0 = w * (OS_ITEMS * kWave) + lane;
    int4 nxt[OS_ITEMS];
    auto load_tile = [&](int64_t tb) {
        const int tn = (int)((cend - tb) < (int64_t)OS_TILE ? (cend - tb) : (int64_t)OS_TILE);
#pragma unroll
        for (int j = 0; j < OS_ITEMS; ++j) {
            const int il = el0 + j * kWave;
            nxt[j] = il < tn ? os_load_record(true, contig, start, end, row_id, nullptr, tb + il, n_contigs) : make_int4(0, 0, 0, 0);
        }
    };
    if (cbase < cend) load_tile(cbase);
    for (int64_t tbase = cbase; tbase < cend; tbase += OS_TILE) {
        const int tile_n = (int)((cend - tbase) < (int64_t)OS_TILE ? (cend - tbase) : (int64_t)OS_TILE);
        int4 r[OS_ITEMS];
        uint32_t d[OS_ITEMS], rank[OS_ITEMS];
#pragma unroll
        for (int j = 0; j < OS_ITEMS; ++j) {
            r[j] = nxt[j];
            d[j] = v3_bucket(l_base[r[j].w] + (flip(r[j].x) - l_cmin[r[j].w]), M);
        }
        if (tbase + OS_TILE < cend) load_tile(tbase + OS_TILE);
#pragma unroll
        for (int j = 0; j < OS_ITEMS; ++j) {
            const bool valid = el0 + j * kWave < tile_n;
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < OS_BITS; ++b) {
                const bool bit = (d[j] >> b) & 1u;
                const uint64_t m = __ballot(valid && bit);
                peers &= bit ? m : ~m;
            }
            const uint32_t rk = (uint32_t)__popcll(peers & lt);
            const uint32_t before = valid ? (uint32_t)my[d[j]] : 0u;
            rank[j] = before + rk;
            __builtin_amdgcn_wave_barrier();
            if (valid && rk == 0) my[d[j]] = (unsigned short)(before + (uint32_t)__popcll(peers));
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        uint32_t x0 = 0, x1 = 0;
        {
            uint32_t* row32 = reinterpret_cast<uint32_t*>(wcnt) + tid;
#pragma unroll
            for (int k = 0; k < OS_WAVES; ++k) {
                const uint32_t v = row32[k * (OS_RADIX / 2)];
                row32[k * (OS_RADIX / 2)] = x0 | (x1 << 16);
                x0 += v & 0xffffu; x1 += v >> 16;
            }
        }
        uint32_t tsum;
        const uint32_t pre = sl_block_exclusive_sum(x0 + x1, wsum, &tsum);
        lstart[2 * tid] = pre;
        lstart[2 * tid + 1] = pre + x0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < OS_ITEMS; ++j) {
            if (el0 + j * kWave < tile_n) {
                const uint32_t pos = lstart[d[j]] + (uint32_t)my[d[j]] + rank[j];
                l_rec[pos] = r[j];
                l_d[pos] = (unsigned short)d[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < OS_ITEMS; ++j) {
            const int il = j * OS_THREADS + tid;
            if (il < tile_n) {
                const uint32_t dd = l_d[il];
                const int4 rr = l_rec[il];
                dst[base[dd] + ((uint32_t)il - lstart[dd])] = rr;
            }
        }
        for (int k = tid; k < OS_RADIX * OS_WAVES / 2; k += OS_THREADS) reinterpret_cast<uint32_t*>(wcnt)[k] = 0;
        __syncthreads();
        base[2 * tid] += x0;
        base[2 * tid + 1] += x1;
        __syncthreads();
    }
}

// ---- one workgroup per bucket: LDS sort + index arrays ------------------------------------------------------------------------------
// cap = rows the launch's buckets hold at most (the host knows the largest bucket: a multiple of 1024), bins = 1 << bin_bits >= cap.
// STAGE: the rows themselves wait in LDS (16 bytes per row) -- two workgroups share a CU up to cap = 2048; beyond, the rows are
// fetched again from the bucket's 16-byte records (contiguous, L2-resident) where the index arrays are written, and 8 bytes per row
// of LDS keep two workgroups per CU at cap = 3072 as well.
struct V3LocalLds { int rec, bst, cur, tk, ts, ifin, tbase, tcmin, wsum, total; };
__host__ __device__ inline V3LocalLds v3_local_lds(int cap, int bins, bool stage) {
    V3LocalLds L;
    int o = 0;
    L.rec = o; o += stage ? 16 * cap : 0;
    L.bst = o; o += 4 * (bins + 4);                    // bin counts, then bin starts (exclusive scan); [bins] = rows of the bucket
    L.cur = o; o += 4 * bins;                          // running cursor of every bin
    L.tk = o; o += 4 * cap;                            // keys in bin order (after the ranking: the prefix maxima per sorted position)
    L.ts = o; o += 2 * cap;                            // slots in bin order
    L.ifin = o; o += 2 * cap;                          // slot of the row at sorted position q
    L.tbase = o; o += 4 * V3_MAX_KEYS;
    L.tcmin = o; o += 4 * V3_MAX_KEYS;
    L.wsum = (o + 15) & ~15; o = L.wsum + 8 * OS_WAVES;
    L.total = o;
    return L;
}

template <bool STAGE>
__global__ __launch_bounds__(OS_THREADS, 8) void k_v3_local(const int4* __restrict__ rec, const uint32_t* __restrict__ off, int nchunks, int64_t n,
                                                        int32_t n_contigs, int cap, int bin_bits, int ms, V3Meta* __restrict__ meta, unsigned long long* __restrict__ status64,
                                                        int32_t* __restrict__ b_start, int2* __restrict__ ep, int32_t* __restrict__ b_row,
                                                        int32_t* __restrict__ b_contig, int32_t* __restrict__ seg, int32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char os_lds[];
    const int bins = 1 << bin_bits;
    const V3LocalLds L = v3_local_lds(cap, bins, STAGE);
    int4* l_rec = reinterpret_cast<int4*>(os_lds + L.rec);
    uint32_t* bst = reinterpret_cast<uint32_t*>(os_lds + L.bst);
    uint32_t* cur = reinterpret_cast<uint32_t*>(os_lds + L.cur);
    uint32_t* tk = reinterpret_cast<uint32_t*>(os_lds + L.tk);
    unsigned short* ts = reinterpret_cast<unsigned short*>(os_lds + L.ts);
    unsigned short* iF = reinterpret_cast<unsigned short*>(os_lds + L.ifin);
    uint32_t* l_base = reinterpret_cast<uint32_t*>(os_lds + L.tbase);
    uint32_t* l_cmin = reinterpret_cast<uint32_t*>(os_lds + L.tcmin);
    unsigned long long* wmax = reinterpret_cast<unsigned long long*>(os_lds + L.wsum);
    uint32_t* wsum = reinterpret_cast<uint32_t*>(wmax);
    uint32_t* l_pm = tk;
    __shared__ int l_tile;
    __shared__ uint32_t l_lo;
    __shared__ unsigned long long s_carry;
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    if (tid == 0) {
        const int t = (int)atomicAdd(&meta->ticket, 1u);
        l_tile = t;
        const unsigned long long M = meta->M;
        l_lo = (uint32_t)((((unsigned long long)((unsigned)t << ms) << 32) + M - 1ull) / M);   // smallest linear key of bucket t << ms (merged buckets: the first of them)
    }
    v3_load_tables(meta, n_contigs + 1, l_base, l_cmin);
    for (int k = tid; k < bins; k += OS_THREADS) { bst[k] = 0u; cur[k] = 0u; }
    __syncthreads();
    const int tile = l_tile;
    const uint32_t lo = l_lo;
    const int fb0 = tile << ms, fb1 = (tile + 1) << ms;                          // the 2^ms fine buckets of this workgroup
    const uint32_t b0 = off[(int64_t)fb0 * nchunks];
    const uint32_t b1 = fb1 < V3_BUCKETS ? off[(int64_t)fb1 * nchunks] : (uint32_t)n;
    const int nb = (int)(b1 - b0);                                              // <= cap: the host sized cap from the largest bucket
    const int wbits = (int)meta->wbits + ms;                                     // merged buckets: 2^ms key ranges side by side
    const int bshift = wbits > bin_bits ? wbits - bin_bits : 0;                 // bin = key >> bshift < bins
    const int4* __restrict__ grec = rec + (int64_t)b0;                          // the bucket's rows in input order (slot p)
    auto row_at = [&](int p) -> int4 { if constexpr (STAGE) return l_rec[p]; else return grec[p]; };

    // rows of the bucket -> LDS (slot p = input order inside the bucket: item j of thread t = j * 1024 + t), bin counts, bucket maximum
    uint32_t kv[V3_ITEMS];
    unsigned long long cmx = 0ull;
#pragma unroll
    for (int j = 0; j < V3_ITEMS; ++j) {
        const int p = j * OS_THREADS + tid;
        kv[j] = 0u;
        if (p < nb) {
            const int4 r = grec[p];
            if constexpr (STAGE) l_rec[p] = r;
            kv[j] = l_base[r.w] + (flip(r.x) - l_cmin[r.w]) - lo;
            atomicAdd(&bst[kv[j] >> bshift], 1u);
            const unsigned long long c = ((unsigned long long)(uint32_t)r.w << 32) | (unsigned long long)flip(r.y);
            cmx = c > cmx ? c : cmx;
        }
    }
    // the bucket's (contig, end) maximum does not depend on the order of its rows: it is published BEFORE the sort, so that by the
    // time a later bucket looks back every running predecessor has at least its aggregate out
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(cmx, d, kWave); cmx = o > cmx ? o : cmx; }
    if (lane == 0) wmax[w] = cmx;
    __syncthreads();                                                            // (A) counts + wavefront maxima complete
    if (tid == 0) {
        unsigned long long tmax = 0;
#pragma unroll
        for (int k = 0; k < OS_WAVES; ++k) tmax = wmax[k] > tmax ? wmax[k] : tmax;
        os_st64(status64 + tile, ((tile == 0 ? 2ull : 1ull) << 62) | tmax);
    }
    // exclusive scan of the bin counts: thread t owns the bins 4 t .. 4 t + 3 (threads beyond the bins hold zeros)
    {
        const bool own = 4 * tid < bins;
        uint4 c4 = make_uint4(0u, 0u, 0u, 0u);
        if (own) c4 = *reinterpret_cast<const uint4*>(bst + 4 * tid);
        const uint32_t s = c4.x + c4.y + c4.z + c4.w;
        __syncthreads();                                                        // (wmax read by thread 0 before wsum is reused)
        uint32_t tsum;
        uint32_t pre = sl_block_exclusive_sum<uint32_t>(s, wsum, &tsum);
        uint4 o4;
        o4.x = pre; pre += c4.x; o4.y = pre; pre += c4.y; o4.z = pre; pre += c4.z; o4.w = pre;
        if (own) *reinterpret_cast<uint4*>(bst + 4 * tid) = o4;
        if (tid == 0) bst[bins] = (uint32_t)nb;
    }
    __syncthreads();                                                            // (B) bin starts
#pragma unroll
    for (int j = 0; j < V3_ITEMS; ++j) {
        const int p = j * OS_THREADS + tid;
        if (p < nb) {
            const uint32_t bin = kv[j] >> bshift;
            const uint32_t q = bst[bin] + atomicAdd(&cur[bin], 1u);             // arrival order inside the bin: settled by the ranking below
            tk[q] = kv[j];
            ts[q] = (unsigned short)p;
        }
    }
    __syncthreads();                                                            // (C) rows in bin order
    // rank inside the bin by (key, slot): equal keys keep their input order whatever the arrival order was
#pragma unroll
    for (int j = 0; j < V3_ITEMS; ++j) {
        const int p = j * OS_THREADS + tid;
        if (p < nb) {
            const uint32_t bin = kv[j] >> bshift;
            const uint32_t a = bst[bin], z = bst[bin + 1];
            uint32_t r = a;
            for (uint32_t q = a; q < z; ++q) {
                const uint32_t k2 = tk[q];
                const uint32_t s2 = ts[q];
                r += (k2 < kv[j] || (k2 == kv[j] && s2 < (uint32_t)p)) ? 1u : 0u;
            }
            iF[r] = (unsigned short)p;
        }
    }
    __syncthreads();                                                            // (D) iF complete; tk is free (l_pm)

    // prefix max of (contig, end) in sorted order: thread t holds the positions 4 t .. 4 t + 3
    unsigned long long comp[V3_ITEMS];
    int32_t cc[V3_ITEMS];
    unsigned long long run = 0;
#pragma unroll
    for (int k = 0; k < V3_ITEMS; ++k) {
        const int q = V3_ITEMS * tid + k;
        unsigned long long c = 0ull;
        cc[k] = 0;
        if (q < nb) {
            const int4 r = row_at(iF[q]);
            c = ((unsigned long long)(uint32_t)r.w << 32) | (unsigned long long)flip(r.y);
            cc[k] = r.w;
        }
        run = c > run ? c : run;
        comp[k] = run;
    }
    int32_t prev_c = -1;
    if (tid > 0 && V3_ITEMS * tid < nb) prev_c = row_at(iF[V3_ITEMS * tid - 1]).w;
    unsigned long long inc = run;
#pragma unroll
    for (int dd = 1; dd < kWave; dd <<= 1) {
        const unsigned long long o = __shfl_up(inc, dd, kWave);
        if (lane >= dd) inc = o > inc ? o : inc;
    }
    if (lane == kWave - 1) wmax[w] = inc;
    // wavefront 0: look-back over the earlier buckets, 64 status words per step (one thread walking 200 concurrently running
    // predecessors pays a memory round trip per bucket)
    if (w == 0) {
        const unsigned long long VAL = (1ull << 62) - 1ull;
        unsigned long long carry = 0;
        for (int t0 = tile - 1; t0 >= 0; t0 -= kWave) {
            const int t = t0 - lane;
            unsigned long long v = 2ull << 62;                                  // before bucket 0: a complete prefix of nothing
            if (t >= 0) {
                v = os_ld64(status64 + t);
                while ((v >> 62) == 0) { __builtin_amdgcn_s_sleep(2); v = os_ld64(status64 + t); }
            }
            const unsigned long long pre_mask = __ballot((v >> 62) == 2ull);
            const int first = pre_mask ? __builtin_ctzll(pre_mask) : kWave;     // nearest predecessor that carries a complete prefix
            unsigned long long x = lane <= first ? (v & VAL) : 0ull;
#pragma unroll
            for (int d = kWave / 2; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(x, d, kWave); x = o > x ? o : x; }
            carry = x > carry ? x : carry;
            if (pre_mask) break;
        }
        if (lane == 0) s_carry = carry;
    }
    __syncthreads();                                                            // (E) wavefront maxima + carry
    unsigned long long wpre = 0, tmax = 0;
#pragma unroll
    for (int k = 0; k < OS_WAVES; ++k) { const unsigned long long x = wmax[k]; if (k < w) wpre = x > wpre ? x : wpre; tmax = x > tmax ? x : tmax; }
    unsigned long long excl = __shfl_up(inc, 1, kWave);
    if (lane == 0) excl = 0;
    excl = wpre > excl ? wpre : excl;
    if (tid == 0) {
        if (tile > 0) os_st64(status64 + tile, (2ull << 62) | (s_carry > tmax ? s_carry : tmax));
        if (tile == 0) flags[0] = (int32_t)meta->inverted;
    }
    const unsigned long long before = s_carry > excl ? s_carry : excl;
    // the row before this bucket's first one carries the largest contig key so far: the high word of the carried composite
    if (tid == 0) prev_c = b0 == 0u ? -1 : (int32_t)(s_carry >> 32);
#pragma unroll
    for (int k = 0; k < V3_ITEMS; ++k) {
        const int q = V3_ITEMS * tid + k;
        if (q < nb) {
            const unsigned long long pm = before > comp[k] ? before : comp[k];
            l_pm[q] = (uint32_t)pm;
            // segment offsets: seg[key] = first position whose contig key is >= key (keys 0 .. n_contigs; n_contigs + 1 = n)
            const int64_t p = (int64_t)b0 + q;
            for (int32_t kk = prev_c + 1; kk <= cc[k]; ++kk) seg[kk] = (int32_t)p;
            prev_c = cc[k];
            if (p == n - 1) for (int32_t kk = cc[k] + 1; kk <= n_contigs + 1; ++kk) seg[kk] = (int32_t)n;
        }
    }
    __syncthreads();                                                            // (F) prefix maxima per position
    // index arrays, coalesced: position q = j * 1024 + tid
#pragma unroll
    for (int j = 0; j < V3_ITEMS; ++j) {
        const int q = j * OS_THREADS + tid;
        if (q < nb) {
            const int4 r = row_at(iF[q]);
            const int64_t g = (int64_t)b0 + q;
            b_start[g] = r.x;
            ep[g] = make_int2(r.y, unflip(l_pm[q]));
            b_row[g] = r.z;
            b_contig[g] = r.w;
        }
    }
}

}  // namespace ivj
