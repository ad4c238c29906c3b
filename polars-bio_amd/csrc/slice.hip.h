// slice.hip.h -- pb.overlap on LDS-resident index slices (the round-2 hot path).
//
// The 256-bucket window-scan kernels (overlap.hip.h) are bound by the vector L1's miss queue: ~5 L1-missing 64-byte
// lines per probe (table record, window rows, build rows) at ~64 outstanding requests per CU.  This path removes the
// per-probe gathers from global memory altogether:
//
//   * the sorted build side is cut into `nb` SLICES of R consecutive rows (equal ROW COUNT, so a slice always fits
//     the LDS whatever the genomic density: clustered real data cannot overflow it);
//   * one stable, LDS-staged partition pass (k_slice_hist / k_slice_scatter) sends every probe to the slice that holds
//     its hi-bound -- the bucket of a probe is the number of slice boundaries ("splitters", 8 bytes each, in LDS)
//     below its (contig, end) key -- and writes ONE 16-byte record {start, end, row, contig} per probe, so a bucket
//     run of four probes is a full 64-byte line even at > 1000 buckets;
//   * the join kernel (k_slice_join) gives a workgroup one bucket chunk: it loads the slice (start / (end, pmax) / row
//     of R rows, coalesced) into LDS ONCE and streams tiles of 4096 probe records through it.  The hi-bound is a
//     fixed-trip-count bound search in LDS, the window scan and the build rows of the matches come from LDS; the only
//     global accesses are the coalesced record stream, the coalesced slice load and the coalesced result stream.
//     Windows that reach below the slice (rare: a few rows per slice boundary) read the global arrays.
//
// Count / fill / fused are ONE kernel template: COUNT writes tile totals, FILL emits at scanned tile bases
// (deterministic), FUSED reserves the tile's output range with one atomic (single pass, tile order not reproducible).
#pragma once
#include "index_view.hip.h"

namespace ivj {

constexpr int SL_THREADS = 1024;
constexpr int SL_WAVES = SL_THREADS / kWave;
constexpr int SL_ITEMS = 4;
constexpr int SL_TILE = SL_THREADS * SL_ITEMS;       // probes per tile of the partition and of the join
constexpr int SL_MAX_BUCKETS = 1536;                  // slices (+ 1 bucket for probes without any candidate row)
constexpr int SL_MAX_ROWS = 5120;                     // rows per slice (16 bytes of LDS each)
constexpr int SL_LDS_CONTIGS = 1022;                  // segment offsets are staged in LDS up to this many contigs
constexpr int SL_TAB_CONTIGS = 64;                    // the direct-address bucket table serves dictionaries up to this size
constexpr int SL_WIN = 12;                            // rows below hi examined by the branch-free window code

struct SliceGeom {
    int nb;            // number of slices = buckets 0 .. nb-1; bucket nb = probes that cannot match
    int R;             // rows per slice (multiple of 64); the last slice may be shorter
    int nbits;         // bits of a bucket id (match-any ranking)
    int p2;            // largest power of two <= nb   (fixed-trip-count bucket search)
    int p2r;           // largest power of two <= R    (fixed-trip-count hi-bound search)
    int ncells;        // cells of the direct-address bucket table (0: none, plain bound search over the splitters)
    int cps;           // cells per splitter the table was sized with
};

// Direct-address table over the splitters: per contig a uniform grid over the starts of its rows, about four cells
// per splitter; a cell holds the range [lo, hi] of "number of splitters below" values its keys can have, so the
// bucket of a probe is one 16-byte contig record + one 4-byte cell read + (expected) one splitter compare instead
// of an 11-step bound search of dependent LDS reads.
struct SliceTab {
    const int4* cm;          // per contig {ulo, uhi, shift, first cell}
    const uint32_t* cell;    // per cell lo | hi << 16
};

// ---- workgroup scan over SL_THREADS threads ---------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T sl_block_exclusive_sum(T v, T* wsum /* SL_WAVES */, T* total) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const T inc = wave_inclusive_scan(v, SumOp());
    if (lane == kWave - 1) wsum[w] = inc;
    __syncthreads();
    T pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SL_WAVES; ++k) { const T x = wsum[k]; if (k < w) pre += x; tot += x; }
    __syncthreads();
    *total = tot;
    return pre + inc - v;
}

// One-barrier variant for per-thread values whose WAVEFRONT sums fit 32 bits (the slice join: a probe matches at most
// nbuild <= SL_MAX_BUCKETS * SL_MAX_ROWS rows, times 128 probes per wavefront < 2^31); the cross-wavefront part is 64-bit.
// The caller alternates between two partial arrays (`wsum` = parity-selected SL_WAVES ints, 16-byte aligned), so the
// next scan may start before the slowest wavefront has read this one's partials.
template <int WAVES = SL_WAVES>
__device__ __forceinline__ long long sl_block_exclusive_sum_i32(int v, int* wsum, long long* total) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const int inc = wave_inclusive_scan(v, SumOp());
    if (lane == kWave - 1) wsum[w] = inc;
    __syncthreads();
    long long pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < WAVES; k += 4) {
        const int4 x = *reinterpret_cast<const int4*>(wsum + k);
        pre += (long long)(k < w ? x.x : 0) + (k + 1 < w ? x.y : 0) + (k + 2 < w ? x.z : 0) + (k + 3 < w ? x.w : 0);
        tot += (long long)x.x + x.y + x.z + x.w;
    }
    *total = tot;
    return pre + (inc - v);
}

// ---- splitters ---------------------------------------------------------------------------------------------------
// spl[j] = composite key (contig, start) of sorted row j * R: the first row of slice j.
__global__ void k_slice_splitters(const int32_t* __restrict__ b_contig, const int32_t* __restrict__ b_start, int64_t n, int R, int nb,
                                  unsigned long long* __restrict__ spl) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nb) return;
    const int64_t p = (int64_t)j * R;
    spl[j] = p < n ? (((unsigned long long)(uint32_t)b_contig[p] << 32) | (unsigned long long)flip(b_start[p])) : ~0ull;
}

// Bucket of a probe: hi = number of build rows whose (contig, start) key lies below the probe's (contig, end) key
// [+1 for Weak: start <= end counts]; rows are sorted by that key, so row p is below the key iff p < hi, and the
// number of splitters (rows 0, R, 2R, ...) below the key is ceil(hi / R): hi lies in (kR, (k+1)R] for bucket
// k = ceil(hi / R) - 1.  hi = 0 (k = -1) or a contig outside the dictionary: no candidate row at all -> bucket nb.
template <bool STRICT>
__device__ __forceinline__ uint32_t slice_bucket(const unsigned long long* __restrict__ l_spl, const SliceGeom& g, int32_t n_contigs,
                                                 int32_t c, int32_t qe) {
    const unsigned long long key = (((unsigned long long)(uint32_t)c << 32) | (unsigned long long)flip(qe)) + (STRICT ? 0ull : 1ull);
    int pos = 0;                                             // number of splitters < key
    for (int step = g.p2; step > 0; step >>= 1) {
        const int t = pos + step;
        if (t <= g.nb && l_spl[t - 1] < key) pos = t;
    }
    return ((uint32_t)c >= (uint32_t)n_contigs || pos == 0) ? (uint32_t)g.nb : (uint32_t)(pos - 1);
}

template <bool STRICT>
__device__ __forceinline__ uint32_t slice_bucket_tab(const unsigned long long* __restrict__ l_spl, const int4* __restrict__ l_cm,
                                                     const uint32_t* __restrict__ l_cell, const SliceGeom& g, int32_t n_contigs, int32_t c,
                                                     int32_t qe) {
    if ((uint32_t)c >= (uint32_t)n_contigs) return (uint32_t)g.nb;
    const unsigned long long tu = (unsigned long long)flip(qe) + (STRICT ? 0ull : 1ull);
    const unsigned long long key = ((unsigned long long)(uint32_t)c << 32) + tu;        // (c, INT32_MAX) + 1 carries into the contig
    const int4 m = l_cm[c];
    const uint32_t ulo = (uint32_t)m.x, uhi = (uint32_t)m.y;
    uint32_t k;
    if (tu <= ulo) k = 0;
    else if (tu > uhi) k = (uhi - ulo) >> m.z;
    else k = ((uint32_t)tu - ulo) >> m.z;
    const uint32_t lh = l_cell[m.w + k];
    int pos = (int)(lh & 0xffffu);
    const int hi = (int)(lh >> 16);
    while (pos < hi && l_spl[pos] < key) ++pos;
    return pos == 0 ? (uint32_t)g.nb : (uint32_t)(pos - 1);
}

// One workgroup: per-contig grid metadata and the cells.  nspl_c = splitters inside contig c = ceil(seg[c+1] / R) -
// ceil(seg[c] / R); cells_c = max(2, cps * nspl_c).
__global__ __launch_bounds__(SL_THREADS) void k_slice_tab(const unsigned long long* __restrict__ spl, SliceGeom g, int cps,
                                                         const int32_t* __restrict__ seg, const int32_t* __restrict__ b_start,
                                                         int32_t n_contigs, int4* __restrict__ cm, uint32_t* __restrict__ cell) {
    __shared__ int4 l_cm[SL_TAB_CONTIGS];
    __shared__ int l_j[2 * SL_TAB_CONTIGS];
    __shared__ int l_total;
    if (threadIdx.x == 0) {
        int tb = 0;
        for (int c = 0; c < n_contigs; ++c) {
            const int a = seg[c], b = seg[c + 1];
            const int jlo = (a + g.R - 1) / g.R, jhi = (b + g.R - 1) / g.R;
            int nc = cps * (jhi - jlo);
            if (nc < 2) nc = 2;                                  // two cells keep shift <= 31 for any int32 span (a shift by 32 is undefined)
            uint32_t ulo = 0, uhi = 0;
            int shift = 0;
            if (b > a) {
                ulo = flip(b_start[a]); uhi = flip(b_start[b - 1]);
                while (shift < 31 && ((unsigned long long)(uhi - ulo) >> shift) + 1ull > (unsigned long long)nc) ++shift;
            }
            l_cm[c] = make_int4((int)ulo, (int)uhi, shift, tb);
            l_j[2 * c] = jlo; l_j[2 * c + 1] = jhi;
            tb += nc;
        }
        l_total = tb;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n_contigs; c += SL_THREADS) cm[c] = l_cm[c];
    const int total = l_total;
    for (int i = threadIdx.x; i < total; i += SL_THREADS) {
        int lo = 0, hi = n_contigs;                                   // last contig whose first cell is <= i
        while (lo < hi) { const int m = (lo + hi) >> 1; if (l_cm[m].w <= i) lo = m + 1; else hi = m; }
        const int c = lo - 1;
        const int4 m = l_cm[c];
        const int k = i - m.w;
        const int jlo = l_j[2 * c], jhi = l_j[2 * c + 1];
        const int last = (c + 1 < n_contigs ? l_cm[c + 1].w : total) - 1;
        auto below = [&](int kk) {                                   // splitters of contig c below the lower edge of cell kk
            if (kk <= 0) return jlo;
            const unsigned long long edge = ((unsigned long long)(uint32_t)c << 32) + (unsigned long long)(uint32_t)m.x +
                                            ((unsigned long long)kk << m.z);
            int a = jlo, b = jhi;
            while (a < b) { const int mid = (a + b) >> 1; if (spl[mid] < edge) a = mid + 1; else b = mid; }
            return a;
        };
        const int l = below(k);
        const int h = i == last ? jhi : below(k + 1);
        cell[i] = (uint32_t)l | ((uint32_t)h << 16);
    }
}

// lanes of this wavefront that are valid and hold the same bucket id (nbits <= 11)
__device__ __forceinline__ uint64_t wave_match_n(uint32_t d, bool valid, int nbits) {
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// ---- partition, pass 1: per-chunk bucket histogram -----------------------------------------------------------------
// Workgroup g owns the probes [g * chunk, (g + 1) * chunk); blk_hist is bucket-major: blk_hist[b * nchunks + g].
template <bool STRICT>
__global__ __launch_bounds__(SL_THREADS) void k_slice_hist(const unsigned long long* __restrict__ spl, SliceTab tab, SliceGeom g, int32_t n_contigs,
                                                          const int32_t* __restrict__ pc, const int32_t* __restrict__ pe, int64_t n,
                                                          int chunk, int nchunks, bool vec_ok, uint32_t* __restrict__ blk_hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sl_lds[];
    // dynamic LDS: cm[SL_TAB_CONTIGS] | spl[nb] | cells[ncells] | hist[nb + 1]
    int4* l_cm = reinterpret_cast<int4*>(sl_lds);
    unsigned long long* l_spl = reinterpret_cast<unsigned long long*>(l_cm + SL_TAB_CONTIGS);
    uint32_t* l_cell = reinterpret_cast<uint32_t*>(l_spl + g.nb);
    uint32_t* h = l_cell + g.ncells;
    for (int k = threadIdx.x; k < g.nb; k += SL_THREADS) l_spl[k] = spl[k];
    for (int k = threadIdx.x; k < g.ncells; k += SL_THREADS) l_cell[k] = tab.cell[k];
    if (g.ncells) for (int k = threadIdx.x; k < n_contigs; k += SL_THREADS) l_cm[k] = tab.cm[k];
    for (int k = threadIdx.x; k <= g.nb; k += SL_THREADS) h[k] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * chunk;
    const int64_t end = base + chunk < n ? base + chunk : n;
    for (int64_t i0 = base + (int64_t)threadIdx.x * 4; i0 < end; i0 += (int64_t)SL_THREADS * 4) {
        int32_t c[4], e[4];
        load_items_nt(pc, i0, end, vec_ok, -1, c);
        load_items_nt(pe, i0, end, vec_ok, 0, e);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < end)
                atomicAdd(&h[g.ncells ? slice_bucket_tab<STRICT>(l_spl, l_cm, l_cell, g, n_contigs, c[k], e[k])
                                      : slice_bucket<STRICT>(l_spl, g, n_contigs, c[k], e[k])], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= g.nb; k += SL_THREADS) blk_hist[(int64_t)k * nchunks + blockIdx.x] = h[k];
}

// ---- chunk table of the join ------------------------------------------------------------------------------------------
// From the scanned histogram (blk_off[b * nchunks] = first probe of bucket b): bucket starts, the number of join
// workgroups per bucket (chunks of `jchunk` probes) and the map workgroup -> (bucket, chunk inside the bucket).
// meta[0] = number of join workgroups.  One workgroup.
__global__ __launch_bounds__(SL_THREADS) void k_slice_chunks(const uint32_t* __restrict__ blk_off, int nchunks, int nb, int64_t n,
                                                            int jchunk, uint32_t* __restrict__ bstart, int32_t* __restrict__ meta,
                                                            int2* __restrict__ wg_map) {
    __shared__ int l_cpre[SL_MAX_BUCKETS + 2];
    __shared__ int wsum[SL_WAVES];
    // two buckets per thread (nb + 1 <= 2 * SL_THREADS)
    int cnt2[2];
    int v = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int b = 2 * threadIdx.x + k;
        cnt2[k] = 0;
        if (b <= nb) {
            const uint32_t s = blk_off[(int64_t)b * nchunks];
            const uint32_t e = b < nb ? blk_off[(int64_t)(b + 1) * nchunks] : (uint32_t)n;
            bstart[b] = s;
            if (b == nb) bstart[nb + 1] = (uint32_t)n;
            if (b < nb) cnt2[k] = (int)((e - s + (uint32_t)jchunk - 1u) / (uint32_t)jchunk);
        }
        v += cnt2[k];
    }
    int total;
    const int pre = sl_block_exclusive_sum(v, wsum, &total);
    if (2 * threadIdx.x <= nb) l_cpre[2 * threadIdx.x] = pre;
    if (2 * threadIdx.x + 1 <= nb) l_cpre[2 * threadIdx.x + 1] = pre + cnt2[0];
    if (threadIdx.x == 0) { meta[0] = total; l_cpre[nb + 1] = total; }
    __syncthreads();
    for (int w = threadIdx.x; w < total; w += SL_THREADS) {
        // last bucket b in [0, nb) with l_cpre[b] <= w  (buckets without probes repeat the prefix: take the last)
        int lo = 0, hi = nb;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (l_cpre[m + 1] <= w) lo = m + 1; else hi = m; }
        wg_map[w] = make_int2(lo, w - l_cpre[lo]);
    }
}

// ---- partition, pass 2: stable scatter of 16-byte probe records ---------------------------------------------------
// blk_off = exclusive scan of blk_hist (bucket-major).  The workgroup walks its chunk in tiles of SL_TILE probes:
// bucket ids (splitter search in LDS) -> rank inside (wavefront, bucket) with match-any ballots against the
// wavefront's PRIVATE counter row (no workgroup barrier inside the rounds) -> per-bucket prefix over the wavefronts
// + tile-local bucket starts -> the records are placed in LDS at their sorted tile-local position and copied out as
// contiguous bucket runs (consecutive threads -> consecutive records of one run: coalesced 16-byte stores).  The
// running global offset of every bucket lives in LDS across the tiles of the chunk.
struct SlicePartLds {
    // byte offsets into the dynamic LDS block
    int cm, cell, spl, base, lstart, tot, wcnt, rec, d, wsum, total;
};
__host__ __device__ inline SlicePartLds slice_part_lds(int nb, int ncells) {
    SlicePartLds L;
    const int nbp = (nb + 2 + 1) & ~1;                          // counters per wavefront row, even
    int o = 0;
    L.cm = o; o += ncells ? 16 * SL_TAB_CONTIGS : 0;
    L.spl = o; o += 8 * nb;
    L.cell = o; o += 4 * ncells;
    L.rec = (o + 15) & ~15; o = L.rec + 16 * SL_TILE;
    L.base = o; o += 4 * (nb + 2);
    L.lstart = o; o += 4 * (nb + 2);
    L.tot = o; o += 4 * (nb + 2);
    L.wcnt = (o + 3) & ~3; o = L.wcnt + 2 * nbp * SL_WAVES;
    L.d = (o + 3) & ~3; o = L.d + 2 * SL_TILE;
    L.wsum = (o + 3) & ~3; o = L.wsum + 4 * SL_WAVES;
    L.total = o;
    return L;
}

template <bool STRICT>
__global__ __launch_bounds__(SL_THREADS) void k_slice_scatter(const unsigned long long* __restrict__ spl, SliceTab tab, SliceGeom g, int32_t n_contigs,
                                                             const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                             const int32_t* __restrict__ pe, const int32_t* __restrict__ row_id, int64_t n,
                                                             int chunk, int nchunks, const uint32_t* __restrict__ blk_off,
                                                             int4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sl_lds[];
    const SlicePartLds L = slice_part_lds(g.nb, g.ncells);
    unsigned long long* l_spl = reinterpret_cast<unsigned long long*>(sl_lds + L.spl);
    int4* l_cm = reinterpret_cast<int4*>(sl_lds + L.cm);
    uint32_t* l_cell = reinterpret_cast<uint32_t*>(sl_lds + L.cell);
    int4* l_rec = reinterpret_cast<int4*>(sl_lds + L.rec);
    uint32_t* base = reinterpret_cast<uint32_t*>(sl_lds + L.base);
    uint32_t* lstart = reinterpret_cast<uint32_t*>(sl_lds + L.lstart);
    uint32_t* tot = reinterpret_cast<uint32_t*>(sl_lds + L.tot);
    unsigned short* wcnt = reinterpret_cast<unsigned short*>(sl_lds + L.wcnt);
    unsigned short* l_d = reinterpret_cast<unsigned short*>(sl_lds + L.d);
    uint32_t* wsum = reinterpret_cast<uint32_t*>(sl_lds + L.wsum);
    const int nbp = (g.nb + 2 + 1) & ~1;
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    const int nbk = g.nb + 1;                                   // buckets incl. the "no candidate" one

    for (int k = tid; k < g.nb; k += SL_THREADS) l_spl[k] = spl[k];
    for (int k = tid; k < g.ncells; k += SL_THREADS) l_cell[k] = tab.cell[k];
    if (g.ncells) for (int k = tid; k < n_contigs; k += SL_THREADS) l_cm[k] = tab.cm[k];
    for (int k = tid; k < nbk; k += SL_THREADS) { base[k] = blk_off[(int64_t)k * nchunks + blockIdx.x]; tot[k] = 0; }
    for (int k = tid; k < nbp * SL_WAVES / 2; k += SL_THREADS) reinterpret_cast<uint32_t*>(wcnt)[k] = 0;
    __syncthreads();

    const int64_t cbase = (int64_t)blockIdx.x * chunk;
    const int64_t cend = cbase + chunk < n ? cbase + chunk : n;
    const uint64_t lt = lanemask_lt();
    unsigned short* my = wcnt + w * nbp;
    // wavefront w owns the contiguous quarter-KiB [w * 256, (w + 1) * 256) of the tile: item j of lane l is tile
    // element w * 256 + j * 64 + l (every load is one contiguous 256-byte segment)
    const int el0 = w * (SL_ITEMS * kWave) + lane;

    int32_t nc[SL_ITEMS], ns[SL_ITEMS], ne[SL_ITEMS], nr[SL_ITEMS];
    auto load_tile = [&](int64_t tbase) {
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) {
            const int64_t i = tbase + el0 + j * kWave;
            const bool valid = i < cend;
            nc[j] = valid ? __builtin_nontemporal_load(pc + i) : -1;
            ns[j] = valid ? __builtin_nontemporal_load(ps + i) : 0;
            ne[j] = valid ? __builtin_nontemporal_load(pe + i) : 0;
            nr[j] = valid ? (row_id ? __builtin_nontemporal_load(row_id + i) : (int32_t)i) : -1;
        }
    };
    load_tile(cbase);
    for (int64_t tbase = cbase; tbase < cend; tbase += SL_TILE) {
        int32_t c[SL_ITEMS], s[SL_ITEMS], e[SL_ITEMS], r[SL_ITEMS];
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) { c[j] = nc[j]; s[j] = ns[j]; e[j] = ne[j]; r[j] = nr[j]; }
        if (tbase + SL_TILE < cend) load_tile(tbase + SL_TILE);        // next tile's columns are in flight during this one
        const int tile_n = (int)((cend - tbase) < (int64_t)SL_TILE ? (cend - tbase) : (int64_t)SL_TILE);
        uint32_t d[SL_ITEMS], rank[SL_ITEMS];
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) {
            const bool valid = el0 + j * kWave < tile_n;
            d[j] = !valid ? 0u : (g.ncells ? slice_bucket_tab<STRICT>(l_spl, l_cm, l_cell, g, n_contigs, c[j], e[j])
                                           : slice_bucket<STRICT>(l_spl, g, n_contigs, c[j], e[j]));
        }
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) {
            const bool valid = el0 + j * kWave < tile_n;
            const uint64_t peers = wave_match_n(d[j], valid, g.nbits);
            const uint32_t rk = (uint32_t)__popcll(peers & lt);
            const uint32_t before = valid ? (uint32_t)my[d[j]] : 0u;
            rank[j] = before + rk;
            __builtin_amdgcn_wave_barrier();
            if (valid && rk == 0) my[d[j]] = (unsigned short)(before + (uint32_t)__popcll(peers));
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                                                        // (A) all wavefront rows counted
        // thread t owns buckets 2t, 2t+1 (one 32-bit word of every wavefront row): advance the global offsets by
        // the previous tile's totals, exclusive prefix over the wavefronts, tile totals, tile-local bucket starts
        uint32_t x0 = 0, x1 = 0;
        {
            const int b0 = 2 * tid;
            if (b0 < nbk) {
                base[b0] += tot[b0];
                if (b0 + 1 < nbk) base[b0 + 1] += tot[b0 + 1];
                uint32_t* row32 = reinterpret_cast<uint32_t*>(wcnt) + tid;
#pragma unroll
                for (int k = 0; k < SL_WAVES; ++k) {
                    const uint32_t v = row32[k * (nbp / 2)];
                    row32[k * (nbp / 2)] = x0 | (x1 << 16);
                    x0 += v & 0xffffu; x1 += v >> 16;
                }
                tot[b0] = x0;
                if (b0 + 1 < nbk) tot[b0 + 1] = x1;
            }
        }
        uint32_t tsum;
        const uint32_t pre = sl_block_exclusive_sum(x0 + x1, wsum, &tsum);      // (B), (C)
        if (2 * tid < nbk) { lstart[2 * tid] = pre; if (2 * tid + 1 < nbk) lstart[2 * tid + 1] = pre + x0; }
        __syncthreads();                                                        // (D) lstart / prefixes visible
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) {
            if (el0 + j * kWave < tile_n) {
                const uint32_t pos = lstart[d[j]] + (uint32_t)my[d[j]] + rank[j];
                l_rec[pos] = make_int4(s[j], e[j], r[j], c[j]);
                l_d[pos] = (unsigned short)d[j];
            }
        }
        __syncthreads();                                                        // (E) tile sorted in LDS
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) {
            const int il = j * SL_THREADS + tid;
            if (il < tile_n) {
                const uint32_t dd = l_d[il];
                out[(int64_t)base[dd] + ((uint32_t)il - lstart[dd])] = l_rec[il];
            }
        }
        for (int k = tid; k < nbp * SL_WAVES / 2; k += SL_THREADS) reinterpret_cast<uint32_t*>(wcnt)[k] = 0;
        __syncthreads();                                                        // (F) counters clear, staging free
    }
}

// ---- partition, pass 2, unordered variant (fused single-pass join only) ---------------------------------------------
// Same tiles, same output layout, but the rank of a record inside (tile, bucket) comes from ONE returning LDS atomic instead
// of match-any ballots + per-wavefront counter rows + a prefix over the wavefronts: less than half the instructions.  The
// order of the records inside a bucket then depends on the timing of the wavefronts, i.e. is not reproducible from run to
// run -- which the fused join's output is not either (its tiles land in reservation order); the pairs of one probe row stay
// contiguous because a probe is one record.  The count -> fill pair keeps the stable scatter above.
// (Writing the records straight from registers to their global slot -- no staging, two barriers per tile, 32 KB of LDS -- was
// measured as well: 1.38 ms against 0.99 ms for config 3; scattered 16-byte stores cost more than the staging they save.)
struct SlicePartULds { int cm, cell, spl, base, lstart, delta, cnt, rec, d, wsum, total; };
__host__ __device__ inline SlicePartULds slice_part_u_lds(int nb, int ncells, int threads) {
    const int tile = threads * SL_ITEMS;
    SlicePartULds L;
    int o = 0;
    L.cm = o; o += ncells ? 16 * SL_TAB_CONTIGS : 0;
    L.spl = o; o += 8 * nb;
    L.cell = o; o += 4 * ncells;
    L.rec = (o + 15) & ~15; o = L.rec + 16 * tile;
    L.base = o; o += 4 * (nb + 2);
    L.lstart = o; o += 4 * (nb + 2);
    L.delta = o; o += 4 * (nb + 2);
    L.cnt = o; o += 4 * (nb + 2);
    L.d = (o + 3) & ~3; o = L.d + 2 * tile;
    L.wsum = (o + 15) & ~15; o = L.wsum + 4 * 2 * SL_WAVES;
    L.total = o;
    return L;
}

template <bool STRICT, int THREADS>
__global__ __launch_bounds__(THREADS) void k_slice_scatter_u(const unsigned long long* __restrict__ spl, SliceTab tab, SliceGeom g, int32_t n_contigs,
                                                               const int32_t* __restrict__ pc, const int32_t* __restrict__ ps,
                                                               const int32_t* __restrict__ pe, const int32_t* __restrict__ row_id, int64_t n,
                                                               int chunk, int nchunks, const uint32_t* __restrict__ blk_off,
                                                               int4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sl_lds[];
    constexpr int TILE = THREADS * SL_ITEMS;
    const SlicePartULds L = slice_part_u_lds(g.nb, g.ncells, THREADS);
    unsigned long long* l_spl = reinterpret_cast<unsigned long long*>(sl_lds + L.spl);
    int4* l_cm = reinterpret_cast<int4*>(sl_lds + L.cm);
    uint32_t* l_cell = reinterpret_cast<uint32_t*>(sl_lds + L.cell);
    int4* l_rec = reinterpret_cast<int4*>(sl_lds + L.rec);
    uint32_t* base = reinterpret_cast<uint32_t*>(sl_lds + L.base);
    uint32_t* lstart = reinterpret_cast<uint32_t*>(sl_lds + L.lstart);
    uint32_t* delta = reinterpret_cast<uint32_t*>(sl_lds + L.delta);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(sl_lds + L.cnt);
    unsigned short* l_d = reinterpret_cast<unsigned short*>(sl_lds + L.d);
    int* wsum = reinterpret_cast<int*>(sl_lds + L.wsum);
    const int tid = threadIdx.x;
    const int nbk = g.nb + 1;
    for (int k = tid; k < g.nb; k += THREADS) l_spl[k] = spl[k];
    for (int k = tid; k < g.ncells; k += THREADS) l_cell[k] = tab.cell[k];
    if (g.ncells) for (int k = tid; k < n_contigs; k += THREADS) l_cm[k] = tab.cm[k];
    for (int k = tid; k < nbk + 1; k += THREADS) { base[k] = k < nbk ? blk_off[(int64_t)k * nchunks + blockIdx.x] : 0u; cnt[k] = 0; }
    __syncthreads();
    const int64_t cbase = (int64_t)blockIdx.x * chunk;
    const int64_t cend = cbase + chunk < n ? cbase + chunk : n;
    int32_t nc[SL_ITEMS], ns[SL_ITEMS], ne[SL_ITEMS], nr[SL_ITEMS];
    auto load_tile = [&](int64_t tbase) {
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) {
            const int64_t i = tbase + j * THREADS + tid;
            const bool valid = i < cend;
            nc[j] = valid ? __builtin_nontemporal_load(pc + i) : -1;
            ns[j] = valid ? __builtin_nontemporal_load(ps + i) : 0;
            ne[j] = valid ? __builtin_nontemporal_load(pe + i) : 0;
            nr[j] = valid ? (row_id ? __builtin_nontemporal_load(row_id + i) : (int32_t)i) : -1;
        }
    };
    load_tile(cbase);
    int tix = 0;
    for (int64_t tbase = cbase; tbase < cend; tbase += TILE, ++tix) {
        int32_t c[SL_ITEMS], s[SL_ITEMS], e[SL_ITEMS], r[SL_ITEMS];
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) { c[j] = nc[j]; s[j] = ns[j]; e[j] = ne[j]; r[j] = nr[j]; }
        if (tbase + TILE < cend) load_tile(tbase + TILE);
        const int tile_n = (int)((cend - tbase) < (int64_t)TILE ? (cend - tbase) : (int64_t)TILE);
        uint32_t d[SL_ITEMS], rank[SL_ITEMS];
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) {
            const bool valid = j * THREADS + tid < tile_n;
            d[j] = !valid ? 0u : (g.ncells ? slice_bucket_tab<STRICT>(l_spl, l_cm, l_cell, g, n_contigs, c[j], e[j])
                                           : slice_bucket<STRICT>(l_spl, g, n_contigs, c[j], e[j]));
            rank[j] = valid ? atomicAdd(&cnt[d[j]], 1u) : 0u;
        }
        __syncthreads();                                                        // (A) bucket counts of the tile complete
        // thread t owns buckets 2t, 2t+1: advance the global offsets by the previous tile's totals (kept in `lstart` deltas),
        // tile-local starts, copy-out deltas; the counters are cleared for the next tile
        constexpr int OWN = (SL_MAX_BUCKETS + 1 + THREADS - 1) / THREADS;      // consecutive buckets per thread
        int x[OWN];
        int xs = 0;
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
            const int b = OWN * tid + q;
            x[q] = 0;
            if (b < nbk) { x[q] = (int)cnt[b]; cnt[b] = 0; }
            xs += x[q];
        }
        long long tsum;
        int pre = (int)sl_block_exclusive_sum_i32<THREADS / kWave>(xs, wsum + (tix & 1) * SL_WAVES, &tsum);      // (B)
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
            const int b = OWN * tid + q;
            if (b < nbk) { lstart[b] = (uint32_t)pre; delta[b] = base[b] - (uint32_t)pre; base[b] += (uint32_t)x[q]; }
            pre += x[q];
        }
        __syncthreads();                                                        // (C)
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) {
            if (j * THREADS + tid < tile_n) {
                const uint32_t pos = lstart[d[j]] + rank[j];
                l_rec[pos] = make_int4(s[j], e[j], r[j], c[j]);
                l_d[pos] = (unsigned short)d[j];
            }
        }
        __syncthreads();                                                        // (D) tile sorted in LDS
#pragma unroll
        for (int j = 0; j < SL_ITEMS; ++j) {
            const int il = j * THREADS + tid;
            if (il < tile_n) out[(int64_t)((uint32_t)il + delta[l_d[il]])] = l_rec[il];
        }
        // no barrier here: the next tile's barrier (A) separates this copy-out from the next placement
    }
}

// ---- the join ---------------------------------------------------------------------------------------------------------
enum { SL_COUNT = 0, SL_FILL = 1, SL_FUSED = 2 };

// Row p of the sorted build side: from the slice in LDS, or -- a window reaching below the slice, rare -- from the
// global array.  The LDS read is unconditional (clamped index) and the global read sits in its own branch: a
// `cond ? lds[..] : global[..]` select of two address spaces trips hipcc 7.2 ("Operand has incorrect register class").
__device__ __forceinline__ int2 slice_ep(const int32_t* l_end, const int32_t* l_pmax, const int2* __restrict__ g_ep, int p, int r0) {
    const int i = p - r0;
    int2 v = make_int2(l_end[i < 0 ? 0 : i], l_pmax[i < 0 ? 0 : i]);
    if (i < 0) v = g_ep[p];
    return v;
}
__device__ __forceinline__ int32_t slice_row(const int32_t* l_row, const int32_t* __restrict__ g_row, int p, int r0) {
    const int i = p - r0;
    int32_t v = l_row[i < 0 ? 0 : i];
    if (i < 0) v = g_row[p];
    return v;
}

// The bounded waits of the fused tile protocol.  A protocol bug must neither hang the box nor pass unseen: a wait that runs out
// raises bit 1 of the call's state word (bit 0: the capacity was exceeded) and the host reports IVJ_EHIP "tile protocol timeout"
// instead of pairs that may be wrong.  IVJ_SLICE_ABLATE bit 4096 (test knob): workgroup 0 never publishes the base of its first
// tile and the bound drops to SL_SPIN_BOUND_TEST, so the failure path runs in milliseconds.
constexpr int SL_SPIN_BOUND = 1 << 24;
constexpr int SL_SPIN_BOUND_TEST = 1 << 8;
constexpr int SL_ABLATE_TILE_FAULT = 4096;
#define IVJ_TILE_WAIT(STILL_WAITING, BOUND, STATE, LANE, ON_TIMEOUT)                                                            \
    {                                                                                                                           \
        int spin_ = 0;                                                                                                          \
        for (; (STILL_WAITING) && spin_ < (BOUND); ++spin_) __builtin_amdgcn_s_sleep(1);                                        \
        if (__builtin_expect(spin_ >= (BOUND), 0)) {                                                                            \
            if (STILL_WAITING) {                                                                                                \
                if ((LANE) == 0) atomicOr((STATE) + 1, 2ull);                                                                   \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* (the flag is out before this wavefront may end) */         \
                ON_TIMEOUT;                                                                                                     \
            }                                                                                                                   \
        }                                                                                                                       \
    }

// host words (host_core.hip.h): a value the host reads after the kernel, stored with system scope into pinned host-coherent memory
__device__ __forceinline__ void hw_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

struct SliceJoinArgs {
    // the build index as this kernel needs it (a slim copy: the full IndexView costs ~50 SGPRs of kernel arguments)
    const int32_t* b_start;
    const int2* ep;
    const int32_t* b_row;
    const int32_t* b_contig;
    const int32_t* seg;
    int32_t n_contigs;
    const int4* rec;                  // bucket-ordered probe records {start, end, row, contig}
    const uint32_t* bstart;           // nb + 2 bucket starts
    const int32_t* meta;              // [0] = number of join workgroups
    const int2* wg_map;               // workgroup -> (bucket, chunk inside the bucket)
    int jchunk;                       // probes per join workgroup (multiple of the tile)
    int stage;                        // pairs of LDS staging (multiple of SL_THREADS)
    int lds_seg;                      // 1: segment offsets staged in LDS
    int use_bins;                     // 1: direct-address table over the slice's starts (single-contig slices)
    int ablate;                       // profiling only (IVJ_SLICE_ABLATE): 1 skip the hi lookup, 2 skip the window, 4 skip the tile scan, 8 no slice load, 16 no staging writes, 32 no copy-out
    long long capacity;
    long long* tile_tot;              // COUNT: out; FILL: scanned tile bases
    unsigned long long* state;        // FUSED: [0] cursor, [1] overflow flag
    int32_t* out_probe;
    int32_t* out_build;
};

// suffix minimum over the SL_THREADS threads of the workgroup, exclusive (min over the threads AFTER this one)
__device__ __forceinline__ uint32_t sl_block_suffix_min_excl(uint32_t v, uint32_t* wmin /* SL_WAVES */) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t o = __shfl_down(x, d, kWave);
        if (lane + d < kWave) x = x < o ? x : o;
    }
    if (lane == 0) wmin[w] = x;                      // inclusive suffix min of the wavefront
    uint32_t nxt = __shfl_down(x, 1, kWave);
    if (lane == kWave - 1) nxt = 0xffffffffu;
    __syncthreads();
    uint32_t later = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < SL_WAVES; ++k) { const uint32_t y = wmin[k]; if (k > w) later = later < y ? later : y; }
    __syncthreads();
    return nxt < later ? nxt : later;
}

constexpr int SL_PAD = 16;       // ints in front of l_end: the branch-free window may read up to SL_WIN rows below the slice (ignored)

template <bool STRICT, int MODE, int ITEMS>
__global__ __launch_bounds__(SL_THREADS) void k_slice_join(SliceGeom g, int64_t nbuild, SliceJoinArgs A) {
    constexpr int TILE = SL_THREADS * ITEMS;
    static_assert(SL_WIN < SL_PAD, "front padding must cover the window");
    extern __shared__ __attribute__((aligned(16))) unsigned char sl_lds[];
    // dynamic LDS: start[R] | pad | end[R] | pmax[R] | row[R] | bins[2R + 8 (u16)] | staging | seg | scan scratch
    int32_t* l_start = reinterpret_cast<int32_t*>(sl_lds);
    int32_t* l_end = l_start + g.R + SL_PAD;
    int32_t* l_pmax = l_end + g.R;
    int32_t* l_row = l_pmax + g.R;
    unsigned short* l_bin = reinterpret_cast<unsigned short*>(l_row + g.R);
    int2* st = reinterpret_cast<int2*>(l_bin + (A.use_bins ? 2 * g.R + 8 : 0));
    int32_t* l_seg = reinterpret_cast<int32_t*>(st + (MODE == SL_COUNT ? 0 : A.stage));
    long long* wsum = reinterpret_cast<long long*>(l_seg + (A.lds_seg ? ((A.n_contigs + 2 + 3) & ~3) : 0));     // 16-byte aligned
    long long* s_base = wsum + 2 * SL_WAVES;

    // XCD-affine order: workgroup b runs on XCD b % 8 (observed); XCD x takes the contiguous eighth
    // [x * per, (x + 1) * per) of the (bucket, chunk) list, so its L2 only ever holds its own slices
    const int total_wg = A.meta[0];
    const int per = (total_wg + 7) / 8;
    const int slot = (int)(blockIdx.x >> 3);
    const int v = (int)(blockIdx.x & 7) * per + slot;
    if (slot >= per || v >= total_wg) return;                                  // uniform
    const int2 bc = A.wg_map[v];
    const int k = bc.x;
    const int64_t q0 = (int64_t)A.bstart[k] + (int64_t)bc.y * A.jchunk;
    const int64_t qend = (int64_t)A.bstart[k + 1];
    const int64_t q1 = q0 + A.jchunk < qend ? q0 + A.jchunk : qend;
    const int tid = threadIdx.x;

    // slice k = sorted rows [r0, r0 + rk)
    const int r0 = k * g.R;
    const int rk = (int)((nbuild - r0) < (int64_t)g.R ? (nbuild - r0) : (int64_t)g.R);
    for (int i = tid * 4; i < ((A.ablate & 8) ? 0 : rk); i += SL_THREADS * 4) {
        if (i + 4 <= rk) {
            *reinterpret_cast<int4*>(l_start + i) = *reinterpret_cast<const int4*>(A.b_start + r0 + i);
            *reinterpret_cast<int4*>(l_row + i) = *reinterpret_cast<const int4*>(A.b_row + r0 + i);
            const int4 e01 = *reinterpret_cast<const int4*>(A.ep + r0 + i);
            const int4 e23 = *reinterpret_cast<const int4*>(A.ep + r0 + i + 2);
            *reinterpret_cast<int4*>(l_end + i) = make_int4(e01.x, e01.z, e23.x, e23.z);
            *reinterpret_cast<int4*>(l_pmax + i) = make_int4(e01.y, e01.w, e23.y, e23.w);
        } else {
            for (int j = i; j < rk; ++j) {
                l_start[j] = A.b_start[r0 + j]; l_row[j] = A.b_row[r0 + j];
                const int2 e = A.ep[r0 + j];
                l_end[j] = e.x; l_pmax[j] = e.y;
            }
        }
    }
    if (tid < SL_PAD) l_end[tid - SL_PAD] = 0;
    if (A.lds_seg) for (int i = tid; i < A.n_contigs + 2; i += SL_THREADS) l_seg[i] = A.seg[i];
    // direct-address table over the starts of the slice (all of one contig): two cells per row, cell -> first row at
    // or above its lower edge; the hi-bound of a probe is then one 2-byte read + (expected) one compare
    const int32_t cs = A.b_contig[r0];                                         // contig of the slice's first row
    const bool bins = A.use_bins != 0 && rk >= 2 && cs == A.b_contig[r0 + rk - 1] && (uint32_t)cs < (uint32_t)A.n_contigs;      // uniform
    uint32_t s0 = 0, s1 = 0;
    int bshift = 0;
    const int ncell = 2 * rk;
    // single-contig slice: the segment bounds are the same for every probe of the slice's contig (others cannot match here)
    int u_a = 0, u_la = 0, u_lb = 0;
    if (bins) {
        u_a = A.seg[cs];
        const int b = A.seg[cs + 1];
        u_la = u_a - r0; u_la = u_la < 0 ? 0 : (u_la > rk ? rk : u_la);
        u_lb = b - r0; u_lb = u_lb < 0 ? 0 : (u_lb > rk ? rk : u_lb);
        if (u_lb < u_la) u_lb = u_la;
    }
    __syncthreads();
    if (bins) {
        s0 = flip(l_start[0]); s1 = flip(l_start[rk - 1]);
        while ((unsigned long long)((s1 - s0) >> bshift) + 1ull > (unsigned long long)ncell) ++bshift;
        for (int i = tid; i <= ncell; i += SL_THREADS) l_bin[i] = i == ncell ? (unsigned short)rk : (unsigned short)0xffff;
        __syncthreads();
        for (int i = tid; i < rk; i += SL_THREADS) {
            const uint32_t c1 = (flip(l_start[i]) - s0) >> bshift;
            const bool head = i == 0 || ((flip(l_start[i - 1]) - s0) >> bshift) != c1;
            if (head) l_bin[c1] = (unsigned short)i;
        }
        __syncthreads();
        // empty cells take the next head: suffix minimum (heads ascend with the cell index)
        const int per_t = (ncell + 1 + SL_THREADS - 1) / SL_THREADS;
        const int c_lo = tid * per_t, c_hi = (c_lo + per_t) < (ncell + 1) ? (c_lo + per_t) : (ncell + 1);
        uint32_t mn = 0xffffffffu;
        for (int c = c_lo; c < c_hi; ++c) { const uint32_t x = l_bin[c]; mn = x < mn ? x : mn; }
        uint32_t run = sl_block_suffix_min_excl(mn, reinterpret_cast<uint32_t*>(wsum));
        for (int c = c_hi - 1; c >= c_lo; --c) { const uint32_t x = l_bin[c]; run = x < run ? x : run; l_bin[c] = (unsigned short)run; }
        __syncthreads();
    }

    int4 nxt[ITEMS];
    auto load_tile = [&](int64_t tb) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int64_t q = tb + j * SL_THREADS + tid;
            typedef int v4i __attribute__((ext_vector_type(4)));
            if (q < q1) { const v4i t = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(A.rec + q)); nxt[j] = make_int4(t.x, t.y, t.z, t.w); }
            else nxt[j] = make_int4(0, 0, 0, -1);
        }
    };
    // per-probe state of the tile whose matches are known but not yet emitted
    int32_t qs[ITEMS], qrow[ITEMS];
    int seg_a[ITEMS], hi[ITEMS], lo_s[ITEMS];
    uint32_t mask[ITEMS];
    int cnt[ITEMS];
    bool lng[ITEMS];

    // Phase A of a tile: hi-bound + window of every probe -> cnt / mask / lng.  Consumes the prefetched records and
    // puts the next tile's records in flight.  Touches only the slice in LDS, never the staging buffer.
    auto match_tile = [&](int64_t tb) {
        int32_t qe[ITEMS];
        bool valid[ITEMS];
        int lb[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            qs[j] = nxt[j].x; qe[j] = nxt[j].y; qrow[j] = nxt[j].z;
            const int32_t c = nxt[j].w;
            if (bins) {
                valid[j] = (tb + j * SL_THREADS + tid < q1) && c == cs;
                seg_a[j] = u_a; lo_s[j] = u_la; lb[j] = u_lb;
            } else {
                valid[j] = (tb + j * SL_THREADS + tid < q1) && (uint32_t)c < (uint32_t)A.n_contigs;
                int a = 0, b = 0;
                if (valid[j]) {
                    if (A.lds_seg) { a = l_seg[c]; b = l_seg[c + 1]; }
                    else { a = A.seg[c]; b = A.seg[c + 1]; }
                }
                seg_a[j] = a;
                // the contig's rows inside the slice, slice-local: [la, lb)
                int la = a - r0; la = la < 0 ? 0 : (la > rk ? rk : la);
                int lbb = b - r0; lbb = lbb < 0 ? 0 : (lbb > rk ? rk : lbb);
                lo_s[j] = la; lb[j] = lbb < la ? la : lbb;
            }
        }
        if (tb + TILE < q1) load_tile(tb + TILE);                        // next tile's records in flight
        if (A.ablate & 1) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) lo_s[j] = lb[j] > 20 ? 20 + (qe[j] & 1023) % (lb[j] - 19) : lb[j];
        } else if (bins) {
            // hi-bound through the table: first row whose start reaches q.end, clamped to the contig's rows [la, lb)
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const unsigned long long tu = (unsigned long long)flip(qe[j]) + (STRICT ? 0ull : 1ull);
                int r;
                if (tu <= s0) r = 0;
                else if (tu > s1) r = rk;
                else {
                    const uint32_t cl = ((uint32_t)tu - s0) >> bshift;
                    r = l_bin[cl];
                    const int rend = l_bin[cl + 1];
                    while (r < rend && (unsigned long long)flip(l_start[r]) < tu) ++r;
                }
                r = r < lo_s[j] ? lo_s[j] : r;
                lo_s[j] = r > lb[j] ? lb[j] : r;
            }
        } else {
            // hi-bound: number of rows of [la, lb) whose start fails to reach q.end, fixed trip count, probes interleaved
            for (int step = g.p2r; step > 0; step >>= 1) {
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const int t = lo_s[j] + step;
                    if (t <= lb[j]) {
                        const int32_t sv = l_start[t - 1];
                        if (STRICT ? (sv < qe[j]) : (sv <= qe[j])) lo_s[j] = t;
                    }
                }
            }
        }
        // window below hi, branch-free.  Every row below hi already satisfies start (<) q.end, so row p matches iff
        // q.start (<) end[p] -- and then q.start (<) pmax[p] holds for it and for every row between it and hi (pmax is the
        // prefix max of the ends): the prefix max is only needed to know whether rows BELOW the examined ones can still
        // match.  The SL_WIN ends below hi are read unconditionally (front padding covers the slice's lower edge), bit t
        // <=> row hi-1-t; one prefix-max read of the first row that was not examined decides whether the exact per-lane
        // loop has to redo the probe (window longer than SL_WIN rows, or running on below the slice: rare).
        bool need = false;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            hi[j] = r0 + lo_s[j];
            // sixteen ends from the 16-byte aligned address at or below row hi-SL_WIN: four ds_read_b128 instead of twelve
            // ds_read_b32 (random lanes: ~7 LDS cycles per b32 instruction against ~10 per b128 one).  Element i <=> slice row
            // al + i; rows at or above hi are shifted out, bit reversal turns "ascending row" into "bit t <=> row hi-1-t".
            const int al = (lo_s[j] - SL_WIN) & ~3;
            const int dd = lo_s[j] - al;                                       // SL_WIN .. SL_WIN + 3 rows of the sixteen lie below hi
            uint32_t m = 0;
            if (A.ablate & 2) m = (uint32_t)qs[j] & 3u;
            else {
                const int4* p4 = reinterpret_cast<const int4*>(l_end + al);
                uint32_t m16 = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int4 v = p4[q];
                    m16 |= (lt_op<STRICT>(qs[j], v.x) ? 1u : 0u) << (4 * q);
                    m16 |= (lt_op<STRICT>(qs[j], v.y) ? 1u : 0u) << (4 * q + 1);
                    m16 |= (lt_op<STRICT>(qs[j], v.z) ? 1u : 0u) << (4 * q + 2);
                    m16 |= (lt_op<STRICT>(qs[j], v.w) ? 1u : 0u) << (4 * q + 3);
                }
                m = __brev(m16 << (32 - dd));
            }
            const int lowlim = seg_a[j] > r0 ? seg_a[j] : r0;
            int nrows = hi[j] - lowlim;                                        // rows of the window that exist in LDS
            nrows = nrows < 0 ? 0 : (nrows > SL_WIN ? SL_WIN : nrows);
            m &= (1u << nrows) - 1u;
            const int below = hi[j] - 1 - nrows;                               // first row that was not examined
            bool fb = false;
            if (below >= seg_a[j] && !(A.ablate & 2)) fb = below < r0 ? true : lt_op<STRICT>(qs[j], l_pmax[below - r0]);
            lng[j] = valid[j] && fb;
            mask[j] = valid[j] ? m : 0u;
            need |= lng[j];
            cnt[j] = __popc(mask[j]);
        }
        if (__any(need)) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                if (lng[j]) {
                    int c2 = 0;
                    for (int p = hi[j] - 1; p >= seg_a[j]; --p) {
                        const int2 vv = slice_ep(l_end, l_pmax, A.ep, p, r0);
                        if (!lt_op<STRICT>(qs[j], vv.y)) break;
                        c2 += lt_op<STRICT>(qs[j], vv.x) ? 1 : 0;
                    }
                    cnt[j] = c2;
                }
            }
        }
    };

    if constexpr (MODE == SL_FUSED) {
        // ---- fused single pass, barrier-free tile loop ------------------------------------------------------------------
        // A workgroup barrier per tile costs this kernel ~0.3 ms on config 3 (16 wavefronts with data-dependent timing wait
        // for the slowest one, and nothing else is resident on the CU).  Here the wavefronts run the tile loop independently:
        //   * a wavefront's pairs of a tile are contiguous in the tile's output range, at the offset a returning LDS atomic on
        //     the tile's cursor gives it (arrival order: the fused output order is not reproducible anyway);
        //   * the LAST wavefront to arrive at a tile reserves the tile's range with the one global atomic and publishes the
        //     base in LDS; the others only look at it one iteration later, after they have staged their pairs (wavefront-
        //     private staging region) and matched the next tile -- by then it is there (split-phase: arrive early, wait late);
        //   * two control blocks {cursor, base; arrived, done, ready, seq} alternate; a block is recycled by the last
        //     wavefront that finishes its tile.
        // Every wait is for an event of an EARLIER tile, and all 16 wavefronts of the workgroup are resident: no deadlock.
        // two control blocks: 64-bit {cursor, base} and 32-bit {arrived, done, ready, seq}
        unsigned long long* lc = reinterpret_cast<unsigned long long*>(wsum);  // [2][2]
        int* li = reinterpret_cast<int*>(wsum + 4);                            // [2][4]
        if (tid < 4) lc[tid] = 0;
        if (tid < 8) li[tid] = (tid == 7) ? 1 : 0;                             // block 1 serves tile 1 first (seq = li[1][3])
        __syncthreads();
        const int lane = tid & (kWave - 1), wv = tid / kWave;
        const int wcap = A.stage / SL_WAVES;                                   // pairs of wavefront-private staging
        int2* stw = st + wv * wcap;
        auto ld = [](const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
        auto stv = [](int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
        load_tile(q0);
        const int ntile = (int)((q1 - q0 + TILE - 1) / TILE);
        const int spin_bound = (A.ablate & SL_ABLATE_TILE_FAULT) ? SL_SPIN_BOUND_TEST : SL_SPIN_BOUND;
        const bool tile_fault = (A.ablate & SL_ABLATE_TILE_FAULT) && v == 0;
        int pend_wtot = -1;                                                    // this wavefront's staged pairs of the previous tile
        long long pend_woff = 0;
        // (the last tile's finish is peeled off the loop: one loop of ntile + 1 iterations with the matching under a condition made the
        // compiler carry every per-tile register from iteration to iteration -- cslice.hip.h, round 6)
        auto finish_tile = [&](int t) {
            int* c = li + (t & 1) * 4;
            unsigned long long* c64 = lc + (t & 1) * 2;
            bool timed_out = false;
            IVJ_TILE_WAIT(ld(c + 2) == 0, spin_bound, A.state, lane, timed_out = true);
            long long tb = (long long)__hip_atomic_load(c64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (timed_out) tb = -1;
            if (tb >= 0 && pend_wtot > 0 && pend_wtot <= wcap && !(A.ablate & 32)) {
                for (int i = lane; i < pend_wtot; i += kWave) {
                    const int2 pr = stw[i];
                    __builtin_nontemporal_store(pr.x, A.out_probe + tb + pend_woff + i);
                    __builtin_nontemporal_store(pr.y, A.out_build + tb + pend_woff + i);
                }
            }
            if (lane == 0) {
                if (atomicAdd(c + 1, 1) == SL_WAVES - 1) {                     // last wavefront out: recycle the block for tile t + 2
                    __hip_atomic_store(c64, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    stv(c + 0, 0); stv(c + 1, 0); stv(c + 2, 0);
                    stv(c + 3, t + 2);
                }
            }
        };
        for (int tix = 0; tix < ntile; ++tix) {
            match_tile(q0 + (int64_t)tix * TILE);
            if (tix > 0) finish_tile(tix - 1);                                 // its base was requested one iteration ago
            int* c = li + (tix & 1) * 4;
            unsigned long long* c64 = lc + (tix & 1) * 2;
            IVJ_TILE_WAIT(ld(c + 3) != tix, spin_bound, A.state, lane, return);   // the block is ours (recycled after tile tix - 2)
            int lsum = 0;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) lsum += cnt[j];
            const int linc = wave_inclusive_scan(lsum, SumOp());
            const int wtot = __shfl(linc, kWave - 1, kWave);
            long long woff = 0;
            if (lane == 0) {
                woff = (long long)atomicAdd(c64, (unsigned long long)wtot);
                if (atomicAdd(c + 0, 1) == SL_WAVES - 1) {                     // last wavefront in: the tile's total is complete
                    const long long total = (long long)__hip_atomic_load(c64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    long long base = 0;
                    if (total > 0) {
                        base = (long long)atomicAdd(&A.state[0], (unsigned long long)total);
                        if (base + total > A.capacity) { atomicOr(&A.state[1], 1ull); base = -1; }
                    }
                    __hip_atomic_store(c64 + 1, (unsigned long long)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!(tile_fault && tix == 0)) stv(c + 2, 1);
                }
            }
            woff = ((long long)__shfl((int)(woff >> 32), 0, kWave) << 32) | (unsigned long long)(unsigned int)__shfl((int)(woff & 0xffffffffll), 0, kWave);
            if (wtot > 0 && wtot <= wcap) {
                int off = linc - lsum;                                         // this lane's first slot in the wavefront's region
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    if (cnt[j] != 0 && !(A.ablate & 128)) {
                        if (!lng[j]) {
                            uint32_t m = mask[j];
                            int o = off;
                            const int32_t* pr = l_row + (lo_s[j] - 1);
                            while (m) {
                                const int jj = 31 - __clz(m);
                                m &= ~(1u << jj);
                                if (!(A.ablate & 16)) stw[o] = make_int2(qrow[j], pr[-jj]);
                                ++o;
                            }
                        } else {
                            int o = off + cnt[j] - 1;
                            for (int pp = hi[j] - 1; o >= off; --pp) {
                                const int2 vv = slice_ep(l_end, l_pmax, A.ep, pp, r0);
                                if (lt_op<STRICT>(qs[j], vv.x)) { stw[o] = make_int2(qrow[j], slice_row(l_row, A.b_row, pp, r0)); --o; }
                            }
                        }
                    }
                    off += cnt[j];
                }
            } else if (wtot > wcap) {
                // dense wavefront: its pairs do not fit the staging region -- wait for the base now and write them from the lanes
                bool timed_out = false;
                IVJ_TILE_WAIT(ld(c + 2) == 0, spin_bound, A.state, lane, timed_out = true);
                long long tb = (long long)__hip_atomic_load(c64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (timed_out) tb = -1;
                if (tb >= 0) {
                    long long off = tb + woff + (linc - lsum);
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) {
                        long long o = off + cnt[j] - 1;                        // the f-th match from the top of the window owns slot end - 1 - f
                        for (int pp = hi[j] - 1; o >= off; --pp) {
                            const int2 vv = slice_ep(l_end, l_pmax, A.ep, pp, r0);
                            if (lt_op<STRICT>(qs[j], vv.x)) { A.out_probe[o] = qrow[j]; A.out_build[o] = slice_row(l_row, A.b_row, pp, r0); --o; }
                        }
                        off += cnt[j];
                    }
                }
            }
            pend_wtot = wtot; pend_woff = woff;
        }
        if (ntile > 0) finish_tile(ntile - 1);
        return;
    }
    // Tile loop, rotated so that the matching code exists once: iteration t matches tile t, THEN finishes tile t - 1 (reads
    // the output base its atomic reserved one iteration ago, copies its staged pairs out), then scans / reserves / stages
    // tile t.  The atomic's round trip (microseconds under load) is thus hidden behind the staging of its own tile and the
    // matching of the next one.
    load_tile(q0);
    const int tiles_per_chunk = A.jchunk / TILE;
    const int ntile = (int)((q1 - q0 + TILE - 1) / TILE);
    long long pend_tot = 0, pend_reserved = 0, pend_base = 0;                  // tile whose pairs sit in `st`, waiting for copy-out
    bool pending = false;
    auto finish_pending = [&]() {
        if (!pending) return;                                                  // uniform
        long long tbase = pend_base;
        bool skip = false;
        if (MODE == SL_FUSED) {
            if (tid == 0) {
                if (pend_reserved + pend_tot > A.capacity) { atomicExch(&A.state[1], 1ull); *s_base = -1; }
                else *s_base = pend_reserved;
            }
            __syncthreads();                                                   // staging + base visible
            tbase = *s_base;
            skip = tbase < 0;                                                  // uniform: over capacity, nothing is written
        } else if (!(A.ablate & 64)) __syncthreads();
        if (!skip && !(A.ablate & 32)) {
            const int t = (int)pend_tot;
            for (int i = tid; i < t; i += SL_THREADS) {
                const int2 pr = st[i];
                __builtin_nontemporal_store(pr.x, A.out_probe + tbase + i);
                __builtin_nontemporal_store(pr.y, A.out_build + tbase + i);
            }
        }
        if (!(A.ablate & 64)) __syncthreads();                                 // staging buffer (and s_base) free again
        pending = false;
    };
    for (int tix = 0; tix < ntile; ++tix) {                                      // (the last finish_pending is peeled off: see the fused loop above)
        match_tile(q0 + (int64_t)tix * TILE);
        if (MODE != SL_COUNT) finish_pending();
        const long long tile_id = (long long)v * tiles_per_chunk + tix;
        if (MODE == SL_COUNT) {
            // only the tile total is needed: wavefront sums go straight to the (zeroed) tile slot, no workgroup barrier
            long long wsum_c = 0;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) wsum_c += cnt[j];
#pragma unroll
            for (int d = kWave / 2; d > 0; d >>= 1) wsum_c += __shfl_xor(wsum_c, d, kWave);
            if ((tid & (kWave - 1)) == 0 && wsum_c) atomicAdd(reinterpret_cast<unsigned long long*>(A.tile_tot + tile_id), (unsigned long long)wsum_c);
            continue;
        }
        // exclusive offsets of the tile's pairs (one barrier)
        long long tot, loc0;
        {
            int tsum32 = 0;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) tsum32 += cnt[j];
            if (A.ablate & 4) { loc0 = (long long)tsum32 * tid; tot = (long long)tsum32 * SL_THREADS; }
            else loc0 = sl_block_exclusive_sum_i32(tsum32, reinterpret_cast<int*>(wsum) + (tix & 1) * SL_WAVES, &tot);
        }
        if (tot == 0) continue;                                                // uniform
        // FUSED: the tile reserves its output range with ONE atomic; the value is only read in finish_pending()
        long long tbase = MODE == SL_FILL ? A.tile_tot[tile_id] : 0;
        long long reserved = 0;
        if (MODE == SL_FUSED && tid == 0) reserved = (long long)atomicAdd(&A.state[0], (unsigned long long)tot);
        // emission: pairs staged in LDS at their tile-local offset, then copied out with fully coalesced non-temporal
        // stores.  Mask probes: bit t <=> row hi-1-t, ascending (start, row) order = descending t.  Long windows are
        // rescanned by their lane.  Usual case: the tile's pairs fit ONE staging window (32-bit offsets, no range checks).
        if (tot <= (long long)A.stage) {
            int off = (int)loc0;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                if (cnt[j] != 0 && !(A.ablate & 128)) {
                    if (!lng[j]) {
                        uint32_t m = mask[j];
                        int o = off;
                        const int32_t* pr = l_row + (lo_s[j] - 1);
                        while (m) {
                            const int jj = 31 - __clz(m);
                            m &= ~(1u << jj);
                            if (!(A.ablate & 16)) st[o] = make_int2(qrow[j], pr[-jj]);
                            ++o;
                        }
                    } else {
                        int o = off + cnt[j] - 1;
                        for (int pp = hi[j] - 1; o >= off; --pp) {
                            const int2 vv = slice_ep(l_end, l_pmax, A.ep, pp, r0);
                            if (lt_op<STRICT>(qs[j], vv.x)) { st[o] = make_int2(qrow[j], slice_row(l_row, A.b_row, pp, r0)); --o; }
                        }
                    }
                }
                off += cnt[j];
            }
            pending = true; pend_tot = tot; pend_reserved = reserved; pend_base = tbase;
            continue;
        }
        // dense tile: several staging windows, emitted at once
        bool have_base = MODE == SL_FILL;
        for (long long w0 = 0; w0 < tot; w0 += A.stage) {
            const long long w1 = w0 + A.stage;
            long long off = loc0;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const long long end = off + cnt[j];
                if (cnt[j] != 0 && end > w0 && off < w1) {
                    if (!lng[j]) {
                        uint32_t m = mask[j];
                        long long o = off;
                        while (m) {
                            const int jj = 31 - __clz(m);
                            m &= ~(1u << jj);
                            if (o >= w0 && o < w1) st[o - w0] = make_int2(qrow[j], l_row[hi[j] - 1 - jj - r0]);
                            ++o;
                        }
                    } else {
                        // the f-th match counted from the top of the window owns slot end - 1 - f
                        long long o = end - 1;
                        for (int pp = hi[j] - 1; o >= off; --pp) {
                            const int2 vv = slice_ep(l_end, l_pmax, A.ep, pp, r0);
                            if (lt_op<STRICT>(qs[j], vv.x)) {
                                if (o >= w0 && o < w1) st[o - w0] = make_int2(qrow[j], slice_row(l_row, A.b_row, pp, r0));
                                --o;
                            }
                        }
                    }
                }
                off = end;
            }
            if (!have_base) {                                                  // FUSED, first window: now the range is needed
                if (tid == 0) {
                    if (reserved + tot > A.capacity) { atomicExch(&A.state[1], 1ull); *s_base = -1; }
                    else *s_base = reserved;
                }
                have_base = true;
                __syncthreads();
                tbase = *s_base;
                if (tbase < 0) { __syncthreads(); break; }                     // uniform: over capacity, nothing is written
            } else __syncthreads();
            const int t = (int)((tot - w0) < (long long)A.stage ? (tot - w0) : (long long)A.stage);
            for (int i = tid; i < t; i += SL_THREADS) {
                const int2 pr = st[i];
                __builtin_nontemporal_store(pr.x, A.out_probe + tbase + w0 + i);
                __builtin_nontemporal_store(pr.y, A.out_build + tbase + w0 + i);
            }
            __syncthreads();
        }
    }
    if (MODE != SL_COUNT) finish_pending();
}

}  // namespace ivj
