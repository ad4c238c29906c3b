// ssort.hip.h -- index build, round 3: sample sort.  TWO passes over the build records instead of three LSD passes.
//
// Round 2's build (onesweep.hip.h) sorts the 16-byte records {start, end, row, contig} by three 11-bit LSD passes, each a
// histogram + scan + scatter over all records: 0.50 ms for 5 M rows, launch- and latency-bound.  Here:
//
//   k_ix_minmax    (onesweep.hip.h) key geometry: key = (contig << sbits) | (start - min start)
//   k_ss_sample    ONE workgroup: 16 384 evenly spaced rows -> 64-bit entries {key prefix, position}, bitonic sort in LDS,
//                  every 8th entry is a splitter: 2 048 buckets of about n / 2048 rows whatever the distribution of the
//                  coordinates (the position is part of the order, so even a run of identical keys is cut evenly)
//   k_ss_hist      per-(bucket, chunk) histogram (bucket = bound search over the splitters in LDS)
//   k_scan_lb_u32  one look-back scan
//   k_ss_scatter   STABLE scatter of the records into their buckets (match-any ranking as in k_os_scatter): inside a bucket
//                  the records keep their input order
//   k_ss_sort      one workgroup per bucket: 64-bit keys {full key relative to the bucket's minimum, position inside the
//                  bucket} sorted by a bitonic network in LDS (unique keys: the position breaks ties = input order, the order
//                  the LSD sort produces), records gathered in sorted order and written out
//   k_ix_final     (onesweep.hip.h) sorted records -> index arrays
//
// A bucket larger than SS_CAP rows (probability ~1e-7 per bucket with 8 samples per bucket, none for sorted or uniform
// input) raises meta->overflow; the host then rebuilds with the LSD sort (host_index.hip.h).
#pragma once
#include "onesweep.hip.h"

namespace ivj {

constexpr int SS_BUCKETS = 2048;
constexpr int SS_SAMPLES = 16384;                      // 8 per bucket
constexpr int SS_CAP = 8192;                           // rows a bucket may hold (bitonic network in 64 KB of LDS)
constexpr int SS_ITEMS = 3;                            // records per thread and tile of the stable scatter
constexpr int SS_TILE = OS_THREADS * SS_ITEMS;
constexpr int SS_MIN_ROWS = 1 << 17;                   // below this the LSD sort is used (fewer rows than a few per bucket and sample)

// key prefix that fits 32 bits: the sort key shifted right by max(0, total bits - 32)
__device__ __forceinline__ uint32_t ss_prefix(const OsKey& kg, uint32_t c, int32_t start) {
    const unsigned long long k = os_key(kg, c, start);
    return (uint32_t)(kg.total > 32 ? (k >> (kg.total - 32)) : k);
}
__device__ __forceinline__ uint32_t ss_ckey(int32_t c0, int32_t n_contigs) { return (uint32_t)c0 < (uint32_t)n_contigs ? (uint32_t)c0 : (uint32_t)n_contigs; }

// bucket of entry e = {prefix << 32 | position}: number of splitters <= e  (spl[0 .. SS_BUCKETS - 2] ascending)
__device__ __forceinline__ uint32_t ss_bucket(const unsigned long long* __restrict__ l_spl, unsigned long long e) {
    int pos = 0;
#pragma unroll
    for (int step = SS_BUCKETS / 2; step > 0; step >>= 1) {
        const int t = pos + step;
        if (t <= SS_BUCKETS - 1 && l_spl[t - 1] <= e) pos = t;
    }
    return (uint32_t)pos;
}

// bitonic sort of P (power of two) 64-bit keys in LDS by a workgroup of OS_THREADS threads
__device__ __forceinline__ void ss_bitonic(unsigned long long* __restrict__ a, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += OS_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));          // lower index of pair t at distance j
                const int l = i | j;
                const unsigned long long x = a[i], y = a[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(OS_THREADS) void k_ss_sample(const int32_t* __restrict__ contig, const int32_t* __restrict__ start, int64_t n,
                                                         int32_t n_contigs, int cbits, const OsMeta* __restrict__ meta,
                                                         unsigned long long* __restrict__ spl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_lds[];
    unsigned long long* a = reinterpret_cast<unsigned long long*>(ss_lds);     // SS_SAMPLES entries
    const OsKey kg = os_key_geom(meta, cbits);
    for (int j = threadIdx.x; j < SS_SAMPLES; j += OS_THREADS) {
        const int64_t pos = (int64_t)(((unsigned __int128)(unsigned long long)j * (unsigned long long)n) / SS_SAMPLES);
        a[j] = ((unsigned long long)ss_prefix(kg, ss_ckey(contig[pos], n_contigs), start[pos]) << 32) | (unsigned long long)(uint32_t)pos;
    }
    __syncthreads();
    ss_bitonic(a, SS_SAMPLES);
    for (int k = threadIdx.x; k < SS_BUCKETS - 1; k += OS_THREADS) spl[k] = a[(k + 1) * (SS_SAMPLES / SS_BUCKETS)];
}

__global__ __launch_bounds__(OS_THREADS) void k_ss_hist(const int32_t* __restrict__ contig, const int32_t* __restrict__ start, int64_t n,
                                                       int32_t n_contigs, int cbits, const OsMeta* __restrict__ meta,
                                                       const unsigned long long* __restrict__ spl, int chunk, int nchunks,
                                                       uint32_t* __restrict__ hist) {
    __shared__ unsigned long long l_spl[SS_BUCKETS];
    __shared__ uint32_t h[SS_BUCKETS];
    const OsKey kg = os_key_geom(meta, cbits);
    for (int k = threadIdx.x; k < SS_BUCKETS; k += OS_THREADS) { l_spl[k] = k < SS_BUCKETS - 1 ? spl[k] : ~0ull; h[k] = 0; }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * chunk;
    const int64_t end = base + chunk < n ? base + chunk : n;
    for (int64_t i = base + threadIdx.x; i < end; i += OS_THREADS) {
        const unsigned long long e = ((unsigned long long)ss_prefix(kg, ss_ckey(contig[i], n_contigs), start[i]) << 32) | (unsigned long long)(uint32_t)i;
        atomicAdd(&h[ss_bucket(l_spl, e)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SS_BUCKETS; k += OS_THREADS) hist[(int64_t)k * nchunks + blockIdx.x] = h[k];
}

// stable scatter into the buckets: k_os_scatter's tile machinery (match-any ranking against per-wavefront counter rows,
// records staged in LDS in bucket order, copied out as runs) with the bucket as the digit and tiles of SS_TILE records
struct SsPassLds { int spl, rec, wcnt, d, base, lstart, wsum, total; };
__host__ __device__ inline SsPassLds ss_pass_lds() {
    SsPassLds L;
    int o = 0;
    L.spl = o; o += 8 * SS_BUCKETS;
    L.rec = o; o += 16 * SS_TILE;
    L.wcnt = o; o += 2 * SS_BUCKETS * OS_WAVES;
    L.d = o; o += 2 * SS_TILE;
    L.base = o; o += 4 * SS_BUCKETS;
    L.lstart = o; o += 4 * SS_BUCKETS;
    L.wsum = (o + 15) & ~15; o = L.wsum + 4 * OS_WAVES;
    L.total = o;
    return L;
}

__global__ __launch_bounds__(OS_THREADS) void k_ss_scatter(const int32_t* __restrict__ contig, const int32_t* __restrict__ start,
                                                          const int32_t* __restrict__ end, const int32_t* __restrict__ row_id,
                                                          int4* __restrict__ dst, int64_t n, int32_t n_contigs, int cbits,
                                                          const OsMeta* __restrict__ meta, const unsigned long long* __restrict__ spl,
                                                          int chunk, int nchunks, const uint32_t* __restrict__ off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_lds[];
    const SsPassLds L = ss_pass_lds();
    unsigned long long* l_spl = reinterpret_cast<unsigned long long*>(ss_lds + L.spl);
    int4* l_rec = reinterpret_cast<int4*>(ss_lds + L.rec);
    unsigned short* wcnt = reinterpret_cast<unsigned short*>(ss_lds + L.wcnt);
    unsigned short* l_d = reinterpret_cast<unsigned short*>(ss_lds + L.d);
    uint32_t* base = reinterpret_cast<uint32_t*>(ss_lds + L.base);
    uint32_t* lstart = reinterpret_cast<uint32_t*>(ss_lds + L.lstart);
    uint32_t* wsum = reinterpret_cast<uint32_t*>(ss_lds + L.wsum);
    const OsKey kg = os_key_geom(meta, cbits);
    const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    for (int k = tid; k < SS_BUCKETS; k += OS_THREADS) { l_spl[k] = k < SS_BUCKETS - 1 ? spl[k] : ~0ull; base[k] = off[(int64_t)k * nchunks + blockIdx.x]; }
    for (int k = tid; k < SS_BUCKETS * OS_WAVES / 2; k += OS_THREADS) reinterpret_cast<uint32_t*>(wcnt)[k] = 0;
    __syncthreads();
    const int64_t cbase = (int64_t)blockIdx.x * chunk;
    const int64_t cend = cbase + chunk < n ? cbase + chunk : n;
    const uint64_t lt = lanemask_lt();
    unsigned short* my = wcnt + w * SS_BUCKETS;
    // wavefront w owns elements [w * 192, (w + 1) * 192) of the tile: item j of lane l = w * 192 + j * 64 + l (input order)
    const int el0 = w * (SS_ITEMS * kWave) + lane;
    for (int64_t tbase = cbase; tbase < cend; tbase += SS_TILE) {
        const int tile_n = (int)((cend - tbase) < (int64_t)SS_TILE ? (cend - tbase) : (int64_t)SS_TILE);
        int4 r[SS_ITEMS];
        uint32_t d[SS_ITEMS], rank[SS_ITEMS];
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) {
            const int il = el0 + j * kWave;
            d[j] = 0;
            r[j] = make_int4(0, 0, 0, 0);
            if (il < tile_n) {
                const int64_t i = tbase + il;
                const uint32_t ck = ss_ckey(contig[i], n_contigs);
                r[j] = make_int4(start[i], end[i], row_id ? row_id[i] : (int32_t)i, (int32_t)ck);
                d[j] = ss_bucket(l_spl, ((unsigned long long)ss_prefix(kg, ck, r[j].x) << 32) | (unsigned long long)(uint32_t)i);
            }
        }
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) {
            const bool valid = el0 + j * kWave < tile_n;
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 11; ++b) {
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
        // thread t owns buckets 2t, 2t+1: exclusive prefix over the wavefronts, tile totals, tile-local starts
        uint32_t x0 = 0, x1 = 0;
        {
            uint32_t* row32 = reinterpret_cast<uint32_t*>(wcnt) + tid;
#pragma unroll
            for (int k = 0; k < OS_WAVES; ++k) {
                const uint32_t v = row32[k * (SS_BUCKETS / 2)];
                row32[k * (SS_BUCKETS / 2)] = x0 | (x1 << 16);
                x0 += v & 0xffffu; x1 += v >> 16;
            }
        }
        uint32_t tsum;
        const uint32_t pre = sl_block_exclusive_sum(x0 + x1, wsum, &tsum);
        lstart[2 * tid] = pre;
        lstart[2 * tid + 1] = pre + x0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) {
            if (el0 + j * kWave < tile_n) {
                const uint32_t pos = lstart[d[j]] + (uint32_t)my[d[j]] + rank[j];
                l_rec[pos] = r[j];
                l_d[pos] = (unsigned short)d[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SS_ITEMS; ++j) {
            const int il = j * OS_THREADS + tid;
            if (il < tile_n) {
                const uint32_t dd = l_d[il];
                dst[base[dd] + ((uint32_t)il - lstart[dd])] = l_rec[il];
            }
        }
        for (int k = tid; k < SS_BUCKETS * OS_WAVES / 2; k += OS_THREADS) reinterpret_cast<uint32_t*>(wcnt)[k] = 0;
        __syncthreads();
        base[2 * tid] += x0;                                                   // this thread's two buckets: nobody else touches them
        base[2 * tid + 1] += x1;
        __syncthreads();
    }
}

// one workgroup per bucket: sort the bucket's records by (contig, start, position inside the bucket) and write them to `dst`
__global__ __launch_bounds__(OS_THREADS) void k_ss_sort(const int4* __restrict__ src, int4* __restrict__ dst, int64_t n, int nchunks,
                                                       const uint32_t* __restrict__ off, OsMeta* __restrict__ meta, int cap /* SS_CAP; tests lower it to reach the fallback */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_lds[];
    unsigned long long* a = reinterpret_cast<unsigned long long*>(ss_lds);     // up to SS_CAP keys
    __shared__ unsigned long long wmin[OS_WAVES];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t s0 = off[(int64_t)b * nchunks];
    const int64_t s1 = b + 1 < SS_BUCKETS ? (int64_t)off[(int64_t)(b + 1) * nchunks] : n;
    const int cnt = (int)(s1 - s0);
    if (cnt <= 0) return;                                                      // uniform
    if (cnt > cap) {                                                           // the host rebuilds with the LSD sort
        if (tid == 0) atomicOr(&meta->overflow, 1u);
        return;
    }
    int P = 64;
    while (P < cnt) P <<= 1;
    // composite (contig, start) keys; the bucket's minimum makes them short enough to carry the position in the low 13 bits
    unsigned long long mn = ~0ull;
    for (int i = tid; i < cnt; i += OS_THREADS) {
        const int4 r = src[s0 + i];
        const unsigned long long k = ((unsigned long long)(uint32_t)r.w << 32) | (unsigned long long)flip(r.x);
        a[i] = k;
        mn = k < mn ? k : mn;
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(mn, d, kWave); mn = o < mn ? o : mn; }
    if ((tid & (kWave - 1)) == 0) wmin[tid / kWave] = mn;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < OS_WAVES; ++k) mn = wmin[k] < mn ? wmin[k] : mn;
    for (int i = tid; i < P; i += OS_THREADS) a[i] = i < cnt ? (((a[i] - mn) << 13) | (unsigned long long)i) : ~0ull;
    __syncthreads();
    ss_bitonic(a, P);
    for (int j = tid; j < cnt; j += OS_THREADS) dst[s0 + j] = src[s0 + (int)(a[j] & (SS_CAP - 1))];
}

}  // namespace ivj
